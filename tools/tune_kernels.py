"""Measures which eclipse kernel serves which launch under the default conventions (rule 1, `cut slant`, five ray
angles) and writes the table the library reads: bart_amd/csrc/kernel_table.inc (+ the figures behind it, as JSON).

    python tools/tune_kernels.py [--out profiles/r06_kernel_table.json] [--write-table]      (on the GPU box, repo root)
    python tools/tune_kernels.py --check W M L walkers...      times the default choice and every forced variant on ONE grid
                                                               (what tests/test_gpu_kernel_choice.py runs on untuned grids)

Every variant is forced in turn (BARTRT_KERNEL: the switch is read once per process, so each (grid, variant) is a child
process of this script) on each tuning grid and walker count; the figure is the STEP -- preparation + RT kernel, a hundred
steps queued back to back, wall clock -- median of --repeats windows (the RT kernel's own HIP-event time is recorded too).  Per table-molecule count (1 .. 6; seven and more read the sixth table) the winner at every measured column count becomes an interval of the table; a variant must beat the interval's
current holder by 3 % to take over (no flapping on noise); where an adjacent-rows variant wins, the best OTHER variant is
recorded as the entry's fallback.  Beyond the last measured column count the single-wave kernel serves everything.

VERDICT r5 item 8: "replace the threshold thicket with a measured table"."""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# BARTRT_KERNEL value -> the table's variant name
VARIANTS = {"r32": "rows32", "hexa": "rows16", "octo": "rows8", "quad": "rows4", "adj16": "adj16", "adj8": "adj8",
            "mono": "single"}
ENUM = {"single": "kVarSingle", "rows4": "kVarRows4", "rows8": "kVarRows8", "rows16": "kVarRows16",
        "rows32": "kVarRows32", "adj8": "kVarAdj8", "adj16": "kVarAdj16"}
MOLS = ("H2O", "CO", "CO2", "CH4", "NH3", "HCN", "C2H2", "TiO", "VO")
# tuning grids: (samples, table molecules, layers) and the walker counts timed on each (columns = walkers x ceil(W / 64))
def _walkers(percol, upto):
    """Walker counts whose column counts cover [percol, upto] about every 40 columns (every count while that is finer)."""
    out, last = [], -10 ** 9
    n = 1
    while n * percol <= upto:
        if n * percol - last >= 38 or n <= 4:
            out.append(n)
            last = n * percol
        n += 1
    return tuple(out)


# tuning grids: (samples, table molecules, layers) and the walker counts timed on each (columns = walkers x ceil(W / 64)).
# One table per table-molecule count, 1 .. 6 (seven and more read the sixth): the crossovers move with the weight of a
# step and not monotonically -- two variants 5 % apart with four molecules are 24 % apart with three.  The sweep goes to
# 700 columns, past the single-wave kernel's take-over at every count.
TUNE = [(2501, 1, 100, _walkers(40, 700)), (5000, 1, 100, _walkers(79, 700))]
for _m in (2, 3, 4, 5, 6):
    TUNE += [(10000, _m, 100, _walkers(157, 700)), (5000, _m, 100, _walkers(79, 700)), (2424, _m, 100, _walkers(38, 700))]
# classes of table-molecule count: (key, name of the table's array, note, predicate)
CLASSES = [("m%d" % m, "kSlantSimpsonM%d" % m, "%d table molecule%s%s" % (m, "s" if m > 1 else "", " and more" if m == 6 else ""),
            (lambda mm: (lambda x: x == mm))(m)) for m in range(1, 7)]


def case_for(W, M, L):
    from bart_amd import synth
    mols = MOLS[:M]
    d = os.path.join(tempfile.gettempdir(), "bartrt_tune_W%d_M%d_L%d" % (W, M, L))
    return synth.make_case(d, nlayers=L, nwave=W, opmol=mols, species=("He", "H2") + mols,
                           abund=(0.15, 0.85) + (1e-4,) * M, kappa_model="survey8d", reuse=True)


def child(W, M, L, walkers, repeats):
    """One process = one BARTRT_KERNEL: RT-kernel microseconds per walker count, one JSON line each."""
    import numpy as np
    import torch
    import bench
    from bart_amd import engine, transit_module as trm
    case = case_for(W, M, L)
    engine.init(case.tcfg)
    try:
        for n in walkers:
            profs = bench.make_profiles(case, n * 8, seed=3).reshape(8, n, -1)
            d = torch.from_numpy(profs).cuda()
            out = torch.empty((n, W), dtype=torch.float64, device="cuda")
            for i in range(40):
                engine.run_batch_dev(d[i % 8], out)
            torch.cuda.synchronize()
            engine.walked_begin()
            engine.run_batch_dev(d[0], out)
            torch.cuda.synchronize()
            kname = engine.walked_end()[2]
            # the figure that counts is the STEP (preparation + RT kernel, queued back to back): the layer-parallel forms
            # may prepare their walkers in their own prologue, the others need a prep_profiles launch in front
            us, rt = [], []
            for _ in range(repeats):
                engine.timing_begin(1)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(100):
                    engine.run_batch_dev(d[i % 8], out)
                torch.cuda.synchronize()
                us.append((time.perf_counter() - t0) / 100 * 1e6)
                ms, nl = engine.timing_end()
                rt.append(ms / nl * 1e3)
            print(json.dumps({"W": W, "M": M, "L": L, "walkers": n, "columns": n * ((W + 63) // 64),
                              "forced": os.environ.get("BARTRT_KERNEL", ""), "us": float(np.median(us)),
                              "rt_kernel_us": float(np.median(rt)), "us_all": [round(x, 2) for x in us], "kernel": kname}),
                  flush=True)
    finally:
        trm.free_memory()


def run_child(W, M, L, walkers, forced, repeats):
    env = dict(os.environ)
    env.pop("BARTRT_KERNEL", None)
    if forced:
        env["BARTRT_KERNEL"] = forced
    cmd = [sys.executable, os.path.abspath(__file__), "--child", str(W), str(M), str(L), "--repeats", str(repeats),
           *[str(n) for n in walkers]]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200)
    if r.returncode != 0:
        raise RuntimeError("child failed (%s on W=%d M=%d): %s" % (forced or "default", W, M, r.stderr[-1500:]))
    return [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]


def measure(grids, repeats, with_default=False):
    rows = []
    for W, M, L, walkers in grids:
        case_for(W, M, L)       # (written once, by this process)
        for forced in ([""] if with_default else []) + list(VARIANTS):
            t0 = time.time()
            rows += run_child(W, M, L, walkers, forced, repeats)
            print("  W=%d M=%d L=%d %-6s %.0f s" % (W, M, L, forced or "default", time.time() - t0), file=sys.stderr, flush=True)
    return rows


def build_table(rows, margin=0.03):
    """-> {"few": [(max_columns, variant, fallback), ...], "many": [...]} and the per-point figures."""
    out, points = {}, {}
    for cls, _arr, _note, pick in CLASSES:
        by_col = {}
        for r in rows:
            if pick(r["M"]) and r["forced"]:
                by_col.setdefault(r["columns"], {}).setdefault(VARIANTS[r["forced"]], []).append(r["us"])
        cols = sorted(by_col)
        pts = []
        holder = None
        for c in cols:
            t = {v: sum(x) / len(x) for v, x in by_col[c].items()}   # (two grids may share a column count: every variant ran on both)
            best = min(t, key=t.get)
            if holder is None or holder not in t or t[best] < t[holder] * (1.0 - margin):
                holder = best
            others = {v: x for v, x in t.items() if not v.startswith("adj")}
            pts.append({"columns": c, "us": {v: round(x, 2) for v, x in sorted(t.items())}, "best": best,
                        "chosen": holder, "fallback": min(others, key=others.get)})
        # intervals: a run of points with one chosen variant ends half way to the next point
        entries = []
        for i, p in enumerate(pts):
            last = i + 1 == len(pts)
            if last or pts[i + 1]["chosen"] != p["chosen"] or \
                    (p["chosen"].startswith("adj") and pts[i + 1]["fallback"] != p["fallback"]):
                bound = None if last else (p["columns"] + pts[i + 1]["columns"]) // 2
                entries.append([bound, p["chosen"], p["fallback"] if p["chosen"].startswith("adj") else p["chosen"]])
        # beyond the last measured point: the single-wave kernel (the last point must have chosen it, or stay as it is
        # up to that point and hand over right after)
        if entries[-1][1] != "single":
            entries[-1][0] = pts[-1]["columns"]
            entries.append([None, "single", "single"])
        out[cls], points[cls] = entries, pts
    return out, points


def write_inc(table, record, grids):
    def arr(name, entries, note):
        body = ", ".join("{%s, %s, %s}" % ("kAllColumns" if b is None else str(b), ENUM[v], ENUM[f]) for b, v, f in entries)
        return "static constexpr KernelChoice %s[] = {   // %s\n    %s};\n" % (name, note, body)
    txt = """// Which kernel serves a launch of rule 1 under `cut slant` on the five-angle ray grid (the default conventions), by the
// launch's 64-sample columns (walkers the variant is chosen for x ceil(samples / 64)) and the number of table molecules.
// GENERATED by tools/tune_kernels.py from launch times measured on an MI355X -- every variant forced in turn
// (BARTRT_KERNEL) on each tuning grid and walker count; the figures behind each boundary are in the JSON record named
// below.  Do not edit by hand: re-run the tool (rt_eclipse.hpp reads the table through slant_simpson_choice()).
//   record: %s
//   tuned on (samples, table molecules, layers): %s
// An entry {max_columns, variant, fallback} serves columns <= max_columns; `fallback` is what runs where the adjacent-rows
// kernel cannot (BARTRT_ADJ=0, a shape with neither an instantiation nor a run-time compiler): the best of the OTHER
// variants in that range.  The last entry of a class is the single-wave kernel for everything beyond.
//   variants: kVarSingle rt_eclipse_simpson_slant; kVarRows4 / 8 / 16 / 32 rt_eclipse_quad<ALLR> with that many layers
//   per step; kVarAdj8 / 16 rt_eclipse_qadj (rows on adjacent lanes)
""" % (record, ", ".join("(%d, %d, %d)" % g[:3] for g in grids))
    for cls, name, note, _pick in CLASSES:
        txt += arr(name, table[cls], note)
    open(os.path.join(ROOT, "bart_amd", "csrc", "kernel_table.inc"), "w").write(txt)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--child", nargs=3, type=int, metavar=("W", "M", "L"))
    ap.add_argument("--check", nargs=3, type=int, metavar=("W", "M", "L"))
    ap.add_argument("walkers", nargs="*", type=int)
    ap.add_argument("--repeats", type=int, default=5)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "kernel_table.json"))
    ap.add_argument("--write-table", action="store_true", help="replace bart_amd/csrc/kernel_table.inc (then rebuild)")
    ap.add_argument("--record-name", default="profiles/r06_kernel_table.json", help="the record's name inside the table's header")
    a = ap.parse_args()
    if a.child:
        return child(*a.child, a.walkers, a.repeats)
    if a.check:
        W, M, L = a.check
        rows = measure([(W, M, L, tuple(a.walkers))], a.repeats, with_default=True)
        res = []
        for n in a.walkers:
            t = {(r["forced"] or "default"): r for r in rows if r["walkers"] == n}
            forced = {VARIANTS[k]: v["us"] for k, v in t.items() if k != "default"}
            res.append({"W": W, "M": M, "L": L, "walkers": n, "columns": t["default"]["columns"],
                        "default_us": t["default"]["us"], "default_kernel": t["default"]["kernel"],
                        "best_forced": min(forced, key=forced.get), "best_forced_us": min(forced.values()), "forced_us": forced})
        print(json.dumps(res))
        return
    rows = measure(TUNE, a.repeats)
    table, points = build_table(rows)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump({"table": table, "points": points, "grids": [list(g[:3]) + [list(g[3])] for g in TUNE], "rows": rows,
               "margin": 0.03, "repeats": a.repeats}, open(a.out, "w"), indent=1)
    print(json.dumps(table))
    if a.write_table:
        write_inc(table, a.record_name, TUNE)


if __name__ == "__main__":
    main()
