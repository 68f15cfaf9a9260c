"""Copies the rocprofv3 summaries bench.py's numbers come from out of the
scratch gpurun_out/ tree into profiles/ (tracked).  Usage:
    python tools/collect_profiles.py <round-tag> <stats_dir> [<pmc_fetch_dir> <pmc_write_dir> <calib_dir>]
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def ours(name):
    return "bartrt::" in name


def kernel_stats(d, out):
    f = glob.glob(os.path.join(d, "*", "*_kernel_stats.csv"))[0]
    rows = list(csv.reader(open(f)))
    with open(out, "w", newline="") as g:
        w = csv.writer(g)
        w.writerow(rows[0])
        for r in rows[1:]:
            if ours(r[0]) or "copyBuffer" in r[0]:
                w.writerow(r)


def pmc_mean(d, kernel, counter):
    """Mean counter value, mean duration (us), launch count and the kernel's own name
    over the dispatches whose name contains `kernel`."""
    f = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))[0]
    v, t, names = [], [], set()
    for r in csv.DictReader(open(f)):
        if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
            v.append(float(r["Counter_Value"]))
            t.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            names.add(r["Kernel_Name"])
    return sum(v) / len(v), sum(t) / len(t) / 1e3, len(v), sorted(names)


if __name__ == "__main__":
    tag, stats = sys.argv[1], sys.argv[2]
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    kernel_stats(stats, os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv"))
    if len(sys.argv) >= 6:
        fetch, write, calib = sys.argv[3:6]
        # the calibration run prints the bytes its launch has to bring in from HBM
        expected = None
        log = calib.rstrip("/") + ".log"
        if os.path.exists(log):
            for line in open(log):
                if line.startswith("expected HBM bytes per launch:"):
                    expected = float(line.split(":")[1])
        if len(sys.argv) > 6:
            expected = float(sys.argv[6])
        if expected is None:
            raise SystemExit("no 'expected HBM bytes per launch' line in " + log)
        sys.path.insert(0, ROOT)
        import bench
        cal_kb, _, _, cal_names = pmc_mean(calib, "rt_eclipse", "FETCH_SIZE")
        factor = expected / (cal_kb * 1024.0)
        f_kb, f_us, n, names = pmc_mean(fetch, "rt_eclipse", "FETCH_SIZE")
        w_kb, w_us, _, _ = pmc_mean(write, "rt_eclipse", "WRITE_SIZE")
        res = {
            "tag": tag, "source_id": bench.source_id(),
            # the workload of tools/profile_round.sh's PMC passes (bench.py defaults)
            "walkers": int(os.environ.get("PMC_WALKERS", 10)), "nwave": 10000, "nlayers": 100,
            "integ": int(os.environ.get("PMC_INTEG", 1)),      # the engine's default rule (bench.py without --integ)
            # bench.py's defaults since round 4: SURVEY 8d's literal opacities, every step its own prep_profiles,
            # the engine's default cut and CIA interpolation
            "kappa": os.environ.get("PMC_KAPPA", "survey8d"), "cut": os.environ.get("PMC_CUT", "slant"),
            "prefetch": bool(int(os.environ.get("PMC_PREFETCH", 0))),
            "kernel": names[0] if len(names) == 1 else names,
            "calibration_kernel": cal_names[0] if len(cal_names) == 1 else cal_names,
            "launches_averaged": n,
            "FETCH_SIZE_KB_raw": f_kb, "WRITE_SIZE_KB_raw": w_kb,
            "fetch_calibration": {
                "known_bytes": expected, "FETCH_SIZE_KB_reported": cal_kb, "factor": factor,
                "method": "tools/pmc_calib.py: 1 walker, toomuch=1e30: every layer's two grid planes come in "
                          "from HBM once (64.0 MB) + the distinct CIA pair planes + records, with the "
                          "kernel's own 16-byte loads; MI355X_MICROARCH.md gives x2 for this load width"},
            "traffic_bytes_per_launch": f_kb * 1024.0 * factor + w_kb * 1024.0,
            "avg_launch_us_in_pmc_pass": f_us,
        }
        json.dump(res, open(os.path.join(ROOT, "profiles", tag + "_pmc.json"), "w"), indent=1)
        json.dump(res, open(os.path.join(ROOT, "profiles", "pmc_latest.json"), "w"), indent=1)
        print(json.dumps(res, indent=1))
