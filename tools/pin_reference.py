"""The day the reference engine's outputs arrive: pin the RT oracle in one command.

    python tools/pin_reference.py <refdir> [--write-golden] [--tol 1e-6] [--samples 96]

<refdir> holds what a run of the reference's engine (`transit -c X.cfg`, exosports/transit -- the empty submodule of
/root/reference/.gitmodules:8-10) read and wrote: one or more transit configuration files `*.cfg`, the inputs they name
(atmosphere, molecule, opacity-grid, cross-section, TLI files; absolute paths or relative to <refdir>) and, per cfg, the
spectrum it produced -- the file its `outspec` key names (or `X.spec` next to `X.cfg`), in the layout
code/readtransit.py:23-64 reads (one comment line, then rows whose first column is the wavelength in microns and whose
last is the flux) -- and optionally `tau.dat` (code/cf.py:68-94: per sample a `wn` line, a line of nlayers optical
depths, one more line).

What it does, per cfg:
 (a) every input file goes through the PRODUCT's readers (bart_amd/csrc/io.cpp, built here as a small host program
     from tools/fuzz_readers.cpp): the first file that fails validation is reported with the reader's message;
 (b) the oracle (oracle/rt_oracle.py, oracle/lbl_oracle.py) is run on the atmosphere file's own profile under the full
     cross-product of the conventions that no in-tree evidence settles: `integ` 0/1/2 (C6/C8),
     `cut` vertical/slant (C19), `cia_interp` linear/spline (C20); line-by-line cfgs: `voigt` exact/grid (C18) x
     `BARTRT_OSAMP_RULE` divisor/full (C15); and, where the cfg carries the key, with the key honoured or ignored:
     `cloudtop` / `scattering` (C11), `transparent` (C16), `cloudrad`+`cloudext` (C17).  Printed: the maximum relative
     error of every combination against the reference spectrum, best first (and of tau.dat for the best ones);
 (c) --write-golden: the winning combination's vectors go to tests/golden/transit_ref_<name>/ -- a block of
     `--samples` wavenumbers of the inputs (sliced opacity grid, the small input files, a cfg with local paths) and
     expected.npz (wavenumbers, reference spectrum, the combination).  tests/test_reference_pin.py picks every such
     directory up and holds the oracle to it; until one exists that test is skipped and parity stays UNPINNED.

This is readiness, not parity evidence: tests/test_pin_reference.py exercises it with the oracle playing the reference
under a hidden combination."""
from __future__ import annotations

import argparse
import itertools
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

INPUT_KINDS = (("atm", "atm"), ("molfile", "mol"), ("opacityfile", "opacity"))


def read_keys(cfg):
    from oracle import rt_oracle as orc
    return orc.read_tcfg(cfg)


def resolve(path, base):
    return path if os.path.isabs(path) else os.path.join(base, path)


def localised_cfg(cfg, workdir, drop=(), set_keys=None):
    """A copy of the cfg with its file keys made absolute (relative to the cfg's directory), some keys dropped / set."""
    base = os.path.dirname(os.path.abspath(cfg))
    out = []
    seen = set()
    for line in open(cfg):
        s = line.strip()
        if not s or s[0] in "#;":
            continue
        key, _, val = s.partition(" ")
        val = val.strip()
        if key in drop:
            continue
        if set_keys and key in set_keys:
            continue
        if key in ("atm", "molfile", "opacityfile", "outspec"):
            val = resolve(val, base)
        elif key in ("csfile", "linedb"):
            val = ",".join(resolve(x.strip(), base) for x in val.replace(",", " ").split())
        out.append("%s %s" % (key, val))
        seen.add(key)
    for k, v in (set_keys or {}).items():
        out.append("%s %s" % (k, v))
    path = os.path.join(workdir, "%s.%d.cfg" % (os.path.basename(cfg), len(os.listdir(workdir))))
    open(path, "w").write("\n".join(out) + "\n")
    return path


def validate_inputs(cfg, exe):
    """-> list of (key, file, verdict); verdict 'ok' or the reader's message.  Stops describing at the first failure."""
    keys = read_keys(cfg)
    base = os.path.dirname(os.path.abspath(cfg))
    files = [("cfg", cfg, "cfg")]
    for key, kind in INPUT_KINDS:
        if keys.get(key):
            files.append((key, resolve(keys[key], base), kind))
    for key, kind in (("csfile", "cia"), ("linedb", "tli")):
        for f in [x for x in keys.get(key, "").replace(",", " ").split() if x]:
            files.append((key, resolve(f, base), kind))
    rep = []
    for key, path, kind in files:
        if not os.path.exists(path):
            rep.append((key, path, "MISSING"))
            break
        r = subprocess.run([exe, kind, path], capture_output=True, text=True, errors="replace")
        verdict = "ok" if r.returncode == 0 and r.stdout.startswith("ok") else (r.stdout.strip() or r.stderr.strip()[-300:])
        rep.append((key, path, verdict))
        if verdict != "ok":
            break
    return rep


def read_spectrum(path):
    """code/readtransit.py:23-64's layout -> (wavenumber ascending, flux)."""
    rows = [l.split() for l in open(path).read().split("\n")[1:] if l.strip()]
    wl = np.array([float(r[0]) for r in rows])
    fl = np.array([float(r[-1]) for r in rows])
    wn = 1e4 / wl
    o = np.argsort(wn)
    return wn[o], fl[o]


def read_tau_dat(path, nlayers):
    """code/cf.py:68-94 -> (tau[nlayers][nsamples], wn)."""
    lines = open(path).read().split("\n")
    while lines and (lines[0].startswith("#") or not lines[0].strip()):
        lines.pop(0)
    tau_lines, wn_lines = lines[1:-1:3], lines[0:-1:3]
    tau = np.array([[float(x) for x in l.split()] for l in tau_lines if l.strip()])
    wns = np.array([float(l.split()[1]) for l in wn_lines[:len(tau)]])
    assert tau.shape[1] == nlayers, "tau.dat: %d values per sample, the atmosphere has %d layers" % (tau.shape[1], nlayers)
    return tau.T, wns


def combinations(keys):
    dims = {"integ": [0, 1, 2], "cut": ["vertical", "slant"], "cia_interp": ["linear", "spline"] if keys.get("csfile") else ["linear"]}
    if keys.get("linedb") and not keys.get("opacityfile"):
        dims["voigt"] = ["exact", "grid"]
        dims["osamp_rule"] = ["divisor", "full"]
    if "cloudtop" in keys or "scattering" in keys:
        dims["C11_cloud_scattering_keys"] = ["honoured", "ignored"]
    if keys.get("solution", "eclipse") == "transit" and "transparent" in keys:
        dims["C16_transparent"] = ["honoured", "ignored"]
    if float(keys.get("cloudext", 0) or 0) != 0.0:
        dims["C17_ramp_cloud"] = ["honoured", "ignored"]
    names = list(dims)
    return [dict(zip(names, v)) for v in itertools.product(*[dims[n] for n in names])]


def ignored_keys(combo):
    drop = []
    if combo.get("C11_cloud_scattering_keys") == "ignored":
        drop += ["cloudtop", "scattering"]
    if combo.get("C16_transparent") == "ignored":
        drop += ["transparent"]
    if combo.get("C17_ramp_cloud") == "ignored":
        drop += ["cloudext", "cloudrad", "cloudfct"]
    return drop


def oracle_spectrum(cfg, combo, workdir, want_tau=False, wn_slice=None):
    from oracle import rt_oracle as orc
    c = localised_cfg(cfg, workdir, drop=ignored_keys(combo))
    lo, hi = wn_slice if wn_slice else (None, None)
    o = orc.OracleEngine(c, wn_lo=lo, wn_hi=hi, integ=combo["integ"], cut=combo["cut"], cia_interp=combo["cia_interp"])
    atm = orc.read_atm(o.keys["atm"])
    prof = np.vstack([atm["temp"][None, :], atm["abund"].T])
    if "voigt" in combo:
        from oracle import lbl_oracle
        lb = lbl_oracle.LblOracle(c, osamp_rule=combo["osamp_rule"], voigt=combo["voigt"], wn_slice=wn_slice)
        o.set_extra_extinction(lb.extinction(prof))
    if want_tau:
        spec, tau, last = o.run(prof, want_tau=True)
        return o, prof, spec, tau
    return o, prof, o.run(prof), None


def rel_err(got, ref):
    scale = np.abs(ref).max()
    m = np.abs(ref) > 1e-12 * scale
    return float(np.max(np.abs(got[m] / ref[m] - 1.0))) if m.any() else float(np.abs(got - ref).max())


def pin_one(cfg, exe, tol, workdir, verbose=True):
    keys = read_keys(cfg)
    base = os.path.dirname(os.path.abspath(cfg))
    rep = {"cfg": cfg, "inputs": validate_inputs(cfg, exe)}
    bad = [r for r in rep["inputs"] if r[2] != "ok"]
    if bad:
        rep["error"] = "input validation failed at %s (%s): %s" % bad[0]
        return rep
    stem = os.path.splitext(cfg)[0]
    cands = [resolve(keys["outspec"], base)] if keys.get("outspec") else []
    cands += [stem + ".spec", stem + "_spectrum.dat", stem + ".dat"]
    spec_file = next((c for c in cands if os.path.exists(c)), None)
    if not spec_file:
        rep["error"] = "no reference spectrum found (looked for %s)" % ", ".join(cands)
        return rep
    wn_ref, fl_ref = read_spectrum(spec_file)
    rep["spectrum_file"] = spec_file
    rows = []
    for combo in combinations(keys):
        o, prof, spec, _ = oracle_spectrum(cfg, combo, workdir)
        if len(o.wn) != len(wn_ref) or np.max(np.abs(o.wn / wn_ref - 1.0)) > 1e-6:
            rep["error"] = ("the reference spectrum's grid (%d samples, %.4f..%.4f cm-1) is not the grid the inputs give "
                            "(%d samples, %.4f..%.4f)" % (len(wn_ref), wn_ref[0], wn_ref[-1], len(o.wn), o.wn[0], o.wn[-1]))
            return rep
        rows.append((rel_err(spec, fl_ref), combo))
    rows.sort(key=lambda r: r[0])
    rep["combinations"] = [{"max_rel_err": e, **c} for e, c in rows]
    rep["winner"] = rep["combinations"][0]
    rep["pinned"] = bool(rows[0][0] <= tol)
    rep["runner_up_max_rel_err"] = rows[1][0] if len(rows) > 1 else None
    tau_file = next((c for c in (os.path.join(base, "tau.dat"), stem + ".tau.dat") if os.path.exists(c)), None)
    if tau_file:
        L = len(prof[0])
        tau_ref, wn_tau = read_tau_dat(tau_file, L)
        toomuch = float(keys.get("toomuch", 20.0))
        for r in rep["combinations"][:4]:
            combo = {k: v for k, v in r.items() if k != "max_rel_err" and k != "tau_max_rel_err"}
            o, prof, spec, tau = oracle_spectrum(cfg, combo, workdir, want_tau=True)
            idx = np.array([int(np.argmin(np.abs(o.wn - w))) for w in wn_tau])
            mine = tau[idx].T            # [L][nsamples], layer 0 = top
            m = (tau_ref > 0) & (tau_ref < toomuch)
            r["tau_max_rel_err"] = float(np.max(np.abs(mine[m] / tau_ref[m] - 1.0))) if m.any() else None
        rep["tau_file"] = tau_file
    if verbose:
        print("== %s" % cfg)
        for key, path, verdict in rep["inputs"]:
            print("   input %-12s %-50s %s" % (key, os.path.basename(path), verdict))
        print("   reference spectrum: %s (%d samples)" % (spec_file, len(wn_ref)))
        for r in rep["combinations"]:
            print("   %.3e  %s" % (r["max_rel_err"], {k: v for k, v in r.items() if k != "max_rel_err"}))
        print("   -> %s (tolerance %.1e)" % ("PINNED by the first row" if rep["pinned"] else "NO combination reproduces the reference", tol))
    return rep


def write_golden(cfg, rep, nsamples, workdir):
    """tests/golden/transit_ref_<name>/: a contiguous block of the grid around its middle -- sliced opacity grid, the
    small input files, a cfg with local file names -- and expected.npz."""
    from oracle import rt_oracle as orc
    keys = read_keys(cfg)
    base = os.path.dirname(os.path.abspath(cfg))
    name = os.path.splitext(os.path.basename(cfg))[0]
    out = os.path.join(ROOT, "tests", "golden", "transit_ref_" + name)
    if keys.get("linedb") and not keys.get("opacityfile"):
        big = sum(os.path.getsize(resolve(f, base)) for f in keys["linedb"].replace(",", " ").split())
        if big > 8 << 20:
            return "line-by-line case: the TLI files are %.0f MB, too large for a fixture -- cut the line list to the block first" % (big / 1e6)
    os.makedirs(out, exist_ok=True)
    wn_ref, fl_ref = read_spectrum(rep["spectrum_file"])
    W = len(wn_ref)
    i0 = max(0, W // 2 - nsamples // 2)
    i1 = min(W, i0 + nsamples)
    lines = []
    for line in open(cfg):
        s = line.strip()
        if not s or s[0] in "#;":
            continue
        key, _, val = s.partition(" ")
        val = val.strip()
        if key in ("outspec", "outtoomuch", "outsample", "outintens", "outtau") or key in ignored_keys(rep["winner"]):
            continue
        if key in ("atm", "molfile"):
            shutil.copy(resolve(val, base), os.path.join(out, os.path.basename(val)))
            val = os.path.basename(val)
        elif key in ("csfile", "linedb"):
            names = []
            for f in val.replace(",", " ").split():
                shutil.copy(resolve(f, base), os.path.join(out, os.path.basename(f)))
                names.append(os.path.basename(f))
            val = ",".join(names)
        elif key == "opacityfile":
            op = orc.read_opacity(resolve(val, base))
            from bart_amd import synth
            synth.write_opacity(os.path.join(out, "opacity_block.dat"), op["ids"], op["temps"], op["press"], op["wn"][i0:i1],
                                kappa=np.ascontiguousarray(op["kappa"][:, :, :, i0:i1]))
            val = "opacity_block.dat"
        elif key in ("wnlow", "wnhigh") and not keys.get("opacityfile"):
            fct = float(keys.get("wnfct", 1.0))
            val = repr(float((wn_ref[i0] if key == "wnlow" else wn_ref[i1 - 1]) / fct))
        lines.append("%s %s" % (key, val))
    open(os.path.join(out, "transit.cfg"), "w").write("# block [%d, %d) of %s; file names are relative to this directory\n" % (i0, i1, os.path.basename(cfg))
                                                      + "\n".join(lines) + "\n")
    w = rep["winner"]
    np.savez(os.path.join(out, "expected.npz"), wn=wn_ref[i0:i1], spectrum=fl_ref[i0:i1], block=np.array([i0, i1]),
             combination=json.dumps({k: v for k, v in w.items() if k != "max_rel_err" and k != "tau_max_rel_err"}),
             max_rel_err_when_pinned=w["max_rel_err"])
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("refdir")
    ap.add_argument("--write-golden", action="store_true")
    ap.add_argument("--tol", type=float, default=1e-6, help="north_star's relative tolerance")
    ap.add_argument("--samples", type=int, default=96)
    ap.add_argument("--json", default=None, help="also write the full report here")
    a = ap.parse_args(argv)
    import fuzz_readers
    exe = os.path.join(tempfile.gettempdir(), "bartrt_validate_inputs")
    subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tools", "fuzz_readers.cpp"),
                           os.path.join(ROOT, "bart_amd", "csrc", "io.cpp"), "-o", exe])
    del fuzz_readers
    cfgs = sorted(os.path.join(a.refdir, f) for f in os.listdir(a.refdir) if f.endswith(".cfg"))
    if not cfgs:
        print("no *.cfg in %s" % a.refdir)
        return 2
    work = tempfile.mkdtemp(prefix="bartrt_pin_")
    reports = []
    for cfg in cfgs:
        rep = pin_one(cfg, exe, a.tol, work)
        if "error" in rep:
            print("== %s\n   %s" % (cfg, rep["error"]))
        elif a.write_golden and rep["pinned"]:
            rep["golden"] = write_golden(cfg, rep, a.samples, work)
            print("   golden vectors: %s" % rep["golden"])
        elif a.write_golden:
            print("   no golden vectors written: nothing reproduces the reference within %.1e" % a.tol)
        reports.append(rep)
    if a.json:
        json.dump(reports, open(a.json, "w"), indent=1, default=str)
    ok = all(r.get("pinned") for r in reports)
    print("parity: %s" % ("PINNED for %d configuration(s)" % len(reports) if ok else "UNPINNED"))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
