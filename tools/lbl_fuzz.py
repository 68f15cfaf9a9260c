"""Random line-by-line cases against the scipy-Faddeeva oracle (GPU box):
    python tools/lbl_fuzz.py [ncases] [seed0]
Each case draws the grid (length, spacing, start), the line list (size, molecules), the
pressure range, nwidth / ethresh, wnosamp and its rule, and a shard split; it compares the
device extinction with the oracle at 1e-7 and the concatenated shard blocks with the
unsharded array bit for bit.  Prints one line per case and a summary; exit code 1 on a
failure."""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402


def draw(rng):
    wndelt = float(rng.choice([0.02, 0.05, 0.1, 0.25, 0.5, 1.0]))
    ptop = 10 ** rng.uniform(-7, -1)
    return dict(molecules=tuple(rng.choice(["H2O", "CO", "CO2", "CH4"], size=rng.integers(1, 4), replace=False)),
                nlines=int(rng.integers(20, 2500)), nwave=int(rng.integers(3, 420)), wndelt=wndelt,
                wnlow=float(rng.uniform(500, 9000)), nlayers=int(rng.integers(2, 14)),
                ptop=ptop, pbottom=ptop * 10 ** rng.uniform(0.5, 7.0),
                nwidth=float(rng.choice([5, 20, 50])), ethresh=float(rng.choice([1e-30, 1e-6, 1e-3])),
                wnosamp=int(rng.choice([1, 1, 2, 6, 45, 2160])), cia=bool(rng.integers(0, 2)),
                seed=int(rng.integers(1, 1 << 30)))


def run_case(s):
    """One seeded case -> (ok, description)."""
    from bart_amd import engine, synth_lbl, transit_module as trm
    from oracle import lbl_oracle
    rng = np.random.default_rng(s)
    kw = draw(rng)
    kw["pbottom"] = min(kw["pbottom"], 300.0)
    rule = "full" if (kw["wnosamp"] in (2, 6) and rng.integers(0, 2)) else "divisor"
    nsh = min(int(rng.integers(2, 5)), kw["nwave"])       # a shard without samples is refused by the engine
    prev = os.environ.get("BARTRT_OSAMP_RULE")
    os.environ["BARTRT_OSAMP_RULE"] = rule
    try:
        d = os.path.join(tempfile.gettempdir(), "lbl_fuzz_%d_%d" % (os.getpid(), s))
        c = synth_lbl.make_lbl_case(d, **kw)
        prof = c.profiles()
        engine.init(c.tcfg)
        ext = engine.lbl_extinction(prof)
        trm.free_memory()
        blocks = []
        for r in range(nsh):
            engine.init(c.tcfg, shard=(r, nsh))
            blocks.append(engine.lbl_extinction(prof))
            trm.free_memory()
        ref = lbl_oracle.LblOracle(c.tcfg, osamp_rule=rule).extinction(prof)
    finally:
        if prev is None:
            os.environ.pop("BARTRT_OSAMP_RULE", None)
        else:
            os.environ["BARTRT_OSAMP_RULE"] = prev
    scale = max(ref.max(), 1e-300)
    err = np.max(np.abs(ext - ref) / np.maximum(np.abs(ref), 1e-12 * scale))
    same = np.array_equal(np.concatenate(blocks, axis=1), ext)
    ok = bool(err < 1e-7 and same and np.all(np.isfinite(ext)))
    return ok, ("seed %d  W %d  dnu %g  lines %d x %d  L %d  p %.1e..%.1e  nwidth %g  ethresh %g  wnosamp %d/%s  "
                "shards %d  err %.2e  %s" % (s, kw["nwave"], kw["wndelt"], kw["nlines"], len(kw["molecules"]),
                                             kw["nlayers"], kw["ptop"], kw["pbottom"], kw["nwidth"], kw["ethresh"],
                                             kw["wnosamp"], rule, nsh, err, "" if same else "SHARDS DIFFER"))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    bad = 0
    for s in range(seed0, seed0 + n):
        ok, line = run_case(s)
        bad += not ok
        print(("ok   " if ok else "FAIL ") + line, flush=True)
    print("%d cases, %d failed" % (n, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
