"""Columns that migrate between SIMDs (bart_amd/csrc/rt_eclipse_s1s.hpp, MIG; include/bartrt.h,
bartrt_get_migration_stats): a launch whose single-wave columns do not divide evenly over the GPU's SIMDs hands columns
from SIMDs that walk two to SIMDs that have run out of work, at boundaries of six layers.  Whoever walks a column runs
the same instructions on the same numbers, so the spectra must not change by a BIT against the kernel that does not
migrate (BARTRT_MIG=0) -- not with the default rule (hand over when this SIMD holds two waves at work), not when every
wave hands over whenever somebody waits (BARTRT_MIG=force: columns move several times, also back to SIMDs they came
from), not with a cloud deck or a low `toomuch` that ends columns early.  The rule-1 walk itself is held to the oracle
(oracle/rt_oracle.c column_eclipse) by tests/test_gpu_parity.py; one slice is checked here again.
BARTRT_MIG is read once per process: each setting runs in a child."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from test_gpu_parity import walkers, RTOL

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = """
import json, sys
import numpy as np
sys.path.insert(0, %r)
from bart_amd import engine, transit_module as trm
rep = []
for pfile, tcfg, out in %r:
    p = np.load(pfile)
    engine.init(tcfg)
    a = engine.run_batch(p)
    for _ in range(3):                       # the control words are left as the next launch needs them
        assert np.array_equal(engine.run_batch(p), a)
    few = engine.run_batch(p[:3])            # a launch that does not take this form, in between
    trm.set_cloudtop(-1.0)
    b = engine.run_batch(p)
    engine.walked_begin()                    # the launch's own record: layers walked per column, the kernel's name
    assert np.array_equal(engine.run_batch(p), b)
    walked, _, kernel = engine.walked_end()
    st = trm.get_migration_stats()
    np.save(out, np.array([a, b]))
    np.save(out + ".few.npy", few)
    rep.append(dict(st, kernel=kernel, walked_min=int(walked.min()), walked_max=int(walked.max())))
    trm.free_memory()
print("REP " + json.dumps(rep))
"""


@pytest.fixture(scope="module")
def cases(tmp_path_factory):
    from bart_amd import synth
    d = tmp_path_factory.mktemp("mig")
    # 32 columns of 64 samples x 40 walkers = 1 280 columns (1 024 SIMDs); toomuch 1.5 ends rays in mid-column
    c1 = synth.make_case(str(d / "a"), nlayers=100, nwave=2048, tempdelt=200.0)
    c2 = synth.make_case(str(d / "b"), nlayers=67, nwave=2000, tempdelt=200.0, toomuch=1.5)
    jobs = []
    for c, n in ((c1, 40), (c2, 44)):
        profs = walkers(c, n, seed=5)
        np.save(os.path.join(c.dir, "p.npy"), profs)
        jobs.append((os.path.join(c.dir, "p.npy"), c.tcfg))
    return jobs


def run(jobs, tag, env):
    full = [(p, t, os.path.join(os.path.dirname(p), "s_%s.npy" % tag)) for p, t in jobs]
    e = dict(os.environ)
    for k in ("BARTRT_MIG", "BARTRT_MIG_CB", "BARTRT_KERNEL"):
        e.pop(k, None)
    e.update(env)
    out = subprocess.run([sys.executable, "-c", CHILD % (ROOT, full)], env=e, timeout=900, check=True,
                         stdout=subprocess.PIPE, text=True).stdout
    import json
    rep = json.loads(re.search(r"^REP (.*)$", out, re.M).group(1))
    return [(np.load(o), np.load(o + ".few.npy")) for _, _, o in full], rep


def test_migrating_columns_leave_every_bit(cases):
    from oracle import rt_oracle as orc
    ref, rep0 = run(cases, "off", {"BARTRT_MIG": "0"})
    assert all(r["moves"] == 0 and not r["gave_up"] and "migrate" not in r["kernel"] for r in rep0), rep0
    for tag, env in (("on", {}), ("cb3", {"BARTRT_MIG_CB": "3"}), ("force", {"BARTRT_MIG": "force"}),
                     ("force2", {"BARTRT_MIG": "force", "BARTRT_MIG_CB": "2"})):
        got, rep = run(cases, tag, env)
        for (g, gf), (r, rf), st in zip(got, ref, rep):
            assert not st["gave_up"], (tag, st)
            assert np.array_equal(g, r), (tag, np.abs(g / r - 1).max())
            assert np.array_equal(gf, rf)
            assert not np.array_equal(g[0], g[1])            # the deck matters
        assert all("migrate" in st["kernel"] and st["walked_min"] >= 1 for st in rep), rep
        if "force" in tag:
            assert all(st["moves"] > 0 for st in rep), rep   # columns did move
    # ... and the numbers are the oracle's
    (pfile, tcfg), (r, _) = cases[1], ref[1]
    o = orc.OracleEngine(tcfg)
    sl = slice(0, 6)
    np.testing.assert_allclose(r[0][sl], o.run_batch(np.load(pfile)[sl]), rtol=RTOL)
