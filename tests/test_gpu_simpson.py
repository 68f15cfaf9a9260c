"""Integration rule 1 (SURVEY.md App. A-4 as recalled) through its tuned single-wave kernel
(csrc/rt_eclipse_s1.hpp) against the oracle (oracle/rt_oracle.c column_eclipse): the places
where that kernel's arrangement differs from a plain walk -- the first block's trapezoid, the
column's last, masked block at every length modulo four, the `toomuch` cut on every position
of a block (the padded point then falls inside the block, on its masked overrun layers, or
to the epilogue), a cut on the bottom layer (no padded point), the cloud deck's surface
term read back after the loop, and zero-width panels (the non-finite fallback).

Small batches would run the quad-layer kernel, so the sweeps run in a child process
under BARTRT_KERNEL=mono_ilp (the switch is read once per process)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import re, sys, os
import numpy as np
sys.path.insert(0, %(root)r)
sys.path.insert(0, os.path.join(%(root)r, "tests"))
from bart_amd import engine, synth, transit_module as trm
from oracle import rt_oracle as orc
from test_gpu_parity import walkers

mode = sys.argv[1]
tmp = sys.argv[2]
CUT = sys.argv[3]                 # vertical | slant (DESIGN.md C19)
INTEG = int(sys.argv[4])

def check(cfg, profs, what, clouds=(None,)):
    engine.init(cfg)
    try:
        trm.set_integ(INTEG); trm.set_cut(CUT)
        o = orc.OracleEngine(cfg, integ=INTEG, cut=CUT)
        if CUT == "slant":          # the per-ray cut runs its own single-wave kernels, not the generic one
            engine.walked_begin(); engine.run_batch(profs); kname = engine.walked_end()[2]
            assert "slant" in kname.lower() or "per lane" in kname, kname
            km = os.environ.get("BARTRT_KERNEL")
            assert ("per lane" in kname) == (km in ("quad", "octo", "hexa", "r32", "quadrays", "octorays")), kname
            assert ("all rays per lane" in kname) == (km in ("quad", "octo", "hexa", "r32") and INTEG == 1), kname
            assert ("team" in kname) == (os.environ.get("BARTRT_KERNEL") == "team"), kname
        for ct in clouds:
            if ct is not None:
                trm.set_cloudtop(float(ct)); o.set_cloudtop(float(ct))
            ref = o.run_batch(profs)
            got = engine.run_batch(profs)
            assert np.all(np.isfinite(got)), what
            # a dozen layers with tau steps >> 1: panels of both signs, see test_cloud_deck_sweep
            np.testing.assert_allclose(got, ref, rtol=1e-10, atol=1e-12 * np.abs(ref).max(),
                                       err_msg="%%s cloud=%%s" %% (what, ct))
    finally:
        trm.free_memory()

if mode == "lengths":
    for L in (2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 29, 30, 31, 32, 100):
        c = synth.make_case(os.path.join(tmp, "L%%d" %% L), nlayers=L, nwave=130)
        profs = walkers(c, 6, seed=L)
        lp = np.log10(c.press_bar)
        clouds = [None] + list(np.unique(np.concatenate([lp[:: max(1, L // 9)], lp[-4:], [lp.min() - 0.3, lp.max() + 0.3]])))
        check(c.tcfg, profs, "L=%%d" %% L, clouds)
elif mode == "cuts":
    # toomuch from "the second layer already" to "never": the cut lands on every block position
    c = synth.make_case(os.path.join(tmp, "cuts"), nlayers=43, nwave=260)
    profs = walkers(c, 6, seed=4)
    txt = open(c.tcfg).read()
    for i, tm in enumerate([1e-9, 1e-6, 1e-4, 3e-3, 0.05, 0.3, 0.7, 2.0, 5.0, 10.0, 40.0, 300.0, 1e30]):
        cfg = c.tcfg + ".tm%%d" %% i
        open(cfg, "w").write(re.sub(r"(?m)^toomuch .*$", "toomuch %%r" %% tm, txt))
        lp = np.log10(c.press_bar)
        check(cfg, profs, "toomuch=%%g" %% tm, [None, lp[20], lp[3]])
elif mode == "zero":
    # exactly zero extinction in the top layers (tau stays 0: zero-width panels), everywhere, and
    # in a band in the middle of the column
    for tag, zero in (("top", slice(-9, None)), ("all", slice(None)), ("mid", slice(10, 17))):
        c = synth.make_case(os.path.join(tmp, "z" + tag), nlayers=30, nwave=130, cia=False)
        op = orc.read_opacity(c.opacity)
        k = op["kappa"].copy()
        k[zero] = 0.0                      # atm order: index 0 = bottom
        ids, tg, pr, wnn = op["ids"].copy(), op["temps"].copy(), op["press"].copy(), op["wn"].copy(); del op
        synth.write_opacity(c.opacity, ids, tg, pr, wnn, kappa=k)
        profs = walkers(c, 6, seed=11)
        engine.init(c.tcfg); trm.set_integ(INTEG); trm.set_cut(CUT)
        o = orc.OracleEngine(c.tcfg, integ=INTEG, cut=CUT)
        ref, got = o.run_batch(profs), engine.run_batch(profs)
        trm.free_memory()
        assert np.all(np.isfinite(got)), tag
        if tag == "all":
            assert np.all(got == 0.0) and np.all(ref == 0.0)
        np.testing.assert_allclose(got, ref, rtol=1e-10, atol=1e-12 * np.abs(ref).max(), err_msg=tag)
print("ok")
"""


@pytest.mark.parametrize("cut,integ,kernel", [("vertical", 1, "mono_ilp"), ("slant", 1, "mono_ilp"), ("slant", 0, "mono_ilp"),
                                              ("slant", 2, "mono_ilp"), ("slant", 1, "quad"), ("slant", 0, "quad"),
                                              ("slant", 2, "quad"), ("slant", 1, "team"), ("slant", 1, "octo"),
                                              ("slant", 1, "quadrays"), ("slant", 1, "octorays"), ("slant", 1, "hexa"),
                                              ("slant", 1, "r32")])
@pytest.mark.parametrize("mode", ["lengths", "cuts", "zero"])
def test_simpson_single_wave_kernel(tmp_path, mode, cut, integ, kernel):
    """(cut slant: the same sweeps through rt_eclipse_simpson_slant / rt_eclipse_fast<SLANT>, where every ray
    angle ends on its own layer -- the deaths, pads and decks of five rays land on every block position -- and,
    kernel = quad / octo / hexa / r32, through the layer-parallel walk with all rays per lane (rule 1: rt_eclipse_quad<..., ALLR>,
    4 / 8 / 16 / 32 layers per step) or
    one ray per lane (rules 0 / 2, and rule 1 as quadrays / octorays: <..., RAYS>); kernel = team, through the three waves
    per column of rt_eclipse_slant_team.)"""
    env = dict(os.environ, BARTRT_KERNEL=kernel)
    out = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}, mode, str(tmp_path), cut, str(integ)],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout[-2000:] + out.stderr[-4000:]


def test_simpson_headline_batches_match_oracle_slices(small_case):
    """The default kernel choice at 110 walkers x 13 columns (past the quad-layer and producer /
    consumer ranges: the single-wave kernel) on the 777-sample grid, against the oracle."""
    from bart_amd import engine, transit_module as trm
    from oracle import rt_oracle as orc
    from test_gpu_parity import walkers
    c = small_case
    engine.init(c.tcfg)
    try:
        trm.set_integ(1)
        o = orc.OracleEngine(c.tcfg, integ=1)
        profs = walkers(c, 110, seed=77)      # 110 x 13 columns: past the producer / consumer range
        got = engine.run_batch(profs)
        np.testing.assert_allclose(got[::9], o.run_batch(profs[::9]), rtol=1e-10)
    finally:
        trm.free_memory()
