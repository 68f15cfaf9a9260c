"""Synthetic line-by-line input sets: a seeded TLI plus a transit cfg that names
it through ``linedb`` (and, optionally, an ``opacityfile`` to be generated)."""
from __future__ import annotations

import os

from . import synth


def make_lbl_case(outdir: str, molecules=("H2O", "CO"), nlines=2000, nwave=400, wnlow=2000.0,
                  wndelt=0.05, nlayers=20, with_table=False, cia=False, seed=20260104,
                  nwidth=20, ethresh=1e-6, wnosamp=1, **kw):
    """Engine inputs without an opacity table + a TLI covering the grid with a
    margin.  ``with_table=True`` adds ``opacityfile`` (a file that does not exist
    yet: the engine builds it from the lines on first init, like
    ``transit --justOpacity``)."""
    os.makedirs(outdir, exist_ok=True)
    wnhigh = wnlow + wndelt * (nwave - 1)
    tli = os.path.join(outdir, "lines.tli")
    dbs = synth.synth_linelist(molecules, nlines, wnlow - 30.0, wnhigh + 30.0, seed=seed)
    synth.write_tli(tli, dbs, wnlow - 30.0, wnhigh + 30.0)
    # wnosamp: the oversampling of the line sums (reference cfgs carry 2160,
    # examples/demo/transit_demo.cfg:27-29); 1 = evaluated on the output points
    extra = {"linedb": tli, "nwidth": nwidth, "ethresh": ethresh, "wnosamp": wnosamp}
    if with_table:
        extra["opacityfile"] = os.path.join(outdir, "opacity_from_lines.dat")
    extra.update(kw.pop("extra_keys", {}) or {})
    case = synth.make_case(outdir, nlayers=nlayers, nwave=nwave, wnlow=wnlow, wndelt=wndelt,
                           opmol=(), cia=cia, extra_keys=extra, **kw)
    case.tli = tli
    case.linedbs = dbs
    return case
