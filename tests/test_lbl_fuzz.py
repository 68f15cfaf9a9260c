"""Seeded random line-by-line cases (tools/lbl_fuzz.py): grid length / spacing / start, line
list, pressure range, nwidth, ethresh, wnosamp and its rule -- the device extinction against
the scipy-Faddeeva oracle at 1e-7, and the blocks of a 2-4-way sharded engine against the
unsharded array bit for bit."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(100, 130))
def test_random_lbl_case(seed):
    import lbl_fuzz
    ok, line = lbl_fuzz.run_case(seed)
    assert ok, line
