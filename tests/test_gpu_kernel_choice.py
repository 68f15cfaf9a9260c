"""The kernel the library picks by default (csrc/kernel_table.inc, a table MEASURED by tools/tune_kernels.py -- VERDICT
r5 item 8: the hand-measured column intervals of rounds 4-5 were tuned on three grids, "any other (W, M, L) is
extrapolation") held to the best forced variant on grids the table was NOT tuned on: another sample count with two
molecules, six molecules on 7 000 samples, sixty layers."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_the_table_is_well_formed():
    """(no GPU) every column count has an answer, the answers beyond the last boundary are the single-wave kernel, and
    the compiled table is the one the committed record describes."""
    sys.path.insert(0, ROOT)
    from bart_amd import transit_module as trm
    L = trm.lib()
    names = {"single", "rows4", "rows8", "rows16", "rows32", "adj8", "adj16"}
    for M in (1, 2, 3, 4, 6, 9):
        seen = [L.bartrt_kernel_choice(M, c).decode() for c in range(0, 4000)]
        assert set(seen) <= names
        assert seen[-1] == "single" and L.bartrt_kernel_choice(M, 10 ** 9).decode() == "single"
        assert L.bartrt_kernel_choice(M, -5).decode() == seen[0]
    for c in (40, 100, 200, 300, 400, 900):              # seven and more molecules read the sixth table; none, the first
        assert L.bartrt_kernel_choice(6, c) == L.bartrt_kernel_choice(9, c)
        assert L.bartrt_kernel_choice(0, c) == L.bartrt_kernel_choice(1, c)
    rec = os.path.join(ROOT, "profiles", "r06_kernel_table.json")
    if os.path.exists(rec):
        table = json.load(open(rec))["table"]
        for cls, M in [("m%d" % m, m) for m in range(1, 7)]:
            lo = 0
            for bound, variant, _fallback in table[cls]:
                hi = bound if bound is not None else lo + 1000
                for c in {lo, (lo + hi) // 2, hi}:
                    assert L.bartrt_kernel_choice(M, c).decode() == variant, (cls, c, variant)
                lo = hi + 1


@pytest.mark.gpu
@pytest.mark.parametrize("grid, walkers", [
    ((3333, 2, 100), (1, 2, 3, 4, 5)),        # 53 columns per walker, two molecules
    ((7000, 6, 100), (1, 2, 3)),              # 110 per walker, six molecules
    ((4000, 3, 100), (1, 2, 3, 4, 5)),        # 63 per walker, three molecules
    ((6200, 5, 80), (1, 2, 3)),               # 97 per walker, five molecules, eighty layers
    ((10000, 4, 60), (1, 2, 3)),              # the headline grid with sixty layers
])
def test_default_choice_is_within_seven_percent_of_the_best_forced_variant(grid, walkers):
    W, M, Lyr = grid

    def check():
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "tune_kernels.py"), "--check", str(W), str(M), str(Lyr),
                            *[str(n) for n in walkers], "--repeats", "5"], capture_output=True, text=True, timeout=1500)
        assert r.returncode == 0, r.stderr[-3000:]
        return json.loads([l for l in r.stdout.splitlines() if l.startswith("[")][-1])

    # (0.7 us of absolute slack: the run-to-run spread of a 20-40 us step on one box)
    ok = lambda p: p["default_us"] <= 1.07 * p["best_forced_us"] + 0.7
    res = check()
    if not all(ok(p) for p in res):
        # a timing test on a shared box: a point that fails is measured once more, and the better figure of each side counts
        again = check()
        for p, q in zip(res, again):
            p["default_us"] = min(p["default_us"], q["default_us"])
            p["best_forced_us"] = min(p["best_forced_us"], q["best_forced_us"])
    for p in res:
        assert ok(p), p
