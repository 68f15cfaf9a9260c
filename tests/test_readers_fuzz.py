"""Host-side readers (bart_amd/csrc/io.cpp) under AddressSanitizer + UBSan on corrupted copies of valid inputs, and
the two cross-section layouts -- the sectioned text layout and the HITRAN CIA layout the reference's manual names
(doc/BART_user_manual/BART_user_manual.tex:506-510) -- read to the same table by the product's reader and by the
oracle's."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    import fuzz_readers
    e = str(tmp_path_factory.mktemp("fz") / "fuzz_readers")
    fuzz_readers.build(e)
    return e


def test_corrupted_inputs_end_in_a_parsed_file_or_an_ioerror(exe, tmp_path):
    import fuzz_readers
    rep = fuzz_readers.sweep(exe, 25, seed=11, workdir=str(tmp_path))
    assert rep["defect"] == 0, json.dumps(rep["defects"], indent=1)
    assert rep["ok"] + rep["IoError"] == 25 * len(rep["files"]) and rep["IoError"] > 20
    assert ("cia", "hitran.cia") in [tuple(f) for f in rep["files"]]


def two_layouts(tmp_path, nt=4, nw=60, seed=3):
    """One table written in both layouts so that both readers arrive at the same doubles: the HITRAN file carries
    k (cm5 molecule-2), the sectioned file the doubles k * N_L * N_L printed exactly."""
    from bart_amd import synth
    rng = np.random.default_rng(seed)
    temps = np.linspace(400.0, 2800.0, nt)
    wn = 900.0 + 7.5 * np.arange(nw)
    k = 1e-45 * np.exp(rng.normal(size=(nt, nw)))
    alpha = k * synth.LOSCHMIDT * synth.LOSCHMIDT
    a, b = str(tmp_path / "sectioned.dat"), str(tmp_path / "hitran.cia")
    synth.write_cia(a, "H2", "He", temps, wn, alpha, fmt="%.17e")
    synth.write_cia_hitran(b, "H2", "He", temps[::-1], wn, k[::-1])     # (blocks in any order: sorted on reading)
    return a, b, temps, wn, alpha


def test_both_layouts_read_to_one_table_by_the_oracle(tmp_path):
    from oracle import rt_oracle as orc
    a, b, temps, wn, alpha = two_layouts(tmp_path)
    ca, cb = orc.read_cia(a), orc.read_cia(b)
    assert ca["species"] == cb["species"] == ["H2", "He"]
    for k in ("temps", "wn", "alpha"):
        assert np.array_equal(ca[k], cb[k]), k
    assert np.array_equal(ca["alpha"], alpha)


def test_product_reader_accepts_the_hitran_layout_and_rejects_what_is_not(exe, tmp_path):
    a, b, *_ = two_layouts(tmp_path)
    for f in (a, b):
        r = subprocess.run([exe, "cia", f], capture_output=True, text=True)
        assert r.returncode == 0 and r.stdout.startswith("ok"), (r.stdout, r.stderr[-500:])
    lines = open(b).read().split("\n")
    bad = {
        "short block": "\n".join(lines[:30]),
        "two pairs": "\n".join(lines[:61] + [lines[61].replace("H2-He", "H2-H2")] + lines[62:]),
        "no dash": "\n".join([lines[0].replace("H2-He", "H2_He")] + lines[1:]),
        "grids differ": "\n".join(lines[:62] + [lines[62].replace(lines[62].split()[0], "901.0000", 1)] + lines[63:]),
    }
    for what, txt in bad.items():
        p = tmp_path / "bad.cia"
        p.write_text(txt)
        r = subprocess.run([exe, "cia", str(p)], capture_output=True, text=True)
        assert r.returncode == 0 and r.stdout.startswith("IoError"), (what, r.stdout, r.stderr[-300:])
