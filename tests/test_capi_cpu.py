"""CPU: the C-ABI library loads, exports every symbol include/bartrt.h declares,
refuses to compute without a GPU (no CPU fallback) and reports input errors.
No compute calls are made here."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from bart_amd import build, transit_module as trm
    build.build()
    return trm.lib()


def _declared():
    txt = open(os.path.join(ROOT, "include", "bartrt.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(bartrt_\w+)\s*\(", txt)))


def test_every_declared_symbol_is_exported(lib):
    names = _declared()
    assert len(names) >= 25 and "bartrt_run_transit" in names
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_header_is_plain_c_and_the_c_host_links(lib, tmp_path):
    """include/bartrt.h through a strict C99 compiler with every declared entry point
    referenced, and examples/c_host.c linked against the library."""
    import subprocess
    names = _declared()
    tu = tmp_path / "all.c"
    tu.write_text('#include "bartrt.h"\n'
                  "void *table[] = {%s};\nint main(void) { return table[0] == 0; }\n"
                  % ", ".join("(void *)%s" % n for n in names))
    inc = os.path.join(ROOT, "include")
    libdir = os.path.join(ROOT, "bart_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-Wno-pedantic", "-I" + inc, str(tu),
                           "-L" + libdir, "-lbartrt", "-Wl,-rpath," + libdir, "-o", str(tmp_path / "all")])
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I" + inc,
                           os.path.join(ROOT, "examples", "c_host.c"), "-L" + libdir, "-lbartrt",
                           "-Wl,-rpath," + libdir, "-o", str(tmp_path / "c_host")])
    r = subprocess.run([str(tmp_path / "c_host"), str(tmp_path / "none.cfg"), str(tmp_path / "o.txt")],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "cannot open" in r.stderr


def test_reference_module_names_present():
    from bart_amd import transit_module as trm
    for n in ("transit_init", "get_no_samples", "get_waveno_arr", "set_radius", "set_cloudtop",
              "set_scattering", "run_transit", "free_memory"):      # BARTfunc.py:230-234,351-363,406
        assert callable(getattr(trm, n))


def test_calls_before_init_fail_loudly(lib):
    from bart_amd import transit_module as trm
    trm.free_memory()
    assert lib.bartrt_get_no_samples() < 0
    assert b"not initialised" in lib.bartrt_last_error()
    with pytest.raises(trm.TransitError):
        trm.run_transit([1.0, 2.0], 2)
    with pytest.raises(trm.TransitError):
        trm.set_radius(1.0)


def test_missing_and_malformed_inputs(lib, tmp_path):
    from bart_amd import synth, transit_module as trm
    with pytest.raises(trm.TransitError, match="cannot open"):
        trm.transit_init(3, ["transit", "-c", str(tmp_path / "nope.cfg")])
    with pytest.raises(trm.TransitError, match="no '-c"):
        trm.transit_init(1, ["transit"])
    case = synth.make_case(str(tmp_path / "c"), nwave=16, nlayers=10)
    bad = dict(case.keys)
    bad["solution"] = "transit"                 # transit geometry needs the stellar radius
    synth.write_tcfg(str(tmp_path / "t.cfg"), bad)
    with pytest.raises(trm.TransitError, match="starrad"):
        trm.transit_init(3, ["transit", "-c", str(tmp_path / "t.cfg")])
    bad["solution"] = "sideways"
    synth.write_tcfg(str(tmp_path / "t.cfg"), bad)
    with pytest.raises(trm.TransitError, match="unknown solution"):
        trm.transit_init(3, ["transit", "-c", str(tmp_path / "t.cfg")])
    bad = dict(case.keys)
    del bad["gsurf"]
    synth.write_tcfg(str(tmp_path / "g.cfg"), bad)
    with pytest.raises(trm.TransitError, match="gsurf"):
        trm.transit_init(3, ["transit", "-c", str(tmp_path / "g.cfg")])
    bad = dict(case.keys)
    bad.update(cloudrad="7.0e4 7.5e4", cloudfct="1e5", cloudext="1e-6")   # ramp cloud with its top below its bottom
    synth.write_tcfg(str(tmp_path / "cl.cfg"), bad)
    with pytest.raises(trm.TransitError, match="cloudrad"):
        trm.transit_init(3, ["transit", "-c", str(tmp_path / "cl.cfg")])
    with open(case.opacity, "r+b") as f:
        f.truncate(4000)
    with pytest.raises(trm.TransitError, match="truncated"):
        trm.transit_init(3, ["transit", "-c", case.tcfg])


def test_corrupt_sample_count_is_refused_before_it_sizes_a_buffer(lib, tmp_path):
    """A sample count in the opacity header that the file cannot hold is an input
    error at once -- not a 70 GB allocation (the TLI's transition counts: tests/test_lbl.py)."""
    import struct
    import time
    from bart_amd import synth, transit_module as trm
    case = synth.make_case(str(tmp_path / "o"), nwave=16, nlayers=10)
    with open(case.opacity, "r+b") as f:
        f.seek(24)
        f.write(struct.pack("=q", 1 << 33))             # Nwave
    t0 = time.time()
    with pytest.raises(trm.TransitError, match="truncated|not an opacity grid"):
        trm.transit_init(3, ["transit", "-c", case.tcfg])
    assert time.time() - t0 < 20


def test_no_gpu_means_error_not_fallback(lib, tmp_path):
    """On a box without a GPU a valid configuration must fail with ENODEV."""
    import torch
    from bart_amd import synth, transit_module as trm
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    case = synth.make_case(str(tmp_path), nwave=16, nlayers=10)
    with pytest.raises(trm.TransitError, match="GPU only"):
        trm.transit_init(3, ["transit", "-c", case.tcfg])
    assert lib.bartrt_get_no_samples() < 0


def test_product_does_not_import_the_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkg = os.path.join(ROOT, "bart_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("PT oracle", ""), os.path.join(dirpath, f)


def test_generated_coefficient_tables_are_reproducible():
    """csrc/voigt_coef.hpp is what tools/gen_voigt_coef.py produces (Weideman's FFT recipe), csrc/imw_tab.hpp what
    tools/gen_imw_table.py does (40-digit interpolants of Im w)."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_voigt_coef.py"), "--check"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "max |recipe - header| = 0.0" in r.stdout
    # csrc/imw_tab.hpp (Im w on the real axis, the small-y branch of the Voigt function) from tools/gen_imw_table.py
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_imw_table.py"), "--check"],
                       capture_output=True, text=True)
    assert r.returncode == 0 and "identical to the header: True" in r.stdout, r.stdout + r.stderr


def test_embedded_kernel_sources_compile_at_run_time(lib):
    """csrc/rtc.hpp: shapes outside the ahead-of-time set are instantiated from the kernel headers embedded in the library
    (hiprtc).  No GPU is needed to COMPILE: one kernel of each family the launcher may ask for, on shapes the library
    does not ship -- the embedded text, its !__HIPCC_RTC__ guards and the compiler at hand agree."""
    import ctypes as C
    avail = C.c_int(0)
    assert lib.bartrt_get_rtc_stats(C.byref(avail), None, None, None, None) == 0
    if not avail.value:
        pytest.skip("no libhiprtc.so on this machine")
    for expr, ilp in (("rt_eclipse_simpson_slant<5, 9, 4, true, 1>", 1),
                      ("rt_eclipse_quad<5, 7, 4, false, 16, 1, false, true>", 0),
                      ("rt_eclipse_qadj<5, 8, 4, true, 16>", 0),
                      ("rt_eclipse_fast<12, 4, 2, false, 0, 1, false, true>", 1)):
        n = C.c_long(0)
        rc = lib.bartrt_rtc_compile(expr.encode(), ilp, C.byref(n))
        assert rc == 0 and n.value > 10000, (expr, lib.bartrt_last_error().decode()[:2000])
    n = C.c_long(0)
    assert lib.bartrt_rtc_compile(b"rt_eclipse_no_such_kernel<1>", 0, C.byref(n)) == -4 and n.value < 0
