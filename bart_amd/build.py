"""Builds libbartrt.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libbartrt.so")
CLI = os.path.join(HERE, "transit")
SOURCES = ["rt_eclipse_angles.hip", "rt_eclipse_slant_ilp.hip", "rt_eclipse_team.hip", "rt_eclipse_i0.hip", "rt_eclipse_i1.hip", "rt_eclipse_i2.hip", "rt_eclipse_i0_ilp.hip", "rt_eclipse_i1_ilp.hip", "lbl.hip",
           "transit_geom.hip",
           "kernels.hip", "capi.hip", "engine.hip", "step.hip", "mcmc.hip", "share.hip", "svc.hip", "io.cpp"]   # longest first
HEADERS = ["engine.hpp", "kernels.hpp", "rt_eclipse.hpp", "rt_eclipse_s1.hpp", "rt_eclipse_s1s.hpp", "rt_eclipse_s1t.hpp", "imw_tab.hpp", "integ.hpp", "step.hpp", "lbl.hpp", "voigt_coef.hpp", "expint_coef.hpp", "prep.hpp", "io.hpp", "share.hpp", "svc.hpp", "svc_core.hpp",
           "transit_main.cpp", "../../include/bartrt.h"]


# per-file compiler options (see the comment on rt_eclipse_fast in csrc/rt_eclipse.hpp)
ILP = ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]
EXTRA_FLAGS = {"rt_eclipse_i0_ilp.hip": ILP, "rt_eclipse_i1_ilp.hip": ILP, "rt_eclipse_slant_ilp.hip": ILP}
# rt_eclipse_angles.hip is compiled once per ray-grid size other than five: (object name, flags)
ANGLE_SIZES = (1, 2, 3, 4, 6, 7, 8, 9)
VARIANTS = {"rt_eclipse_angles.hip": [("rt_eclipse_a%d" % n, ["-DBARTRT_ANGLES=%d" % n, *ILP]) for n in ANGLE_SIZES]}


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False) -> str:
    flags = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", 
             "-Wall", "-Wno-unused-result", *os.environ.get("BARTRT_CXXFLAGS", "").split()]
    def stale(obj: str, src: str) -> bool:
        """The object is older than its source, this script (the flags) or a header it includes (the
        compiler's own dependency file, written next to the object; without one: any header)."""
        if force or not os.path.exists(obj):
            return True
        t = os.path.getmtime(obj)
        deps = [os.path.join(CSRC, src), os.path.abspath(__file__)]
        try:
            txt = open(obj + ".d").read().replace("\\\n", " ")
            deps += [d for d in txt.split(":", 1)[1].split() if d.startswith(HERE) or d.startswith(os.path.dirname(HERE))]
        except (OSError, IndexError):
            deps += [os.path.join(CSRC, h) for h in HEADERS]
        return any((not os.path.exists(d)) or os.path.getmtime(d) > t for d in deps)

    def compile_one(job) -> str:
        src, name, extra = job
        obj = os.path.join(CSRC, name + ".o")
        if not stale(obj, src):
            return obj
        cmd = [_hipcc(), *flags, *extra, "-MD", "-MF", obj + ".d", "-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        return obj

    jobs = []
    for src in SOURCES:
        if src in VARIANTS:
            jobs += [(src, name, extra) for name, extra in VARIANTS[src]]
        else:
            jobs.append((src, os.path.splitext(src)[0], EXTRA_FLAGS.get(src, [])))
    # translation units are independent: a few compilers side by side
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_one, jobs))
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-o", LIB, *objs]
        subprocess.check_call(cmd)
    # the standalone `transit` executable (C ABI only), found next to the library
    main_cpp = os.path.join(CSRC, "transit_main.cpp")
    if force or not os.path.exists(CLI) or os.path.getmtime(CLI) < max(os.path.getmtime(LIB), os.path.getmtime(main_cpp)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", main_cpp,
                               "-o", CLI, "-L" + HERE, "-lbartrt", "-Wl,-rpath,$ORIGIN"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
