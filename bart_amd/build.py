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
SOURCES = ["rt_eclipse_angles.hip", "rt_eclipse_slant_ilp.hip", "rt_eclipse_qadj.hip", "rt_eclipse_team.hip", "rt_eclipse_i0.hip", "rt_eclipse_i1.hip", "rt_eclipse_i2.hip", "rt_eclipse_i0_ilp.hip", "rt_eclipse_i1_ilp.hip", "lbl.hip",
           "transit_geom.hip",
           "kernels.hip", "capi.hip", "engine.hip", "step.hip", "mcmc.hip", "share.hip", "svc.hip", "rtc.hip", "io.cpp"]   # longest first
HEADERS = ["engine.hpp", "kernels.hpp", "rt_eclipse.hpp", "rt_eclipse_s1.hpp", "rt_eclipse_s1s.hpp", "rt_eclipse_s1t.hpp", "rt_eclipse_qadj.hpp", "kernel_table.inc", "imw_tab.hpp", "integ.hpp", "step.hpp", "lbl.hpp", "voigt_coef.hpp", "expint_coef.hpp", "prep.hpp", "io.hpp", "share.hpp", "svc.hpp", "svc_core.hpp", "rtc.hpp",
           "transit_main.cpp", "../../include/bartrt.h"]


# per-file compiler options (see the comment on rt_eclipse_fast in csrc/rt_eclipse.hpp)
ILP = ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]
EXTRA_FLAGS = {"rt_eclipse_i0_ilp.hip": ILP, "rt_eclipse_i1_ilp.hip": ILP, "rt_eclipse_slant_ilp.hip": ILP}
# rt_eclipse_angles.hip is compiled once per ray-grid size other than five that is wanted AHEAD OF TIME: (object name,
# flags).  None by default since round 6: every BASELINE config and every reference example uses the five-angle grid
# (examples/demo/BART_eclipse.cfg:135), the eight other sizes were 39 MB of objects, and hiprtc instantiates any size
# from the same templates at its first launch (csrc/rtc.hpp; tests/test_gpu_rtc.py).  BARTRT_AOT_ANGLES="3 7" at build
# time brings sizes back for machines without a run-time compiler.
ANGLE_SIZES = tuple(int(x) for x in os.environ.get("BARTRT_AOT_ANGLES", "").split())
VARIANTS = {"rt_eclipse_angles.hip": [("rt_eclipse_a%d" % n, ["-DBARTRT_ANGLES=%d" % n, *ILP]) for n in ANGLE_SIZES]}
ANGLE_FLAG = "-DBARTRT_ANGLE_SIZES(X)=" + " ".join("X(%d)" % n for n in ANGLE_SIZES)

# The objects the eclipse RT launch is made of: bartrt_build_id() is a hash of THEIR code (device code objects and host
# text), so the committed profiler figures (PMC traffic, SQ pass, instruction mix) are tied to the code that ships, not
# to the sources' text (a comment edit keeps the id).
RT_OBJECT_PREFIXES = ("rt_eclipse_", "kernels", "engine")
BUILD_ID_INC = os.path.join(CSRC, "build_id.inc")


def _elf_section(blob: bytes, want: str):
    """Bytes of the named section of an ELF64 little-endian object (None if absent)."""
    import struct
    if blob[:4] != b"\x7fELF" or blob[4] != 2:
        return None
    shoff, = struct.unpack_from("<Q", blob, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", blob, 0x3A)
    if shnum == 0:                     # more than 0xff00 sections (the kernel objects have thousands): in section 0
        shnum, = struct.unpack_from("<Q", blob, shoff + 0x20)
    if shstrndx == 0xFFFF:
        shstrndx, = struct.unpack_from("<I", blob, shoff + 0x28)
    hdr = lambda i: struct.unpack_from("<IIQQQQIIQQ", blob, shoff + i * shentsize)
    names = hdr(shstrndx)[4]
    out = []
    for i in range(shnum):
        h = hdr(i)
        n = blob[names + h[0]: blob.index(b"\0", names + h[0])].decode()
        if n == want or n.startswith(want + "."):
            if h[1] != 8:              # (SHT_NOBITS has no bytes)
                out.append(blob[h[4]: h[4] + h[5]])
    return b"".join(out) if out else None


def code_id(objs) -> str:
    """Hash of the code in the RT objects: their device code objects (.hip_fatbin) and host text.  Compilation is
    deterministic and, with -cuid=<object name>, independent of the checkout's path (measured: same bytes from two
    directories and with comment lines added)."""
    import hashlib
    h = hashlib.sha1()
    for o in sorted(objs, key=os.path.basename):
        name = os.path.basename(o)
        if not name.startswith(RT_OBJECT_PREFIXES):
            continue
        blob = open(o, "rb").read()
        h.update(name.encode())
        for sec in (".hip_fatbin", ".text", ".rodata"):
            d = _elf_section(blob, sec)
            h.update(hashlib.sha1(d or b"").digest())
    return h.hexdigest()[:12]


# the kernel headers embedded in the library as text: csrc/rtc.hip instantiates shapes outside the ahead-of-time set from
# them with hiprtc (csrc/rtc.hpp)
RTC_HEADERS = ["kernels.hpp", "integ.hpp", "prep.hpp", "rt_eclipse.hpp", "rt_eclipse_s1.hpp", "rt_eclipse_s1s.hpp",
               "rt_eclipse_s1t.hpp", "rt_eclipse_qadj.hpp"]
RTC_INC = os.path.join(CSRC, "rtc_sources.inc")


def write_rtc_sources() -> None:
    """csrc/rtc_sources.inc (generated, git-ignored): the headers of RTC_HEADERS as string literals + a hash of them."""
    import hashlib
    texts = [open(os.path.join(CSRC, h)).read() for h in RTC_HEADERS]
    sid = hashlib.sha1("\0".join(texts).encode()).hexdigest()[:16]

    def lit(t):
        # adjacent literals of at most 8 kB (a compiler's limit on one literal is not ours to test)
        parts = [t[i:i + 8000] for i in range(0, len(t), 8000)] or [""]
        out = []
        for part in parts:
            assert ")BARTRTC\"" not in part
            out.append('R"BARTRTC(' + part + ')BARTRTC"')
        return "\n".join(out)
    body = ["// generated by bart_amd/build.py from %s -- do not edit" % ", ".join(RTC_HEADERS),
            "static const int kRtcNumHeaders = %d;" % len(RTC_HEADERS),
            "static const char *kRtcHeaderNames[] = {%s};" % ", ".join('"%s"' % h for h in RTC_HEADERS),
            "static const char kRtcSourceId[] = \"%s\";" % sid,
            "static const char *kRtcHeaderSources[] = {"]
    body += [lit(t) + "," for t in texts]
    body.append("};")
    new = "\n".join(body) + "\n"
    if not os.path.exists(RTC_INC) or open(RTC_INC).read() != new:
        open(RTC_INC, "w").write(new)


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
    flags = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", ANGLE_FLAG,
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
        # -cuid: the compilation unit's id (suffix of its internal symbols) from the object's name instead of a hash
        # of the source PATH: the code object then is the same bytes wherever the checkout lies
        cmd = [_hipcc(), *flags, *extra, "-cuid=" + name, "-MD", "-MF", obj + ".d", "-x", "hip", "-c",
               os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        return obj

    write_rtc_sources()
    jobs = []
    for src in SOURCES:
        if src in VARIANTS:
            jobs += [(src, name, extra) for name, extra in VARIANTS[src]]
        else:
            jobs.append((src, os.path.splitext(src)[0], EXTRA_FLAGS.get(src, [])))
    # translation units are independent: a few compilers side by side
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_one, jobs))
    # objects this script does not know (experiments, dropped translation units) do not stay behind
    known = {os.path.basename(o) for o in objs} | {"build_id.o"}
    for f in os.listdir(CSRC):
        if f.endswith(".o") and f not in known:
            for g in (f, f + ".d"):
                try:
                    os.remove(os.path.join(CSRC, g))
                except OSError:
                    pass
    # the build's id, compiled in last (csrc/build_id.cpp includes the generated build_id.inc)
    new = '#define BARTRT_BUILD_ID "%s"\n' % code_id(objs)
    if not os.path.exists(BUILD_ID_INC) or open(BUILD_ID_INC).read() != new:
        open(BUILD_ID_INC, "w").write(new)
    id_obj = os.path.join(CSRC, "build_id.o")
    if force or not os.path.exists(id_obj) or os.path.getmtime(id_obj) < os.path.getmtime(BUILD_ID_INC):
        subprocess.check_call(["g++", "-O2", "-fPIC", "-c", os.path.join(CSRC, "build_id.cpp"), "-o", id_obj])
    objs.append(id_obj)
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-o", LIB, *objs, "-ldl"]
        subprocess.check_call(cmd)
    # the standalone `transit` executable (C ABI only), found next to the library
    main_cpp = os.path.join(CSRC, "transit_main.cpp")
    if force or not os.path.exists(CLI) or os.path.getmtime(CLI) < max(os.path.getmtime(LIB), os.path.getmtime(main_cpp)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", main_cpp,
                               "-o", CLI, "-L" + HERE, "-lbartrt", "-Wl,-rpath,$ORIGIN"])
    return LIB


def size_report() -> str:
    """Bytes per translation unit's object (what the 60 MB library is made of)."""
    rows = sorted(((os.path.getsize(os.path.join(CSRC, f)), f) for f in os.listdir(CSRC) if f.endswith(".o")), reverse=True)
    lines = ["%10d  %s" % r for r in rows]
    lines.append("%10d  libbartrt.so" % os.path.getsize(LIB))
    return "\n".join(lines)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--sizes" in sys.argv:
        print(size_report())
