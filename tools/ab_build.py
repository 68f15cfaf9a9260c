"""A second build of libbartrt.so that differs in ONE translation unit's compiler flags, for same-box A/B runs:
    python tools/ab_build.py <name> <source.hip> [extra hipcc flags ...]
compiles bart_amd/csrc/<source.hip> with the library's flags + the extra ones into bart_amd/csrc/ab_<name>.o and links
bart_amd/libbartrt_<name>.so from it and the other objects of the regular build (which must be up to date).  Run a
tool against it with BARTRT_LIBPATH=bart_amd/libbartrt_<name>.so."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bart_amd import build as b  # noqa: E402

name, src, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
b.build()
own = [] if "--no-file-flags" in extra else b.EXTRA_FLAGS.get(src, [])     # (e.g. the max-ILP scheduling option)
extra = [x for x in extra if x != "--no-file-flags"]
flags = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", b.ANGLE_FLAG, "-cuid=ab_" + name, *own, *extra]
obj = os.path.join(b.CSRC, "ab_%s.o" % name)
subprocess.check_call([b._hipcc(), *flags, "-x", "hip", "-c", os.path.join(b.CSRC, src), "-o", obj])
objs = []
for s in b.SOURCES:
    if s in b.VARIANTS:
        objs += [os.path.join(b.CSRC, n + ".o") for n, _ in b.VARIANTS[s]]
    else:
        objs.append(obj if s == src else os.path.join(b.CSRC, os.path.splitext(s)[0] + ".o"))
objs.append(os.path.join(b.CSRC, "build_id.o"))      # (the regular build's id: an A/B library is not a shipped one)
out = os.path.join(b.HERE, "libbartrt_%s.so" % name)
subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-o", out, *objs])
print(out)
