"""Instruction mix of an RT kernel's layer loop from the gfx950 assembly (no GPU needed).
usage: python tools/isa_stats.py [--ilp] [--json out.json] [mangled-kernel-substring] > profiles/<tag>_isa_<kernel>.txt
Compiles csrc/rt_eclipse_i<rule>.hip (the rule follows the SQ flag in the name; rt_eclipse_simpson is
rule 1) -- with --ilp csrc/rt_eclipse_i<rule>_ilp.hip under the build's max-ILP scheduling option -- to assembly, takes
the largest basic block of the kernel (the straight-line block of four layers) and counts
instructions by class."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FP64 = ("v_fma_f64", "v_fmac_f64", "v_mul_f64", "v_add_f64", "v_min_f64", "v_max_f64", "v_rcp_f64")


def main():
    ilp = "--ilp" in sys.argv
    pos = [a for i, a in enumerate(sys.argv[1:], 1) if not a.startswith("--") and sys.argv[i - 1] != "--json"]
    want = pos[0] if pos else ("rt_eclipse_fastILi5ELi4ELi1ELb1ELi0ELi1ELb0E" if ilp
                               else "rt_eclipse_fastILi5ELi4ELi1ELb1ELi0ELi0ELb0E")
    m = re.search(r"Lb[01]ELi(\d)E", want)
    integ = "1" if "simpson" in want else (m.group(1) if m else "0")   # rule 1's single-wave kernel has its own name
    src = "rt_eclipse_i%s_ilp.hip" % integ if ilp else "rt_eclipse_i%s.hip" % integ
    nlay = 4            # layers per straight-line block of the kernel
    if "simpson_slant" in want:   # the `cut slant` kernel of rule 1: its own translation unit, six-layer blocks
        src, ilp, nlay = "rt_eclipse_slant_ilp.hip", True, 6
    extra = ["-mllvm", "-amdgpu-sched-strategy=max-ilp"] if ilp else []
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", *extra,
                               "--cuda-device-only", "-S", "-x", "hip",
                               os.path.join(ROOT, "bart_amd", "csrc", src), "-o", out],
                              stderr=subprocess.DEVNULL)
        s = open(out).read()
    name = re.search(r"^(_ZN6bartrt\w*%s\w*):" % re.escape(want), s, re.M).group(1)
    body = s[s.index(name + ":"):]
    body = body[:body.index(".Lfunc_end")].splitlines()
    blocks, cur, labels, lab, loops = [], [], [], "", []
    for line in body:
        t = line.strip()
        if re.match(r"^\.LBB\d+_\d+:", t):
            blocks.append(cur)
            labels.append(lab)
            cur, lab = [], t.split(":")[0]
        elif t and not t.startswith((";", ".", "//")):
            cur.append(t.split()[0])
            if t.startswith(("s_cbranch", "s_branch")) and t.split()[-1] == lab:
                loops.append(len(blocks))       # this block branches back to its own label: a loop body
    blocks.append(cur)
    labels.append(lab)
    # the layer loop: the largest block that loops on itself (a peeled first block or a masked last
    # block may be larger: they run once per column); else the largest block
    cand = [blocks[i] for i in loops if len(blocks[i]) > 300]     # (a fused prep_body brings small loops of its own)
    big = max(cand, key=len) if cand else max(blocks, key=len)
    cls = collections.Counter()
    for m in big:
        if m.startswith(FP64):
            cls["fp64 VALU"] += 1
        elif m.startswith("v_"):
            cls["other VALU"] += 1
        elif m.startswith("s_waitcnt"):
            cls["s_waitcnt"] += 1
        elif m.startswith("s_"):
            cls["SALU"] += 1
        elif m.startswith("buffer_load"):
            cls["buffer_load"] += 1
        elif m.startswith("ds_"):
            cls["LDS"] += 1
        else:
            cls[m] += 1
    print("kernel:", name)
    print("instructions in the kernel: %d; layer loop body (%d layers): %d = %.1f per layer"
          % (sum(len(b) for b in blocks), nlay, len(big), len(big) / nlay))
    for k, v in cls.most_common():
        print("  %-12s %4d  %6.1f per layer" % (k, v, v / nlay))
    print("by mnemonic:")
    for k, v in collections.Counter(big).most_common():
        print("  %-26s %4d" % (k, v))
    vm = re.findall(r"s_waitcnt vmcnt\((\d+)\)", "\n".join(l.strip() for l in body))
    print("counted vmcnt waits in the kernel:", dict(collections.Counter(int(x) for x in vm)))
    if "--json" in sys.argv:
        # the figures bench.py's fp64 line is computed from, tied to the sources they were taken on
        import json
        sys.path.insert(0, ROOT)
        import bench
        out = sys.argv[sys.argv.index("--json") + 1]
        fma = sum(1 for m in big if m.startswith(("v_fma_f64", "v_fmac_f64")))
        oth = sum(1 for m in big if m.startswith(FP64)) - fma
        fam = "rt_eclipse_simpson_slant" if "simpson_slant" in name else ("rt_eclipse_simpson" if "simpson" in name else "rt_eclipse_fast")
        json.dump({"kernel": name, "kernel_family": fam,
                   "source_id": bench.source_id(),
                   "fp64_fma_per_layer": fma / nlay, "fp64_other_per_layer": oth / nlay,
                   "instructions_per_layer": len(big) / nlay, "file": os.path.basename(out)},
                  open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
