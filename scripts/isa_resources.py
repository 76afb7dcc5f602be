#!/usr/bin/env python3
"""scripts/isa_resources.py [OUT] -- compile kernels.hip for gfx950 with --save-temps and list
what the code object's metadata says every kernel uses (VGPRs, AGPRs, SGPRs, static LDS, scratch,
spills) plus the occupancy that follows (512 VGPRs per SIMD lane on CDNA4, granule 8, at most 8
waves per SIMD).  Runs without a GPU.  The summary is written to OUT (default
profiles/isa_resources.txt) so that statements about registers / occupancy in DESIGN.md can be
checked against the compiler's own numbers rather than against profiler dispatch fields
(rocprofv3's VGPR_Count column reports the arch-VGPR allocation only and LDS_Block_Size does
not include dynamic LDS)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "isa_resources.txt")
    src = os.path.join(ROOT, "cobs_amd", "csrc", "kernels.hip")
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                               "-I" + os.path.dirname(src), "--save-temps", "-c", src, "-o", "k.o"], cwd=tmp,
                              stderr=subprocess.DEVNULL)
        asm = open(os.path.join(tmp, "kernels-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    rows = []
    for m in re.finditer(r"- \.agpr_count:.*?(?=\n  - \.agpr_count:|\namdhsa\.target)", asm, re.S):
        blk = m.group(0)

        def g(k):
            r = re.search(r"\.%s:\s*(\S+)" % k, blk)
            return r.group(1) if r else "?"
        sym = g("name")
        dem = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip()
        dem = dem.replace("cobs_amd::", "").replace("unsigned char", "u8").replace("unsigned short", "u16") \
                 .replace("unsigned int", "u32").replace("unsigned long", "u64")
        dem = re.sub(r"\(.*\)$", "", dem).replace("void ", "")
        vg, ag = int(g("vgpr_count")), int(g("agpr_count"))
        # unified register file: arch VGPRs + AGPRs, allocated in granules of 8, 512 per SIMD lane
        alloc = (vg + 7) // 8 * 8      # .vgpr_count already includes the AGPRs (unified file)
        waves = min(8, 512 // max(alloc, 1))
        rows.append((dem, vg, ag, int(g("sgpr_count")), int(g("group_segment_fixed_size")),
                     int(g("private_segment_fixed_size")), g("vgpr_spill_count"), waves))
    rows.sort()
    lines = ["# kernels.hip @ %s, hipcc -O3 --offload-arch=gfx950; from the code object metadata (.s of --save-temps)" % head,
             "# static_lds excludes the dynamic LDS a launch adds (scan_kernel: merge buffers + expansion table)",
             "%-64s %5s %5s %5s %10s %8s %6s %10s" % ("kernel", "vgpr", "agpr", "sgpr", "static_lds", "scratch", "spill",
                                                     "waves/SIMD")]
    for r in rows:
        lines.append("%-64s %5d %5d %5d %10d %8d %6s %10d" % r)
    text = "\n".join(lines) + "\n"
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as f:
        f.write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
