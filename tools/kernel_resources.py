#!/usr/bin/env python3
"""Per-kernel resources of the built library, from the gfx950 code objects inside csrc/*.o (no GPU needed): VGPRs (arch +
accumulation), SGPRs, static LDS, scratch, and the waves per SIMD those allow (512 VGPRs per lane of a SIMD on gfx950,
in steps of 8; at most 8 waves).  The occupancy arguments of DESIGN.md section 3 quote this table.

    python tools/kernel_resources.py [filter-regex] > profiles/r05_kernel_resources.txt
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def kernels_of(obj, tmp):
    fat = os.path.join(tmp, "fat.bin")
    co = os.path.join(tmp, "k.co")
    r = subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", obj, fat], capture_output=True)
    if r.returncode != 0 or not os.path.exists(fat) or os.path.getsize(fat) == 0:
        return []
    r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        "--input=" + fat, "--output=" + co], capture_output=True, text=True)
    if r.returncode != 0:
        return []
    notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
    out = []
    for block in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
        block = ".agpr_count:" + block
        get = lambda k: (re.search(r"\.%s:\s*(\S+)" % k, block) or [None, "0"])[1]
        out.append({"name": get("name"), "vgpr": int(get("vgpr_count")), "agpr": int(get("agpr_count")), "sgpr": int(get("sgpr_count")),
                    "lds": int(get("group_segment_fixed_size")), "scratch": int(get("private_segment_fixed_size")),
                    "wg": int(get("max_flat_workgroup_size"))})
    for f in (fat, co):
        if os.path.exists(f):
            os.remove(f)
    return out


def main():
    want = re.compile(sys.argv[1]) if len(sys.argv) > 1 else None
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for obj in sorted(glob.glob(os.path.join(ROOT, "sybil_amd", "csrc", "*.o"))):
            for k in kernels_of(obj, tmp):
                k["file"] = os.path.basename(obj)
                rows.append(k)
    names = demangle([k["name"] for k in rows])
    print("# kernel resources of sybil_amd/csrc/*.o (gfx950 code objects; tools/kernel_resources.py)")
    print("# waves/SIMD = min(8, 512 // roundup(vgpr + agpr, 8)); dynamic LDS (the cell tables) is the launch's, not listed here")
    print("%-14s %5s %5s %5s %7s %7s %5s %6s  %s" % ("file", "vgpr", "agpr", "sgpr", "lds", "scratch", "wg", "waves", "kernel"))
    for k in rows:
        nm = names.get(k["name"], k["name"])
        nm = re.sub(r"^void ", "", nm)
        nm = re.sub(r"\(.*$", "", nm)
        if want and not want.search(nm):
            continue
        regs = k["vgpr"] + k["agpr"]
        waves = min(8, 512 // max(8, (regs + 7) // 8 * 8))
        print("%-14s %5d %5d %5d %7d %7d %5d %6d  %s" % (k["file"].replace("kernels_", "k_")[:14], k["vgpr"], k["agpr"], k["sgpr"], k["lds"], k["scratch"], k["wg"],
                                                          waves, nm[:150]))
    print("# %d kernels" % len(rows))


if __name__ == "__main__":
    main()
