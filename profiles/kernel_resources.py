#!/usr/bin/env python3
"""Register / scratch / LDS budget of every gfx950 kernel in the built objects (no GPU needed).

  python profiles/kernel_resources.py [--spills-only] [--md]      reads bitnetmcu_amd/_build/*.o

Each object's .hip_fatbin section is an offload bundle; the gfx950 code object's AMDGPU metadata note holds, per kernel,
.vgpr_count / .sgpr_count / .vgpr_spill_count / .sgpr_spill_count / .private_segment_fixed_size (scratch bytes per lane) /
.group_segment_fixed_size (static LDS).  Exit status 1 if any kernel spills VGPRs or uses scratch (the product rule:
no spilling instantiation ships)."""
import argparse
import glob
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return out.strip().split("\n")


def kernels_of(obj, tmp):
    fat = os.path.join(tmp, "fat.bin")
    co = os.path.join(tmp, "k.co")
    r = subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", obj], capture_output=True)
    if r.returncode != 0 or not os.path.exists(fat):
        return []
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o",
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}"])
    notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
    os.remove(fat)
    os.remove(co)
    ks, cur = [], None
    for line in notes.splitlines():
        m = re.match(r"\s+- \.agpr_count:\s+(\d+)", line)
        if m:
            cur = {"agpr": int(m.group(1))}
            ks.append(cur)
            continue
        m = re.match(r"\s+\.(\w+):\s+(.*)", line)
        if m and cur is not None and m.group(1) in ("name", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count",
                                                    "private_segment_fixed_size", "group_segment_fixed_size",
                                                    "max_flat_workgroup_size"):
            v = m.group(2).strip()
            cur[m.group(1)] = v if m.group(1) == "name" else int(v)
    return [k for k in ks if "name" in k]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spills-only", action="store_true")
    ap.add_argument("--md", action="store_true", help="markdown table")
    ap.add_argument("--dir", default=os.path.join(REPO, "bitnetmcu_amd", "_build"))
    a = ap.parse_args()
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for obj in sorted(glob.glob(os.path.join(a.dir, "*.o"))):
            for k in kernels_of(obj, tmp):
                k["obj"] = os.path.basename(obj)
                rows.append(k)
    names = demangle([k["name"] for k in rows])
    bad = 0
    sep = " | " if a.md else "  "
    hdr = ["object", "kernel", "vgpr", "sgpr", "vgpr_spill", "sgpr_spill", "scratch B/lane", "static LDS", "max threads"]
    print(("| " if a.md else "") + sep.join(hdr) + (" |" if a.md else ""))
    if a.md:
        print("|" + "---|" * len(hdr))
    for k, nm in zip(rows, names):
        spill = k.get("vgpr_spill_count", 0) or k.get("private_segment_fixed_size", 0)
        bad += 1 if spill else 0
        if a.spills_only and not spill:
            continue
        nm = re.sub(r"^void ", "", nm)
        nm = re.sub(r"\(.*", "", nm)
        f = [k["obj"], f"`{nm}`" if a.md else nm, k.get("vgpr_count", 0), k.get("sgpr_count", 0), k.get("vgpr_spill_count", 0),
             k.get("sgpr_spill_count", 0), k.get("private_segment_fixed_size", 0), k.get("group_segment_fixed_size", 0),
             k.get("max_flat_workgroup_size", 0)]
        print(("| " if a.md else "") + sep.join(str(x) for x in f) + (" |" if a.md else ""))
    print(f"\n{len(rows)} kernels, {bad} with VGPR spills or scratch")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
