#!/usr/bin/env python3
"""Static instruction-class breakdown of the CNN front end's item loop (no GPU needed): disassembles
bitnetmcu_amd/_build/bnm_cnn.o, takes the basic blocks of cnn_front_mfma_kernel<true, false> that form one work item
(image x 32-channel block: the block with the 7 conv1 MFMAs and the stage-3 block behind it) and the per-image block behind
them, and counts wave instructions per class.  An image of a 64-channel model = 2 items + 1 per-image block.
Prints a markdown table (committed as profiles/r03/cnn_instruction_classes.md)."""
import collections
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
KERNEL = "_Z21cnn_front_mfma_kernelILb1ELb0EEvPKamPKijjjjPajPiPjj"

CLASSES = [
    ("conv2 MACs: v_dot2_i32_i16 (2 MACs per lane)", lambda o, a: o.startswith("v_dot2")),
    ("stage 1 shift + int16 pack: v_ashrrev_i32_sdwa", lambda o, a: o == "v_ashrrev_i32_sdwa"),
    ("stage 1 ReLU on int16 pairs: v_pk_max_i16", lambda o, a: o.startswith("v_pk_max")),
    ("conv3 MACs: v_mad_i32_i24 / v_mul_i32_i24", lambda o, a: o.startswith(("v_mad_i32_i24", "v_mul_i32_i24"))),
    ("pool / ReLU maxima: v_max_i32 / v_max3_i32", lambda o, a: o.startswith(("v_max_i32", "v_max3_i32", "v_max_u32", "v_max3_u32"))),
    ("shifts after the pools, ReLUNorm shift / round / clip", lambda o, a: o.startswith(("v_ashrrev_i32", "v_lshrrev_b32", "v_min_", "v_add_u32", "v_add3", "v_lshl_or", "v_and_or", "v_or_b32", "v_or3", "v_lshlrev_b32"))),
    ("weight pairs / patch offsets: v_alignbit, v_bfe, v_perm, v_and", lambda o, a: o.startswith(("v_alignbit", "v_bfe", "v_perm", "v_and_b32"))),
    ("selects / moves: v_cndmask, v_mov, v_accvgpr", lambda o, a: o.startswith(("v_cndmask", "v_mov", "v_accvgpr"))),
    ("lane bookkeeping: v_mbcnt, v_readlane / readfirstlane, DPP steps, compares", lambda o, a: o.startswith(("v_mbcnt", "v_readlane", "v_readfirstlane", "v_cmp")) or "dpp" in o or "row_" in a or "quad_perm" in a),
]


def disasm():
    obj = os.path.join(REPO, "bitnetmcu_amd", "_build", "bnm_cnn.o")
    with tempfile.TemporaryDirectory() as t:
        fat, co = os.path.join(t, "fat"), os.path.join(t, "k.co")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", obj], stderr=subprocess.DEVNULL)
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                               f"--input={fat}", f"--output={co}"])
        return subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], capture_output=True, text=True, check=True).stdout


def blocks(text):
    lines = text.splitlines()
    start = [i for i, l in enumerate(lines) if re.match(r"^[0-9a-f]+ <" + re.escape(KERNEL) + ">:", l)][0]
    end = next((i for i in range(start + 1, len(lines)) if re.match(r"^[0-9a-f]+ <", lines[i])), len(lines))
    ins = []
    for l in lines[start + 1:end]:
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):", l)
        if m:
            ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
    addr = {a: i for i, (a, _, _) in enumerate(ins)}
    leaders = {0}
    for i, (a, op, args) in enumerate(ins):
        if op.startswith(("s_cbranch", "s_branch")):
            off = int(args.split()[0])
            off = off - 65536 if off >= 32768 else off
            t = a + 4 + 4 * off
            if t in addr:
                leaders.add(addr[t])
            leaders.add(i + 1)
    cut = sorted(x for x in leaders if x < len(ins))
    return [ins[s:(cut[k + 1] if k + 1 < len(cut) else len(ins))] for k, s in enumerate(cut)]


def classify(block):
    c = collections.Counter()
    for _, op, args in block:
        if op.startswith("v_mfma"):
            c["conv1 on the matrix cores: v_mfma_i32_32x32x32_i8"] += 1
        elif op.startswith("v_"):
            for name, pred in CLASSES:
                if pred(op, args):
                    c[name] += 1
                    break
            else:
                c["other VALU: " + op] += 1
        elif op.startswith("ds_"):
            c["LDS (partner exchange): " + op] += 1
        elif op.startswith(("global_", "buffer_")):
            c["vector memory: " + op.split("_dword")[0]] += 1
        elif op == "s_nop":
            c["s_nop wait states (cycles)"] += int(args.split()[0]) + 1
        elif op.startswith("s_"):
            c["scalar"] += 1
    return c


def main():
    bl = blocks(disasm())
    big = [b for b in bl if sum(o.startswith("v_") for _, o, _ in b) > 20]
    item = [b for b in big if any(o.startswith("v_mfma") for _, o, _ in b)]
    assert len(item) == 1, "expected ONE block with the conv1 MFMAs"
    k = bl.index(item[0])
    parts = [("item, first basic block (the 7 conv1 tiles, two thirds of conv2)", bl[k])]
    rest = [b for b in bl[k + 1:] if sum(o.startswith("v_") for _, o, _ in b) > 20]
    if rest:
        parts.append(("item, second basic block (rest of conv2, pools, conv3, partner exchange)", rest[0]))
    if len(rest) > 1:
        parts.append(("per image: ReLUNorm of the 4 C features, act bytes", rest[1]))
    tot = collections.Counter()
    per = []
    for name, b in parts:
        c = classify(b)
        per.append((name, c))
    mult = [2, 2, 1]
    print("# CNN front end: wave instructions per class (static, `cnn_front_mfma_kernel<true, false>`)\n")
    print("An image of the 64-channel model = 2 items (32 channels each; a lane = one channel x one band) + the per-image block.\n")
    keys = []
    for _, c in per:
        for kk in c:
            if kk not in keys:
                keys.append(kk)
    print("| class | " + " | ".join(n for n, _ in per) + " | per image (2 items + 1) |")
    print("|---|" + "---|" * (len(per) + 1))
    valu_total = 0
    for kk in keys:
        row = [c.get(kk, 0) for _, c in per]
        img = sum(r * m for r, m in zip(row, mult))
        if not kk.startswith(("scalar", "s_nop", "LDS", "vector memory", "conv1 on")):
            valu_total += img
        print(f"| {kk} | " + " | ".join(str(r) for r in row) + f" | {img} |")
    print(f"\nVALU (without the MFMAs) per image: **{valu_total}**  (counters: SQ_INSTS_VALU per image incl. 14 MFMAs)\n")


if __name__ == "__main__":
    sys.exit(main())
