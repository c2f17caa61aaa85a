#!/usr/bin/env python3
"""Static check of the built gfx950 objects: no vector instruction reads or writes a register of an MFMA result still in flight.

hipcc places the wait states between a v_mfma and the first VALU / memory instruction that touches its result registers - but it
does not look inside inline-asm statements, and its register allocator hands the DEAD registers of a result (rows nobody reads:
padding rows of a Toeplitz tile, classes beyond the model's) to other values at once.  An inline-asm instruction whose output
lands in such a register shortly behind the MFMA is overwritten by the late write-back: round 4's cnn_li_kernel lost a plane-2
operand dword that way, about one image in 50,000 and not reproducibly.  This script walks the disassembly of
every kernel: behind each v_mfma it counts wait states (an instruction = 1, s_nop N = N + 1) and reports any non-MFMA vector
instruction that names a register of the result before the required count has passed (v_mfma_i32_32x32x32_i8 on gfx950: 12, what
hipcc itself pads to; MFMA -> MFMA forwarding is interlocked and not checked).  Straight-line code only: the walk stops at
branches.

Round 6 added the second hazard hipcc cannot see through inline asm: the result of a TRANSCENDENTAL operation (v_rcp / v_rsq / v_sqrt /
v_exp / v_log / v_sin / v_cos) needs one wait state before a non-transcendental VALU instruction reads it.  The first version of the
one-kernel convolution front fed v_rcp_f32 results straight into a hand-written `v_pk_mul_f32 ... clamp` and read stale registers
(every feature wrong by up to 35 %).

usage: python profiles/check_mfma_hazards.py [object ...]      (default: every object under bitnetmcu_amd/_build)"""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
REQUIRED = {"v_mfma_i32_32x32x32_i8": 12, "v_mfma_f32_32x32x2_f32": 20}
TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_")
REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")


def disassemble(obj):
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "k.co")
        r = subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", obj], capture_output=True)
        if r.returncode:
            return None      # host-only object
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        f"--input={fat}", f"--output={co}", "--unbundle"], check=True)
        return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True, check=True).stdout


def kernels_of(text):
    out, cur = {}, None
    for line in text.splitlines():
        if line.endswith(">:") and "<" in line:
            cur = line[line.index("<") + 1:-2]
            out[cur] = []
        elif cur is not None and "\t" in line:
            ins = line.split("//")[0].strip()
            if ins:
                out[cur].append(ins)
    return out


def regs(text):
    s = set()
    for m in REG.finditer(text):
        if m.group(1):
            s.add((m.group(1), int(m.group(2))))
        else:
            s.update((m.group(3), k) for k in range(int(m.group(4)), int(m.group(5)) + 1))
    return s


def check(ins):
    bad = []
    for i, text in enumerate(ins):
        op = text.split()[0]
        need = REQUIRED.get(op)
        trans = need is None and op.startswith(TRANS)
        if trans:
            need = 1
        if need is None:
            continue
        dest = regs(text.split(None, 1)[1].split(",")[0])
        states = 0
        for k in range(i + 1, len(ins)):
            if states >= need:
                break
            t = ins[k]
            o = t.split()[0]
            if o == "s_nop":
                states += int(t.split()[1], 0) + 1
                continue
            if o.startswith(("s_branch", "s_cbranch", "s_endpgm", "s_setpc", "s_swappc")):
                break
            if trans and (not o.startswith("v_") or o.startswith(TRANS)):      # (only VALU readers; transcendental -> transcendental forwards)
                states += 1
                continue
            if not o.startswith(("v_mfma", "v_smfmac")) and not o.startswith("s_") and " " in t:
                hit = dest & regs(t.split(None, 1)[1])
                if hit:
                    bad.append((i, text, k, t, states))
                    break
            states += 1
    return bad


def main():
    objs = sys.argv[1:] or sorted(os.path.join(REPO, "bitnetmcu_amd", "_build", f)
                                  for f in os.listdir(os.path.join(REPO, "bitnetmcu_amd", "_build")) if f.endswith(".o"))
    n_k = n_m = n_bad = 0
    for obj in objs:
        text = disassemble(obj)
        if text is None:
            continue
        for name, ins in kernels_of(text).items():
            n_k += 1
            n_m += sum(1 for t in ins if t.split()[0] in REQUIRED or t.split()[0].startswith(TRANS))
            for i, mf, k, t, states in check(ins):
                n_bad += 1
                print(f"{os.path.basename(obj)} {name[:60]}: +{k - i} instructions, {states} wait states behind\n    {mf}\n    {t}")
    print(f"{n_k} kernels, {n_m} MFMAs and transcendental operations: {n_bad} instructions touch a result register in flight")
    return 1 if n_bad else 0


if __name__ == "__main__":
    sys.exit(main())
