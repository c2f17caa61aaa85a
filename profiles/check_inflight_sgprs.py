#!/usr/bin/env python3
"""Static checks of compiler-invisible scalar memory operations (no GPU needed): between an asm-issued s_load_dwordx16 (streamed
ternary kernels) or s_atomic_add (work-counter takes of the fused kernels) and the s_waitcnt lgkmcnt(0) that retires it, no
instruction may read or write the destination scalar registers (a compiler copy or spill there would move a value that has not
landed).  Reads bitnetmcu_amd/_build/*.o; exit status 1 on a violation."""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def disasm(obj):
    with tempfile.TemporaryDirectory() as t:
        fat, co = os.path.join(t, "fat"), os.path.join(t, "k.co")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", obj], stderr=subprocess.DEVNULL)
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                               f"--input={fat}", f"--output={co}"])
        return subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], capture_output=True, text=True, check=True).stdout


def sregs(text):
    """Scalar registers an instruction names.  Packed-math instructions (v_pk_*) read, per lane, the half of a 64-bit source their
    op_sel / op_sel_hi bits select: a scalar pair with op_sel 0 and op_sel_hi 0 on that source is read in its LOW register only
    (the compiler broadcasts one scalar that way and leaves the pair's high half undefined - it may sit on a register that is
    otherwise in use), with both bits 1 in its high register only."""
    out = set()
    if text.startswith("v_pk_"):
        def bits(name, default):
            m = re.search(name + r":\[([01,]+)\]", text)
            return [int(v) for v in m.group(1).split(",")] if m else default
        ops = re.split(r",\s*", text.split(None, 1)[1].split(" op_sel")[0].split(" neg_")[0].split(" clamp")[0])
        lo_sel, hi_sel = bits("op_sel", [0, 0, 0]), bits("op_sel_hi", [1, 1, 1])
        for k, op in enumerate(ops[1:]):
            m = re.fullmatch(r"s\[(\d+):(\d+)\]", op.strip())
            if m and k < 3 and int(m.group(2)) == int(m.group(1)) + 1:
                sel = {lo_sel[k] if k < len(lo_sel) else 0, hi_sel[k] if k < len(hi_sel) else 1}
                out.update(int(m.group(1)) + half for half in sel)
            else:
                out |= sregs_plain(op)
        return out | sregs_plain(ops[0])
    return sregs_plain(text)


def sregs_plain(text):
    out = set()
    for a, b in re.findall(r"\bs\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bs(\d+)\b", text):
        out.add(int(a))
    return out


def check_takes():
    """work_take_issue / work_take_wait (bnm_device.hpp): between an s_atomic_add and the next s_waitcnt lgkmcnt(0) nothing may
    read or write the atomic's result register."""
    import glob
    bad, takes = [], 0
    for obj in sorted(glob.glob(os.path.join(REPO, "bitnetmcu_amd", "_build", "*.o"))):
        try:
            asm = disasm(obj)
        except subprocess.CalledProcessError:
            continue
        kernel, reg = None, None
        for line in asm.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                kernel, reg = m.group(1), None
                continue
            ins = line.split("//")[0].strip()
            if not ins:
                continue
            m = re.match(r"s_atomic_add s(\d+),", ins)
            if m:
                reg = int(m.group(1))
                takes += 1
                continue
            if ins.startswith("s_waitcnt") and "lgkmcnt(0)" in ins:
                reg = None
                continue
            if reg is not None and reg in sregs(ins):
                bad.append((os.path.basename(obj), (kernel or "")[:40], ins))
    print(f"{takes} scalar work-counter takes checked, {len(bad)} instructions touch a result register in flight")
    for b in bad[:20]:
        print("  ", b)
    return bad


def main():
    if check_takes():
        return 1
    asm = disasm(os.path.join(REPO, "bitnetmcu_amd", "_build", "bnm_ternary.o"))
    bad, kernel, inflight, loads = [], None, set(), 0
    for line in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            kernel, inflight = m.group(1), set()
            continue
        if "ternary_stream_kernel" not in (kernel or ""):
            continue
        ins = line.split("//")[0].strip()
        if not ins:
            continue
        if ins.startswith("s_load_dwordx16"):
            dst = re.match(r"s_load_dwordx16 s\[(\d+):(\d+)\]", ins)
            inflight |= set(range(int(dst.group(1)), int(dst.group(2)) + 1))
            loads += 1
            continue
        if ins.startswith("s_waitcnt") and "lgkmcnt(0)" in ins:
            inflight = set()
            continue
        if inflight and sregs(ins) & inflight:
            bad.append((kernel[:40], ins))
    print(f"{loads} asm-issued chunk loads checked, {len(bad)} instructions touch registers of a load in flight")
    for b in bad[:20]:
        print("  ", b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
