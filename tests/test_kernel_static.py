"""Static checks of the built gfx950 code objects (no GPU): no kernel spills, the streamed ternary kernels' compiler-invisible
scalar loads are never touched while in flight (profiles/check_inflight_sgprs.py), and no vector instruction touches a register of
an MFMA result in flight (profiles/check_mfma_hazards.py)."""
import os
import subprocess
import sys

import pytest

import util

BUILD = os.path.join(util.REPO, "bitnetmcu_amd", "_build")
pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(BUILD, "bnm_ternary.o")), reason="objects not built")


def test_no_kernel_spills_or_scratch():
    r = subprocess.run([sys.executable, os.path.join(util.REPO, "profiles", "kernel_resources.py"), "--spills-only"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "0 with VGPR spills or scratch" in r.stdout


def test_ternary_chunk_loads_untouched_in_flight():
    r = subprocess.run([sys.executable, os.path.join(util.REPO, "profiles", "check_inflight_sgprs.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    assert " 0 instructions touch registers of a load in flight" in r.stdout
    assert " 0 instructions touch a result register in flight" in r.stdout and " 0 scalar work-counter takes" not in r.stdout


def test_inflight_checker_reads_packed_math_operand_halves():
    """Packed fp32 instructions read the half of a scalar pair their op_sel bits select (the whole-model QAT kernel broadcasts one
    scalar as `s[94:95] op_sel_hi 0`, the pair's undefined high half lying on the take's result register): the checker counts
    exactly the registers read."""
    sys.path.insert(0, os.path.join(util.REPO, "profiles"))
    import check_inflight_sgprs as C
    assert C.sregs("v_pk_fma_f32 v[58:59], v[110:111], s[94:95], v[142:143] op_sel_hi:[1,0,1]") == {94}
    assert C.sregs("v_pk_fma_f32 v[58:59], v[110:111], s[94:95], v[142:143]") == {94, 95}
    assert C.sregs("v_pk_fma_f32 v[58:59], v[110:111], s[94:95], v[142:143] op_sel:[0,1,0] op_sel_hi:[1,1,1]") == {95}
    assert C.sregs("v_pk_mul_f32 v[6:7], s[94:95], v[58:59] op_sel_hi:[0,1]") == {94}
    assert C.sregs("v_mov_b64_e32 v[2:3], s[94:95]") == {94, 95} and C.sregs("s_add_i32 s0, s95, s0") == {0, 95}


def test_no_instruction_touches_an_mfma_result_in_flight():
    """Inline-asm outputs allocated to the dead rows of an MFMA result are overwritten by its late write-back (hipcc pads nothing
    in front of inline asm): round 4's lane = image CNN kernel lost operand bytes that way, one image in 50,000."""
    sys.path.insert(0, os.path.join(util.REPO, "profiles"))
    import check_mfma_hazards as H
    # the checker sees the pattern that was the bug (the asm write 4 wait states behind the MFMA), not the padded one
    mf = "v_mfma_i32_32x32x32_i8 v[14:29], v[76:79], v[84:87], v[14:29]"
    wr = "v_lshrrev_b32_sdwa v28, v110, v117 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD"
    assert len(H.check([mf, "v_mov_b32_e32 v78, v35", "v_mov_b32_e32 v79, v35", wr])) == 1
    assert H.check([mf, "v_mov_b32_e32 v78, v35", "s_nop 10", wr]) == []
    assert H.check([mf, "v_mfma_i32_32x32x32_i8 v[14:29], v[0:3], v[4:7], v[14:29]", "s_nop 11", "v_lshlrev_b32_e32 v79, 8, v14"]) == []
    assert len(H.check([mf, "s_nop 9", "v_lshlrev_b32_e32 v79, 8, v14"])) == 1
    # ... and round 6's: a transcendental result read by the very next VALU instruction (an inline-asm one: hipcc pads its own)
    rcp = "v_rcp_f32_e32 v197, v11"
    assert len(H.check([rcp, "v_pk_mul_f32 v[10:11], v[60:61], v[196:197] clamp"])) == 1
    assert H.check([rcp, "s_nop 0", "v_pk_mul_f32 v[10:11], v[60:61], v[196:197] clamp"]) == []
    assert H.check([rcp, "v_mov_b32_e32 v1, v2", "v_pk_mul_f32 v[10:11], v[60:61], v[196:197] clamp"]) == []
    assert H.check([rcp, "v_sqrt_f32_e32 v3, v197"]) == []
    r = subprocess.run([sys.executable, os.path.join(util.REPO, "profiles", "check_mfma_hazards.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    assert " 0 instructions touch a result register in flight" in r.stdout


def _disassemble(obj):
    import tempfile
    llvm = "/opt/rocm/lib/llvm/bin"
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "k.co")
        subprocess.run([os.path.join(llvm, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", obj], check=True)
        subprocess.run([os.path.join(llvm, "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        f"--input={fat}", f"--output={co}", "--unbundle"], check=True)
        return subprocess.run([os.path.join(llvm, "llvm-objdump"), "-d", co], capture_output=True, text=True, check=True).stdout


def test_dual_kernel_stores_are_nontemporal_and_the_first_one_is_masked():
    """The headline kernel's class-id store costs 3-11 % of its time as a plain store; and its first, empty issue must be
    masked off, not a placeholder write (the latency path of bnm_infer_host takes a word's first change as the result)."""
    text = _disassemble(os.path.join(BUILD, "bnm_fused_fc.o"))
    kernels, cur = {}, None
    for line in text.splitlines():
        if line.endswith(">:") and "<" in line:
            cur = line[line.index("<") + 1:-2]
            kernels[cur] = []
        elif cur is not None and "\t" in line:
            kernels[cur].append(line.split("//")[0].strip())
    dual = {k: v for k, v in kernels.items() if k.startswith("_Z20fused_fc_dual_kernel")}
    assert len(dual) >= 12, list(kernels)[:5]
    for name, ins in dual.items():
        stores = [i for i in ins if i.startswith("global_store")]
        masked = [k for k, i in enumerate(ins) if i.startswith("s_and_saveexec_b64") and k + 2 < len(ins)
                  and ins[k + 1].startswith("global_store_dword ") and ins[k + 1].rstrip().endswith(" nt")
                  and ins[k + 2].startswith("s_mov_b64 exec")]
        assert len(masked) == 1, (name, len(masked))
        # the loop's masked store + the store of the last pair after the loop; the whole-line logits stores (16 bytes per lane).
        # (plain stores remain in the piecewise logits fallback for more than 16 classes)
        assert sum(1 for i in stores if i.startswith("global_store_dword ") and i.rstrip().endswith(" nt")) >= 2, name
        assert any(i.startswith("global_store_dwordx4") and i.rstrip().endswith(" nt") for i in stores), name
