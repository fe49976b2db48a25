"""CPU: the shipped device code holds no inline-asm instruction inside the hazard window of a matrix instruction (ADVICE r05 'low' 1).

`split_pair` (csrc/cv_kernels.h) issues v_fma_mixlo / mixhi_f16 through inline asm, which the compiler's hazard recogniser does not
look into.  tools/check_asm_hazards.py disassembles the built library and proves, kernel by kernel, that no such instruction touches a
register of a recent v_mfma unless a compiler-padded instruction touched it first -- a property of the BUILD, checked on the build."""
from __future__ import annotations

import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))

import check_asm_hazards as chk  # noqa: E402

LIB = ROOT / "chessvision-3lc_amd" / "lib" / "libchessvision_hip.so"


def test_checker_flags_a_mix_instruction_that_overwrites_a_live_accumulator_and_accepts_a_settled_one():
    bad = """
0000000000001000 <kernel_a>:
	v_mfma_f32_16x16x32_f16 v[38:41], v[2:5], v[34:37], v[50:53]// 000000001000: D3D40026
	s_nop 1                                                    // 000000001008: BF800001
	v_fma_mixlo_f16 v50, v54, -1.0, v60 op_sel_hi:[1,0,0]      // 00000000100C: D3A10032
	v_fma_mixhi_f16 v61, v54, -1.0, v38 op_sel:[1,0,0] op_sel_hi:[1,0,0]// 000000001014: D3A20032
"""
    kernels, mixes, found = chk.check_disassembly(bad)
    assert (kernels, mixes) == (1, 2) and len(found) == 2
    assert "writes v[50]" in found[0] and "reads v[38]" in found[1]
    ok = """
0000000000002000 <kernel_b>:
	v_mfma_f32_16x16x32_f16 v[38:41], v[2:5], v[34:37], v[50:53]// 000000002000: D3D40026
	s_nop 2                                                    // 000000002008: BF800002
	v_add_u32_e32 v50, s12, v89                                // 00000000200C: 6864B20C
	v_fma_f32 v60, v38, v80, v26                               // 000000002010: D1CB0032
	v_cvt_pk_f16_f32 v54, v50, v60                             // 000000002018: D2670036
	v_fma_mixlo_f16 v50, v54, -1.0, v50 op_sel_hi:[1,0,0]      // 000000002020: D3A10032
	v_fma_mixhi_f16 v50, v54, -1.0, v38 op_sel:[1,0,0] op_sel_hi:[1,0,0]// 000000002028: D3A20032
"""
    assert chk.check_disassembly(ok) == (1, 2, [])


@pytest.mark.skipif(not LIB.exists() or not Path(chk.OBJDUMP).exists(), reason="library not built / no llvm-objdump")
def test_built_library_has_no_inline_asm_inside_a_matrix_instructions_hazard_window():
    kernels, mixes, found = chk.check_disassembly(chk.disassemble(LIB))
    assert kernels > 100 and mixes > 1000, (kernels, mixes)      # the scan really saw the conv kernels and their split epilogues
    assert found == [], found[:10]
