"""The hand-placed MFMA streams of csrc/prl_lmhead_core.h are `asm volatile` statements, invisible to hipcc's hazard recogniser.
`scripts/check_mfma_hazards.py` compiles the two translation units that instantiate them to gfx950 assembly and verifies that no VALU write of an MFMA operand sits
directly in front of an asm MFMA and that nothing touches an asm MFMA's result directly behind it - the two hazards
`mfma_pin_acc` / `mfma_settle` exist for.  (Found on hardware in round 4: hipcc had sunk the zero fill of one accumulator tile
right in front of the first MFMA into it; one register of the tile kept its stale contents.)"""

import importlib.util
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _checker():
    spec = importlib.util.spec_from_file_location("check_mfma_hazards", ROOT / "scripts" / "check_mfma_hazards.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_no_valu_write_in_front_of_and_no_use_behind_an_asm_mfma():
    chk = _checker()
    text = chk.compile_to_asm()
    assert text.count(";;#ASMSTART") > 300  # the streams are there (4 hand-placed cores x 32-48 MFMAs x their instantiations)
    problems = chk.scan(text)
    assert not problems, "\n".join(problems[:10])


def test_the_checker_sees_the_hazard_it_was_written_for():
    chk = _checker()
    bad = """
_ZN4testE:
	v_mov_b64_e32 v[56:57], v[22:23]
	;;#ASMSTART
	v_mfma_f32_32x32x16_bf16 v[56:71], v[158:161], v[12:15], v[56:71]
	;;#ASMEND
	v_mul_f32_e32 v0, v56, v1
"""
    problems = chk.scan(bad)
    assert len(problems) == 2 and "VALU write" in problems[0] and "touches" in problems[1]
    ok = """
_ZN4testE:
	v_mov_b64_e32 v[56:57], v[22:23]
	s_nop 7
	;;#ASMSTART
	v_mfma_f32_32x32x16_bf16 v[56:71], v[158:161], v[12:15], v[56:71]
	;;#ASMEND
	;;#ASMSTART
	s_nop 15
	s_nop 15
	;;#ASMEND
	v_mul_f32_e32 v0, v56, v1
"""
    assert chk.scan(ok) == []
