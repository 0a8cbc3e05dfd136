"""tools/isa_scan.py: the build-time lint for the s_cselect-on-stale-SCC mis-lowering (scanner logic only, no hipcc)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
import isa_scan  # noqa: E402

BAD = """
.LBB0_1:
	s_waitcnt lgkmcnt(0)
	v_cmp_eq_u32_e32 vcc, 0, v15
	s_cselect_b32 s8, 4, 0
	v_mov_b32_e32 v15, s8
"""
GOOD_INTERLEAVED = """
.LBB0_2:
	s_cmp_eq_u32 s23, 0
	v_cmp_ne_u32_e32 vcc, 0, v37
	s_cselect_b64 s[6:7], -1, 0
"""
GOOD_CARRY = """
.LBB0_3:
	s_add_u32 s4, s4, s44
	v_cmp_lt_u64_e32 vcc, s[6:7], v[2:3]
	s_addc_u32 s5, s5, s44
"""
BAD_AFTER_BRANCH = """
.LBB0_4:
	s_add_i32 s0, s26, s0
	s_cbranch_vccnz .LBB0_9
; %bb.5:
	v_cmp_eq_u32_e32 vcc, 0, v15
	s_cselect_b32 s0, 4, 0
"""


def test_flags_vcc_compare_feeding_scc_reader():
    assert len(isa_scan.scan(BAD)) == 1
    assert len(isa_scan.scan(BAD_AFTER_BRANCH)) == 1


def test_accepts_scheduled_interleavings():
    assert isa_scan.scan(GOOD_INTERLEAVED) == []
    assert isa_scan.scan(GOOD_CARRY) == []
