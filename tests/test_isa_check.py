"""CPU: the build-time assembly audit (stein_amd/csrc/isa_check.py) on synthetic gfx950 listings, one per rule, and that
__graft_entry__.build() runs it for the two kernels whose loads are hand-counted inline asm."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stein_amd", "csrc"))
import isa_check  # noqa: E402

NAME = "_Z10k_phi_x3fsILi1EEvv"


def listing(body, spills=0, scratch=0, name=NAME):
    return """\t.text
\t.globl\t%(n)s
%(n)s:
%(b)s
\ts_endpgm
.Lfunc_end0:
\t.size\t%(n)s, .Lfunc_end0-%(n)s
amdhsa.kernels:
  - .args: []
    .name:           %(n)s
    .private_segment_fixed_size: %(p)d
    .vgpr_count:     64
    .vgpr_spill_count: %(s)d
""" % {"n": name, "b": body, "s": spills, "p": scratch}


CLEAN = """\tv_add_u32_e32 v1, v2, v3
\t;;#ASMSTART
\ts_nop 4
\tglobal_load_dwordx4 v[10:13], v1, s[4:5]
\t;;#ASMEND
\tv_add_u32_e32 v20, v21, v22
\t;;#ASMSTART
\ts_waitcnt vmcnt(0)
\t;;#ASMEND
\tv_add_f32_e32 v30, v10, v11"""


def test_a_clean_listing_passes():
    assert isa_check.check_text(listing(CLEAN)) == []


def test_rule_1_spills_and_scratch_are_refused():
    probs = isa_check.check_text(listing(CLEAN, spills=3))
    assert len(probs) == 1 and "vgpr_spill_count = 3" in probs[0]
    probs = isa_check.check_text(listing(CLEAN, scratch=16))
    assert len(probs) == 1 and "private_segment_fixed_size = 16" in probs[0]


def test_rule_2_valu_written_sgpr_read_by_asm_vmem_needs_wait_states():
    body = """\tv_readlane_b32 s4, v40, 3
\tv_readlane_b32 s5, v40, 4
\t;;#ASMSTART
\tglobal_load_dwordx4 v[10:13], v1, s[4:5]
\t;;#ASMEND
\t;;#ASMSTART
\ts_waitcnt vmcnt(0)
\t;;#ASMEND"""
    probs = isa_check.check_text(listing(body))
    assert probs and all("wait state" in p for p in probs)
    # the statement's own leading s_nop 4 supplies the five wait states
    assert isa_check.check_text(listing(body.replace("\tglobal_load", "\ts_nop 4\n\tglobal_load", 1))) == []
    # so do five unrelated instructions between the write and the statement
    pad = "\n".join("\tv_mov_b32_e32 v%d, 0" % (50 + k) for k in range(5))
    assert isa_check.check_text(listing(body.replace("\t;;#ASMSTART\n\tglobal_load", pad + "\n\t;;#ASMSTART\n\tglobal_load", 1))) == []
    # an SGPR pair the load does not read is nobody's hazard
    assert isa_check.check_text(listing(body.replace("s[4:5]", "s[8:9]"))) == []


def test_rule_3_a_destination_touched_before_the_wait_is_refused():
    moved = CLEAN.replace("\tv_add_u32_e32 v20, v21, v22", "\tv_mov_b32_e32 v20, v11")
    probs = isa_check.check_text(listing(moved))
    assert len(probs) == 1 and "in flight" in probs[0] and "[11]" in probs[0]
    # overwriting one is as bad as reading one
    probs = isa_check.check_text(listing(CLEAN.replace("\tv_add_u32_e32 v20, v21, v22", "\tv_mov_b32_e32 v13, 0")))
    assert len(probs) == 1 and "[13]" in probs[0]


BRANCHY = """\t;;#ASMSTART
\ts_nop 4
\tglobal_load_dwordx4 v[10:13], v1, s[4:5]
\t;;#ASMEND
\ts_cbranch_scc1 .LBB0_2
\t;;#ASMSTART
\ts_waitcnt vmcnt(0)
\t;;#ASMEND
\ts_branch .LBB0_3
.LBB0_2:
\tv_add_f32_e32 v30, v10, v10
.LBB0_3:
\t;;#ASMSTART
\ts_waitcnt vmcnt(0)
\t;;#ASMEND"""


def test_rule_3_follows_every_path_only_for_the_kernels_written_for_it():
    # straight-line strength (k_phi_x3fs): the scan stops at the branch
    assert isa_check.check_text(listing(BRANCHY)) == []
    # all-paths strength (k_distance_panel): the taken side of the branch uses v10 without a wait
    name = "_Z16k_distance_panelILb1ELi2EEvv"
    probs = isa_check.check_text(listing(BRANCHY, name=name))
    assert len(probs) == 1 and "[10]" in probs[0]
    # with the wait in front of the use on that side as well, every path is covered
    fixed = BRANCHY.replace(".LBB0_2:\n", ".LBB0_2:\n\t;;#ASMSTART\n\ts_waitcnt vmcnt(0)\n\t;;#ASMEND\n")
    assert isa_check.check_text(listing(fixed, name=name)) == []


def test_a_listing_without_the_kernel_is_a_failure_not_a_pass():
    probs = isa_check.check_text(listing(CLEAN, name="_Z7k_otherv"))
    assert len(probs) == 1 and "no kernel" in probs[0]


def test_build_audits_the_hand_counted_kernels():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    assert set(ge.ISA_CHECKED) == {"stein_x3.hip", "stein_dpanel.hip"}
    # every source with inline-asm vector-memory instructions is on the list
    csrc = os.path.join(ROOT, "stein_amd", "csrc")
    for fn in sorted(os.listdir(csrc)):
        if fn.endswith(".hip"):
            with open(os.path.join(csrc, fn)) as f:
                text = f.read()
            uses = "stream_load16(" in text or "global_load_dword" in text
            assert uses == (fn in ge.ISA_CHECKED), fn
    # and the listings the last build() left behind pass (they exist wherever build() compiled those sources)
    for base, prefixes in ge.ISA_CHECKED.items():
        asm = os.path.join(ge.OBJDIR, base[:-4] + "-hip-amdgcn-amd-amdhsa-gfx950.s")
        if os.path.exists(asm):
            assert isa_check.check_file(asm, prefixes) == []
