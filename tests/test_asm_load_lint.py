"""The build's lint of kernels that issue global loads from inline asm (focus_amd/build.py lint_hand_loads): the compiler
does not know such a load is in flight, so a spill or an AGPR park of its destination before the hand-counted wait copies
garbage (seen once: slot_bwd_defer_kernel<8, false, true>, v_accvgpr_write right after the global_load).  The build
refuses such kernels; this test pins the lint itself and re-checks the device assembly the last build left behind."""
import glob
import os

from focus_amd import build

CLEAN = """
_ZN1a6kernelEv:
\ts_load_dwordx2 s[0:1], s[4:5], 0x0
\t;;#ASMSTART
\tglobal_load_dwordx4 v[2:5], v[0:1], off offset:0
\t;;#ASMEND
\t;;#ASMSTART
\ts_waitcnt vmcnt(0)
\t;;#ASMEND
\tv_accvgpr_write_b32 a0, 0
\ts_endpgm
.Lfunc_end0:
\t.size\t_ZN1a6kernelEv, .Lfunc_end0-_ZN1a6kernelEv
; ScratchSize: 0
"""
COPY = CLEAN.replace("\tv_accvgpr_write_b32 a0, 0\n", "\tv_accvgpr_write_b32 a7, v5\n")
SPILL = CLEAN.replace("; ScratchSize: 0", "; ScratchSize: 20")
# the same copy in a kernel WITHOUT hand-issued loads is the compiler's own business
PLAIN = COPY.replace("\t;;#ASMSTART\n\tglobal_load_dwordx4 v[2:5], v[0:1], off offset:0\n\t;;#ASMEND\n",
                     "\tglobal_load_dwordx4 v[2:5], v[0:1], off offset:0\n")


def _lint(tmp_path, text):
    f = tmp_path / "k.s"
    f.write_text(text)
    return build.lint_hand_loads(str(f))


def test_lint_flags_copies_and_scratch_only_in_hand_load_kernels(tmp_path):
    assert _lint(tmp_path, CLEAN) == []
    assert _lint(tmp_path, COPY) == [("_ZN1a6kernelEv", "1 VGPR->AGPR copies")]
    assert _lint(tmp_path, SPILL) == [("_ZN1a6kernelEv", "scratch 20 B/lane")]
    assert _lint(tmp_path, PLAIN) == []


def test_the_built_kernels_pass_the_lint():
    """Every source with hand-issued loads leaves its device assembly next to its object (lib/obj is not shipped to the
    GPU box: there this only checks that the sources are still recognised)."""
    srcs = [s for s in glob.glob(os.path.join(build.CSRC, "*.hip")) if build._hand_loads(s)]
    assert {os.path.basename(s) for s in srcs} >= {"traj_time2.hip", "slot_attn.hip"}
    for s in srcs:
        asm = build._device_asm(s)
        if os.path.exists(asm):
            assert build.lint_hand_loads(asm) == [], asm
