"""The build's lint of kernels that issue global loads from inline asm (focus_amd/asm_lint.py, run by focus_amd/build.py
on the device assembly of every such source): the compiler does not know a hand-issued load is in flight, so a copy, spill,
AGPR park or any other read of its destination before the hand-counted wait uses whatever the register held before.  The
build refuses such kernels; this file pins the lint on small hand-written listings and re-checks the device assembly the
last build left behind."""
import glob
import os

import pytest

from focus_amd import asm_lint, build


def listing(body, scratch=0):
    return ("_ZN1a6kernelEv:\n" + "".join("\t%s\n" % l if not l.endswith(":") else l + "\n" for l in body)
            + ".Lfunc_end0:\n\t.size\t_ZN1a6kernelEv, .Lfunc_end0-_ZN1a6kernelEv\n; ScratchSize: %d\n" % scratch)


HAND = [";;#ASMSTART", "global_load_dwordx4 v[2:5], v[0:1], off offset:0", ";;#ASMEND"]
WAIT0 = [";;#ASMSTART", "s_waitcnt vmcnt(0)", ";;#ASMEND"]
END = ["s_endpgm"]


def _lint(tmp_path, body, scratch=0):
    f = tmp_path / "k.s"
    f.write_text(listing(body, scratch))
    return [p for _, p in asm_lint.lint_hand_loads(str(f))]


def test_a_register_is_the_compilers_again_after_the_counted_wait(tmp_path):
    body = HAND + WAIT0 + ["v_accvgpr_write_b32 a7, v5", "v_mov_b32_e32 v9, v2", "global_store_dword v[0:1], v9, off"] + END
    assert _lint(tmp_path, body) == []
    assert _lint(tmp_path, body, scratch=20) == ["scratch 20 B/lane"]


def test_copies_made_before_the_wait_are_followed_to_their_use(tmp_path):
    # the miscompile seen in slot_bwd_defer_kernel<8, false, true>: AGPR park right after the load, read back after the wait
    park = HAND + ["v_accvgpr_write_b32 a7, v5"] + WAIT0 + ["v_accvgpr_read_b32 v9, a7", "global_store_dword v[0:1], v9, off"] + END
    assert any("global_store_dword uses v9" in p for p in _lint(tmp_path, park))
    # plain v_mov / v_pk_mov copies count the same, through any chain of arithmetic
    mov = HAND + ["v_mov_b32_e32 v9, v3", "v_pk_mov_b32 v[10:11], v[4:5], v[4:5]"] + WAIT0 + [
        "v_add_f32_e32 v12, v9, v10", "ds_write_b32 v20, v12"] + END
    assert any("ds_write_b32 uses v12" in p for p in _lint(tmp_path, mov))
    # a spill of the destination, a store of it and its use as an address are effects themselves
    assert any("scratch_store_dword uses v2" in p for p in _lint(tmp_path, HAND + ["scratch_store_dword off, v2, off"] + WAIT0 + END))
    assert any("global_load_dword uses v2,v3" in p for p in _lint(tmp_path, HAND + ["global_load_dword v30, v[2:3], off"] + WAIT0 + END))
    # a copy whose result is never used (dead after the exit) is not reported
    assert _lint(tmp_path, HAND + ["v_mov_b32_e32 v9, v3"] + WAIT0 + END) == []
    # ... and a copy overwritten by a clean value before its use is clean again
    assert _lint(tmp_path, HAND + ["v_mov_b32_e32 v9, v3"] + WAIT0 + ["v_mov_b32_e32 v9, v40", "ds_write_b32 v20, v9"] + END) == []


def test_writes_under_a_load_in_flight_and_partial_waits(tmp_path):
    assert any("writes v4" in p for p in _lint(tmp_path, HAND + ["v_mov_b32_e32 v4, 0"] + WAIT0 + END))
    two = HAND + [";;#ASMSTART", "global_load_dwordx4 v[6:9], v[0:1], off offset:16", ";;#ASMEND",
                  ";;#ASMSTART", "s_waitcnt vmcnt(1)", ";;#ASMEND"]
    assert _lint(tmp_path, two + ["ds_write_b32 v20, v2"] + WAIT0 + END) == []            # the older load has landed
    assert any("uses v6" in p for p in _lint(tmp_path, two + ["ds_write_b32 v20, v6"] + WAIT0 + END))
    # younger loads the compiler issued itself only make the counted wait stricter; stores do not count
    mixed = HAND + ["global_load_dword v30, v[0:1], off", ";;#ASMSTART", "s_waitcnt vmcnt(1)", ";;#ASMEND", "ds_write_b32 v20, v2"] + WAIT0 + END
    assert _lint(tmp_path, mixed) == []
    store = HAND + ["global_store_dword v[0:1], v40, off", ";;#ASMSTART", "s_waitcnt vmcnt(1)", ";;#ASMEND", "ds_write_b32 v20, v2"] + WAIT0 + END
    assert any("uses v2" in p for p in _lint(tmp_path, store))


def test_loop_back_edges_and_exit_flags(tmp_path):
    # ring registers rotated with v_mov on the back edge while their load is in flight (time2_dx_lds_kernel once)
    loop = [".LBB0_1:"] + [";;#ASMSTART", "s_waitcnt vmcnt(0)", ";;#ASMEND", "ds_write_b128 v20, v[6:9]"] + HAND + [
        "v_mov_b32_e32 v6, v2", "v_mov_b32_e32 v7, v3", "v_mov_b32_e32 v8, v4", "v_mov_b32_e32 v9, v5",
        "s_cmp_lt_i32 s0, s1", "s_cbranch_scc1 .LBB0_1"] + WAIT0 + END
    assert any("ds_write_b128 uses v6,v7,v8,v9" in p for p in _lint(tmp_path, loop))
    # the same loop consuming the ring register itself after the wait is clean
    good = [".LBB0_1:"] + [";;#ASMSTART", "s_waitcnt vmcnt(0)", ";;#ASMEND", "ds_write_b128 v20, v[2:5]"] + HAND + [
        "s_cmp_lt_i32 s0, s1", "s_cbranch_scc1 .LBB0_1"] + WAIT0 + END
    assert _lint(tmp_path, good) == []
    # hipcc's exit flag: the early exit sets s[16:17] = -1 and leaves through the block that tests it; without following the
    # flag the walk would re-enter the loop with the load still in flight and report the (then unwaited) use
    flag = HAND + [".LBB0_1:", "s_mov_b64 s[16:17], 0", "s_cmp_lt_i32 s0, s1", "s_cbranch_scc1 .LBB0_3",
                   ";;#ASMSTART", "s_waitcnt vmcnt(0)", ";;#ASMEND", "ds_write_b128 v20, v[2:5]"] + HAND + [
        "s_branch .LBB0_2", ".LBB0_3:", "s_mov_b64 s[16:17], -1", ".LBB0_2:", "s_and_b64 vcc, exec, s[16:17]",
        "s_cbranch_vccnz .LBB0_4", "ds_write_b128 v20, v[10:13]", "s_branch .LBB0_1", ".LBB0_4:"] + WAIT0 + END
    assert _lint(tmp_path, flag) == []


def test_kernels_without_hand_loads_are_the_compilers_business(tmp_path):
    plain = ["global_load_dwordx4 v[2:5], v[0:1], off", "v_accvgpr_write_b32 a7, v5", "s_waitcnt vmcnt(0)",
             "global_store_dword v[0:1], v2, off"] + END
    assert _lint(tmp_path, plain, scratch=64) == []


def test_the_built_kernels_pass_the_lint():
    """Every source with hand-issued loads leaves its device assembly next to its object; a source whose assembly is gone
    cannot have been linted and the build refuses it (lib/obj does not travel to the GPU box: there only the sources are
    checked for still being recognised)."""
    srcs = [s for s in glob.glob(os.path.join(build.CSRC, "*.hip")) if build._hand_loads(s)]
    assert {os.path.basename(s) for s in srcs} >= {"traj_time2.hip", "slot_attn.hip"}
    if not os.path.isdir(build.OBJ):
        pytest.skip("no build directory here (GPU box: the library was built where the lint ran)")
    for s in srcs:
        asm = build._device_asm(s)
        assert os.path.exists(asm), "%s was built without its device assembly: not linted" % os.path.basename(s)
        assert build.lint_hand_loads(asm) == [], asm


def test_build_refuses_a_hand_load_source_without_its_assembly(tmp_path, monkeypatch):
    src = tmp_path / "k.hip"
    src.write_text('__global__ void k(int* p) { int v; asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p)); }\n')
    monkeypatch.setattr(build, "OBJ", str(tmp_path))
    (tmp_path / "k.o").write_bytes(b"")                                   # an up-to-date object, no .s beside it
    os.utime(str(src), (0, 0))
    with pytest.raises(RuntimeError, match="device assembly .* is missing"):
        build._compile(str(src), [])
    assert not (tmp_path / "k.o").exists()
