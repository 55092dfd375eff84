"""CPU (hipcc cross-compiles): no kernel may read an MFMA result with a vector-ALU / memory / LDS instruction before the
wait states the hardware needs have passed -- the compiler pads such pairs inside a basic block but left them open across
block boundaries in one round-1 build of the cross-attention kernel, which then returned wrong values
(tools/check_isa_hazards.py has the story; tests/golden/isa_mfma_valu_hazard_r01.s is that build's ISA)."""
import glob
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _scanner():
    spec = importlib.util.spec_from_file_location("check_isa_hazards", os.path.join(ROOT, "tools", "check_isa_hazards.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="hipcc not available")
def test_no_mfma_result_read_too_early():
    files = sorted(glob.glob(os.path.join(ROOT, "gram_amd", "csrc", "*.hip")))
    assert len(files) >= 6
    assert _scanner().main(files) == 0


def test_scanner_flags_the_round1_build(golden_dir):
    """The real thing: the last MFMAs of a 32-key step are copied out by v_mov_b64 two to seven wait states later, through
    branches -- including the v[102:103] <- o[3] copy (dims 48-63) four wait states after its MFMA."""
    hits = _scanner().scan_asm(os.path.join(golden_dir, "isa_mfma_valu_hazard_r01.s"))
    assert len(hits) == 6
    assert all("v_mov_b64" in h[4] and h[5] < h[6] for h in hits)
    assert any("v[102:105], v[42:45]" in h[2] and "v[46:47], v[102:103]" in h[4] and h[5] == 4 for h in hits)


def test_scanner_control_flow_and_wait_states(tmp_path):
    p = tmp_path / "k.s"
    p.write_text(
        "_Zkern:\n"
        "\tv_mfma_f32_16x16x32_bf16 v[0:3], v[8:11], v[12:15], v[0:3]\n"
        "\ts_nop 6\n"
        "\tv_add_f32_e32 v20, v0, v1\n"                 # 7 wait states: too early
        "\tv_mfma_f32_16x16x32_bf16 v[4:7], v[8:11], v[12:15], v[4:7]\n"
        "\ts_cbranch_vccnz .LBB0_2\n"
        "\ts_nop 7\n"
        "\tv_mov_b32_e32 v21, v4\n"                     # fall-through path: 9 wait states, fine
        "\ts_endpgm\n"
        ".LBB0_2:\n"
        "\ts_mov_b32 s0, 0\n"
        "\tglobal_store_dword v[30:31], v5, off\n"     # taken path: 2 wait states
        "\tv_mfma_f32_16x16x32_bf16 v[16:19], v[8:11], v[12:15], v[4:7]\n"   # matrix op reading the result: not this check's business
        "\ts_endpgm\n")
    hits = _scanner().scan_asm(str(p))
    assert [(h[4].split()[0], h[5]) for h in hits] == [("v_add_f32_e32", 7), ("global_store_dword", 2)]


def test_scanner_flags_a_valu_written_sgpr_read_by_a_load(tmp_path):
    """gfx9: a vector-memory instruction may not read an SGPR within 5 wait states of the VALU instruction that wrote it (hipcc pads its
    own loads; a load inside an asm statement it cannot see).  The round-4 pattern, its padded form, a scalar-ALU rewrite in between and a
    hazard that arrives through a branch."""
    p = tmp_path / "k.s"
    p.write_text(
        "_Zkern:\n"
        "\tv_readlane_b32 s4, v255, 2\n"
        "\tv_lshlrev_b32_e32 v2, 2, v2\n"
        "\tv_readlane_b32 s5, v255, 3\n"
        "\ts_mov_b32 m0, s70\n"
        "\ts_nop 0\n"
        "\tglobal_load_lds_dword v2, s[4:5]\n"          # s5 two wait states old, s4 four: both too young
        "\tv_readlane_b32 s9, v255, 5\n"
        "\ts_mov_b32 m0, s70\n"
        "\ts_nop 3\n"
        "\tglobal_load_lds_dwordx4 v3, s[8:9]\n"        # 5 wait states: fine
        "\tv_readfirstlane_b32 s12, v7\n"
        "\ts_mov_b32 s12, s20\n"
        "\tglobal_load_dword v9, v8, s[12:13]\n"        # rewritten by the scalar ALU (interlocked): fine
        "\tv_readfirstlane_b32 s16, v7\n"
        "\ts_cbranch_scc1 .LBB0_2\n"
        "\ts_nop 7\n"
        ".LBB0_2:\n"
        "\tbuffer_load_dword v1, v2, s[16:19], 0 offen\n"  # one wait state on the taken path
        "\ts_endpgm\n")
    hits = _scanner().scan_asm_sgpr_vmem(str(p))
    assert [(h[2].split()[0], h[2].split()[1], h[4].split()[0], h[5]) for h in hits] == [
        ("v_readlane_b32", "s4,", "global_load_lds_dword", 4), ("v_readlane_b32", "s5,", "global_load_lds_dword", 2),
        ("v_readfirstlane_b32", "s16,", "buffer_load_dword", 1)]
