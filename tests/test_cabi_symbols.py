"""CPU: libgram_hip.so loads and exports every symbol include/gram_hip.h declares (no compute calls)."""
import ctypes
import os
import re

from gram_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "gram_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gram_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    names = _declared()
    assert len(names) >= 20
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/gram_hip.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature in gram_amd/_lib.py"
    assert set(_lib.SIGNATURES) == set(names)


def test_abi_version_and_struct_sizes():
    lib = _lib.load()
    assert lib.gram_abi_version() == _lib.ABI_VERSION == 7
    # field layout sanity (pointer + int32 packing as in the C header)
    assert ctypes.sizeof(_lib.KVBank) == 48
    assert ctypes.sizeof(_lib.Trie) == 40
    assert ctypes.sizeof(_lib.BeamState) == 24 + 12 * 8 + 8 + 8 + 8  # (+ cand_logits, its user count (padded), its stride)
    assert ctypes.sizeof(_lib.ModelDesc) == 48 + 7 * 8 + 15 * 8 + 8 + 16 + 8
    assert ctypes.sizeof(_lib.Split) == 32
    assert lib.gram_piece_format() in (0, 1)


def test_argument_errors_without_gpu():
    """Shape validation happens on the host before any launch: callable without a GPU."""
    lib = _lib.load()
    assert lib.gram_gemm_bf16(None, None, None, 0, 128, 64, 64, 128, 0, None, None) == _lib.E_ARG
    assert lib.gram_gemm_bf16(None, None, None, 16, 100, 64, 64, 100, 0, None, None) == _lib.E_ARG
    assert lib.gram_enc_self_attn(None, None, None, None, 1, 130, 2, None) == _lib.E_ARG
    assert lib.gram_cross_attn_decode(None, None, None, None, None, 1, 65, 2, 64, None) == _lib.E_ARG
    assert lib.gram_model_create(None) is None
    assert lib.gram_workspace_bytes(None, 1, 1, 32, 1, 4) == _lib.E_ARG
    # 16-column ("quarter") sum-of-squares partials are the streaming small-M GEMM's layout only: refused for more rows than it takes
    # (the pointers are never dereferenced: the refusal comes before any launch)
    fake = ctypes.c_void_p(0x1000)
    big = lib.gram_gemm_stream_max_m() + 1
    assert big > 1
    nf = _lib.NormFusion(None, None, 0x1000, 12, 768, 1e-6, 1)
    assert lib.gram_gemm_bf16_ex(fake, fake, fake, big, 768, 768, 768, 768, _lib.EPI_BF16, None, ctypes.byref(nf), None) == _lib.E_ARG


def test_ctypes_structs_match_the_header(tmp_path):
    """Every struct of include/gram_hip.h against its ctypes mirror in gram_amd/_lib.py: total size and the offset of every field,
    from a C program compiled against the real header (gcc; no GPU)."""
    import subprocess
    pairs = {"gram_kv_bank_t": _lib.KVBank, "gram_norm_fusion_t": _lib.NormFusion, "gram_split_t": _lib.Split, "gram_trie_t": _lib.Trie,
             "gram_beam_state_t": _lib.BeamState, "gram_live_rows_t": _lib.LiveRows, "gram_model_desc_t": _lib.ModelDesc,
             "gram_compaction_t": _lib.Compaction}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "gram_hip.h"', "int main(void) {"]
    for cname, st in pairs.items():
        lines.append(f'  printf("{cname} size %zu\\n", sizeof({cname}));')
        for fname, _ in st._fields_:
            lines.append(f'  printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    got = {tuple(ln.split()[:2]): int(ln.split()[2]) for ln in out if ln.strip()}
    for cname, st in pairs.items():
        assert got[(cname, "size")] == ctypes.sizeof(st), cname
        for fname, _ in st._fields_:
            assert got[(cname, fname)] == getattr(st, fname).offset, (cname, fname)


def test_f16_aliases_refuse_the_bfloat16_build():
    """Every `_bf16` entry point has an `_f16` alias (gram_hip.h) so that a binding by name cannot pass the wrong 16-bit type: in the
    bfloat16 A/B build the aliases return GRAM_E_ARG before anything is launched (no GPU needed to see that)."""
    import ctypes as C
    import os

    from gram_amd import _lib
    _lib.load()  # (torch's HIP runtime first, as for the product library)
    path = os.path.join(os.path.dirname(_lib.LIB_PATH), "libgram_hip_bf16.so")
    if not os.path.exists(path):
        import pytest
        pytest.skip("libgram_hip_bf16.so not built (make -C gram_amd/csrc PIECE=bf16)")
    lib = C.CDLL(path)
    assert lib.gram_piece_format() == 0
    vp = C.c_void_p
    lib.gram_gemm_f16.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
    assert lib.gram_gemm_f16(None, None, None, 128, 128, 64, 64, 128, 0, None, None) == _lib.E_ARG
    lib.gram_rmsnorm_f16.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_float, C.c_float, vp, C.c_int, C.c_int, vp]
    assert lib.gram_rmsnorm_f16(None, None, None, 4, 128, 1e-6, 1.0, None, 1, 1, None) == _lib.E_ARG
    assert _lib.load().gram_piece_format() == 1 and all(hasattr(_lib.load(), n) for n in ("gram_gemm_f16_split", "gram_rmsnorm_f16_split"))
