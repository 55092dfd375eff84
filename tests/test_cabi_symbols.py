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
    assert lib.gram_abi_version() == _lib.ABI_VERSION == 4
    # field layout sanity (pointer + int32 packing as in the C header)
    assert ctypes.sizeof(_lib.KVBank) == 48
    assert ctypes.sizeof(_lib.Trie) == 40
    assert ctypes.sizeof(_lib.BeamState) == 24 + 12 * 8
    assert ctypes.sizeof(_lib.ModelDesc) == 48 + 7 * 8 + 15 * 8 + 8 + 16
    assert ctypes.sizeof(_lib.Split) == 40


def test_argument_errors_without_gpu():
    """Shape validation happens on the host before any launch: callable without a GPU."""
    lib = _lib.load()
    assert lib.gram_gemm_bf16(None, None, None, 0, 128, 64, 64, 128, 0, None, None) == _lib.E_ARG
    assert lib.gram_gemm_bf16(None, None, None, 16, 100, 64, 64, 100, 0, None, None) == _lib.E_ARG
    assert lib.gram_enc_self_attn(None, None, None, None, 1, 130, 2, None) == _lib.E_ARG
    assert lib.gram_cross_attn_decode(None, None, None, None, None, 1, 65, 2, 64, None) == _lib.E_ARG
    assert lib.gram_model_create(None) is None
    assert lib.gram_workspace_bytes(None, 1, 1, 32, 1, 4) == _lib.E_ARG
