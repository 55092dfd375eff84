"""CPU (hipcc cross-compiles): the attention kernels must not contain the packed-fp32 pattern that coincided with
silently wrong cross-attention outputs once (a tripwire, see tools/check_isa_hazards.py; elsewhere hits are notes)."""
import glob
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="hipcc not available")
def test_no_packed_fp32_self_overwrite():
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_isa_hazards", os.path.join(ROOT, "tools", "check_isa_hazards.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    files = sorted(glob.glob(os.path.join(ROOT, "gram_amd", "csrc", "*.hip")))
    assert len(files) >= 6
    assert mod.main(files) == 0


def test_scanner_recognises_the_pattern(tmp_path):
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_isa_hazards", os.path.join(ROOT, "tools", "check_isa_hazards.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    p = tmp_path / "k.s"
    p.write_text("_Zkern:\n\tv_pk_mul_f32 v[138:139], v[46:47], v[138:139] op_sel_hi:[1,0]\n"      # the observed bad one
                 "\tv_pk_mul_f32 v[108:109], v[108:109], v[140:141] op_sel_hi:[1,0]\n"              # in-place, harmless
                 "\tv_pk_mul_f32 v[8:9], v[0:1], v[8:9] op_sel_hi:[0,1]\n"                          # broadcast of v0, harmless
                 "\tv_pk_fma_f32 v[4:5], v[3:4], v[10:11], v[12:13]\n")                               # hi lane reads v4 == dst lo
    hits = mod.scan_asm(str(p))
    assert len(hits) == 2 and "v[138:139], v[46:47], v[138:139]" in hits[0][1] and "v[3:4]" in hits[1][1]
