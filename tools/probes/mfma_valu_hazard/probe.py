#!/usr/bin/env python3
"""Hardware probe for the round-1 wrong-result build of the cross-attention kernel (VERDICT r1 "weak" #6).

tests/golden/isa_mfma_valu_hazard_r01.s is the ISA of that build.  tools/check_isa_hazards.py finds MFMA results read by
v_mov_b64 two to seven wait states after the MFMA issued, across basic-block boundaries (the compiler keeps eight inside
a block).  This probe settles whether that is the cause: it assembles the fixture twice -- as is, and with ``s_nop 7``
inserted in front of the accumulator copies (nothing else changes: same registers, same packed multiplies, same
schedule) -- loads both code objects with hipModuleLoad and runs the shapes that failed (K <= 16, more than one 32-key
step per wave) against a torch fp32 reference, reporting the error in dims 0-47 and 48-63 separately.

Run on the GPU box:  python tools/probes/mfma_valu_hazard/probe.py   (writes gpurun_out/mfma_valu_hazard_probe.json)
"""
import ctypes as C
import json
import os
import re
import subprocess
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
FIXTURE = os.path.join(ROOT, "tests", "golden", "isa_mfma_valu_hazard_r01.s")
KERNEL = b"_ZN12_GLOBAL__N_117cross_attn_kernelILi1EEEvPKDF16bS2_S2_PKhPDF16biii"
LLVM = "/opt/rocm/lib/llvm/bin"


def patched_source(text):
    """s_nop 7 in front of every run of accumulator copies (v_mov_b64) that opens a basic block."""
    out, lines = [], text.split("\n")
    for i, line in enumerate(lines):
        out.append(line)
        if re.match(r"^\.LBB\d+_\d+:", line):
            j = i + 1
            while j < len(lines) and (not lines[j].strip() or lines[j].strip().startswith(";")):
                j += 1
            if j < len(lines) and lines[j].strip().startswith("v_mov_b64_e32"):
                out.append("\ts_nop 7")
    return "\n".join(out)


def assemble(src_text, tmp, name):
    s, o, co = (os.path.join(tmp, name + ext) for ext in (".s", ".o", ".co"))
    open(s, "w").write(src_text)
    subprocess.run([f"{LLVM}/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s, "-o", o], check=True)
    subprocess.run([f"{LLVM}/ld.lld", "-shared", o, "-o", co], check=True)
    return s, co


def main():
    from tools import check_isa_hazards as scan
    hip = C.CDLL("libamdhip64.so")
    text = open(FIXTURE).read()
    res = {"fixture": os.path.relpath(FIXTURE, ROOT), "builds": {}}
    with tempfile.TemporaryDirectory() as tmp:
        builds = {"as_built": assemble(text, tmp, "as_built"), "with_s_nop": assemble(patched_source(text), tmp, "with_s_nop")}
        torch.zeros(1, device="cuda")  # context
        for name, (s_path, co) in builds.items():
            hazards = scan.scan_asm(s_path)
            mod, fn = C.c_void_p(), C.c_void_p()
            assert hip.hipModuleLoad(C.byref(mod), co.encode()) == 0
            assert hip.hipModuleGetFunction(C.byref(fn), mod, KERNEL) == 0
            cases = []
            for (K, S) in [(8, 384), (16, 2688), (1, 384), (8, 160), (16, 96), (4, 1024)]:
                B, H = 3, 2
                inner = H * 64
                g = torch.Generator().manual_seed(K * 1000 + S)
                q = (torch.randn(B * K, inner, generator=g) * 0.3).to(torch.bfloat16).cuda()
                kb = torch.randn(B, H, S, 64, generator=g).to(torch.bfloat16).cuda()
                vt = torch.randn(B, H, S, 64, generator=g).transpose(2, 3).contiguous().to(torch.bfloat16).cuda()
                mask = torch.ones(B, S, dtype=torch.uint8, device="cuda")
                out = torch.zeros(B * K, inner, dtype=torch.bfloat16, device="cuda")
                args = [C.c_void_p(q.data_ptr()), C.c_void_p(kb.data_ptr()), C.c_void_p(vt.data_ptr()), C.c_void_p(mask.data_ptr()),
                        C.c_void_p(out.data_ptr()), C.c_int(K), C.c_int(H), C.c_int(S)]
                argv = (C.c_void_p * len(args))(*[C.cast(C.byref(a), C.c_void_p) for a in args])
                smem = (2 * 4 * 16 + 4 * 16 * 64) * 4 + 16
                worst_lo = worst_hi = 0.0
                for rep in range(20):  # the failure is timing dependent: repeat
                    out.zero_()
                    rc = hip.hipModuleLaunchKernel(fn, H, B, 1, 256, 1, 1, smem, None, argv, None)
                    assert rc == 0, rc
                    torch.cuda.synchronize()
                    qh = q.float().view(B, K, H, 64).permute(0, 2, 1, 3)
                    p = torch.softmax(qh @ kb.float().transpose(2, 3), -1)
                    ref = (p @ vt.float().transpose(2, 3)).permute(0, 2, 1, 3).reshape(B * K, inner)
                    err = (out.float() - ref).abs().view(B * K, H, 64)
                    worst_lo = max(worst_lo, float(err[..., :48].max()))
                    worst_hi = max(worst_hi, float(err[..., 48:].max()))
                cases.append({"K": K, "S": S, "max_abs_err_dims_0_47": worst_lo, "max_abs_err_dims_48_63": worst_hi})
            res["builds"][name] = {"mfma_valu_hazards_found_by_scanner": len(hazards), "cases": cases}
            hip.hipModuleUnload(mod)
    txt = json.dumps(res, indent=1)
    print(txt)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", "mfma_valu_hazard_probe.json"), "w").write(txt + "\n")


if __name__ == "__main__":
    main()
