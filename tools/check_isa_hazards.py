#!/usr/bin/env python3
"""Scan the gfx950 ISA of every kernel in gram_amd/csrc for a packed-fp32 self-overwrite pattern.

`v_pk_{mul,add,fma}_f32 vD[lo:hi], ...` executes its low lane before its high lane; if the register the
HIGH lane reads (a source's low register when that operand is broadcast with op_sel_hi = 0, or its high
register otherwise) is the destination's LOW register, the high lane sees the low lane's result.  hipcc
(ROCm 7.2) emitted exactly one such instruction in the cross-attention kernel (wrong outputs, no fault).
Usage: tools/check_isa_hazards.py [file.hip ...]  -> exit status 1 if any hazard is found."""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAT = re.compile(r'^\s*(v_pk_(?:mul|add|fma)_f32)\s+v\[(\d+):(\d+)\],\s*(.*?)(?:\s+op_sel:\[([\d,]+)\])?(?:\s+op_sel_hi:\[([\d,]+)\])?\s*$')


def scan_asm(path):
    found, kern = [], None
    for line in open(path):
        m = re.match(r'^(_Z\w+):', line)
        if m:
            kern = m.group(1)
        m = PAT.match(line)
        if not m:
            continue
        d0 = int(m.group(2))
        ops = [o.strip() for o in m.group(4).split(',')]
        hi = [int(x) for x in m.group(6).split(',')] if m.group(6) else [1] * len(ops)
        for i, o in enumerate(ops):
            mm = re.match(r'v\[(\d+):(\d+)\]', o)
            if mm and int(mm.group(1)) + (1 if hi[i] else 0) == d0:
                found.append((kern, line.strip()))
    return found


def main(files):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    bad = []
    with tempfile.TemporaryDirectory() as tmp:
        for f in files:
            base = os.path.splitext(os.path.basename(f))[0]
            subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-c", f, "-o", os.path.join(tmp, base + ".o"),
                            "-save-temps=obj"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=os.path.dirname(f))
            for asm in glob.glob(os.path.join(tmp, base + "-hip-amdgcn-*.s")):
                bad += [(base,) + h for h in scan_asm(asm)]
    for b in bad:
        print("HAZARD", *b)
    print(f"{len(files)} files scanned, {len(bad)} hazards")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "gram_amd", "csrc", "*.hip")))))
