#!/usr/bin/env python3
"""Scan the gfx950 ISA of the kernels in gram_amd/csrc for packed-fp32 instructions whose destination pair
overlaps the register its HIGH lane reads (`v_pk_{mul,add,fma}_f32 vD[lo:hi], ..., vD... op_sel_hi:[..,0]`).

History: one build of the cross-attention kernel produced wrong outputs (no fault) and differed from a
correct build by exactly one such instruction, so the pattern is kept as a TRIPWIRE for that kernel
(dec_attn.hip: a hit fails the CPU test-suite).  tools/probes/pk_mul_self.hip later showed that the
instruction form by itself executes correctly on MI355X (64/64 lanes), and the ping-pong GEMM contains it
while matching the other GEMM kernel bit for bit -- so elsewhere a hit is reported as a note, and what
guards those kernels is their parity tests.
Usage: tools/check_isa_hazards.py [file.hip ...]  -> exit status 1 if a STRICT file (dec_attn.hip) has a hit."""
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


STRICT = ("dec_attn",)


def main(files, strict=STRICT):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    bad = []
    with tempfile.TemporaryDirectory() as tmp:
        for f in files:
            base = os.path.splitext(os.path.basename(f))[0]
            subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-c", f, "-o", os.path.join(tmp, base + ".o"),
                            "-save-temps=obj"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=os.path.dirname(f))
            for asm in glob.glob(os.path.join(tmp, base + "-hip-amdgcn-*.s")):
                bad += [(base,) + h for h in scan_asm(asm)]
    fatal = [b for b in bad if b[0] in strict]
    for b in bad:
        print("HAZARD" if b[0] in strict else "note", *b)
    print(f"{len(files)} files scanned, {len(bad)} pattern hits, {len(fatal)} in strict files {strict}")
    return 1 if fatal else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "gram_amd", "csrc", "*.hip")))))
