#!/usr/bin/env python3
"""Static check of the gfx950 ISA of gram_amd/csrc/*.hip for MFMA -> VALU/VMEM/LDS register hazards that the
compiler's hazard recognizer left open across basic-block boundaries.

CDNA3/4 has no hardware interlock between a matrix instruction's VGPR write-back and a later vector-ALU, memory or
LDS instruction that reads (or overwrites) those VGPRs: software must keep a minimum number of wait states (issued
instructions / s_nop cycles) between them.  hipcc pads such pairs inside a basic block (e.g. ``v_mfma ...; s_nop 6;
v_cndmask``: 8 wait states after a 4-pass v_mfma_f32_16x16x32_bf16), but one round-1 build of the cross-attention
kernel (commit a152048^) read an accumulator FOUR wait states after its MFMA, through two branches:

    v_mfma_f32_16x16x32_bf16 v[102:105], v[42:45], v[102:105], v[138:141]     ; last MFMA of a 32-key step (dims 48-63)
    s_cbranch_vccnz .LBB1_34
  .LBB1_34:  s_mov_b64 ...;  s_and_b64 vcc, ...;  s_cbranch_vccz .LBB1_38
  .LBB1_38:  v_mov_b64_e32 v[46:47], v[102:103]     <- reads the result 4 wait states after issue, 8 are required
             ... s_nop 1; v_mov_b64_e32 v[48:49], v[104:105]

and returned wrong values in exactly those dims (element 1 of the tile: the MFMA writes its four result registers pass
by pass), for K <= 16 with more than one step per wave.  tests/golden/isa_mfma_valu_hazard_r01.s keeps that ISA; this
scanner walks every kernel's control-flow graph from each MFMA and reports any non-matrix instruction that touches the
MFMA's destination registers before the required wait states have passed.  Any hit fails the CPU test-suite; the fix at
source level is an explicit ``s_nop`` after the last MFMA of the loop body (dec_attn.hip).

The packed-fp32 pattern the round-1 scanner looked for (``v_pk_mul_f32 vD, vA, vD op_sel_hi:[1,0]``) is NOT a hazard:
tools/probes/pk_mul_self.hip shows it executes correctly; it differed between the two builds only because the register
allocation did.

A second scan (scan_kernel_sgpr_vmem, below) looks for vector-memory instructions that read an SGPR fewer than 5 wait states after a
VALU instruction wrote it -- the one pair hipcc cannot pad when the load sits inside an inline-asm statement.

Usage: tools/check_isa_hazards.py [file.hip | file.s ...]   -> exit status 1 on any hit."""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# wait states hipcc itself keeps between an MFMA's issue and a VALU read of its result on gfx950 (measured from its own
# in-block padding: s_mov + s_nop 6 after v_mfma_f32_16x16x32_bf16); by number of passes
REQUIRED = {"16x16x32": 8, "32x32x16": 12, "16x16x16": 8, "32x32x8": 12, "4x4x4": 6}
DEFAULT_REQUIRED = 20

REG = re.compile(r'\b([va])(?:(\d+)|\[(\d+):(\d+)\])')
LABEL = re.compile(r'^([.\w$]+):')


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(2) is not None:
            out.add((m.group(1), int(m.group(2))))
        else:
            for r in range(int(m.group(3)), int(m.group(4)) + 1):
                out.add((m.group(1), r))
    return out


def parse_kernels(path):
    """-> {kernel: [(mnemonic, operands_text, label_or_None)]}; labels are attached to the instruction they precede."""
    kernels, cur, pending = {}, None, []
    for raw in open(path):
        line = raw.split(';')[0].rstrip()
        if not line.strip():
            continue
        m = LABEL.match(line.strip())
        if m:
            name = m.group(1)
            if name.startswith('_Z') or (not name.startswith('.') and cur is None):
                cur = kernels.setdefault(name, [])
                pending = []
            elif name.startswith('.Lfunc_end'):
                cur = None
            elif cur is not None:
                pending.append(name)
            continue
        s = line.strip()
        if s.startswith('.'):
            if s.startswith('.end_amdhsa_kernel') or s.startswith('.section') or s.startswith('.text'):
                pass
            continue
        if cur is None:
            continue
        parts = s.split(None, 1)
        cur.append((parts[0], parts[1] if len(parts) > 1 else '', tuple(pending)))
        pending = []
    return kernels


def wait_states(mn, ops):
    if mn == 's_nop':
        try:
            return int(ops.strip(), 0) + 1
        except ValueError:
            return 1
    return 1


def scan_kernel(name, ins):
    label_at = {}
    for i, (_, _, labels) in enumerate(ins):
        for l in labels:
            label_at[l] = i
    hits = []
    for i, (mn, ops, _) in enumerate(ins):
        if not mn.startswith('v_mfma') and not mn.startswith('v_smfmac'):
            continue
        need = next((v for k, v in REQUIRED.items() if k in mn), DEFAULT_REQUIRED)
        dst = regs_of(ops.split(',')[0])
        best = {}
        work = [(i + 1, 0)]
        while work:
            j, w = work.pop()
            while j < len(ins) and w < need:
                if best.get(j, 1 << 30) <= w:
                    break
                best[j] = w
                m2, o2, _ = ins[j]
                is_matrix = m2.startswith('v_mfma') or m2.startswith('v_smfmac')
                if not is_matrix and not m2.startswith('s_') and (regs_of(o2) & dst):
                    hits.append((name, i, f"{mn} {ops}", j, f"{m2} {o2}", w, need))
                    break
                if is_matrix and (regs_of(o2.split(',')[0]) & dst) == dst:
                    break  # the registers are redefined by a later matrix op; its own entry covers what follows
                w += wait_states(m2, o2)
                if m2 == 's_endpgm' or m2.startswith('s_setpc') or m2.startswith('s_swappc'):
                    break
                if m2 == 's_branch':
                    j = label_at.get(o2.strip(), len(ins))
                    continue
                if m2.startswith('s_cbranch'):
                    t = label_at.get(o2.strip())
                    if t is not None:
                        work.append((t, w))
                j += 1
    return hits


def scan_asm(path):
    hits = []
    for name, ins in parse_kernels(path).items():
        hits += scan_kernel(name, ins)
    return hits


# ---- second check: a vector-memory instruction reading an SGPR that a VALU instruction wrote fewer than 5 wait states earlier.
# gfx9 has no interlock for that pair ("VALU writes SGPR -> VMEM reads that SGPR: 5 wait states"); hipcc pads its OWN loads and stores,
# but a load inside an inline-asm statement (the LDS-DMA of gemm.hip / dec_attn.hip / rowops.hip) is text to it.  Round 4's library had
# `v_readlane_b32 s5, v255, 3` (a spilled SGPR coming back) two wait states ahead of `global_load_lds_dword v2, s[4:5]` in four
# one-piece instantiations of the ping-pong GEMM; the asm statements now carry the padding themselves (s_mov m0 + s_nop 3).
SREG = re.compile(r'\bs(?:(\d+)|\[(\d+):(\d+)\])')
VMEM_PREFIX = ('global_', 'buffer_', 'scratch_', 'flat_')
SGPR_VMEM_WAIT = 5


def sregs_of(text):
    out = set()
    for m in SREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def scan_kernel_sgpr_vmem(name, ins):
    """Walks the control-flow graph BACKWARDS from every vector-memory instruction over all paths, up to 5 wait states."""
    n = len(ins)
    label_at = {l: i for i, (_, _, labels) in enumerate(ins) for l in labels}
    pred = [[] for _ in range(n)]
    for i, (mn, ops, _) in enumerate(ins):
        if mn == 's_endpgm':
            continue
        if mn == 's_branch':
            t = label_at.get(ops.strip())
            if t is not None:
                pred[t].append(i)
            continue
        if mn.startswith('s_cbranch'):
            t = label_at.get(ops.strip().split(',')[-1].strip())
            if t is not None:
                pred[t].append(i)
        if i + 1 < n:
            pred[i + 1].append(i)
    hits = []
    for i, (mn, ops, _) in enumerate(ins):
        if not mn.startswith(VMEM_PREFIX):
            continue
        need = sregs_of(ops)
        if not need:
            continue
        seen, work = set(), [(p, 0, frozenset(need)) for p in pred[i]]
        while work:
            j, w, regs = work.pop()
            if w >= SGPR_VMEM_WAIT or (j, w, regs) in seen:
                continue
            seen.add((j, w, regs))
            m2, o2, _ = ins[j]
            written = sregs_of(o2.split(',')[0]) if (m2.startswith('v_') or m2.startswith('s_')) and not m2.startswith('s_cmp') else set()
            if m2.startswith('v_') and (written & regs):
                hits.append((name, j, f"{m2} {o2}", i, f"{mn} {ops}", w, SGPR_VMEM_WAIT))
            if written & regs:
                regs = regs - written  # (a scalar-ALU write is interlocked, a VALU write has been reported: nothing older matters for these)
                if not regs:
                    continue
            for p in pred[j]:
                work.append((p, w + wait_states(m2, o2), regs))
    return sorted(set(hits), key=lambda h: (h[3], h[1]))


def scan_asm_sgpr_vmem(path):
    hits = []
    for name, ins in parse_kernels(path).items():
        hits += scan_kernel_sgpr_vmem(name, ins)
    return hits


def main(files):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    bad = []
    with tempfile.TemporaryDirectory() as tmp:
        for f in files:
            base = os.path.splitext(os.path.basename(f))[0]
            if f.endswith('.s'):
                bad += [(base,) + h for h in scan_asm(f) + scan_asm_sgpr_vmem(f)]
                continue
            subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-c", f, "-o", os.path.join(tmp, base + ".o"),
                            "-save-temps=obj"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=os.path.dirname(f))
            for asm in glob.glob(os.path.join(tmp, base + "-hip-amdgcn-*.s")):
                bad += [(base,) + h for h in scan_asm(asm) + scan_asm_sgpr_vmem(asm)]
    for b in bad:
        print(f"HAZARD {b[0]} {b[1]}: [{b[2]}] {b[3]}  ->  [{b[4]}] {b[5]}  after {b[6]} wait states (need {b[7]})")
    print(f"{len(files)} files scanned, {len(bad)} MFMA->VALU/VMEM and VALU-SGPR->VMEM hazards")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "gram_amd", "csrc", "*.hip")))))
