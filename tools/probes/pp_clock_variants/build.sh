#!/bin/bash
# Variant builds of gemm.hip for the round-4 hunt of the "clock stamps behind a flag" wrong-result build (DESIGN.md section 4.3b,
# root cause: a missing barrier in the ping-pong kernel's prologue;
# profiles/r04i_*, r04n_*): each variant is today's gemm.hip with one textual change, compiled and linked against the product objects
# into gram_amd/csrc/variants/libgram_hip_<name>.so (git-ignored; GRAM_LIB=... selects it).  Run from the repo root, after `make`.
set -e
cd "$(dirname "$0")/../../.."
C=gram_amd/csrc
V=$C/variants
mkdir -p $V/src
HIPCC=/opt/rocm/bin/hipcc
python3 - <<'P'
import re
src = open('gram_amd/csrc/gemm.hip').read()
good = '  const unsigned long long clk_t0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();\n'
assert src.count(good) == 1
cond = ('  unsigned long long clk_t0 = 0ull, clk_r0 = 0ull;\n  if (clk_on) {\n    clk_t0 = __builtin_amdgcn_s_memtime();\n'
        '    clk_r0 = __builtin_amdgcn_s_memrealtime();\n  }\n')
cond_asm = ('  unsigned long long clk_t0 = 0ull, clk_r0 = 0ull;\n  if (clk_on) {\n'
            '    asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(clk_t0) :: "memory");\n'
            '    asm volatile("s_memrealtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(clk_r0) :: "memory");\n  }\n')
nop0 = lambda s: s.replace('s_nop 3\\n\\tglobal_load_lds', 's_nop 0\\n\\tglobal_load_lds')
drain = lambda s: re.sub(r's_waitcnt vmcnt\(\d+\)', 's_waitcnt vmcnt(0)', s)
# the prologue barrier that closes the race (gemm.hip, "Every wave holds its W_n0(0) fragments before anyone goes on"): the historical
# variants are built WITHOUT it, `fixed_*` with it
fix_a = src.index('  // Every wave holds its W_n0(0) fragments before anyone goes on')
fix_b = src.index('  if (wr == 1) pp_barrier();  // group 1 runs one barrier behind group 0')
assert src[fix_a:fix_b].count('pp_barrier();') == 1
fixed, src = src, src[:fix_a] + src[fix_b:]
bad0 = nop0(src.replace(good, cond))
entry = 'asm volatile("s_waitcnt vmcnt(12)" ::: "memory");  // W_n0(0) and A_m0(0) have landed'
assert bad0.count(entry) == 1
slot_a = bad0.index('  auto end_load_slot = [&](int extra = 0) {')
slot_b = bad0.index('  };', slot_a)
tend_a = '      if (wr == 0) pp_barrier();\n      if constexpr ((GRAM_PP_ABL & 1) != 0) {  // ablation: keep the accumulators live, store (almost) never'
tend_b = '      zero_half(0);\n      zero_half(1);\n      pp_barrier();\n      if (more && wr == 1) pp_barrier();'
assert bad0.count(tend_a) == 1 and bad0.count(tend_b) == 1
full = 'asm volatile("s_waitcnt vmcnt(0)" ::: "memory");'
# the proof: run `nofix_chaos` (the round-4 prologue, -DGRAM_CHAOS=1) with ENTRY_DELAY=20 (the chaos build's test hook holds group 1 back ~5 us between the prologue's
# first barrier and its read of W_n0(0)): that read finds group 0's re-fill of the buffer -- wrong first tiles, every run; with the barrier
# (the product library) nothing changes
variants = {
    'nofix_chaos': src,          # the round-4 kernel (no prologue barrier), compiled with -DGRAM_CHAOS=1: does the chaos build find the race by itself?
    'nofix': src,
    'fixed_bad': fixed.replace(good, cond),                                          # conditional stamp reads + the prologue barrier: clean
    'fixed_bad_nop0': nop0(fixed.replace(good, cond)),
    'bad_drain_entry': bad0.replace(entry, full),                                   # only the prologue's counted wait a full drain
    'bad_drain_slot': bad0[:slot_a] + drain(bad0[slot_a:slot_b]) + bad0[slot_b:],    # only the load slots' counted waits (TEND kernels use no other)
    'bad_vm10': bad0[:slot_a] + bad0[slot_a:slot_b].replace('vmcnt(12)', 'vmcnt(10)') + bad0[slot_b:],  # one half-tile stricter
    'bad_tend_drain': bad0.replace(tend_a, tend_a.replace('pp_barrier();\n', 'pp_barrier();\n      ' + full + '\n', 1)).replace(tend_b, '      ' + full + '\n' + tend_b),
    'bad': src.replace(good, cond),                      # conditional stamp reads, today's tree (DMA statements padded: s_nop 3)
    'bad_nop0': nop0(src.replace(good, cond)),           # ... with the DMA statements as they were (s_nop 0): the round-4 "hyg" build
    'bad_drain': drain(nop0(src.replace(good, cond))),   # ... and every counted vmcnt wait a full drain
    'bad_asm': nop0(src.replace(good, cond_asm)),        # the reads as asm statements that wait for their own result
    'good_nop0': nop0(src),                              # the shipped form with unpadded DMA statements (control)
}
for k, v in variants.items():
    open(f'gram_amd/csrc/variants/src/gemm_{k}.hip', 'w').write(v)
P
for f in $V/src/gemm_*.hip; do
  n=$(basename $f .hip); n=${n#gemm_}
  X=""; case $n in *_chaos) X="-DGRAM_CHAOS=1";; esac
  ( cp $f $C/_variant_$n.hip && $HIPCC --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function $X -c $C/_variant_$n.hip -o $V/gemm_$n.o && rm -f $C/_variant_$n.hip &&
    $HIPCC --offload-arch=gfx950 -shared -fPIC $V/gemm_$n.o $C/build/rowops.o $C/build/enc_attn.o $C/build/dec_attn.o $C/build/beam.o $C/build/generate.o $C/build/prof.o $C/build/aliases.o -o $V/libgram_hip_$n.so && echo built $n ) &
done
wait
ls -la $V/*.so
