"""GPU micro-benchmark (not a test): lm_head in sparse mode (LSE partials only) per GEMM variant."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gram_amd import _lib  # noqa: E402
from tests import gpu_util as G  # noqa: E402

lib = _lib.load()
M, N, K = int(sys.argv[1]) if len(sys.argv) > 1 else 81920, 32128, 768
g = torch.Generator().manual_seed(0)
A = torch.randn(M, K, generator=g).to(G.DEV).to(G.DT)
W = (torch.randn(N, K, generator=g) * K ** -0.5).to(G.DEV).to(G.DT)
part = torch.empty(M, N // 64, 2, dtype=torch.float32, device=G.DEV)
for v in (3, 22, 3, 22):
    lib.gram_debug_set_gemm_variant(v)
    run = lambda: _lib.check(lib.gram_gemm_bf16_lse(G.p(A), G.p(W), None, G.p(part), M, N, K, K, N, G.stream()), "lse")
    for _ in range(3):
        run()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        run()
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 100
    print(f"variant {v:3d}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF", flush=True)
lib.gram_debug_set_gemm_variant(-1)
