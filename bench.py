#!/usr/bin/env python3
"""bench.py -- users/sec of GRAM's generative scoring path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1], the configuration the metric is quoted on): Beauty item Trie
(12 101 items, T = 9 decode steps), T5-base, 3 granularity passages of 128 tokens per user
(S = 384 fused keys), beam = 20 returning top-20.  One "step" = one ``GRAM.generate`` over a batch
of B synthetic users already resident in HBM (random-init weights of the exact architecture,
uniform token ids, all-valid masks: there is no network for checkpoints or tokenisers).  Users are
independent, so N GPUs run N disjoint batches (weak scaling, no data-path collective).

``--gpus N`` without a launcher starts the N ranks itself (children spawned BEFORE this process touches the GPU, one per
device, RCCL over 127.0.0.1); under torchrun (RANK/WORLD_SIZE in the environment) it is one of the ranks.  With N > 1 the
timed region ends with the eval's real exchange: every rank's hit ranks (position of a synthetic gold item in its top-K)
in ONE all_gather_into_tensor over RCCL, cross-checked by the reference's all_reduce(SUM) of metric sums.

One JSON line is printed by rank 0.  Besides the contract keys it carries
  roofline      the dominant kernel by summed device time, measured live with HIP events on the
                launch stream over the timed region (gram_prof_* in libgram_hip.so)
  roofline_cross_attn   the north star's named roofline: the fusion cross-attention kernel
                against the HBM peak, algorithmic bytes = B*H*S*64*2(K,V)*2 B per launch
  kernel_ms_per_step    device-time breakdown by kernel kind
  cpu_baseline  the CPU oracle (reference-faithful mode: beam-replicated KV, per-step cache
                reorder, B = 1 per call like the reference runner) timed on this host's cores on a
                bounded sample, rank 0 at N = 1 only
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16 MFMA peak
# The arithmetic the headline is timed in: the cheapest mode for which tests/test_gpu_precision.py asserts |dRecall@5|, |dNDCG@5|
# <= 1e-4 against the fp32 reference on 16 384 T5-base users of EACH test population (plain, attention-sharpened, ragged masks):
# two IEEE-half pieces per value, three MFMA products per product (profiles/r03*_precision_*.json; no rank flip on any population).
DEFAULT_PRECISION = "f16x3"
PIECES = {"f16": 1, "f16x3": 2, "bf16": 1, "bf16x3": 2}   # (bf16*: the PIECE=bf16 build of the library, GRAM_LIB=...)
NPROD = {"f16": 1, "f16x3": 3, "bf16": 1, "bf16x3": 3}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=4096, help="users per generate() call per GPU")
    ap.add_argument("--backbone", default="t5-base")
    ap.add_argument("--dataset", default="Beauty")
    ap.add_argument("--passages", type=int, default=3)
    ap.add_argument("--passage-len", type=int, default=128)
    ap.add_argument("--beams", type=int, default=20)
    ap.add_argument("--cpu-users", type=int, default=16, help="users in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--precision", default=DEFAULT_PRECISION, choices=sorted(PIECES),
                    help="operand arithmetic (gram_split_t): f16 = one IEEE-half piece per value (11 significant bits); f16x3 = two pieces, "
                         "3 MFMA products per product (~2^-22, fp32-class); accumulation, residual stream, softmax and scores are fp32 in both")
    ap.add_argument("--q-sharpen", type=float, default=4.0,
                    help="every attention q projection of the random-init weights is multiplied by this (1 = plain T5 init).  At the plain "
                         "init every query averages ~140 keys, the encoder is washed out of the scores, all users get the same beams and "
                         "NONE of them sits on one of the Trie's longer ids -- the last decode step then has no live row and the live-row "
                         "compaction drops it (96 instead of 108 cross-attention launches per generate).  With peaky attention the beams "
                         "depend on the passages as a trained model's do and the last step keeps the few percent of live rows the Trie implies")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the secondary measurements of the N = 1 run (all-rows decode, B = 1 latency, the other precision modes)")
    ap.add_argument("--e2e-users", type=int, default=22363,
                    help="users of the synthetic dataset directory the drop-in runner scores end to end in `extras` (Beauty has 22 363; 0 = skip)")
    ap.add_argument("--no-prof", action="store_true", help="do not record per-kernel HIP events in the timed region")
    ap.add_argument("--ragged", action="store_true",
                    help="realistic batch: passage counts drawn from the dataset's histogram (padded to --passages), valid "
                         "lengths U[32, L]; not the headline configuration")
    ap.add_argument("--item-pool", type=int, default=0,
                    help="passages 1..N-1 of every user are drawn from a pool of this many item prompts, registered with "
                         "GRAM.cache_passages before the warmup (SURVEY.md §8f N2); not the headline configuration")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL on ROCm)")
    ap.add_argument("--check-allreduce", action="store_true",
                    help="--gpus > 1: also run the reference's all_reduce(SUM) of the metric sums as a cross-check of the all-gather")
    ap.add_argument("--share-device", action="store_true",
                    help="rehearsal only: all ranks use cuda:0 (with --backend gloo; RCCL needs one GPU per rank)")
    return ap.parse_args()


def _strip(row):
    row = list(row)
    while row and row[-1] == 0:
        row.pop()
    return tuple(row)


def spawn_ranks(args):
    """--gpus N without a launcher: start N ranks of this script (one per device) and relay rank 0's JSON line.  Runs before
    anything here touches the GPU; the parent only waits."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    return max(abs(c) for c in codes)


def e2e_runner(model, dev, n_users):
    """users/s of ``get_runner("single", ...).test()`` on a generated Beauty-sized dataset directory, reference default flags."""
    import shutil
    import tempfile

    from gram_amd.runner import get_runner
    from tools import synth_dataset as SD

    root = tempfile.mkdtemp(prefix="gram_e2e_")
    try:
        a = SD.make(root, n_users=n_users)
        runner = get_runner("single", model, None, SD.SynthTokenizer(), None, None, None, dev, a)
        runs = []
        for _ in range(2):  # the second run is the steady state of a process (weights packed, workspace allocated)
            model.clear_passage_cache()  # every evaluation pays for its own passage-cache fill
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            runner.test(None)  # rebuilds the loaders like the reference (single_runner_gram.py:370-375), then scores
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            r = runner.last_results
            runs.append({"users": r["total"], "test_call_s": round(wall, 3), "score_loader_s": round(r["score_loader_seconds"], 3),
                         "generate_s": round(r["generate_seconds"], 3), "users_per_call": runner.last_host["users_per_call"],
                         "host_phases_s": {k: round(v, 3) for k, v in runner.last_host.get("phases", {}).items()}})
        r = runs[-1]
        return {"users_per_s_end_to_end": r["users"] / r["score_loader_s"], "users_per_s_generate_only": r["users"] / r["generate_s"],
                "users_per_s_test_call_incl_dataset_load": r["users"] / r["test_call_s"],
                "end_to_end_over_generate_only": r["generate_s"] / r["score_loader_s"],
                "host_ms_per_batch_not_overlapped": 1e3 * (r["score_loader_s"] - r["generate_s"]) / max(1, -(-r["users"] // r["users_per_call"])),
                "flags": "--eval_batch_size 1 (reference default), --max_his 2, beam 20, passage cache on; generate-only = the sum of the "
                         "generate() calls, the quantity the reference logs (single_runner_gram.py:640-652,712-714); end-to-end = "
                         "test_dataset_task: candidates -> Trie -> passage cache -> collate -> H2D -> generate -> strings -> metrics",
                "runs": runs}
    finally:
        shutil.rmtree(root, ignore_errors=True)


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if args.share_device:
        local_rank = 0
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(args.backend)
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)

    import gram_amd
    from gram_amd import _lib
    from gram_amd.utils import generation_trie as gt

    lib = _lib.load()
    cfg = gram_amd.T5Config.named(args.backbone)
    torch.manual_seed(2023)
    model = gram_amd.create_model("gram", cfg)
    if args.q_sharpen != 1.0:
        with torch.no_grad():
            for name, p_ in model.named_parameters():
                if name.endswith(".q.weight"):
                    p_.mul_(args.q_sharpen)
    state = {k: v.detach().clone() for k, v in model.state_dict().items()} if (rank == 0 and args.cpu_users > 0 and world == 1) else None
    model = model.to(dev).eval()
    model.set_precision(args.precision)

    z = np.load(os.path.join(ROOT, "tests", "golden", "tries.npz"))
    cands = [[int(x) for x in row if x >= 0] for row in z[f"{args.dataset}_cands"]]
    trie = gt.Trie(cands)
    fn = gt.prefix_allowed_tokens_fn(trie)
    max_length = max(len(c) for c in cands)

    B, N, L, K = args.batch, args.passages, args.passage_len, args.beams
    g = torch.Generator().manual_seed(1000 + rank)
    ids = torch.randint(2, 32100, (B, N, L), generator=g)
    ids[:, :, -1] = 1
    mask = torch.ones(B, N, L, dtype=torch.bool)
    if args.ragged:
        hist = torch.tensor(z[f"{args.dataset}_npassage_hist"][: N + 1].astype("float64"))
        hist[N] += float(z[f"{args.dataset}_npassage_hist"][N + 1:].sum())
        hist[0] = 0
        n_user = torch.multinomial(hist / hist.sum(), B, replacement=True, generator=g)
        lens = torch.randint(32, L + 1, (B, N), generator=g)
        mask = (torch.arange(L)[None, None, :] < lens[:, :, None]) & (torch.arange(N)[None, :, None] < n_user[:, None, None])
        ids[~mask] = 0
    item_cache = None
    if args.item_pool and N > 1:
        pg = torch.Generator().manual_seed(77)  # the same pool on every rank
        P = args.item_pool
        pool_ids = torch.randint(2, 32100, (P, L), generator=pg)
        pool_len = torch.randint(32, L + 1, (P,), generator=pg) if args.ragged else torch.full((P,), L)
        pool_mask = torch.arange(L)[None, :] < pool_len[:, None]
        pool_ids[torch.arange(P), pool_len - 1] = 1
        pool_ids[~pool_mask] = 0
        pick = torch.randint(0, P, (B, N - 1), generator=g)
        slot_on = mask[:, 1:].any(-1)
        ids[:, 1:] = pool_ids[pick]
        mask[:, 1:] = pool_mask[pick] & slot_on[..., None]
        ids[~mask] = 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_cached = model.cache_passages(pool_ids.to(dev), pool_mask.to(dev))
        torch.cuda.synchronize()
        item_cache = {"pool": P, "cached_passages": n_cached, "prefill_s": round(time.perf_counter() - t0, 3),
                      "cache_GiB": round(n_cached * 128 * cfg.d_model * 4 / 2 ** 30, 3)}
    ids_d, mask_d = ids.to(dev), mask.to(dev)
    if item_cache is not None:
        plan = model._plan_encoder(ids_d, mask_d.view(torch.uint8), B, N, L)
        item_cache["passages_from_cache_per_step"] = int(plan[0].n_cached)
        item_cache["passages_encoded_per_step"] = int(plan[0].n_active - plan[0].n_cached)

    def step():
        return model.generate(input_ids=ids_d, attention_mask=mask_d, max_length=max_length, prefix_allowed_tokens_fn=fn,
                              num_beams=K, num_return_sequences=K, output_scores=True, return_dict_in_generate=True,
                              length_penalty=1.0)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    # synthetic gold item per user (uniform over the candidates, seeded per rank) -> hit rank in the returned top-K
    gold_idx = torch.randint(0, len(cands), (B,), generator=g)
    gold = torch.zeros(B, max_length, dtype=torch.int64)
    for b, gi in enumerate(gold_idx.tolist()):
        gold[b, : len(cands[gi])] = torch.tensor(cands[gi])
    gold_d = gold.to(dev)

    def hit_ranks(out):
        seq = out["sequences"]
        pad = torch.zeros(B * K, max_length, dtype=torch.int64, device=dev)
        pad[:, : seq.shape[1]] = seq
        same = (pad.view(B, K, max_length) == gold_d[:, None, :]).all(-1)  # (B, K); sequences are score-sorted
        first = torch.where(same.any(1), same.float().argmax(1), torch.full((B,), -1.0, device=dev))
        return first.to(torch.int16).cpu().numpy()

    for _ in range(args.warmup):
        out = step()
    kinds = {"gemm": _lib.K_GEMM, "enc_attn": _lib.K_ENC_ATTN, "cross_attn": _lib.K_CROSS_ATTN,
             "dec_self_attn": _lib.K_DEC_SELF_ATTN, "rowops": _lib.K_ROWOPS, "lse": _lib.K_LSE, "beam": _lib.K_BEAM}
    prof = not args.no_prof
    if prof:
        launches_per_step = 40 * (cfg.num_layers + cfg.num_decoder_layers * max_length) + 64
        _lib.check(lib.gram_prof_enable(sum(1 << k for k in kinds.values()), launches_per_step * args.steps), "prof_enable")
    lib.gram_prof_pp_clock_enable(1)  # the ping-pong GEMM's clock stamps: a diagnostic, off in the product path (gram_hip.h)
    _lib.check(lib.gram_prof_pp_clock(None, 1), "pp_clock reset")  # (synchronises; outside the timed region)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    exchange = None
    if distributed:
        # the eval's ONE exchange: every rank's packed {user_idx:int32, hit_rank:int16} records in one fixed-width
        # all_gather_into_tensor over RCCL/xGMI; the reference's all_reduce(SUM) of the metric sums
        # (distributed_runner_gram.py:832-838) only as a cross-check under --check-allreduce
        from gram_amd.runner import all_gather_hits
        from gram_amd.utils import evaluate as ev
        cdev = dev if args.backend == "nccl" else torch.device("cpu")
        mine = hit_ranks(out)
        rec = all_gather_hits(rank * B + np.arange(B, dtype=np.int32), mine, world * B, cdev)
        allr = rec["hit_rank"].astype(np.int16)
        names = ["hit@5", "hit@10", "ndcg@5", "ndcg@10"]
        sums = ev.metrics_from_ranks(allr, names, K)
        assert len(allr) == world * B and sorted(rec["user_idx"].tolist()) == list(range(world * B)), "records lost in the all-gather"
        if args.check_allreduce:
            local = torch.tensor(ev.metrics_from_ranks(mine, names, K), dtype=torch.float64, device=cdev)
            dist.all_reduce(local, op=dist.ReduceOp.SUM)
            assert np.allclose(local.cpu().numpy(), sums), "all-gather and all-reduce disagree"
        exchange = {"collective": f"one all_gather_into_tensor of {world} x {B} packed {{user_idx:int32, hit_rank:int16}} records (6 B each)"
                                  + (" + all_reduce(SUM) cross-check" if args.check_allreduce else "") + f", backend {args.backend}",
                    "users_gathered": int(len(allr)), "metrics_vs_synthetic_gold": dict(zip(names, (sums / len(allr)).round(6).tolist()))}
    barrier()
    dt = time.perf_counter() - t0
    if distributed:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        # (outside the timed region) which device each rank ran on: one GPU per rank unless --share-device
        devs = torch.tensor([torch.cuda.current_device()], dtype=torch.int64, device=tmax.device)
        dev_list = [torch.zeros_like(devs) for _ in range(world)]
        dist.all_gather(dev_list, devs)
        exchange["devices"] = [int(t.item()) for t in dev_list]

    pp_ghz = C.c_double(0.0)  # time-weighted in-kernel clock of the ping-pong GEMM launches of the timed region (gram_hip.h)
    _lib.check(lib.gram_prof_pp_clock(C.byref(pp_ghz), 1), "pp_clock")
    lib.gram_prof_pp_clock_enable(0)
    kernel = {}
    if prof:
        for name, kind in kinds.items():
            ms, n, work, dropped = C.c_double(0), C.c_int64(0), C.c_double(0), C.c_int64(0)
            _lib.check(lib.gram_prof_collect(kind, C.byref(ms), C.byref(n), C.byref(work), C.byref(dropped)), "prof_collect")
            kernel[name] = dict(ms=ms.value, launches=n.value, work=work.value, dropped=dropped.value)
        lib.gram_prof_enable(0, 0)

    if rank != 0:
        if distributed:
            dist.destroy_process_group()
        return

    users = world * B * args.steps
    result = {
        "metric": "users/sec @ beam=20 top-20 gen, Beauty T5-base; Recall@5 parity",
        "value": users / dt,
        "unit": "users/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.precision,
        "data": "synthetic",
        "config": {
            "workload": f"{args.dataset} Trie ({len(cands)} items, T={max_length - 1}), {args.backbone}, "
                        f"{N} granularity passages x {L} tokens (S={N * L}), beam={K} top-{K}"
                        + (f"; item passages drawn from {args.item_pool} prompts and served by the passage cache "
                           f"(NOT the headline configuration)" if item_cache else ""),
            "ragged": bool(args.ragged), "item_cache": item_cache, "weights": f"random init (seed 2023), q projections x {args.q_sharpen:g}",
            "live_row_compaction": os.environ.get("GRAM_LIVE_ROWS", "1") != "0", "users_per_step_per_gpu": B, "parallelism": f"dp{world} (users sharded, no data-path collective)",
            "precision": {1: "one 16-bit piece per operand value / fp32 accumulate, fp32 residual stream, softmax and scores (misses the 1e-4 "
                             "metric bound: reported, not the headline)",
                          2: "every operand as 2 IEEE-half pieces (hi + lo), every product as 3 f16 MFMA products on one fetch of the operand "
                             "tiles (~2^-22 relative), fp32 accumulate / residual stream / softmax / scores; bank and activations stored as "
                             "hi + lo pieces"}[PIECES[args.precision]],
            "parity": "tests/test_gpu_precision.py: |dRecall@5|, |dNDCG@5| <= 1e-4 vs the fp32 reference arithmetic on 16 384 users in this "
                      "mode on each population (plain, attention-sharpened; 4 096 with ragged masks); profiles/r03*_precision_*.json",
        },
        "exchange": exchange,
        "output_check": {"sequences_shape": list(out["sequences"].shape),
                         "all_in_trie": bool(set(map(_strip, out["sequences"].cpu().tolist())) <= {tuple(c) for c in cands})},
    }
    if kernel:
        # decode steps that actually launched the cross-attention (the live-row compaction drops a step whose rows are all dead)
        result["config"]["cross_attn_launches_per_generate"] = kernel["cross_attn"]["launches"] // args.steps
        steps = args.steps
        result["kernel_ms_per_step"] = {k: round(v["ms"] / steps, 4) for k, v in kernel.items()}
        result["kernel_ms_per_step"]["sum"] = round(sum(v["ms"] for v in kernel.values()) / steps, 4)
        gm, xa = kernel["gemm"], kernel["cross_attn"]
        # gram_prof counts the EXECUTED MFMA flops (2*M*N*K' with K' = nprod * K in the split modes); the algorithmic flops of
        # the fp32 problem are 2*M*N*K.  `achieved` is the algorithmic rate, `achieved_mfma_executed` what the matrix pipe does.
        gemm_exec_tf = gm["work"] / (gm["ms"] * 1e-3) / 1e12 if gm["ms"] > 0 else 0.0
        gemm_tf = gemm_exec_tf / NPROD[args.precision]
        xa_gbs = xa["work"] / (xa["ms"] * 1e-3) / 1e9 if xa["ms"] > 0 else 0.0
        roof_gemm = {"kernel": "gemm_pp_kernel + gemm_dma_kernel + gemm_stream_kernel + gemm_skinny_kernel (every Linear of the path; all launches in the timed region)", "bound": "mfma", "achieved": gemm_tf, "peak": MFMA_BF16_PEAK_TF,
                     "unit": "TFLOP/s", "frac": gemm_tf / MFMA_BF16_PEAK_TF, "traffic": None,
                     "achieved_mfma_executed": gemm_exec_tf, "frac_mfma_executed": gemm_exec_tf / MFMA_BF16_PEAK_TF,
                     "mfma_products_per_product": NPROD[args.precision],
                     "note": "achieved = flops of the fp32 problem (2MNK) / time, priced against the dense 16-bit MFMA peak (2.5 PFLOP/s at 2.4 GHz); the "
                             "two-piece modes issue mfma_products_per_product MFMA products per product (achieved_mfma_executed); the native fp32 MFMA peak is 157 TFLOP/s",
                     # the chip is power-limited in these kernels: every workgroup of the ping-pong GEMM stamps s_memtime / s_memrealtime around
                     # its tile loop; this is the time-weighted clock over all their launches in the timed region (gram_prof_pp_clock), and the
                     # matrix peak at THAT clock (per-shape table and ablations: profiles/r03d_gemm_x3_ablation_and_clock.txt)
                     "clock_ghz_in_kernel": round(pp_ghz.value, 4) if pp_ghz.value > 0 else None,
                     "peak_at_measured_clock": MFMA_BF16_PEAK_TF * pp_ghz.value / 2.4 if pp_ghz.value > 0 else None,
                     "frac_mfma_executed_at_measured_clock": gemm_exec_tf / (MFMA_BF16_PEAK_TF * pp_ghz.value / 2.4) if pp_ghz.value > 0 else None,
                     "launches": gm["launches"], "avg_launch_us": 1e3 * gm["ms"] / max(gm["launches"], 1),
                     "share_of_kernel_time": gm["ms"] / max(sum(v["ms"] for v in kernel.values()), 1e-9)}
        roof_xa = {"kernel": "cross_attn_kernel", "bound": "hbm", "achieved": xa_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": xa_gbs / HBM_PEAK_GBS, "traffic": None, "launches": xa["launches"],
                   "avg_launch_us": 1e3 * xa["ms"] / max(xa["launches"], 1),
                   "algorithmic_bytes_per_launch": xa["work"] / max(xa["launches"], 1),
                   "share_of_kernel_time": xa["ms"] / max(sum(v["ms"] for v in kernel.values()), 1e-9)}
        # calibration: what a plain streaming read reaches on THIS box (4 GiB swept once per launch by every CU)
        try:
            from gram_amd import _lib as _L
            probe = torch.empty(4 << 30, dtype=torch.uint8, device=dev).fill_(1)
            st_ = torch.cuda.current_stream().cuda_stream
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            _L.check(_L.load().gram_debug_stream_read(probe.data_ptr(), probe.numel(), None, st_), "stream_read")
            e0.record()
            for _ in range(3):
                _L.check(_L.load().gram_debug_stream_read(probe.data_ptr(), probe.numel(), None, st_), "stream_read")
            e1.record()
            torch.cuda.synchronize()
            sweep = 3 * probe.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9
            roof_xa["stream_read_gbs_this_box"] = round(sweep, 1)
            roof_xa["frac_of_stream_read"] = round(xa_gbs / sweep, 4)
            del probe
        except Exception as ex:  # the probe is informational
            roof_xa["stream_read_gbs_this_box"] = None
            roof_xa["stream_read_error"] = str(ex)[:100]
        # HBM traffic per launch from the PMC counters (FETCH_SIZE/WRITE_SIZE, separate rocprofv3 passes of this
        # same command, gfx950 correction applied; profiles/*_pmc_traffic.json) -- only for the matching batch size
        try:
            pmc_files = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_traffic.json"))
            pmc = json.load(open(os.path.join(ROOT, "profiles", pmc_files[-1]))) if pmc_files else None
            if (pmc and not item_cache and not args.ragged and pmc.get("batch") == B and pmc.get("precision", "bf16") == args.precision
                    and args.backbone == "t5-base" and (N, L, K) == (3, 128, 20)):
                roof_xa["traffic"] = pmc["cross_attn_kernel"]["hbm_bytes_per_launch"]
                roof_xa["traffic_source"] = "profiles/" + pmc_files[-1]
                roof_gemm["traffic"] = pmc["gemm_all"]["hbm_bytes_per_launch_avg"]
                roof_gemm["traffic_source"] = "profiles/" + pmc_files[-1]
        except Exception:
            pass
        dominant = max(kernel.items(), key=lambda kv: kv[1]["ms"])[0]
        result["roofline"] = roof_xa if dominant == "cross_attn" else roof_gemm
        result["roofline_cross_attn"] = roof_xa
        result["roofline_gemm"] = roof_gemm
        # the north star's path-level figure: users/s as a fraction of what the cross-attention's HBM stream alone allows,
        # bytes per user = T * decoder layers * 2 (K, V) * S * inner * (2 B * pieces)   
        xa_bytes_user = (max_length - 1) * cfg.num_decoder_layers * 2 * (N * L) * cfg.num_heads * 64 * 2 * PIECES[args.precision]
        result["roofline_path"] = {"cross_attn_bytes_per_user": xa_bytes_user, "users_per_s_at_hbm_peak": world * HBM_PEAK_GBS * 1e9 / xa_bytes_user,
                                   "frac": result["value"] / (world * HBM_PEAK_GBS * 1e9 / xa_bytes_user),
                                   # the same under SURVEY.md §8(d)'s own definition (ONE 16-bit copy of K and V per user: 127.4 MB at this config)
                                   "cross_attn_bytes_per_user_16bit_kv": xa_bytes_user // PIECES[args.precision],
                                   "frac_16bit_kv_definition": result["value"] / (world * HBM_PEAK_GBS * 1e9 / (xa_bytes_user // PIECES[args.precision]))}

    if rank == 0 and world == 1 and not args.no_extras and not item_cache and not args.ragged:
        # secondary measurements, outside the timed region (value / ms_per_step above are the headline)
        def timed(fn_, n):
            torch.cuda.synchronize()
            t0_ = time.perf_counter()
            for _ in range(n):
                fn_()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0_) / n

        extras = {}
        # (1) every row decoded in every step, like the reference (no live-row compaction): with random-init weights all users
        # share nearly the same beams and the LAST step has no live row at all, a trained model keeps ~4 % of them
        lib.gram_debug_set_live_rows(0)
        step()
        extras["users_per_s_all_rows_decoded"] = B / timed(step, 2)
        lib.gram_debug_set_live_rows(-1)
        # (2) the reference's own operating point: --eval_batch_size 1
        one = lambda: model.generate(input_ids=ids_d[:1], attention_mask=mask_d[:1], max_length=max_length, prefix_allowed_tokens_fn=fn,
                                     num_beams=K, num_return_sequences=K, length_penalty=1.0)
        one()
        extras["batch1_ms_per_generate"] = 1e3 * timed(one, 10)
        # (3) the other precision modes on the same batch (weights re-packed; not the headline arithmetic)
        extras["users_per_s_other_modes"] = {}
        for mode in sorted(m_ for m_ in PIECES if m_.startswith("f16") == args.precision.startswith("f16")):
            if mode == args.precision:
                continue
            for Bm in (B, B // 2):  # (half the batch if the full one does not fit this mode's workspace)
                try:
                    model.set_precision(mode)
                    model._workspace = None
                    torch.cuda.empty_cache()
                    sub = lambda: model.generate(input_ids=ids_d[:Bm], attention_mask=mask_d[:Bm], max_length=max_length,
                                                 prefix_allowed_tokens_fn=fn, num_beams=K, num_return_sequences=K, length_penalty=1.0)
                    sub()
                    extras["users_per_s_other_modes"][mode] = {"users_per_s": Bm / timed(sub, 2), "users_per_step": Bm}
                    break
                except Exception as ex:  # the batch does not fit in this mode's workspace
                    extras["users_per_s_other_modes"][mode] = f"failed at B={Bm}: {str(ex)[:60]}"
        model.set_precision(args.precision)
        model._workspace = None
        torch.cuda.empty_cache()  # (before anything small is carved out of the freed block and pins it)
        # (4) SURVEY.md §8(d)'s second metric: end-to-end users/s of the DROP-IN RUNNER -- get_runner("single").test() as
        # main_generative_gram.py:107-127 calls it, with the reference's default --eval_batch_size 1 -- on a synthetic dataset
        # directory of Beauty's size (tools/synth_dataset.py: the real 12 101-item Trie, 22 363 users, N = 3 passages of 128 tokens)
        if (N, L, K) == (3, 128, 20) and args.dataset == "Beauty" and args.e2e_users > 0:
            try:
                extras["e2e_runner"] = e2e_runner(model, dev, args.e2e_users)
            except Exception as ex:  # secondary measurement: never costs the headline line
                extras["e2e_runner"] = {"error": f"{type(ex).__name__}: {str(ex)[:200]}"}
        result["extras"] = extras

    if state is not None:
        from oracle import gram_oracle as O
        oc = O.OracleConfig.named(args.backbone)
        ofn = O.prefix_allowed_tokens_fn(O.Trie(cands))
        threads = torch.get_num_threads()
        t0 = time.perf_counter()
        for u in range(args.cpu_users):  # B = 1 per call, like the reference runner (arguments.py:84-86)
            O.generate(state, oc, ids[u:u + 1], mask[u:u + 1], max_length, ofn, K, K, 1.0, reference_faithful=True)
        cdt = time.perf_counter() - t0
        result["cpu_baseline"] = {
            "value": args.cpu_users / cdt, "unit": "users/s", "cores": threads, "kind": "port",
            "sample": f"{args.cpu_users} users of the same workload, B=1 per call, oracle in reference-faithful mode "
                      f"(fp32, beam-replicated KV, per-step cache reorder), {cdt:.1f} s of CPU work, host has {os.cpu_count()} cpus",
        }
    print(json.dumps(result))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
