"""Generate tests/golden/*.npz by running THE REFERENCE (imported read-only from /root/reference)
on synthetic weights/inputs.  Build-container only: /root/reference does not exist on the GPU
box, and nothing at test/bench time imports this script's dependencies.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

What is captured (inputs + expected outputs only -- never reference source):
  ref_tiny.npz      tiny GRAM: fused encoder output (ragged masks incl. a fully padded passage),
                    cached decode-step logits with _reorder_cache, reference Trie answers,
                    reference evaluate.py answers.
  ref_generate.npz  whole-path generate() = reference forward + reference Trie + the INSTALLED
                    transformers-5.15 beam search, on uniform-length Tries (where 4.26 and 5.15
                    searches coincide).
  ref_t5base.npz    T5-base-shaped slice: encoder rows + first decode-step logits.

Import shims (none touches the reference files): the reference imports two debug-only /
removed symbols that this image lacks (IPython.embed; transformers' model_parallel_utils,
find_pruneable_heads_and_indices, PreTrainedModel.get_head_mask).  They are never executed on
this path; empty stand-ins let the import succeed.
"""
import hashlib
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference/src"
sys.path.insert(0, ROOT)

from oracle import gram_oracle as O  # noqa: E402


def import_reference():
    ip = types.ModuleType("IPython")
    ip.embed = lambda *a, **k: None
    sys.modules.setdefault("IPython", ip)
    import transformers  # noqa: F401
    mpu = types.ModuleType("transformers.utils.model_parallel_utils")
    mpu.assert_device_map = lambda *a, **k: None
    mpu.get_device_map = lambda *a, **k: None
    sys.modules.setdefault("transformers.utils.model_parallel_utils", mpu)
    import transformers.pytorch_utils as pu
    if not hasattr(pu, "find_pruneable_heads_and_indices"):
        pu.find_pruneable_heads_and_indices = lambda *a, **k: None
    from transformers.modeling_utils import PreTrainedModel
    if not hasattr(PreTrainedModel, "get_head_mask"):
        PreTrainedModel.get_head_mask = lambda self, head_mask, n, *a, **k: [None] * n
    sys.path.insert(0, REF)
    import importlib.util
    import model.gram as gram_mod
    from model.gram_t5_config import T5Config

    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m

    trie_mod = load("ref_generation_trie", os.path.join(REF, "utils/generation_trie.py"))
    eval_mod = load("ref_evaluate", os.path.join(REF, "utils/evaluate.py"))
    return gram_mod, T5Config, trie_mod, eval_mod


def ref_model(gram_mod, T5Config, cfg: O.OracleConfig, sd, with_generate=False):
    hf = T5Config(
        vocab_size=cfg.vocab_size, d_model=cfg.d_model, d_kv=cfg.d_kv, d_ff=cfg.d_ff,
        num_layers=cfg.num_layers, num_decoder_layers=cfg.num_decoder_layers, num_heads=cfg.num_heads,
        relative_attention_num_buckets=cfg.relative_attention_num_buckets,
        relative_attention_max_distance=cfg.relative_attention_max_distance,
        layer_norm_epsilon=cfg.layer_norm_epsilon, feed_forward_proj="relu",
        decoder_start_token_id=0, tie_word_embeddings=cfg.tie_word_embeddings, pad_token_id=0, eos_token_id=1,
    )
    hf.max_seq_len = 128
    hf.max_item_num = cfg.max_item_num
    hf.use_position_embedding = cfg.use_position_embedding
    hf.sample_num = 1
    cls = gram_mod.GRAM
    if with_generate:
        from transformers import GenerationMixin

        class OracleGRAM(gram_mod.GRAM, GenerationMixin):
            pass

        cls = OracleGRAM
    m = cls(hf).eval()
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert not missing, missing
    return m


def sd_hash(sd) -> str:
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].contiguous().numpy().tobytes())
    return h.hexdigest()


def cfg_arrays(cfg: O.OracleConfig, seed: int, sd):
    return dict(
        cfg=np.array([cfg.vocab_size, cfg.d_model, cfg.d_kv, cfg.d_ff, cfg.num_layers, cfg.num_decoder_layers,
                      cfg.num_heads, cfg.max_item_num], dtype=np.int64),
        seed=np.array(seed), sd_sha256=np.array(sd_hash(sd)),
    )


def ragged_inputs(g, B, N, L, V, pad_passage=True):
    ids = torch.randint(2, V, (B, N, L), generator=g)
    mask = torch.zeros(B, N, L, dtype=torch.bool)
    for b in range(B):
        for n in range(N):
            ln = int(torch.randint(max(2, L // 3), L + 1, (1,), generator=g))
            if pad_passage and b == B - 1 and n == N - 1:
                ln = 0  # fully padded passage (Collator.py:410-436)
            mask[b, n, :ln] = True
            if ln > 0:
                ids[b, n, ln - 1] = 1  # forced EOS at the valid end (Collator.py:376-380)
            ids[b, n, ln:] = 0
    return ids, mask


def ref_decode_trace(m, enc, mask2, K, prefix_rows, beam_idx_seq):
    """Drive the reference forward step by step with its tuple cache and _reorder_cache."""
    B = enc.shape[0]
    from model.gram_t5_outputs import BaseModelOutput
    enc_rep = enc.repeat_interleave(K, dim=0)
    mask_rep = mask2.repeat_interleave(K, dim=0)
    past = None
    logits_all = []
    for t in range(prefix_rows.shape[1]):
        out = m(
            input_ids=None, attention_mask=mask_rep, decoder_input_ids=prefix_rows[:, t:t + 1],
            encoder_outputs=BaseModelOutput(last_hidden_state=enc_rep), past_key_values=past,
            use_cache=True, return_dict=True,
        )
        logits_all.append(out.logits[:, -1, :].clone())
        past = m._reorder_cache(out.past_key_values, beam_idx_seq[t])
    return torch.stack(logits_all, 0)


def main():
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    gram_mod, T5Config, trie_mod, eval_mod = import_reference()
    torch.set_num_threads(8)

    # ------------------------------------------------------------------ tiny model
    cfg = O.OracleConfig(vocab_size=256, d_model=128, d_kv=64, d_ff=256, num_layers=2, num_decoder_layers=2,
                         num_heads=2, max_item_num=5)
    seed = 11
    sd = O.init_state_dict(cfg, seed)
    m = ref_model(gram_mod, T5Config, cfg, sd)
    g = torch.Generator().manual_seed(5)
    B, N, L, K = 2, 3, 32, 3
    ids, mask = ragged_inputs(g, B, N, L, cfg.vocab_size)
    with torch.no_grad():
        m.encoder.n_passages = N
        enc = m.encoder(input_ids=ids.view(B, -1), attention_mask=mask.view(B, -1), return_dict=True)[0]
        T = 4
        prefix = torch.randint(2, cfg.vocab_size, (B * K, T), generator=g)
        prefix[:, 0] = 0
        beam_idx = [torch.cat([torch.randperm(K, generator=g) + b * K for b in range(B)]) for _ in range(T)]
        # rows after a reorder follow their parents: make the token stream consistent with that
        logits = ref_decode_trace(m, enc, mask.view(B, -1).float(), K, prefix, beam_idx)
    # reference Trie
    cands = [[0, 5, 6, 7, 1], [0, 5, 6, 8, 1], [0, 5, 9, 1], [0, 10, 11, 12, 1], [0, 10, 11, 1]]
    rt = trie_mod.Trie(cands)
    probes = [[0], [0, 5], [0, 5, 6], [0, 5, 9], [0, 10, 11], [0, 10, 11, 1], [0, 99], [3], [], [0, 5, 6, 7, 1]]
    trie_answers = [sorted(rt.get(p)) for p in probes]
    fn = trie_mod.prefix_allowed_tokens_fn(rt)
    assert sorted(fn(0, torch.tensor([0, 5]))) == trie_answers[1]
    # reference metrics: SURVEY A14 known answer + a random case
    rel_known = eval_mod.rel_results(["a", "b", "c", "d"], ["c"], [-1.0, -3.0, -2.0, -4.0], 4)
    met_known = eval_mod.get_metrics_results(rel_known, ["hit@1", "hit@2", "ndcg@2", "ndcg@4"])
    rng = np.random.default_rng(3)
    kk = 6
    preds = [str(x) for x in rng.integers(0, 8, size=4 * kk)]
    golds = [str(x) for x in rng.integers(0, 8, size=4)]
    scs = rng.normal(size=4 * kk).astype(np.float32)
    rel_rand = eval_mod.rel_results(preds, golds, scs, kk)
    mets = ["hit@1", "hit@5", "ndcg@3", "ndcg@6"]
    met_rand = eval_mod.get_metrics_results(rel_rand, mets)
    np.savez_compressed(
        os.path.join(out_dir, "ref_tiny.npz"),
        **cfg_arrays(cfg, seed, sd),
        input_ids=ids.numpy(), attention_mask=mask.numpy(), enc_fused=enc.numpy(),
        K=np.array(K), prefix=prefix.numpy(), beam_idx=torch.stack(beam_idx).numpy(), step_logits=logits.numpy(),
        trie_cands=np.array([c + [-1] * (5 - len(c)) for c in cands]), trie_probes=np.array(
            [p + [-1] * (5 - len(p)) for p in probes]),
        trie_answers=np.array([a + [-1] * (4 - len(a)) for a in trie_answers]),
        rel_known=np.array(rel_known), met_known=met_known,
        rand_preds=np.array(preds), rand_golds=np.array(golds), rand_scores=scs, rand_k=np.array(kk),
        rel_rand=np.array(rel_rand), met_rand=met_rand, rand_metrics=np.array(mets),
    )
    print("ref_tiny.npz: enc", tuple(enc.shape), "logits", tuple(logits.shape))

    # ------------------------------------------------------------------ whole-path generate (5.15 search)
    mg = ref_model(gram_mod, T5Config, cfg, sd, with_generate=True)
    g = torch.Generator().manual_seed(9)
    cases = {}
    for ci, (B, N, L, K, n_items, depth) in enumerate([(2, 3, 16, 4, 40, 3), (1, 2, 32, 6, 80, 4), (3, 1, 16, 5, 30, 2)]):
        ids, mask = ragged_inputs(g, B, N, L, cfg.vocab_size, pad_passage=False)
        items = set()
        while len(items) < n_items:
            items.add(tuple(int(x) for x in torch.randint(2, 40, (depth,), generator=g)))
        cands = [[0] + list(it) + [1] for it in sorted(items)]
        rt = trie_mod.Trie(cands)
        fn = trie_mod.prefix_allowed_tokens_fn(rt)
        with torch.no_grad():
            out = mg.generate(
                input_ids=ids, attention_mask=mask, max_length=depth + 2, prefix_allowed_tokens_fn=fn,
                num_beams=K, num_return_sequences=K, output_scores=True, return_dict_in_generate=True,
                length_penalty=1.0, use_cache=False, do_sample=False, early_stopping=False,
            )
        cases[f"c{ci}_ids"] = ids.numpy()
        cases[f"c{ci}_mask"] = mask.numpy()
        cases[f"c{ci}_cands"] = np.array(cands)
        cases[f"c{ci}_K"] = np.array(K)
        cases[f"c{ci}_sequences"] = out["sequences"].numpy()
        cases[f"c{ci}_scores"] = out["sequences_scores"].numpy()
        print(f"ref_generate c{ci}: seq", tuple(out["sequences"].shape), "top score", float(out["sequences_scores"][0]))
    np.savez_compressed(os.path.join(out_dir, "ref_generate.npz"), **cfg_arrays(cfg, seed, sd), n_cases=np.array(3), **cases)

    # ------------------------------------------------------------------ T5-base-shaped slice
    cfgb = O.OracleConfig.named("t5-base")
    seedb = 2023
    sdb = O.init_state_dict(cfgb, seedb)
    mb = ref_model(gram_mod, T5Config, cfgb, sdb)
    g = torch.Generator().manual_seed(17)
    B, N, L, K = 1, 3, 32, 2
    ids, mask = ragged_inputs(g, B, N, L, 32100, pad_passage=False)
    with torch.no_grad():
        mb.encoder.n_passages = N
        enc = mb.encoder(input_ids=ids.view(B, -1), attention_mask=mask.view(B, -1), return_dict=True)[0]
        prefix = torch.tensor([[0, 77], [0, 4242]])
        beam_idx = [torch.tensor([0, 1]), torch.tensor([1, 0])]
        logits = ref_decode_trace(mb, enc, mask.view(B, -1).float(), K, prefix, beam_idx)
    tok_slice = torch.randint(0, 32128, (64,), generator=g)
    np.savez_compressed(
        os.path.join(out_dir, "ref_t5base.npz"),
        **cfg_arrays(cfgb, seedb, sdb),
        input_ids=ids.numpy(), attention_mask=mask.numpy(),
        enc_rows=enc[0, ::8].numpy(), enc_row_stride=np.array(8),
        K=np.array(K), prefix=prefix.numpy(), beam_idx=torch.stack(beam_idx).numpy(),
        tok_slice=tok_slice.numpy(), step_logits_slice=logits[:, :, tok_slice].numpy(),
        step_lse=torch.logsumexp(logits.float(), -1).numpy(), step_argmax=logits.argmax(-1).numpy(),
    )
    print("ref_t5base.npz: enc", tuple(enc.shape))


if __name__ == "__main__":
    main()
