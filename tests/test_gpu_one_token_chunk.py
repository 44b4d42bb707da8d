"""One-token prefill chunks through the attention WRAPPER and the cache engine (a Sarathi edge case: a prompt whose remainder after
whole chunks is a single token, /root/reference/sarathi-lean/sarathi/model_executor/attention/vattention_flashattention_wrapper.py:
151-166 — cache_flat of one row, then flash_attn_with_kvcache with q [1, 1, Hq, D], causal, no k / v).  Every layer of every
iteration against the oracle, per-layer and megacache layouts:
  * a prompt chunked 512 + 512 + 1;
  * a hybrid iteration whose prefill part is that one token, beside running decodes;
  * an iteration with two prompts' chunks of which one is a single token (the batched variable-length launch with q_lens = 1);
  * an iteration with two one-token chunks (the wrapper's per-prompt path)."""
import pytest
import torch

from oracle.attn import flash_attn_with_kvcache_ref

pytestmark = pytest.mark.gpu
D, Hq, Hkv = 128, 8, 2


def _drive(backend, num_layers, page, schedule, seqs_spec, pool_groups=48):
    """schedule: list of iterations, each a list of (seq index, chunk length or 0 for a decode step)."""
    from vattention_amd.replay import CacheConfig, HotPathRunner, ModelConfig, ParallelConfig, Sequence, SequenceMetadata
    model = ModelConfig(name="tiny", num_layers=num_layers, num_q_heads=Hq, num_kv_heads=Hkv, head_size=D, dtype=torch.float16,
                        max_model_len=4096, attention_backend=backend)
    group = 2 * page if "megacache" in backend else 2 * num_layers * page
    r = HotPathRunner(model, ParallelConfig(1, 1), CacheConfig(page_size=page, max_batch_size=4, memory_for_gpu=pool_groups * group), seed=3)
    r.sample_kv_util = False
    host = {}
    seqs = [Sequence(i, p, t) for i, (p, t) in enumerate(seqs_spec)]
    try:
        for it, entries in enumerate(schedule):
            mds = [SequenceMetadata(seqs[i], n, n > 0) for i, n in entries]
            lens = [md.seq.get_next_prompt_chunk_len(md.prompt_chunk_len) if md.is_prompt else 1 for md in mds]
            T = sum(lens)
            q, k, v = r._qkv(T)
            outs = []
            with torch.cuda.stream(r.stream):
                r.engine.step(mds)
                r.wrapper.begin_forward(mds)
                for l in range(num_layers):
                    s = 1.0 + l / 4.0
                    outs.append(r.wrapper.forward((q * s).half(), (k * s).half(), (v * s).half(), r.engine.gpu_cache[l], r.scale, l))
                r.wrapper.end_forward()
            torch.cuda.synchronize()
            qh, kh, vh = q.float().cpu(), k.float().cpu(), v.float().cpu()
            tok = 0
            for md, n in zip(mds, lens):
                for l in range(num_layers):
                    s = 1.0 + l / 4.0
                    kk = (kh[tok:tok + n] * s).half().view(n, Hkv, D)
                    vv = (vh[tok:tok + n] * s).half().view(n, Hkv, D)
                    pk, pv = host.get((md.seq.seq_id, l), (kk[:0], vv[:0]))
                    host[(md.seq.seq_id, l)] = (torch.cat([pk, kk]), torch.cat([pv, vv]))
                    fk, fv = host[(md.seq.seq_id, l)]
                    ref = flash_attn_with_kvcache_ref((qh[tok:tok + n] * s).half().view(1, n, Hq, D), fk.unsqueeze(0).clone(), fv.unsqueeze(0).clone(),
                                                      cache_seqlens=fk.shape[0], causal=True, softmax_scale=D ** -0.5)
                    got = outs[l][tok:tok + n].view(1, n, Hq, D).double().cpu()
                    err = (got - ref).abs()
                    assert bool((err <= 4e-3 + 4e-3 * ref.abs()).all()), "iteration %d seq %d layer %d (%d tokens at context %d): max err %.3e" % (
                        it, md.seq.seq_id, l, n, fk.shape[0], err.max().item())
                tok += n
            for md, n in zip(mds, lens):
                if md.is_prompt:
                    md.seq.prompt_processed += n
                    if md.seq.prompt_done:
                        md.seq.output_len += 1
                else:
                    md.seq.output_len += 1
            with torch.cuda.stream(r.stream):
                r.engine.on_step_completion(mds)
    finally:
        r.close()


@pytest.mark.parametrize("backend,page", [("fa_vattn", 64 << 10), ("fa_vattn_megacache", 2 << 20)], ids=["per_layer", "megacache"])
def test_prompt_whose_last_chunk_is_one_token(backend, page):
    # 1025 = 512 + 512 + 1, then two decode steps
    _drive(backend, 3, page, [[(0, 512)], [(0, 512)], [(0, 512)], [(0, 0)], [(0, 0)]], [(1025, 1030)])


@pytest.mark.parametrize("backend", ["fa_vattn", "fa_streams", "fa_pod"])
def test_hybrid_iteration_with_a_one_token_chunk(backend):
    # sequence 0 is decoding while sequence 1's prompt (257 tokens) arrives in chunks of 256: the second chunk is ONE token
    sched = [[(0, 300)], [(1, 256), (0, 0)], [(1, 256), (0, 0)], [(0, 0), (1, 0)]]
    _drive(backend, 2, 64 << 10, sched, [(300, 310), (257, 262)])


def test_batched_chunks_of_which_one_is_a_single_token():
    # two prompts' chunks in one iteration: 130 tokens and 1 token -> the batched variable-length launch with q_lens = [130, 1]
    sched = [[(0, 256)], [(1, 384)], [(0, 256), (1, 384)], [(0, 0), (1, 0)]]
    _drive("fa_vattn", 2, 64 << 10, sched, [(386, 390), (385, 390)])


def test_two_one_token_chunks_in_one_iteration():
    sched = [[(0, 200)], [(1, 100)], [(0, 200), (1, 100)], [(0, 0), (1, 0)]]
    _drive("fa_vattn", 2, 64 << 10, sched, [(201, 204), (101, 104)])
