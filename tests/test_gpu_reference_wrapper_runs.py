"""The REFERENCE's OWN fa_vattn wrapper and vATTNCacheEngine, executed unmodified on the MI355X-native stack
(`vattention`, `flash_attn`, `sarathi.cache_ops` resolved to the drop-ins): engine.step + wrapper.forward for chunked prefills,
a hybrid iteration and decode batches, every output compared with the CPU oracle.
north_star: "the sarathi-lean attention-backend wrapper API stay[s] drop-in so sarathi-lean runs unmodified".
Follows /root/reference/sarathi-lean/sarathi/model_executor/attention/vattention_flashattention_wrapper.py:110-224 and
sarathi/worker/cache_engine/vATTN_cache_engine.py:91-124 — by RUNNING those lines (tests/ref_loader.py)."""
import pytest
import torch

from oracle.attn import flash_attn_with_kvcache_ref
from tests import ref_loader

pytestmark = pytest.mark.gpu
D = 128


@pytest.mark.skipif(not ref_loader.available(), reason="reference wrapper/engine not present (build oracle/_ref/pyref where /root/reference exists)")
@pytest.mark.parametrize("backend", ["fa_vattn", "fa_vattn_megacache"])
def test_reference_engine_and_wrapper_execute_on_the_native_stack(backend):
    from vattention_amd.replay import CacheConfig, ModelConfig, ParallelConfig, Sequence, SequenceMetadata
    Hq, Hkv, L = 8, 2, 3
    dev = torch.device("cuda:0")
    torch.zeros(1, device=dev)
    with ref_loader.loaded() as ref:
        print("reference modules loaded from:", ref.how)
        va = ref.vattention
        va.enable_layered_async(False)        # the reference wrapper does not gate layers: plain step_async semantics
        model = ModelConfig(name="tiny", num_layers=L, num_q_heads=Hq, num_kv_heads=Hkv, head_size=D, dtype=torch.float16,
                            max_model_len=4096, attention_backend=backend)
        par = ParallelConfig(1, 1)
        page = 2 << 20 if "megacache" in backend else 64 << 10
        group = 2 * page if "megacache" in backend else 2 * L * page
        cache = CacheConfig(page_size=page, max_batch_size=4, memory_for_gpu=48 * group)
        ref.wrapper.init(model, par, 0, dev)
        engine = ref.engine_mod.vATTNCacheEngine(cache, model, par, "async")
        try:
            assert type(engine).__module__.startswith("sarathi.worker.cache_engine")
            assert len(engine.gpu_cache) == L
            host = {}
            gen = torch.Generator(device="cuda")
            gen.manual_seed(99)

            def iteration(mds):
                T = sum(md.seq.get_next_prompt_chunk_len(md.prompt_chunk_len) if md.is_prompt else 1 for md in mds)
                q = torch.randn(T, Hq * D, device=dev, generator=gen).half()
                k = torch.randn(T, Hkv * D, device=dev, generator=gen).half()
                v = torch.randn(T, Hkv * D, device=dev, generator=gen).half()
                engine.step(mds)                                        # REFERENCE code: slots, step_async, set_batch_idx
                ref.wrapper.begin_forward(mds)
                outs = [ref.wrapper.forward(q * (1 + l), k, v * (1 + l), engine.gpu_cache[l], D ** -0.5, l) for l in range(L)]
                ref.wrapper.end_forward()
                torch.cuda.synchronize()
                qh, kh, vh = q.cpu(), k.cpu(), v.cpu()
                tok = 0
                for md in mds:
                    n = md.seq.get_next_prompt_chunk_len(md.prompt_chunk_len) if md.is_prompt else 1
                    for l in range(L):
                        kk, vv = kh[tok:tok + n].view(n, Hkv, D), (vh[tok:tok + n] * (1 + l)).view(n, Hkv, D)
                        pk, pv = host.get((md.seq.seq_id, l), (kk[:0], vv[:0]))
                        host[(md.seq.seq_id, l)] = (torch.cat([pk, kk]), torch.cat([pv, vv]))
                        fk, fv = host[(md.seq.seq_id, l)]
                        r = flash_attn_with_kvcache_ref((qh[tok:tok + n] * (1 + l)).view(1, n, Hq, D), fk.unsqueeze(0).clone(), fv.unsqueeze(0).clone(),
                                                        cache_seqlens=fk.shape[0], causal=True, softmax_scale=D ** -0.5)
                        got = outs[l][tok:tok + n].view(1, n, Hq, D).double().cpu()
                        err = (got - r).abs()
                        assert bool((err <= 4e-3 + 4e-3 * r.abs()).all()), "seq %d layer %d ctx %d: err %.3e" % (md.seq.seq_id, l, fk.shape[0], err.max().item())
                    tok += n
                for md in mds:
                    if md.is_prompt:
                        md.seq.prompt_processed += md.seq.get_next_prompt_chunk_len(md.prompt_chunk_len)
                        if md.seq.prompt_done:
                            md.seq.output_len += 1
                    else:
                        md.seq.output_len += 1
                engine.on_step_completion(mds)                          # REFERENCE code: free_request of finished sequences

            a, b, c = Sequence(0, 900, 905), Sequence(1, 333, 337), Sequence(2, 1500, 1503)
            iteration([SequenceMetadata(a, 512, True)])                 # chunked prefill (Sarathi): 512 + 388
            iteration([SequenceMetadata(a, 512, True)])
            iteration([SequenceMetadata(b, 333, True)])                 # whole prompt (vLLM)
            while not c.prompt_done:                                    # hybrid iterations: a chunk + the running decodes
                iteration([SequenceMetadata(c, 600, True), SequenceMetadata(a, 0, False), SequenceMetadata(b, 0, False)])
            for _ in range(3):
                iteration([SequenceMetadata(s, 0, False) for s in (a, b, c) if not s.is_finished()])
            assert engine.num_free_blocks() > 0
            st = va.stats()
            assert st["map_calls"] > 0
        finally:
            engine.cleanup_kvcache()
