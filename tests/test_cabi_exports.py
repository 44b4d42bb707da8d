"""The C-ABI shared library loads (no GPU needed) and exports every function include/*.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set(re.findall(r"\b(vattn_[a-z_0-9]+)\s*\(", src))
    # function-pointer fields / typedef'd structs are not exports
    return {n for n in names if not n.endswith("_ops") and n not in ("vattn_config", "vattn_layout", "vattn_stats")}


def test_library_exports_every_declared_symbol():
    from vattention_amd import _lib
    lib = _lib.lib()
    missing = []
    for header in ("vattn.h", "vattn_kernels.h"):
        names = _declared(header)
        assert len(names) >= 6
        for n in sorted(names):
            try:
                getattr(lib, n)
            except AttributeError:
                missing.append(n)
    assert not missing, "declared but not exported: %s" % missing


def test_params_struct_matches_header_size():
    """ctypes mirror of vattn_attn_params must have the C layout (checked against a compiled sizeof)."""
    import subprocess
    import tempfile
    from vattention_amd import kernels as K
    src = '#include "%s/include/vattn_kernels.h"\n#include <stdio.h>\n#include <stddef.h>\nint main(){printf("%%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu", sizeof(vattn_attn_params), offsetof(vattn_attn_params, b), offsetof(vattn_attn_params, softmax_scale), offsetof(vattn_attn_params, split_items), offsetof(vattn_attn_params, pf_items), offsetof(vattn_attn_params, pf_part_rows), sizeof(vattn_decode_item), sizeof(vattn_prefill_item));return 0;}\n' % ROOT
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", c, "-o", exe])
        size, off_b, off_sc, off_si, off_pf, off_pr, sz_di, sz_pi = [int(x) for x in subprocess.check_output([exe]).decode().split()]
    assert ctypes.sizeof(K.AttnParams) == size
    assert K.AttnParams.b.offset == off_b and K.AttnParams.softmax_scale.offset == off_sc
    assert K.AttnParams.split_items.offset == off_si and K.AttnParams.pf_items.offset == off_pf and K.AttnParams.pf_part_rows.offset == off_pr
    assert ctypes.sizeof(K.DecodeItem) == sz_di == 16 and ctypes.sizeof(K.PrefillItem) == sz_pi == 32


def test_params_block_carries_its_size_and_abi_version():
    """the block grows between releases and the kernels branch on pointers inside it.  A caller built against another
    header (wrong struct_size / abi_version), or one that forgot to set them, is refused by every entry point that takes the block
    — never read past."""
    from vattention_amd import kernels as K
    lib = K.klib()
    p = K.AttnParams()
    assert p.struct_size == ctypes.sizeof(K.AttnParams) and p.abi_version == K.ABI_VERSION
    p.b, p.seqlen_q, p.seqlen_k, p.h, p.h_k, p.d = 4, 1, 4096, 8, 2, 128
    assert lib.vattn_attn_workspace_bytes(ctypes.byref(p)) > 0
    for field, bad in (("struct_size", p.struct_size - 16), ("abi_version", K.ABI_VERSION - 1), ("struct_size", 0)):
        q = K.AttnParams()
        q.b, q.seqlen_q, q.seqlen_k, q.h, q.h_k, q.d = 4, 1, 4096, 8, 2, 128
        setattr(q, field, bad)
        assert lib.vattn_attn_workspace_bytes(ctypes.byref(q)) == 0
        assert lib.vattn_flash_attn_with_kvcache(ctypes.byref(q), None) == -11
        assert b"struct_size" in lib.vattn_kernels_last_error()
        assert lib.vattn_attn_plan_describe(ctypes.byref(q), ctypes.byref(K.PlanDesc())) == -11
        assert lib.vattn_decode_plan(ctypes.byref(q), (ctypes.c_int32 * 4)(1, 2, 3, 4), (K.DecodeItem * 8)(), 8, (ctypes.c_int32 * 8)()) == -11


def test_no_gpu_calls_fail_loudly():
    """Without a device the product refuses to run instead of falling back."""
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from vattention_amd.cache_ops import cache_flat
    from vattention_amd.flash_attn import flash_attn_with_kvcache
    q = torch.zeros(1, 1, 2, 128, dtype=torch.float16)
    kc = torch.zeros(1, 8, 2, 128, dtype=torch.float16)
    with pytest.raises(RuntimeError, match="no CPU path"):
        flash_attn_with_kvcache(q, kc, kc, cache_seqlens=4)
    with pytest.raises(RuntimeError, match="no CPU path"):
        cache_flat(kc[0], kc[0], kc[0], kc[0], "auto")


def test_plain_c_client_drives_the_boundary(tmp_path):
    """include/*.h are valid C99 and libvattn_amd.so is usable from plain C (what a cgo / JNI / FFI binding sees): a gcc-built
    client runs init -> reserve -> alloc -> step_async -> free -> cleanup on the fake backend and validates kernel arguments."""
    import subprocess
    exe = str(tmp_path / "cabi_client")
    lib_dir = os.path.join(ROOT, "vattention_amd")
    fake_dir = os.path.join(ROOT, "tests", "native")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(fake_dir, "cabi_client.c"), "-L" + lib_dir, "-lvattn_amd", "-L" + fake_dir,
                           "-lvattn_fake_backend", "-Wl,-rpath," + lib_dir, "-Wl,-rpath," + fake_dir, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    lines = [ln for ln in out.stdout.splitlines() if not ln.startswith("[")]
    text = "\n".join(lines)
    assert "tensors 4 ndim 4" in text and "pool 64" in text and "slot 0" in text
    # 300 tokens = 3 pages of 128 tokens now + look-ahead; 2 layers x (K, V) = 4 pages per group
    assert "step_async 0" in text and "pool_pages 52 mapped_groups 3 needed_groups 3 active_slots 1 map_calls 12" in text
    assert "bad_len -1 err 'seq_lens must have max_batch_size entries'" in text
    assert "cleanup 0" in text and "null_params" in text and "null tensor pointer" in text
    import re
    m = re.search(r"decode_plan items (\d+) first_seq_pieces (\d+) last_seq_pieces (\d+) longest_piece_tiles (\d+)", text)
    assert m and 64 < int(m.group(1)) <= 768 and int(m.group(2)) == 1 and int(m.group(3)) > 4 and int(m.group(4)) < 902 // 3, text
    m = re.search(r"prefill_plan items (\d+) split_blocks (\d+) partial_rows (\d+) first_piece_tiles (\d+) last_piece_tiles (\d+)", text)
    assert m and int(m.group(1)) > 256 and int(m.group(2)) > 0 and int(m.group(3)) % 256 == 0 and int(m.group(4)) >= int(m.group(5)), text
    m2 = re.search(r"prefill_plan_wg items (\d+) queues (\d+) first (\d+) last (\d+) nonempty (\d+)", text)
    # (priced for 64 queues and the chained per-piece overhead: its own cuts) every queue non-empty, the offsets span the list
    assert m2 and int(m2.group(1)) >= 256 and int(m2.group(2)) == 64 and int(m2.group(3)) == 0 and int(m2.group(4)) == int(m2.group(1)) and m2.group(5) == "1", text
    m3 = re.search(r"prefill_plan_drawn items (\d+) queues (\d+)", text)
    assert m3 and int(m3.group(1)) >= 256 and int(m3.group(2)) == 256, text
    assert "pool_ready 0" in text
    assert "lab_variant -11" in text and "measurement build" in text
    # vattn_attn_plan_describe from plain C: configs[1]'s prompt = prefill64 over a 4096-workgroup grid, its batch-16 decode step = the
    # device-planned stream decomposition on 768 workgroups + a merge launch; another header's block is refused
    assert "describe_prefill 0 form 0 path 0 tiling 7 nsplit 1 workgroups 4096 merge 0" in text
    assert "describe_decode 0 form 1 path 2 tiling 1 workgroups 768 merge 1 workspace 6922368" in text
    assert "describe_other_abi -11" in text


def test_split_plans_from_the_workspace_query():
    import ctypes as C
    """vattn_attn_workspace_bytes is pure host code: it exposes the split decisions (decode split-KV, prefill KV split) without
    a GPU.  Pins the documented plans (DESIGN.md §5 / §6b) so a heuristic change shows up here."""
    from vattention_amd import kernels as K
    lib = K.klib()

    def ws(b, sq, sk, h, hk, d=128, hint=0, causal=1, splits=0, variant=0):
        p = K.AttnParams()
        p.b, p.seqlen_q, p.seqlen_k, p.seqlen_knew, p.h, p.h_k, p.d = b, sq, sk, 0, h, hk, d
        p.is_causal, p.dtype, p.num_splits, p.variant, p.max_seqlen_k_hint = causal, 0, splits, variant, hint
        return lib.vattn_attn_workspace_bytes(C.byref(p))

    def dec_splits(b, ctx, h, hk, d=128):      # (variant bit 19: the grid heuristics; the product default for b >= 2 is pinned in test_plan_table.py)
        n = ws(b, 1, ctx, h, hk, d, variant=K.LEGACY_DECODE_PLAN)
        return n // (b * h * (d + 1) * 4) if n else 1

    def pf_splits(sq, sk, h, hk, hint, b=1, d=128, **kw):
        n = ws(b, sq, sk, h, hk, d, hint=hint, **kw)
        return n // (b * sq * h * (d + 1) * 4) if n else 1

    # decode: fill the 768 resident workgroups, never more than 48 splits, none when the batch alone fills the chip
    assert dec_splits(16, 32768, 32, 4) == 12           # c2: 16 x 4 kv heads x 12 = 768
    assert dec_splits(1, 32768, 32, 4) == 48
    assert dec_splits(4, 32768, 32, 4) == 48
    assert dec_splits(1, 131072, 28, 4) == 64           # one 128 k sequence, 4 kv heads (Yi-34B/TP2): measured over rotating caches (round 4)
    assert dec_splits(1, 131072, 14, 2) == 96           # ... 2 kv heads (TP4 shard)
    assert dec_splits(8, 131072, 28, 4) == 24           # batch 8 at 128 k: 32 groups x 24 = 768, as before
    assert dec_splits(256, 2048, 32, 8) == 1
    assert dec_splits(1, 2048, 32, 4) == 16             # one 32-key tile per wave and split at most
    # prefill: shapes that fill the chip stay single-pass
    assert pf_splits(32702, 32768, 32, 4, 32702) == 1   # c2 whole prompt
    assert pf_splits(4096, 32768, 32, 4, 32768) == 1    # 4k chunk, 32 heads: 512 workgroups
    assert pf_splits(16384, 131072, 14, 2, 131072) == 2 # Yi-34B/TP4 16k chunk @ 112k: 896 equal workgroups = 3.5 rounds -> 7 whole rounds
    assert pf_splits(16384, 131072, 28, 4, 131072) == 1 # Yi-34B/TP2: 1792 workgroups = 7 whole rounds already
    assert pf_splits(2048, 2048, 32, 4, 2048) == 1      # 2k prompt, 32 heads: also 256 workgroups, too short to split
    assert pf_splits(8192, 8192, 8, 1, 8192) == 2       # TP8 8k prompt: exactly one 8-wave workgroup per CU, causal whole prompt -> two key-range shares
    # tensor-parallel shards / short chunks on long prefixes are split
    assert pf_splits(2048, 32768, 8, 1, 32768) == 4     # 64 workgroups x 4 = one round
    assert pf_splits(2048, 32768, 8, 1, 0) == 4          # without a host-side bound the view's row count stands in (as in FlashAttention)
    assert pf_splits(512, 16384, 8, 1, 16384) == 8
    assert 2 <= pf_splits(1024, 65536, 28, 4, 65536) <= 8
    lib = K.klib_lab()                                               # (a lab-only kernel: tools/lab/libvattn_lab.so)
    assert pf_splits(2048, 32768, 8, 1, 32768, variant=12) == 1      # the interleaved kernel has no split epilogue
    lib = K.klib()
    assert pf_splits(300, 1400, 8, 2, 1200, splits=5) == 5           # forced


def test_decode_plan_cuts_a_ragged_batch_into_equal_work_items():
    """vattn_decode_plan (pure host code): every sequence's 32-key tiles are covered exactly once by contiguous, ordered pieces; the
    longest piece is far shorter than the longest sequence's share under the uniform split; equal-length batches and single
    sequences keep the uniform split (0 items); at most 128 pieces per sequence."""
    import ctypes as C
    import random
    from vattention_amd import kernels as K
    lib = K.klib()

    def plan(lens, h, hk, knew=1, variant=0):
        p = K.AttnParams()
        p.b, p.seqlen_q, p.seqlen_k, p.seqlen_knew, p.h, p.h_k, p.d = len(lens), 1, 32768, knew, h, hk, 128
        p.variant = variant
        cap = 4 * len(lens) + 1024
        items = (K.DecodeItem * cap)()
        seq = (C.c_int32 * (2 * len(lens)))()
        n = lib.vattn_decode_plan(C.byref(p), (C.c_int32 * len(lens))(*lens), items, cap, seq)
        return n, [(items[i].b, items[i].tile_begin, items[i].tile_end, items[i].index_in_seq) for i in range(max(n, 0))], list(seq)

    rng = random.Random(5)
    # one TP=8 rank of Llama-3-70B: the 256 sequences of the reference's dynamic trace (prompt lengths 4 k .. 29 k, median 7 k) on one
    # kv head, each a few hundred tokens into its decode phase
    import json
    import os
    trace = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c3_arxiv_lengths_256.json")))["requests"]
    lens = [pre + rng.randint(1, 300) for pre, _ in trace]
    n, items, seq = plan(lens, 8, 1)
    assert 700 < n <= 768, n                                     # one round of the 768 resident workgroups, filled
    OPEN = 0x7fffffff                                            # the last piece of a sequence is open-ended (clamped on the device)
    longest = max(min(te, (lens[b] + 1 + 31) // 32) - tb for b, tb, te, _ in items)
    uniform = max((l + 1 + 31) // 32 for l in lens) / 3.0        # the uniform heuristic gives 256 x 3 = 768 workgroups
    assert longest * 1.8 < uniform, (longest, uniform)
    for b, l in enumerate(lens):
        first, cnt = seq[2 * b], seq[2 * b + 1]
        tiles = (l + 1 + 31) // 32
        mine = items[first:first + cnt]
        assert 1 <= cnt <= 128 and all(it[0] == b for it in mine) and [it[3] for it in mine] == list(range(cnt))
        assert mine[0][1] == 0 and mine[-1][2] == OPEN and mine[-1][1] < tiles and all(a[2] == c[1] for a, c in zip(mine, mine[1:]))
    assert sum(seq[1::2]) == n
    # Llama-3-8B, 8 kv heads, 200 sequences: the batch alone exceeds a round; long sequences are cut to about the mean length
    lens = [rng.randint(4000, 32000) for _ in range(200)]
    n, items, seq = plan(lens, 32, 8)
    assert 300 < n <= 384, n                                      # four whole rounds: 384 pieces x 8 kv heads = 3072 workgroups
    # equal lengths, a single sequence, a nearly uniform batch: keep the uniform split
    assert plan([32767] * 16, 32, 4)[0] == 0
    assert plan([20000], 32, 4)[0] == 0
    assert plan([30000 + i for i in range(16)], 32, 4)[0] == 0
    # a prefill-form call has no plan
    p = K.AttnParams()
    p.b, p.seqlen_q, p.h, p.h_k, p.d = 4, 5, 8, 2, 128
    assert lib.vattn_decode_plan(C.byref(p), (C.c_int32 * 4)(1, 2, 3, 4), (K.DecodeItem * 8)(), 8, (C.c_int32 * 8)()) == 0
    # one 128 k sequence beside short ones: never more than 128 pieces of a sequence
    n, items, seq = plan([131000] + [600] * 40, 8, 1)
    assert n > 0 and max(seq[1::2]) <= 128


def test_prefill_plan_lists_every_key_tile_once_and_cuts_only_long_blocks():
    """vattn_prefill_plan (pure host code): the pieces of every (entry, head, 256-row query block) cover its key tiles exactly once;
    pieces are listed longest first; only blocks longer than the per-CU average are cut (at most 16 shares), each split block owns
    256 partial rows per share; grids of several balanced rounds, d != 128 and the decode form keep the default launch."""
    import ctypes as C
    from vattention_amd import kernels as K
    lib = K.klib()

    def plan(q_lens, k_lens, h, hk, causal=1, d=128, uniform_sq=None):
        B = len(k_lens)
        p = K.AttnParams()
        p.b, p.seqlen_q, p.h, p.h_k, p.d, p.is_causal = B, (uniform_sq or max(q_lens)), h, hk, d, causal
        nblk = sum((q + 255) // 256 for q in (q_lens or [uniform_sq] * B)) * h
        ci, cb = 17 * nblk + 16, nblk + 16
        items, blocks, counts = (K.PrefillItem * ci)(), (K.PrefillItem * cb)(), (C.c_int32 * 3)()
        ql = (C.c_int32 * B)(*q_lens) if q_lens else None
        n = lib.vattn_prefill_plan(C.byref(p), ql, (C.c_int32 * B)(*k_lens), items, ci, blocks, cb, counts)
        f = lambda a, m: [(a[i].b, a[i].h, a[i].qb, a[i].tile_begin, a[i].tile_end, a[i].nshares, a[i].part_row) for i in range(m)]
        return n, f(items, max(n, 0)), f(blocks, counts[1]), list(counts)

    def tiles(sq, lk, qb, causal=True):
        n_end = min(lk, qb * 256 + 256 + (lk - sq)) if causal else lk
        return (max(n_end, 0) + 63) // 64

    def check(q_lens, k_lens, h, hk):
        n, items, blocks, counts = plan(q_lens, k_lens, h, hk)
        assert n > 0 and counts[0] == n and counts[1] == len(blocks)
        # a block's last share is open-ended (tile_end = INT32_MAX: the kernel clamps to the tiles the device-side lengths give)
        OPEN = 0x7fffffff
        items = [(b, hh, qb, tb, (tiles(q_lens[b], k_lens[b], qb) if te == OPEN else te), ns, pr) for b, hh, qb, tb, te, ns, pr in items]
        lens = [te - tb for _, _, _, tb, te, _, _ in items]
        assert lens == sorted(lens, reverse=True)
        cover = {}
        for b, hh, qb, tb, te, ns, pr in items:
            cover.setdefault((b, hh, qb), []).append((tb, te, ns, pr))
        W = 0
        for e, (sq, lk) in enumerate(zip(q_lens, k_lens)):
            for qb in range((sq + 255) // 256):
                t = tiles(sq, lk, qb)
                for hh in range(h):
                    pcs = sorted(cover.pop((e, hh, qb)))
                    assert pcs[0][0] == 0 and pcs[-1][1] == t and all(a[1] == c[0] for a, c in zip(pcs, pcs[1:])), (e, hh, qb, pcs, t)
                    assert all(x[2] == len(pcs) for x in pcs) and len(pcs) <= 16
                    if len(pcs) > 1:
                        rows = sorted(x[3] for x in pcs)
                        assert rows == [rows[0] + 256 * i for i in range(len(pcs))] and (e, hh, qb, 0, 0, len(pcs), rows[0]) in blocks
                    else:
                        assert pcs[0][3] == -1
                    W += t
        assert not cover
        assert counts[2] == 256 * sum(ns for *_, ns, _ in blocks)
        return n, lens, counts

    # one TP=8 rank of Llama-3-70B (8 query heads): whole prompts of 8 k / 4 k / 2 k, a 7 k prompt, a batch of three prompts
    n, lens, counts = check([8192], [8192], 8, 1)
    assert 256 < n < 700 and max(lens) <= 90                   # the short blocks stay whole, the long ones are cut (round 5: in two, at 0.7 x the longest)
    check([7344], [7344], 8, 1)
    check([4096], [4096], 8, 1)
    assert plan([2048], [2048], 8, 1)[0] == 0                   # no key walk of 48 tiles: default launch
    check([12001, 900, 600], [12001, 900, 600], 8, 1)           # one long prompt beside short ones: its last blocks outlast the average
    # three rounds of blocks, none longer than a CU's share: nothing is cut, but a RAGGED batch still gets a list — the default grid's
    # exit workgroups starve part of the chip (static striping over shader engines) — of exactly the valid blocks
    n, lens, counts = check([12001, 4119, 7000], [12001, 4119, 7000], 8, 1)
    assert n == (47 + 17 + 28) * 8 and counts[1] == 0 and counts[2] == 0
    n, lens, counts = check([7344, 8347, 7339, 5353], [7344, 8347, 7339, 5353], 32, 8)      # Llama-3-8B heads: XCD-affine head order
    assert counts[1] == 0
    assert plan([8192, 8192], [8192, 8192], 32, 8)[0] == 0                                   # equal lengths, balanced: default grid
    check([2048], [32768], 8, 1)                                # a 2 k chunk on a 30 k prefix: 64 equal blocks, four shares each
    check([300, 1, 700], [9300, 50, 700], 4, 2)                 # ragged small chunks (incl. a one-token entry), one on a long prefix
    # keep the default launch: a grid of many balanced rounds, d = 64, the decode form
    assert plan([32702], [32702], 32, 4)[0] == 0
    assert plan([16384], [131072], 28, 4)[0] == 0
    assert plan([4096], [4096], 8, 1, d=64)[0] == 0
    assert plan(None, [5000, 6000], 8, 1, uniform_sq=1)[0] == 0


def test_product_library_holds_no_measurement_scaffolding():
    """The shipped libvattn_amd.so instantiates the kernels the launch plans choose and nothing else: ONE prefill64 build per dtype
    (the timing ablations with wrong results, the alternative schedules, the hand-interleaved 8-wave kernel, the plain-read operand
    paths and the in-launch merge protocols live in tools/lab/libvattn_lab.so, built with -DVATTN_LAB), and it reads no environment
    variable to pick a kernel build."""
    import re
    import subprocess
    from vattention_amd import _lib as L
    from vattention_amd import kernels as K
    L.lib()
    so = L.LIB_PATH if hasattr(L, "LIB_PATH") else None
    if so is None:
        import os
        so = os.path.join(os.path.dirname(os.path.abspath(L.__file__)), "libvattn_amd.so")
    syms = subprocess.run(["nm", "--defined-only", so], capture_output=True, text=True, check=True).stdout      # (mangled names)
    # prefill64_kernel<T> (round 6): the product source carries ONE schedule as constants — no ablation / schedule / stamp template
    # parameters at all (those live in tools/lab/csrc/prefill64_lab.hip, which only the lab library compiles)
    p64 = set(re.findall(r"prefill64_kernelI(\w+?)EEv", syms))
    assert p64 == {"DF16_", "DF16b"}, p64
    assert "prefill_ilv_kernel" not in syms
    # prefill_kernel<T, HD, USE_TR, WAVES, QC, MSUM> / decode_kernel<T, HD, USE_TR, NB, W>: plain-read (USE_TR = false) operand paths
    # and the row-sums-by-MFMA build are lab-only
    pk = re.findall(r"prefill_kernelI(?:DF16_|DF16b)Li\d+ELb(\d)ELi\d+ELi\d+ELb(\d)E", syms)
    assert pk and all(tr == "1" and msum == "0" for tr, msum in pk), pk
    dk = re.findall(r"decode_kernelI(?:DF16_|DF16b)Li\d+ELb(\d)E", syms)
    assert dk and all(tr == "1" for tr in dk), dk
    strings = subprocess.run(["strings", so], capture_output=True, text=True, check=True).stdout
    assert "VATTN_PREFILL64_BUILD" not in strings
    # which variants need the lab library
    assert not K.needs_lab(0) and not K.needs_lab(14) and not K.needs_lab(8) and not K.needs_lab(2 | 64)
    for v in (1, 4, 12, 16, 782, 526, 512, 1024, 16384, 32768, 65536, 131072, 262144, 14 | (1 << 28), 14 | (7 << 28)):
        assert K.needs_lab(v), v


def test_layout_policy_moves_huge_pools_to_megacache():
    """vattention_amd/policy.py: > 100 k handles -> megacache + >= 8 MiB pages unless the configured layout is pinned."""
    from vattention_amd.policy import choose_layout
    assert choose_layout(2 << 20, False, 64 << 30) == (2 << 20, False, "configured")                 # 32 k handles
    page, mega, what = choose_layout(2 << 20, False, 250 << 30)                                       # 128 k handles
    assert (page, mega) == (8 << 20, True) and what.startswith("auto")
    assert choose_layout(2 << 20, False, 250 << 30, keep=True) == (2 << 20, False, "configured")
    page, mega, _ = choose_layout(64 << 10, False, 243 << 30)                                         # configs[2]: 4 M handles
    assert (page, mega) == (8 << 20, True)
    page, mega, _ = choose_layout(16 << 20, True, 2000 << 30)                                         # already large pages: kept, megacache on
    assert (page, mega) == (16 << 20, True)
