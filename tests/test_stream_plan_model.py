"""The decode launch's device-side plan, restated on the CPU (oracle/stream_plan.py): invariants under arbitrary batches.  The kernel's own
plan is compared with this model on the GPU (tests/test_gpu_attention.py::test_decode_stream_plan_table_matches_the_model)."""
import random

import pytest

from oracle.stream_plan import plan, tiles_of


def _check(lens, seqlen_k, knew, nwg, X=4):
    uniform, pieces, records = plan(lens, seqlen_k, knew, nwg, X)
    B = len(lens)
    lk = [min(max(x, 0) + knew, seqlen_k) for x in lens]
    owned = [[] for _ in range(B)]
    seen_records = set()
    for w, ps in enumerate(pieces):
        assert [p[0] for p in ps] == sorted(p[0] for p in ps)          # a workgroup walks its sequences in batch order
        for b, tb, te, rec in ps:
            assert 0 <= tb < te <= tiles_of(lk[b])
            assert rec not in seen_records, "two pieces publish the same record"
            seen_records.add(rec)
            owned[b].append((tb, te, rec, w))
    for b in range(B):
        ps = sorted(owned[b])
        # every tile exactly once, pieces contiguous; their records consecutive from the table's first record, as many as the table says
        assert ps[0][0] == 0 and ps[-1][1] == tiles_of(lk[b]) and all(a[1] == c[0] for a, c in zip(ps, ps[1:])), (b, ps)
        first, cnt = records[b]
        assert [p[2] for p in ps] == list(range(first, first + cnt)), (b, ps, records[b])
        assert [p[3] for p in ps] == sorted(p[3] for p in ps)          # consecutive workgroups
    assert max(seen_records) < nwg + B                                  # what the workspace is sized for
    return uniform, pieces


def test_equal_lengths_take_the_aligned_decomposition():
    uniform, pieces = _check([32767] * 16, 32768, 1, 192)
    assert uniform and all(len(p) == 1 for p in pieces) and len({p[0][2] - p[0][1] for p in pieces[:11]}) == 1      # 12 pieces of 86 tiles (the last 78)
    assert _check([8191] * 64, 8192, 1, 96)[0] is False                # fewer workgroups than a uniform cut needs: ranges span sequences


def test_ragged_batches_get_equal_ranges():
    rng = random.Random(3)
    lens = [rng.randint(4000, 29000) for _ in range(256)]
    uniform, pieces = _check(lens, max(lens) + 1, 1, 768)
    assert not uniform
    work = [sum(te - tb for _, tb, te, _ in p) for p in pieces if p]
    assert max(work) - min(w for w in work[:-1]) <= 4 * 3 + 1           # equal ranges, minus at most a few switch allowances


@pytest.mark.parametrize("seed", range(40))
def test_invariants_under_random_batches(seed):
    rng = random.Random(seed)
    B = rng.choice([2, 3, 7, 16, 64, 100, 250, 256])
    hi = rng.choice([1, 40, 500, 3000, 30000])
    lens = [rng.randint(0, hi) for _ in range(B)]
    knew = rng.choice([0, 1])
    if knew == 0:
        lens[rng.randrange(B)] = 0                                      # an empty sequence: one masked tile, still written
    nwg = rng.choice([1, 2, 5, 37, 96, 192, 768, 2000])
    _check(lens, max(lens) + knew + rng.choice([0, 5]), knew, nwg, X=rng.choice([0, 2, 4, 9]))
