"""The iteration schedule and seeded inputs shared by oracle/gen_golden_wrapper.py (reference wrapper on CPU) and
tests/test_gpu_wrapper_golden.py (this package's wrapper on the GPU)."""
import torch

HQ, HKV, D, MAX_BATCH, MAX_CTX = 2, 1, 128, 4, 512


class Seq:
    def __init__(self, seq_id, prompt_len, total_len):
        self.seq_id, self.prompt_len, self.total_len = seq_id, prompt_len, total_len
        self.prompt_processed = 0
        self.output_len = 0

    def get_next_prompt_chunk_len(self, chunk):
        return min(chunk, self.prompt_len - self.prompt_processed)

    def get_num_prompt_tokens_processed(self):
        return self.prompt_processed

    def get_len(self):
        return self.prompt_len + self.output_len

    def is_finished(self):
        return self.get_len() >= self.total_len

    @property
    def prompt_done(self):
        return self.prompt_processed >= self.prompt_len


class MD:
    def __init__(self, seq, chunk, is_prompt):
        self.seq, self.prompt_chunk_len, self.is_prompt = seq, chunk, is_prompt


def schedule():
    """Yields (seq_metadata_list, prefill slots, decode slots) and advances the sequences the way seq_manager does.
    Slots are fixed here (the cache engine's choice is checked elsewhere): seq i lives in slot SLOT[i]."""
    SLOT = {0: 2, 1: 0, 2: 3}
    a, b, c = Seq(0, 300, 304), Seq(1, 77, 81), Seq(2, 200, 203)

    def adv(mds):
        for md in mds:
            if md.is_prompt:
                md.seq.prompt_processed += md.seq.get_next_prompt_chunk_len(md.prompt_chunk_len)
                if md.seq.prompt_done:
                    md.seq.output_len += 1
            else:
                md.seq.output_len += 1

    plan = [
        lambda: [MD(a, 128, True)],
        lambda: [MD(a, 128, True)],
        lambda: [MD(a, 128, True)],                              # last chunk: 44 tokens
        lambda: [MD(b, 77, True), MD(a, 0, False)],               # hybrid: whole prompt + one decode
        lambda: [MD(c, 96, True), MD(a, 0, False), MD(b, 0, False)],
        lambda: [MD(c, 96, True), MD(a, 0, False), MD(b, 0, False)],
        lambda: [MD(c, 96, True), MD(b, 0, False)],
        lambda: [MD(s, 0, False) for s in (b, c) if not s.is_finished()],
        lambda: [MD(s, 0, False) for s in (b, c) if not s.is_finished()],
    ]
    for mk in plan:
        mds = mk()
        yield mds, [SLOT[m.seq.seq_id] for m in mds if m.is_prompt], [SLOT[m.seq.seq_id] for m in mds if not m.is_prompt]
        adv(mds)


def make_inputs(it, mds):
    T = sum(m.seq.get_next_prompt_chunk_len(m.prompt_chunk_len) if m.is_prompt else 1 for m in mds)
    g = torch.Generator()
    g.manual_seed(1000 + it)
    mk = lambda h: torch.randn(T, h * D, generator=g).half()
    return mk(HQ), mk(HKV), mk(HKV)
