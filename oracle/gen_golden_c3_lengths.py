#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  Writes tests/golden/c3_arxiv_lengths_256.json: the (prefill, decode) token counts of the first 256
requests of the reference's dynamic trace (BASELINE configs[2] / [4]), produced by the reference's own recipe
(/root/reference/sarathi-lean/sarathi/benchmark/request_generator/trace_request_length_generator.py:16-101 as driven by
/root/reference/scripts/benchmark_e2e_dynamic_trace.py:7-60: scale factors 1, max_tokens 32768 with the proportional
ceil() trim, at least one prefill and one decode token, min_tokens 0, `DataFrame.sample(frac=1, random_state=42)`), applied to
/root/reference/scripts/artifact_asplos25/traces/arxiv_sample.csv (the trace file named by default.yml is not in the tree).
Needs /root/reference; the output travels with the repo."""
import json
import os

import numpy as np
import pandas as pd

REF = os.environ.get("VATTN_REFERENCE_DIR", "/root/reference")
SRC = os.path.join(REF, "scripts/artifact_asplos25/traces/arxiv_sample.csv")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "c3_arxiv_lengths_256.json")
MAX_TOKENS, MIN_TOKENS, SEED, N = 32768, 0, 42, 256

df = pd.read_csv(SRC)
df["num_prefill_tokens"] = (df["num_prefill_tokens"] * 1).astype(int)
df["num_decode_tokens"] = (df["num_decode_tokens"] * 1).astype(int)
total = df["num_prefill_tokens"] + df["num_decode_tokens"]
diff = (total - MAX_TOKENS).clip(lower=0)
df["num_prefill_tokens"] -= np.ceil(diff * (df["num_prefill_tokens"] / total)).astype(int)
df["num_decode_tokens"] -= np.ceil(diff * (df["num_decode_tokens"] / total)).astype(int)
df["num_prefill_tokens"] = df["num_prefill_tokens"].clip(lower=1)
df["num_decode_tokens"] = df["num_decode_tokens"].clip(lower=1)
assert all(df["num_prefill_tokens"] + df["num_decode_tokens"] <= MAX_TOKENS)
df = df[df["num_prefill_tokens"] > MIN_TOKENS]
df = df.sample(frac=1, random_state=SEED)
rows = [[int(r.num_prefill_tokens), int(r.num_decode_tokens)] for r in df.iloc[:N].itertuples()]
json.dump({"source": "arxiv_sample.csv via trace_request_length_generator.py recipe, seed 42, max_tokens 32768, min_tokens 0",
           "requests": rows}, open(OUT, "w"))
p = sorted(r[0] for r in rows)
d = sorted(r[1] for r in rows)
print("wrote %s: %d requests, prefill min/p50/max %d/%d/%d, decode min/p50/max %d/%d/%d" % (OUT, len(rows), p[0], p[len(p) // 2], p[-1], d[0], d[len(d) // 2], d[-1]))
