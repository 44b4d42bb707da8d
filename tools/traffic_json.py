#!/usr/bin/env python3
"""profiles/rNN_hbm_pmc_raw.txt (tools/prof_round.sh: FETCH_SIZE / WRITE_SIZE per dispatch, separate --pmc passes, KiB) -> the per-launch HBM
bytes bench.py reports as roofline.traffic.  FETCH_SIZE is doubled: gfx950 counts 64 B per 128-B request on wide coalesced reads
(MI355X_MICROARCH.md, HBM section).  usage: tools/traffic_json.py profiles/rNN_hbm_pmc_raw.txt > profiles/rNN_traffic.json"""
import json
import re
import sys


def main(path):
    cur, vals = None, {}
    for line in open(path):
        if line.startswith("_ZN"):
            cur = "prefill" if "prefill64" in line else "decode_stream" if "decode_stream_kernel" in line else "decode_combine" if "decode_stream_combine" in line else None
            continue
        m = re.search(r"(FETCH_SIZE|WRITE_SIZE)\s+total \S+\s+per-dispatch (\S+)", line)
        if m and cur:
            vals[(cur, m.group(1))] = float(m.group(2))
    g = lambda k, c: vals.get((k, c), 0.0)
    pf_f, pf_w = g("prefill", "FETCH_SIZE"), g("prefill", "WRITE_SIZE")
    dc_f = g("decode_stream", "FETCH_SIZE") + g("decode_combine", "FETCH_SIZE")
    dc_w = g("decode_stream", "WRITE_SIZE") + g("decode_combine", "WRITE_SIZE")
    n, Hq, Hkv, D, B, ctx = 32702, 32, 4, 128, 16, 32768
    out = {"_source": "%s: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of tools/kbench.py --only 'yi6b whole,yi6b B16@32k' --variants 0 "
                      "(tools/prof_round.sh, tools/pmc_summary.py); KiB per dispatch; FETCH_SIZE doubled (gfx950 counts 64 B per 128-B request on wide "
                      "coalesced reads); decode = decode_stream_kernel + decode_stream_combine_kernel" % path,
           "decode_yi6b_b16_32k": {"fetch_kib": round(dc_f), "write_kib": round(dc_w), "hbm_bytes_per_launch": int((2 * dc_f + dc_w) * 1024),
                                   "algorithmic_bytes": int(B * 2.0 * ctx * Hkv * D * 2 + B * Hq * D * 2 * 2 - B * 2 * Hkv * D * 2 * 0)},
           "prefill_yi6b_n32702": {"fetch_kib": round(pf_f), "write_kib": round(pf_w), "hbm_bytes_per_launch": int((2 * pf_f + pf_w) * 1024),
                                   "algorithmic_bytes": int(2 * n * Hkv * D * 2 + 2 * n * Hq * D * 2)}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
