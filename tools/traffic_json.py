#!/usr/bin/env python3
"""profiles/rNN_hbm_pmc_raw.txt (tools/prof_round.sh: FETCH_SIZE / WRITE_SIZE per dispatch, separate --pmc passes, KiB) -> the per-launch HBM
bytes bench.py reports as roofline.traffic.  FETCH_SIZE is doubled: gfx950 counts 64 B per 128-B request on wide coalesced reads
(MI355X_MICROARCH.md, HBM section).  Round 6: with the SQ pass of the prefill kernel (tools/pmc_prefill.sh) as second argument the file also
carries `mfma_busy_frac` (SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs) over GRBM_GUI_ACTIVE / 8 XCDs), and with the rocprofv3 kernel
statistics of the bench run (tools/rocpd_stats.py) as third the kernels' mean durations (`kernel_us`: what bench.py puts beside its event times).
usage: tools/traffic_json.py profiles/rNN_hbm_pmc_raw.txt [profiles/rNN_prefill_pmc_raw.txt [profiles/rNN_bench_kernel_stats.md]] > profiles/rNN_traffic.json"""
import json
import re
import sys


def main(path, sq_path=None, stats_path=None):
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
    if sq_path:
        c = {}
        for line in open(sq_path):
            m = re.search(r"prefill64_kernelIDF16_\S*\s+(\S+)\s+n=\d+ per-dispatch (\S+)", line)
            if m:
                c[m.group(1)] = float(m.group(2))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
            out["prefill_yi6b_n32702"]["mfma_busy_frac"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (c["GRBM_GUI_ACTIVE"] / 8.0), 4)
            out["prefill_yi6b_n32702"]["mfma_busy_source"] = "%s: SQ_VALU_MFMA_BUSY_CYCLES %.4g / (4 x 256) over GRBM_GUI_ACTIVE %.4g / 8" % (sq_path, c["SQ_VALU_MFMA_BUSY_CYCLES"], c["GRBM_GUI_ACTIVE"])
            if "SQ_INSTS_VALU" in c and "SQ_INSTS_MFMA" in c:
                out["prefill_yi6b_n32702"]["valu_per_mfma"] = round((c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / c["SQ_INSTS_MFMA"], 3)
    if stats_path:
        us = {}
        for line in open(stats_path):
            f = [x.strip() for x in line.split("|")]
            if len(f) > 5 and f[1].startswith("`"):
                for key in ("prefill64_kernel", "decode_stream_kernel", "decode_stream_combine_kernel"):
                    if re.search(r"\d+" + key + "I", f[1]):
                        us[key] = float(f[4])
        if "prefill64_kernel" in us:
            out["prefill_yi6b_n32702"]["kernel_us"] = us["prefill64_kernel"]
        if "decode_stream_kernel" in us:
            out["decode_yi6b_b16_32k"]["kernel_us"] = round(us["decode_stream_kernel"] + us.get("decode_stream_combine_kernel", 0.0), 2)
            out["decode_yi6b_b16_32k"]["kernel_us_parts"] = {k: v for k, v in us.items() if k.startswith("decode")}
        out["_kernel_us_source"] = "%s: rocprofv3 --kernel-trace --stats of `bench.py --steps 2 --warmup 1 --no-dynamic --no-cpu-baseline`, mean duration per launch" % stats_path
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
