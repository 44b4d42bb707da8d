"""Generate tests/golden/pagemgr_*.json: call traces + the REAL reference's answers.

Run in the build container (where /root/reference exists):
    bash oracle/build_ref.sh && python oracle/gen_golden_pagemgr.py
Each file holds {"config", "ops", "expect": [per-op records from oracle/_ref]}; records carry the
return value / error text, mapped_pages[], curr_seq_lengths[], pool size, pool handle order,
the page map and the driver-call log.  tests/test_pagemgr_oracle.py and tests/test_page_manager_product.py replay the ops on the
Python oracle and on the product's C++ manager and compares.  TEST INFRASTRUCTURE ONLY.
"""
import gzip
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import trace as T            # noqa: E402
from oracle.ref_adapter import RefImpl, available   # noqa: E402

MB, KB = 1 << 20, 1 << 10
CONFIGS = {
    # scaled-down shapes that keep the per-config arithmetic of SURVEY §A.3 (tokens/page, pages/req)
    "c2_yi6b_2mb": dict(num_layers=4, num_kv_heads=4, head_size=128, max_batch_size=16, max_context_length=32768,
                        itemsize=2, page_size=2 * MB, megacache=False),       # tpp 2048, 16 pages/req
    "c3_llama8b_64kb": dict(num_layers=3, num_kv_heads=8, head_size=128, max_batch_size=24, max_context_length=2048,
                            itemsize=2, page_size=64 * KB, megacache=False),  # tpp 32, 64 pages/req
    "c5_llama70b_256kb": dict(num_layers=5, num_kv_heads=1, head_size=128, max_batch_size=32, max_context_length=8192,
                              itemsize=2, page_size=256 * KB, megacache=False),  # tpp 1024, 8 pages/req
    "mega_2mb": dict(num_layers=4, num_kv_heads=8, head_size=128, max_batch_size=6, max_context_length=16384,
                     itemsize=2, page_size=2 * MB, megacache=True),           # tpp 256
    "mega_256kb": dict(num_layers=4, num_kv_heads=2, head_size=64, max_batch_size=6, max_context_length=8192,
                       itemsize=2, page_size=256 * KB, megacache=True),
    "fp32_2mb": dict(num_layers=2, num_kv_heads=8, head_size=64, max_batch_size=8, max_context_length=16384,
                     itemsize=4, page_size=2 * MB, megacache=False),
}


def traces_for(name, cfg):
    out = []
    for seed in range(6):
        out.append(T.gen_serving_trace(cfg, 100 + seed, iters=120, pool_groups=[60, 20, 35][seed % 3],
                                       use_async=seed % 3 != 2, chunk=[0, 2048, 512][seed % 3] if cfg["max_context_length"] > 4096 else [0, 256, 64][seed % 3],
                                       p_finish=0.03, disable_deferred=seed == 4, admission=seed != 5))
    for seed in range(3):
        out.append(T.gen_adversarial_trace(cfg, 200 + seed, 150, pool_groups=[30, 8, 16][seed]))
    # prefix-sharing experiment API (vattention.cu:325-373)
    out.append({"config": cfg, "kind": "map_common", "seed": 0,
                "ops": [["reserve", 40 * 2 * cfg["num_layers"] * cfg["page_size"]], ["map_common", 3], ["nfree"],
                        ["alloc", 5], ["step", [5] + [0] * (cfg["max_batch_size"] - 1), False], ["nfree"]]})
    return out


def main():
    if not available():
        raise SystemExit("oracle/_ref not built")
    outdir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    os.makedirs(outdir, exist_ok=True)
    total = 0
    for name, cfg in CONFIGS.items():
        items = []
        for tr in traces_for(name, cfg):
            ops = T.resolve(tr, T.OracleImpl) if tr["kind"] == "serving" else tr["ops"]
            pred = T.replay(T.OracleImpl(cfg), ops)
            ops = T.truncate_for_reference(ops, pred)
            expect = T.replay(RefImpl(cfg), ops, full=True)
            items.append({"kind": tr["kind"], "seed": tr["seed"], "ops": ops, "expect": expect})
            total += len(ops)
        ref = RefImpl(cfg)
        info = ref.tensor_info()
        ref.cleanup()
        path = os.path.join(outdir, "pagemgr_%s.json.gz" % name)
        with gzip.GzipFile(path, "wb", mtime=0) as f:
            f.write(json.dumps({"config": cfg, "tensor_info": info, "traces": items}, separators=(",", ":")).encode())
        print(path, os.path.getsize(path) // 1024, "KiB", len(items), "traces")
    print("total ops", total)


if __name__ == "__main__":
    main()
