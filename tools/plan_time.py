#!/usr/bin/env python3
"""Host time of the prefill work-list planner (vattn_prefill_plan: pure host arithmetic, no GPU) on the launches of the dynamic replay:
one to sixteen prompts of the arxiv-length recipe on a TP8 rank (8 / 1 heads) and on Llama-3-8B's heads.  The plan is built once per engine
iteration in front of layer 0's launch.  usage: python tools/plan_time.py [other libvattn_amd.so to time instead]"""
import ctypes as C, time, sys, json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vattention_amd import kernels as K
lib = K.klib() if len(sys.argv) < 2 else None
if lib is None:
    lib = C.CDLL(sys.argv[1]); K._bind(lib)
reqs = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'c3_arxiv_lengths_256.json')))["requests"]
lens_all = [int(p) for p,_ in reqs]
cases = [[4119],[9441],[14000],[20751],[29092],[6526,14505,5364],[18684,6354],[10159,5174], lens_all[:4], lens_all[4:12], lens_all[:16]]
for Hq,Hkv in ((8,1),(32,8)):
  tot=0
  for q_lens in cases:
    p = K.AttnParams()
    B=len(q_lens)
    p.b, p.seqlen_q, p.h, p.h_k, p.d, p.is_causal = B, max(q_lens), Hq, Hkv, 128, 1
    p.seqlen_k = max(q_lens)
    n_blk = sum((q+255)//256 for q in q_lens)*Hq
    cap_i, cap_b = 17*n_blk+16, n_blk+16
    items, blocks = (K.PrefillItem*cap_i)(), (K.PrefillItem*cap_b)()
    counts=(C.c_int32*3)()
    ql=(C.c_int32*B)(*q_lens); kl=(C.c_int32*B)(*q_lens)
    t0=time.perf_counter()
    for _ in range(20):
        n=lib.vattn_prefill_plan(C.byref(p), ql if B>1 else None, kl, items, cap_i, blocks, cap_b, counts)
    dt=(time.perf_counter()-t0)/20
    tot+=dt
    print("heads %d/%d %-28s blocks %5d -> %5d pieces, %4d split blocks: %.3f ms" % (Hq,Hkv,str(q_lens)[:28], n_blk, n, counts[1], dt*1e3))
  print("sum %.3f ms" % (tot*1e3))
