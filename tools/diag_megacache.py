import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vattention_amd import vattention
torch.zeros(1, device="cuda")
page = 2 << 20
L, kvh, D, B, ctx = 4, 2, 128, 3, 2048
ts = vattention.init_kvcache(L, kvh, D, B, ctx, 0, torch.float16, page, True)
print("shape", ts[0].shape, "stride", ts[0].stride(), "layout", vattention.layout())
vattention.reserve_physical_pages(64 * page)
s = vattention.alloc_new_batch_idx(1500)
lens = [0] * B; lens[s] = 1500
vattention.step_async(lens)
print("slot", s, "state", vattention.state()["mapped"])
K = ts[0]
for l in range(L):
    v = K[:, :, l]
    print("view", l, v.shape, v.stride(), v.storage_offset())
    v[s, :1500].fill_(float(l + 1))
torch.cuda.synchronize()
for pos in (0, 1, 1023, 1024, 1499):
    print(pos, [float(K[s, pos, l, 1, 5]) for l in range(L)])
flat = K[s, :1500].reshape(-1)[:4096].float().cpu()
print("first 4096 elems uniq", flat.unique())
K[s, :1500, 2].fill_(9.0); torch.cuda.synchronize()
print("direct 3-index fill:", float(K[s, 1499, 2, 1, 5]))
vattention.cleanup()
