import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgg_amd import ops
dev = torch.device("cuda", 0)
N, d, h = 100_000, 128, 64
g = torch.Generator(device="cpu").manual_seed(5)
x = torch.randn(N, d, generator=g)
x[20_000:23_000] *= 0.05
x[23_000:23_040] = x[23_000:23_040] * 0.01 + 3.0
x = x.to(dev)
W = (torch.randn(h, d, generator=g) * 0.1).to(dev)
b = (torch.randn(h, generator=g) * 0.1).to(dev)
xp = ops.linear_fwd(x, W, b, ops.ACT_LEAKY)
torch.cuda.synchronize(); print("projected", flush=True)
idx, val, ws = ops.allpairs_topk(xp, 64, noise_mode=ops.NOISE_NONE, return_ws=True)
torch.cuda.synchronize(); print("topk done", flush=True)
print(ops.fast_path_failed_rows(ws, N, h, stats=True))
del ws
for lo, hi in [(0, 300), (22_900, 23_412), (99_700, N)]:
    junk = torch.full((300_000_000,), 0x7f7f7f7f, dtype=torch.int32, device=dev); del junk
    sub_i, sub_v, ws2 = ops.allpairs_topk(xp, 64, noise_mode=ops.NOISE_NONE, rows=(lo, hi), return_ws=True)
    torch.cuda.synchronize(); print("range", lo, hi, "done", ops.fast_path_failed_rows(ws2, N, h, rows=hi - lo, stats=True), bool(torch.equal(sub_i, idx[lo:hi])), flush=True)
