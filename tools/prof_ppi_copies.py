"""Which autograd nodes issue device-to-device copies in the bf16 PPI step (diagnostic)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dgg_amd
from argparse import Namespace
from bench import pubmed_graph
dev = torch.device("cuda:0")
d, hid, C, L = 50, 2048, 121, 9
args = Namespace(extra_edge_dim=0, extra_k_dim=1, dgg_hard=False, deg_mean=3.899, deg_std=5.288, dgg_mode_edge_net="u-v-dist",
                 dgg_mode_k_net="x", dgg_mode_k_select="k_times_edge_prob", debug_step=3, perturb_edge_prob=True,
                 symmetric_noise=False, stochastic_k=False, dgg_adj_input="input_adj", n_dgg_layers=1)
m = dgg_amd.GCNIIppi_DGG(nfeat=d, nlayers=L, nhidden=hid, nclass=C, dropout=0.2, lamda=0.5, alpha=0.5, variant=True, args=args).to(dev)
for conv in m.convs:
    conv.gemm_dtype = torch.bfloat16
m.train()
n = int(os.environ.get('N', 2925))
rows, cols = pubmed_graph(n, n * 14, seed=n)
keep = rows != cols
A = torch.sparse_coo_tensor(torch.from_numpy(np.stack([rows[keep], cols[keep]])), torch.ones(int(keep.sum())), (n, n)).coalesce().to(dev)
x, y = torch.randn(n, d, device=dev), (torch.rand(n, C, device=dev) < 0.3).float()
def step():
    for p in m.parameters():
        p.grad = None
    torch.nn.functional.binary_cross_entropy(m(x, A), y).backward()
for _ in range(2):
    step()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
for e in sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:25]:
    print(e.key[:70].ljust(72), e.count, round(e.device_time_total / 1e3, 3), "ms")
for ev in prof.events():
    if "emcpy" in ev.name or "copyBuffer" in ev.name:
        print("MEMCPY", ev.name, round(ev.device_time_total / 1e3, 3), "ms", "parent:", ev.cpu_parent.name if ev.cpu_parent else None)
