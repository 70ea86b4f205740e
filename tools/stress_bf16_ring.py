#!/usr/bin/env python3
"""Race screen of the LDS-DMA ring product (dgg_gemm_nt_bf16 and its fused epilogues): the same launch repeated many times at several
sizes, beside a second stream that keeps the memory system busy, every result compared bit for bit with the first one and the first one
with the fp32 product of the bf16 operands.  python tools/stress_bf16_ring.py [repeats]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dgg_amd import ops  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(5)
noise = torch.empty(64 << 20, device=dev)
side = torch.cuda.Stream()
bad = 0
for (M, N, K) in ((1783, 2048, 4096), (3478, 4096, 2048), (4096, 2048, 1792), (593, 2048, 4096), (130, 128, 192)):
    A = torch.randn(M, K, generator=g).to(dev).to(torch.bfloat16)
    B = torch.randn(N, K, generator=g).to(dev).to(torch.bfloat16)
    first = ops.gemm_nt_bf16(A, B)
    ref = A.float() @ B.float().t()
    err = float((first - ref).abs().max() / ref.abs().max())
    diff = 0
    for r in range(R):
        with torch.cuda.stream(side):
            noise.add_(1.0)                                     # traffic beside the product
        out = ops.gemm_nt_bf16(A, B)
        if not torch.equal(out, first):
            diff += 1
    torch.cuda.synchronize()
    print(f"M={M} N={N} K={K}: rel err vs fp32 product {err:.2e}, {diff} of {R} repeats differ from the first")
    bad += diff + (err > 1e-5)
print("OK" if bad == 0 else "FAILED")
sys.exit(1 if bad else 0)
