#!/usr/bin/env python3
"""Bench-scale check of the row-sharded step: WORLD ranks share ONE GPU (gloo carries the tensors), each owns --nodes rows
of a (--nodes * WORLD)-node graph, exactly the weak-scaling layout of `bench.py --gpus WORLD`.  Every rank's neighbour
lists / outputs are compared with the SAME rows of a single-process run over the whole graph, gradients with the
single-process gradients.  Everything above the RCCL transport at the sizes the driver's scaling run uses.

    python tools/dist_scale_check.py --world 8 --nodes 100000
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def inputs(N, d, rank_rows):
    """same generators as bench.py main(): per-rank features (seed 1000 + rank), global prior degrees (seed 7)"""
    xs = []
    for r, n in enumerate(rank_rows):
        g = torch.Generator(device="cpu").manual_seed(1000 + r)
        xs.append(torch.randn(n, d, generator=g))
    gd = torch.Generator(device="cpu").manual_seed(7)
    deg = 24 + 16 * torch.rand(N, generator=gd)
    return xs, deg


def step(layer, x_local, deg, P):
    Z = layer.forward(x_local, deg, P)
    g = layer.backward(torch.ones_like(Z), x_local, P)
    torch.cuda.synchronize()
    return Z, g


def worker(rank, world, port, a, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from dgg_amd import ops
    from dgg_amd.parallel import ShardedDGGConv, shard_bounds
    dev = torch.device("cuda", 0)
    N = a.nodes * world
    P = bench.make_params(a.feat, a.latent, dev)
    bounds = [shard_bounds(N, world, r) for r in range(world)]
    xs, deg = inputs(N, a.feat, [b[1] - b[0] for b in bounds])
    x_full = torch.cat(xs).to(dev) if (a.replicate or a.hybrid) else None     # features as data, present on every rank
    layer = ShardedDGGConv(ops, N, group=None, K=64, noise_mode=ops.NOISE_RANKED, seed=(1234, 0), x_grad=a.x_grad, x_full=x_full,
                           hybrid=a.hybrid)
    Z, g = step(layer, xs[rank].to(dev), deg.to(dev), P)
    s = layer.saved
    ret[rank] = dict(idx_crc=int(s["idx"].long().sum().item()), Z_sum=float(Z.double().sum().item()),
                     idx=s["idx"][:: a.sample].cpu().numpy(), Z=Z[:: a.sample].cpu().numpy(),
                     g={k: v.cpu().numpy() for k, v in g.items() if k != "x"},
                     gx=g["x"][:: a.sample].cpu().numpy() if a.x_grad else None, k=float(s["k"].sum().item()))
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--nodes", type=int, default=100_000, help="rows per rank")
    ap.add_argument("--feat", type=int, default=128)
    ap.add_argument("--latent", type=int, default=64)
    ap.add_argument("--sample", type=int, default=97, help="row stride of the compared sample")
    ap.add_argument("--x-grad", action="store_true")
    ap.add_argument("--replicate", action="store_true", help="replicated features (bench.py --exchange replicate)")
    ap.add_argument("--hybrid", action="store_true", help="replicated features, H all-gathered / dH reduce-scattered (bench.py's default)")
    a = ap.parse_args()
    world = a.world
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        procs = [ctx.Process(target=worker, args=(r, world, 29590, a, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(1200)
        assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
        ret = {r: ret[r] for r in range(world)}
    # single process over the whole graph
    import bench
    from dgg_amd import ops
    from dgg_amd.parallel import ShardedDGGConv, shard_bounds
    dev = torch.device("cuda", 0)
    N = a.nodes * world
    P = bench.make_params(a.feat, a.latent, dev)
    bounds = [shard_bounds(N, world, r) for r in range(world)]
    xs, deg = inputs(N, a.feat, [b[1] - b[0] for b in bounds])
    layer = ShardedDGGConv(ops, N, group=None, K=64, noise_mode=ops.NOISE_RANKED, seed=(1234, 0), x_grad=a.x_grad)
    Z, g = step(layer, torch.cat(xs).to(dev), deg.to(dev), P)
    idx = layer.saved["idx"]
    worst = 0.0
    for r, (r0, r1, _) in enumerate(bounds):
        got = ret[r]
        assert np.array_equal(got["idx"], idx[r0:r1][:: a.sample].cpu().numpy()), f"rank {r}: neighbour lists differ"
        assert got["idx_crc"] == int(idx[r0:r1].long().sum().item()), f"rank {r}: neighbour-list checksum differs"
        zr = Z[r0:r1][:: a.sample].cpu().numpy()
        err = np.abs(got["Z"] - zr).max() / max(np.abs(zr).max(), 1e-6)
        assert err <= 1e-5, f"rank {r}: outputs differ {err:.2e}"
        worst = max(worst, err)
        if a.x_grad:
            xr = g["x"][r0:r1][:: a.sample].cpu().numpy()
            e = np.abs(got["gx"] - xr).max() / max(np.abs(xr).max(), 1e-6)
            assert e <= 3e-4, f"rank {r}: dX differs {e:.2e}"
    gerr = {}
    for k_, v in g.items():
        if k_ == "x":
            continue
        ref = v.cpu().numpy()
        for r in range(world):
            assert np.array_equal(ret[r]["g"][k_], ret[0]["g"][k_]), f"grad {k_} differs between ranks"
        gerr[k_] = float(np.abs(ret[0]["g"][k_] - ref).max() / max(np.abs(ref).max(), 1e-6))
        assert gerr[k_] <= 5e-4, f"grad {k_}: {gerr[k_]:.2e}"
    ksum = sum(ret[r]["k"] for r in range(world))
    print(f"OK world={world} N={N}: neighbour lists identical on every rank, outputs within {worst:.1e}, "
          f"mean k {ksum / N:.2f}, worst weight-gradient error {max(gerr.values()):.1e}")


if __name__ == "__main__":
    main()
