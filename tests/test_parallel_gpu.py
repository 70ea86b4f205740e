"""Row-sharded DGG step on the HIP kernels with TWO ranks sharing one GPU (gloo transports the CUDA tensors): the
multi-rank arithmetic -- row offsets into global columns, per-rank partitions, the fused backward, the asynchronous
xp / X gathers, da and weight all-reduces, reduce-scattered dX -- must reproduce the single-process step.  (RCCL itself
needs one GPU per rank and is exercised by the driver's multi-GPU bench; this covers everything above the transport.)"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def make_inputs(N=1500, d=128, h=64):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(N, d, generator=g)
    deg = 24 + 16 * torch.rand(N, generator=g)
    h2, h4 = h // 2, h // 4
    P = dict(We=torch.randn(h, d, generator=g) * 0.1, be=torch.randn(h, generator=g) * 0.1,
             Wk=torch.randn(h, d, generator=g) * 0.1, bk=torch.randn(h, generator=g) * 0.1,
             W1=torch.randn(h2, h + 1, generator=g) * 0.1, b1=torch.randn(h2, generator=g) * 0.1,
             Wmu=torch.randn(h4, h2, generator=g) * 0.1, bmu=torch.randn(h4, generator=g) * 0.1,
             Wp=torch.randn(1, h4, generator=g) * 0.03, bp=torch.tensor([0.05]), Wc=torch.rand(d, 64, generator=g))
    cot = torch.randn(N, 64, generator=g)
    return x, deg, P, cot


def run_step(x_local, deg, P, cot_local, N, x_grad, x_full=None, hybrid=False):
    sys.path.insert(0, ROOT)
    from dgg_amd import ops
    from dgg_amd.parallel import ShardedDGGConv
    dev = torch.device("cuda", 0)
    layer = ShardedDGGConv(ops, N, group=None, K=64, noise_mode=ops.NOISE_RANKED, seed=(5, 6), x_grad=x_grad,
                           x_full=None if x_full is None else x_full.to(dev), hybrid=hybrid)
    Pd = {k: v.to(dev) for k, v in P.items()}
    xl = x_local.to(dev)
    Z = layer.forward(xl, deg.to(dev), Pd)
    g = layer.backward(cot_local.to(dev), xl, Pd)
    torch.cuda.synchronize()
    return Z.cpu(), {k: v.cpu() for k, v in g.items()}, layer.saved["idx"].cpu()


def _worker(rank, world, port, x_grad, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from dgg_amd.parallel import shard_bounds
    x, deg, P, cot = make_inputs()
    N = x.shape[0]
    r0, r1, _ = shard_bounds(N, world, rank)
    # x_grad == "replicated": features are data present on every rank, no per-step exchange of X / xp; "hybrid": the same with H
    # all-gathered and dH reduce-scattered asynchronously (bench.py's default for several GPUs)
    Z, g, idx = run_step(x[r0:r1].contiguous(), deg, P, cot[r0:r1].contiguous(), N, x_grad is True,
                         x if x_grad in ("replicated", "hybrid") else None, x_grad == "hybrid")
    ret[rank] = (r0, r1, Z.numpy(), {k: v.numpy() for k, v in g.items()}, idx.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("x_grad", [False, True, "replicated", "hybrid"])
def test_two_ranks_on_one_gpu_match_single_process(x_grad):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    x, deg, P, cot = make_inputs()
    N = x.shape[0]
    Z1, g1, idx1 = run_step(x, deg, P, cot, N, x_grad is True)
    port = 29500 + os.getpid() % 2000
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, x_grad, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        if p.is_alive():
            p.kill()
            pytest.fail("a rank did not finish")
        assert p.exitcode == 0
    assert ret[0][0] == 0 and ret[0][1] == ret[1][0] and ret[1][1] == N
    assert np.array_equal(np.concatenate([ret[0][4], ret[1][4]]), idx1.numpy()), "selected neighbours differ under sharding"
    Z2 = np.concatenate([ret[0][2], ret[1][2]])
    np.testing.assert_allclose(Z2, Z1.numpy(), rtol=1e-5, atol=1e-5 * float(Z1.abs().max()))
    for k in g1:
        if k == "x":
            got = np.concatenate([ret[0][3]["x"], ret[1][3]["x"]])
        else:
            got = ret[0][3][k]
            np.testing.assert_array_equal(ret[1][3][k], got)      # all-reduced: identical on both ranks
        ref = g1[k].numpy()
        err = np.abs(got.reshape(ref.shape) - ref).max() / max(np.abs(ref).max(), 1e-30)
        assert err <= 3e-4, f"grad {k}: {err:.3e}"


def _rccl_worker(port, x_grad, ret):
    """ONE rank, backend nccl (= RCCL), DGG_FORCE_COLLECTIVES=1: every collective of the N>1 path is issued through RCCL on
    this GPU (asynchronous all_gather_into_tensor of xp / X, all-gather of the row sums, all-reduce of da and of the flat
    weight gradients, reduce_scatter_tensor of dX)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), DGG_FORCE_COLLECTIVES="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    x, deg, P, cot = make_inputs()
    if x_grad == "hybrid":      # asynchronous all_gather_into_tensor of H and reduce_scatter_tensor of dH through RCCL
        Z, g, idx = run_step(x, deg, P, cot, x.shape[0], False, x, True)
    else:
        Z, g, idx = run_step(x, deg, P, cot, x.shape[0], x_grad)
    ret[0] = (Z.numpy(), {k: v.numpy() for k, v in g.items()}, idx.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("x_grad", [False, True, "hybrid"])
def test_rccl_collectives_single_rank_match_plain_step(x_grad):
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        p = ctx.Process(target=_rccl_worker, args=(29571 + {False: 0, True: 1, "hybrid": 2}[x_grad], x_grad, ret))
        p.start()
        p.join(300)
        assert p.exitcode == 0
        Z, g, idx = ret[0]
    x, deg, P, cot = make_inputs()
    os.environ.pop("DGG_FORCE_COLLECTIVES", None)
    Z0, g0, idx0 = run_step(x, deg, P, cot, x.shape[0], x_grad is True)
    assert np.array_equal(idx, idx0.numpy())
    assert np.array_equal(Z, Z0.numpy())
    for k in g0:
        ref = g0[k].numpy()
        np.testing.assert_allclose(g[k], ref, rtol=2e-4, atol=2e-4 * max(np.abs(ref).max(), 1e-6), err_msg=k)


def test_bench_prints_exactly_one_json_line_with_rccl_initialised():
    """the driver parses bench.py's stdout: RCCL writes a version banner to fd 1 when the process group is destroyed, which
    must not reach stdout"""
    import json
    import subprocess
    env = dict(os.environ, DGG_FORCE_COLLECTIVES="1", MASTER_PORT="29575")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--nodes", "20000", "--steps", "3", "--warmup", "1",
                        "--cpu-rows", "-1"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-1000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["roofline"]["frac"] > 0
