"""CPU-side checks: the C-ABI library loads and exports every symbol include/dgg_hip.h declares (no compute
calls without a GPU); the drop-in modules keep the reference's state_dict contract; ops refuse CPU tensors."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from helpers import ROOT, load_fixture


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "dgg_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dgg_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import dgg_amd
    syms = declared_symbols()
    assert len(syms) >= 19
    L = ctypes.CDLL(dgg_amd._lib.SO_PATH)
    for s in syms:
        assert hasattr(L, s), f"libdgg_hip.so does not export {s}"
    # the ctypes prototype table covers the same set (minus the two info functions)
    assert set(dgg_amd._lib.PROTOTYPES) | {"dgg_last_error", "dgg_abi_version"} == set(syms)
    assert dgg_amd._lib.lib().dgg_abi_version() == 6


def test_public_header_is_valid_c():
    """include/dgg_hip.h must compile as plain C (it is what a cgo / JNI / ctypes binding generator would read)"""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    r = subprocess.run(["gcc", "-fsyntax-only", "-x", "c", "-std=c99", "-Wall", "-Werror", os.path.join(ROOT, "include", "dgg_hip.h")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_state_dict_contract_matches_reference_fixture():
    """keys/shapes of the reference's DGG_LearnableK_debug.state_dict() (captured in the golden fixture)"""
    import dgg_amd
    from argparse import Namespace
    fx = load_fixture("allpairs_n256_asym")
    args = Namespace(**fx["meta"]["args"])
    m = dgg_amd.DGG_LearnableK_debug(in_dim=fx["meta"]["d"], latent_dim=fx["meta"]["h"], args=args)
    ref = {k[2:]: v.shape for k, v in fx.items() if k.startswith("p.")}
    mine = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert mine == {k: tuple(s) for k, s in ref.items()}
    m.load_state_dict({k[2:]: torch.from_numpy(v) for k, v in fx.items() if k.startswith("p.")}, strict=True)


def test_model_wrappers_keys():
    import dgg_amd
    from argparse import Namespace
    args = Namespace(extra_edge_dim=2, extra_k_dim=1, dgg_hard=False, deg_mean=3.899, deg_std=5.288,
                     dgg_mode_edge_net="u-v-dist", dgg_mode_k_net="x", dgg_mode_k_select="k_times_edge_prob",
                     debug_step=3, perturb_edge_prob=True, symmetric_noise=False, stochastic_k=False,
                     dgg_adj_input="input_adj", n_dgg_layers=1)
    m = dgg_amd.GCN_DGG(nfeat=1433, nlayers=2, nhidden=64, nclass=7, args=args)
    sd = m.state_dict()
    assert len(sd) == 38                                   # SURVEY.md section 8b [probe]
    assert sum(p.numel() for p in m.parameters()) == 300257
    for k in ("convs.0.W", "convs.1.W", "conv1.W", "conv2.W", "dggs.0.t", "dggs.0.k_W",
              "dggs.0.node_encode_for_edges.0.weight", "dggs.0.k_net.k_project.bias"):
        assert k in sd
    g = dgg_amd.GCNII_DGG(nfeat=50, nlayers=4, nhidden=32, nclass=3, dropout=0.5, lamda=0.5, alpha=0.1, variant=False, args=args)
    ks = g.state_dict().keys()
    assert "convs.3.weight" in ks and "fcs.0.weight" in ks and "fcs.1.bias" in ks and "dggs.0.k_embed.0.weight" in ks


def test_ops_refuse_cpu_tensors_and_unsupported_modes():
    import dgg_amd
    from argparse import Namespace
    with pytest.raises(AssertionError):
        dgg_amd.ops.linear_fwd(torch.zeros(4, 4), torch.zeros(4, 4))
    args = Namespace(extra_edge_dim=2, extra_k_dim=1, dgg_hard=False, deg_mean=3.899, deg_std=5.288,
                     dgg_mode_edge_net="nonsense", dgg_mode_k_net="x", dgg_mode_k_select="k_times_edge_prob",
                     debug_step=3, perturb_edge_prob=True, symmetric_noise=False, stochastic_k=False)
    m = dgg_amd.DGG_LearnableK_debug(8, 16, args)
    with pytest.raises(Exception, match="mode not found"):      # reference dgm.py:1726-1727
        m(torch.zeros(4, 8), dgg_amd.AllPairs(torch.ones(4)))


def test_missing_library_fails_loudly(tmp_path):
    """no CPU fallback: importing the package without libdgg_hip.so must raise as soon as a kernel is requested"""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import dgg_amd; "
            "\ntry:\n    dgg_amd._lib.lib()\n    print('LOADED')\nexcept dgg_amd._lib.DggHipError as e:\n    print('RAISED', 'no CPU fallback' in str(e))") % ROOT
    env = dict(os.environ, DGG_HIP_SO=str(tmp_path / "nope" / "libdgg_hip.so"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert "RAISED True" in r.stdout, r.stdout + r.stderr


def test_chunked_adjacency_container_on_the_host():
    """EllAdjacency with a chunk layout (rows wider than 64 ranks: node i owns the chunks [cptr[i], cptr[i+1])): densification, COO and
    CSR views and row sums are plain torch ops on the container -- checked here without a GPU against a hand-built matrix"""
    import torch
    import dgg_amd
    from dgg_amd import ops
    N = 200
    lens = [40, 150, 64, 100, 3] + [7] * (N - 5)                       # entries per row
    M = torch.tensor([(n + 63) // 64 for n in lens])                   # chunks per row: 1, 3, 1, 2, 1, 1, ...
    cptr = torch.cat([torch.zeros(1, dtype=torch.int64), M.cumsum(0)]).to(torch.int32)
    cnode = torch.repeat_interleave(torch.arange(N), M).to(torch.int32)
    C_ = int(M.sum())
    lay = ops.ChunkLayout(cptr, cnode, torch.tensor([C_, 3, 0, 0], dtype=torch.int32), C_, 3, N)
    assert lay.wide and lay.ranks().shape == (C_, 64) and int(lay.ranks()[2, 5]) == 64 + 5      # chunk 2 = second chunk of node 1
    g = torch.Generator().manual_seed(0)
    dense = torch.zeros(N, N)
    idx = torch.full((C_, 64), -1, dtype=torch.int32)
    val = torch.zeros(C_, 64)
    for i in range(N):
        n = lens[i]
        cols = torch.randperm(N, generator=g)[:n]
        v = torch.rand(n, generator=g) + 0.1
        dense[i, cols] = v
        c0 = int(cptr[i])
        idx[c0:c0 + int(M[i])].view(-1)[:n] = cols.to(torch.int32)
        val[c0:c0 + int(M[i])].view(-1)[:n] = v
    adj = dgg_amd.EllAdjacency(idx, val, N, layout=lay)
    assert adj.shape == (N, N)
    assert torch.equal(adj.to_dense(), dense) and torch.equal(adj.to_sparse().to_dense(), dense)
    csr = adj.to_csr()
    assert torch.equal(csr.to_dense(), dense) and csr.rowptr.tolist()[:6] == [0, 40, 190, 254, 354, 357]
    assert torch.allclose(adj.row_sums(), dense.sum(1))
    flat = dgg_amd.EllAdjacency(idx[:1], val[:1], N, layout=ops.ChunkLayout(cptr[:2], cnode[:1], None, 1, 1, 1))
    assert flat.layout is None, "a layout without a wide row is the plain list"


def test_wide_row_policies_on_the_host():
    """args.dgg_wide_rows: what rows that need more than 64 ranks do, decided on the host (no kernel involved)"""
    import torch
    import dgg_amd
    from argparse import Namespace
    from dgg_amd import ops
    base = dict(extra_edge_dim=0, extra_k_dim=1, dgg_hard=False, deg_mean=3.9, deg_std=5.3, dgg_mode_edge_net="u-v-dist", dgg_mode_k_net="x",
                dgg_mode_k_select="k_times_edge_prob", debug_step=3, perturb_edge_prob=True, symmetric_noise=False, stochastic_k=False,
                dgg_adj_input="input_adj", n_dgg_layers=1)
    m = dgg_amd.DGG_LearnableK_debug(in_dim=8, latent_dim=16, args=Namespace(**base))
    # every counter-based generator and unperturbed scores have a wide-row form (round 6); explicit noise tensors keep the CSR form
    for nm in (ops.NOISE_RANKED, ops.NOISE_HASH, ops.NOISE_HASH_SYM, ops.NOISE_RANKED_SYM, ops.NOISE_NONE):
        assert m._chunk_policy(nm), nm
    assert not m._chunk_policy(ops.NOISE_EXPLICIT)
    for pol, want in (("chunked", True), ("csr", False), ("csr_auto", False), ("ell", False)):
        m.args = Namespace(**base, dgg_wide_rows=pol)
        assert m._chunk_policy(ops.NOISE_RANKED) == want and m._chunk_policy(ops.NOISE_NONE) == want, pol
    m.args = Namespace(**base)
    assert ops.chunk_maxm_for(100_000) == 1564 and ops.chunk_maxm_for(64) == 2, "a row never needs more chunks than hold every column (+ the margin)"
    m2 = dgg_amd.DGG_LearnableK_debug(in_dim=8, latent_dim=16, args=Namespace(**base))
    # edge lists: known before the forward / decided from its learned degrees
    rows = torch.arange(10).repeat_interleave(70)
    A = torch.sparse_coo_tensor(torch.stack([rows, torch.arange(70).repeat(10)]), torch.ones(700), (10, 80)).coalesce()
    rowptr = torch.arange(11) * 70
    assert m2._wide_rows_state(A, rowptr) is None, "rows wider than the list: depends on the learned degrees"
    assert m2._wide_rows(A, rowptr, torch.full((10,), 20.0)) is False and m2._wide_rows_state(A, rowptr) is None
    assert m2._wide_rows(A, rowptr, torch.full((10,), 60.0)) is True and m2._wide_rows_state(A, rowptr) is True, "sticky once needed"
    assert m2._fused_fallback("x") is None and m2.fused_fallback == {"x": 1}


@pytest.mark.parametrize("policy,all_pairs,noise,N,want", [
    ("auto", True, "ranked", 100_000, "chunked"), ("auto", True, "none", 100_000, "chunked"), ("auto", True, "rsym", 3000, "chunked"),
    ("auto", True, "explicit", 3000, "csr_when_needed"), ("auto", True, "explicit", 100_000, "list"), ("auto", False, "hash", 3000, "csr_when_needed"),
    ("chunked", True, "hash", 100_000, "chunked"), ("chunked", True, "explicit", 3000, "list"), ("chunked", False, "hash", 3000, "csr_when_needed"),
    ("csr", True, "ranked", 3000, "csr"), ("csr", True, "ranked", 100_000, "list"), ("csr", False, "hash", 3000, "csr"),
    ("csr_auto", True, "ranked", 3000, "csr_when_needed"), ("csr_auto", False, "hash", 3000, "csr_when_needed"),
    ("ell", True, "ranked", 3000, "list"), ("ell", False, "hash", 3000, "list")])
def test_wide_row_plan_of_every_policy(policy, all_pairs, noise, N, want):
    """args.dgg_wide_rows in one table: what rows wider than 64 ranks do per policy, candidate kind, noise
    generator and graph size -- DGG_LearnableK_debug.wide_row_plan, built from the predicates the forwards themselves use"""
    import dgg_amd
    from argparse import Namespace
    from dgg_amd import ops
    base = dict(extra_edge_dim=0, extra_k_dim=1, dgg_hard=False, deg_mean=3.9, deg_std=5.3, dgg_mode_edge_net="u-v-dist", dgg_mode_k_net="x",
                dgg_mode_k_select="k_times_edge_prob", debug_step=3, perturb_edge_prob=True, symmetric_noise=False, stochastic_k=False,
                dgg_adj_input="input_adj", n_dgg_layers=1, dgg_wide_rows=policy)
    m = dgg_amd.DGG_LearnableK_debug(in_dim=8, latent_dim=16, args=Namespace(**base))
    nm = {"ranked": ops.NOISE_RANKED, "none": ops.NOISE_NONE, "rsym": ops.NOISE_RANKED_SYM, "hash": ops.NOISE_HASH, "explicit": ops.NOISE_EXPLICIT}[noise]
    assert m.wide_row_plan(N, all_pairs, nm) == want
