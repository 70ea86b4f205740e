#!/usr/bin/env python3
"""Generate golden fixtures by IMPORTING the reference (runs only in the build container).

The reference (/root/reference, read-only) is imported unmodified behind harness-side shims
(SURVEY.md Appendix A): torch_geometric / tensorboard stubs, `.cuda()` no-op, `np.float` alias.
Every fixture holds inputs, the reference module's state_dict, the captured Gumbel noise (or the
seed of tests/golden_noise.py that regenerates it), the reference outputs and the reference
gradients for a fixed cotangent.  Fixtures are data only; no reference source is stored.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz
"""
import copy
import json
import os
import sys
import warnings

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")
from unittest.mock import MagicMock  # noqa: E402

for n in ["torch_geometric", "torch_geometric.datasets", "torch_geometric.nn", "torch_geometric.utils",
          "torch_geometric.loader", "torch_geometric.data", "torch_geometric.transforms",
          "torch.utils.tensorboard", "tensorboardX"]:
    sys.modules[n] = MagicMock()
import numpy as np  # noqa: E402
import scipy.sparse  # noqa: E402,F401

np.float = float
import torch  # noqa: E402

torch.Tensor.cuda = lambda self, *a, **k: self
import dgm  # noqa: E402
import model as refmodel  # noqa: E402
from argparse import Namespace  # noqa: E402

from golden_noise import crc, grid_gumbel, grid_normal  # noqa: E402

torch.set_num_threads(8)


def base_args(**kw):
    a = dict(extra_edge_dim=0, extra_k_dim=1, dgg_hard=False, deg_mean=3.899, deg_std=5.288,
             dgg_mode_edge_net="u-v-dist", dgg_mode_k_net="x", dgg_mode_k_select="k_times_edge_prob",
             debug_step=3, perturb_edge_prob=True, symmetric_noise=False, stochastic_k=False,
             dgg_adj_input="input_adj", n_dgg_layers=1)
    a.update(kw)
    return Namespace(**a)


def random_graph(N, avg_deg, gen):
    """symmetric random graph + self loops, coalesced COO (row-major order)."""
    p = avg_deg / N
    A = (torch.rand(N, N, generator=gen) < p / 2).float()
    A = ((A + A.T) > 0).float()
    A.fill_diagonal_(0)
    A = A + torch.eye(N)
    return A.to_sparse().coalesce()


def run_dgg(name, N, d, h, args, in_adj, x, noise, cot, k_scale, store_dense, seed_info=None):
    """noise: None | dense [N,N] float32 (asym) | dense symmetric [N,N] (sym: upper triangle used)."""
    torch.manual_seed(1234)
    m = dgm.DGG_LearnableK_debug(in_dim=d, latent_dim=h, args=args)
    with torch.no_grad():
        m.k_net.k_project.weight.mul_(k_scale)
    m.eval()
    cap = {}
    if noise is not None:
        Gt = torch.from_numpy(noise)
        if args.symmetric_noise:
            iu, ju = torch.triu_indices(N, N, 1)
            m.gumbel.sample = lambda shape: Gt[iu, ju]          # dgm.py:1220 draws len(i) values
        else:
            m.gumbel.sample = lambda shape: Gt.reshape(shape)   # dgm.py:1226 draws [1,N,N]
    _sel = m.select_top_k

    def sel(Nn, k, pert, **kw):
        cap["pert"] = pert.detach().squeeze(0).clone()
        cap["k"] = k.detach().flatten().clone()
        return _sel(Nn, k, pert, **kw)

    m.select_top_k = sel
    x = x.clone().requires_grad_(True)
    out = m(x, in_adj).to_dense()
    loss = (out * cot).sum()
    loss.backward()
    sd = {k_: v.detach().numpy() for k_, v in m.state_dict().items()}
    grads = {k_: (p.grad.detach().numpy() if p.grad is not None else np.zeros_like(p.detach().numpy()))
             for k_, p in m.named_parameters()}
    deg = in_adj.to_dense().sum(-1)
    fx = {
        "x": x.detach().numpy(),
        "deg": deg.numpy(),
        "k": cap["k"].numpy(),
        "g.x": x.grad.numpy(),
    }
    ii = in_adj.indices().numpy().astype(np.int32)
    if in_adj._nnz() < N * N:
        fx["rows"], fx["cols"] = ii[0], ii[1]
        fx["adj_vals"] = in_adj.values().numpy()
    for k_, v in sd.items():
        fx["p." + k_] = v
    for k_, v in grads.items():
        fx["g." + k_] = v
    outd = out.detach()
    if store_dense:
        fx["out"] = outd.numpy()
        fx["pert"] = cap["pert"].numpy()
        fx["cot"] = cot.numpy()
        if noise is not None:
            fx["G"] = noise
    else:
        K = 64
        v, ix = torch.sort(outd, dim=-1, descending=True, stable=True)
        assert (v[:, K:] == 0).all()
        fx["out_idx"] = ix[:, :K].numpy().astype(np.int32)
        fx["out_val"] = v[:, :K].numpy()
        pv = torch.gather(cap["pert"], 1, ix[:, :K])
        fx["pert_val"] = pv.numpy()
    meta = dict(name=name, N=N, d=d, h=h, torch=torch.__version__, args=vars(args), k_scale=k_scale,
                reference="dgm.py:1178-1292 DGG_LearnableK_debug.forward", seed_info=seed_info or {})
    fx["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **fx)
    nnz = (outd != 0).sum(-1).float()
    print(f"{name}: k[{cap['k'].min():.2f},{cap['k'].max():.2f}] nnz/row mean {nnz.mean():.1f} max {nnz.max():.0f}")


def edgelist_cases():
    N, d, h = 96, 24, 16
    gen = torch.Generator().manual_seed(7)
    in_adj = random_graph(N, 24, gen)
    x = torch.randn(N, d, generator=gen)
    cot = torch.from_numpy(grid_normal(11, (N, N)))
    Gasym = grid_gumbel(21, (N, N))
    Gsym = grid_gumbel(22, (N, N))
    Gsym = np.triu(Gsym, 1) + np.triu(Gsym, 1).T
    for ksel in ["k_times_edge_prob", "k_only"]:
        for nz in ["none", "asym", "sym"]:
            a = base_args(dgg_mode_k_select=ksel, perturb_edge_prob=(nz != "none"),
                          symmetric_noise=(nz == "sym"))
            noise = None if nz == "none" else (Gasym if nz == "asym" else Gsym)
            run_dgg(f"edgelist_{ksel}_{nz}", N, d, h, a, in_adj, x, noise, cot, 1.0, True)
    a = base_args(dgg_mode_k_net="input_deg", perturb_edge_prob=False, deg_mean=20.0, deg_std=4.0)
    run_dgg("edgelist_inputdeg_none", N, d, h, a, in_adj, x, None, cot, 1.0, True)
    a = base_args(dgg_mode_k_net="learn_normalized_degree", perturb_edge_prob=True)
    run_dgg("edgelist_lnd_asym", N, d, h, a, in_adj, x, Gasym, cot, 1.0, True)
    a = base_args(dgg_mode_k_net="gcn-x-deg", perturb_edge_prob=False)
    run_dgg("edgelist_gcnxdeg_none", N, d, h, a, in_adj, x, None, cot, 1.0, True)


def edgemlp_cases():
    """edge-MLP scorers of the live class (dgm.py:1628-1725) on a weighted candidate graph"""
    N, d, h = 96, 24, 16
    gen = torch.Generator().manual_seed(17)
    A = random_graph(N, 24, gen).to_dense()
    Wt = 0.5 + torch.rand(N, N, generator=gen)
    Wt = (Wt + Wt.T) / 2
    in_adj = (A * Wt).to_sparse().coalesce()            # non-unit edge values: a_uv and the degrees are informative
    x = torch.randn(N, d, generator=gen)
    cot = torch.from_numpy(grid_normal(61, (N, N)))
    G = grid_gumbel(62, (N, N))
    for mode, extra, nz in [("u-v-deg", 2, "none"), ("u-v-deg", 2, "asym"), ("u-v-A_uv", 1, "none"),
                            ("u-v-deg-dist", 3, "asym"), ("edge_conv", 0, "none"), ("A_uv", 0, "none")]:
        a = base_args(dgg_mode_edge_net=mode, extra_edge_dim=extra, perturb_edge_prob=(nz != "none"))
        run_dgg(f"edgemlp_{mode.replace('-', '').replace('_', '')}_{nz}", N, d, h, a, in_adj, x,
                None if nz == "none" else G, cot, 1.0, True)


def dggclass_cases():
    """`DGG` "for ICLR" (dgm.py:1730-1815) and its wrapper GCN_DGG_00 (model.py:1314-1433); hub rows exceed 64 entries"""
    N, d, h, C = 160, 20, 16, 5
    gen = torch.Generator().manual_seed(27)
    A = random_graph(N, 12, gen).to_dense()
    A[:3, :] = (torch.rand(3, N, generator=gen) < 0.6).float()      # three hubs with ~100 neighbours (> ELL width)
    A = ((A + A.T) > 0).float()
    A.fill_diagonal_(1.0)
    in_adj = A.to_sparse().coalesce()
    x = torch.randn(N, d, generator=gen)
    a = base_args()
    torch.manual_seed(99)
    m = dgm.DGG(in_dim=d, latent_dim=h, args=a)
    m.eval()
    xr = x.clone().requires_grad_(True)
    out, xe = m(xr, in_adj)
    outd = out.to_dense()
    cot = torch.from_numpy(grid_normal(71, (N, N)))
    cote = torch.from_numpy(grid_normal(72, (N, h)))
    ((outd * cot).sum() + (xe * cote).sum()).backward()
    ii = in_adj.indices().numpy().astype(np.int32)
    fx = {"x": x.numpy(), "rows": ii[0], "cols": ii[1], "adj_vals": in_adj.values().numpy(), "out": outd.detach().numpy(),
          "xe": xe.detach().numpy(), "cot": cot.numpy(), "cote": cote.numpy(), "g.x": xr.grad.numpy()}
    for k_, v in m.state_dict().items():
        fx["p." + k_] = v.detach().numpy()
    for k_, p_ in m.named_parameters():
        fx["g." + k_] = p_.grad.numpy()
    meta = dict(name="dggclass", N=N, d=d, h=h, torch=torch.__version__, args=vars(a), reference="dgm.py:1758-1815 DGG.forward")
    fx["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "dggclass.npz"), **fx)
    print("dggclass ok: max row nnz", int((outd != 0).sum(-1).max()))
    # wrapper: in_adj WITHOUT self loops (the wrapper adds them, model.py:1380-1383)
    A2 = (A - torch.eye(N)).to_sparse().coalesce()
    torch.manual_seed(4321)
    mm = refmodel.GCN_DGG_00(nfeat=d, nlayers=2, nhidden=h, nclass=C, args=a)
    mm.eval()
    logp, unnorm, x_dgg = mm(x, A2)
    cot2 = torch.from_numpy(grid_normal(73, (N, C)))
    (logp * cot2).sum().backward()
    fx = {"x": x.numpy(), "rows": A2.indices()[0].numpy().astype(np.int32), "cols": A2.indices()[1].numpy().astype(np.int32),
          "adj_vals": A2.values().numpy(), "cot": cot2.numpy(), "out": logp.detach().numpy(),
          "unnorm": unnorm.detach().to_dense().numpy(), "x_dgg": x_dgg.detach().numpy()}
    for k_, v in mm.state_dict().items():
        fx["p." + k_] = v.detach().numpy()
    for k_, p_ in mm.named_parameters():
        fx["g." + k_] = p_.grad.numpy() if p_.grad is not None else np.zeros_like(p_.detach().numpy())
    meta = dict(name="model_gcn_dgg_00", N=N, d=d, h=h, C=C, torch=torch.__version__, args=vars(a))
    fx["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "model_gcn_dgg_00.npz"), **fx)
    print("model_gcn_dgg_00 ok", tuple(logp.shape))


def cora_cases():
    """BASELINE configs[0] (SURVEY 8b'): Cora through the reference's own loader + add_noisy_edges, one eval-mode forward of
    GCN_DGG_00 (the wrapper the small-graph script can drive as shipped: it accepts the script's edge_index kwarg)."""
    import utils as refutils
    cwd = os.getcwd()
    os.chdir("/root/reference")                                  # load_citation opens data/ind.*.test.index relatively
    try:
        adj, feats, labels, itr, iva, ite = refutils.load_citation("cora", "/root/reference")
    finally:
        os.chdir(cwd)
    adj = adj.coalesce()
    N = feats.shape[0]
    keep = adj.values() != 0                                     # the loader stores explicit zeros on the diagonal
    ei = adj.indices()[:, keep]
    A0 = scipy.sparse.coo_matrix((np.ones(ei.shape[1], np.float32), (ei[0].numpy(), ei[1].numpy())), shape=(N, N))
    noisy = refutils.add_noisy_edges(A0, noise_level=0.00014).tocoo()             # train_small_graphs.py default level
    order = np.lexsort((noisy.col, noisy.row))
    nr, nc, nv = noisy.row[order].astype(np.int32), noisy.col[order].astype(np.int32), noisy.data[order].astype(np.float32)
    fnz = feats.to_sparse().coalesce()
    h, C = 16, int(labels.max()) + 1
    a = base_args()
    torch.manual_seed(4321)
    m = refmodel.GCN_DGG_00(nfeat=feats.shape[1], nlayers=2, nhidden=h, nclass=C, args=a)
    m.eval()
    A = torch.sparse_coo_tensor(torch.from_numpy(np.stack([nr, nc]).astype(np.int64)), torch.from_numpy(nv), (N, N)).coalesce()
    with torch.no_grad():
        logp, unnorm, x_dgg = m(feats, A)
    loss = torch.nn.functional.nll_loss(logp[itr], labels[itr])
    fx = {"feat_rows": fnz.indices()[0].numpy().astype(np.int16), "feat_cols": fnz.indices()[1].numpy().astype(np.int16),
          "feat_vals": fnz.values().numpy(), "rows": ei[0].numpy().astype(np.int32), "cols": ei[1].numpy().astype(np.int32),
          "noisy_rows": nr, "noisy_cols": nc, "labels": labels.numpy().astype(np.int16), "train_idx": itr.numpy(),
          "val_idx": iva.numpy(), "test_idx": ite.numpy(), "out": logp.numpy(), "loss": np.float32(loss.item())}
    for k_, v in m.state_dict().items():
        fx["p." + k_] = v.detach().numpy()
    meta = dict(name="cora_gcn_dgg_00", N=N, d=int(feats.shape[1]), h=h, C=C, torch=torch.__version__, args=vars(a),
                edge_noise_level=0.00014, reference="utils.load_citation + add_noisy_edges + model.GCN_DGG_00 (eval)")
    fx["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "cora_gcn_dgg_00.npz"), **fx)
    print("cora ok: edges", ei.shape[1], "noisy", len(nr), "loss", float(loss), "max row", int(np.bincount(nr).max()))


def edgelist512_cases():
    """SURVEY 8(c) G2 at N = 512: sparse candidates (avg degree 40: rows wider than the ramp support), noise / cotangent
    regenerated from seeds at test time, outputs stored as the top-64 of every row (exact: the rest is zero)"""
    N, d, h = 512, 48, 32
    gen = torch.Generator().manual_seed(71)
    in_adj = random_graph(N, 40, gen)
    x = torch.from_numpy(grid_normal(72, (N, d)))
    G = grid_gumbel(73, (N, N))
    cotn = grid_normal(74, (N, N))
    for nz in ["none", "asym"]:
        a = base_args(perturb_edge_prob=(nz != "none"))
        run_dgg(f"edgelist_n512_{nz}", N, d, h, a, in_adj, x, None if nz == "none" else G, torch.from_numpy(cotn), 1.0, False,
                seed_info=dict(x_seed=72, G_seed=73, cot_seed=74, G_crc=crc(G), cot_crc=crc(cotn)))


def hard_cases():
    """dgg_hard=True through the live class's LITERAL return_hard_or_soft (dgm.py:1294-1311) with k_times_edge_prob -- the one
    select mode it runs for.  Stored: the dense hard output, the soft adjacency it was derived from, the sort
    permutation select_top_k returned (idxs) and the gradients of a fixed cotangent."""
    for tag, N, d, h, allp in [("edgelist", 96, 24, 16, False), ("allpairs", 128, 32, 16, True)]:
        gen = torch.Generator().manual_seed(81 + N)
        x = torch.randn(N, d, generator=gen)
        if allp:
            c = 8 + 10 * torch.rand(N, 1, generator=gen)
            in_adj = (c / N * torch.ones(N, N)).to_sparse().coalesce()
        else:
            in_adj = random_graph(N, 24, gen)
        G = grid_gumbel(83 + N, (N, N))
        cot = torch.from_numpy(grid_normal(84 + N, (N, N)))
        a = base_args(dgg_hard=True)
        torch.manual_seed(1234)
        m = dgm.DGG_LearnableK_debug(in_dim=d, latent_dim=h, args=a)
        with torch.no_grad():
            m.k_net.k_project.weight.mul_(30.0 if allp else 1.0)
        m.eval()
        Gt = torch.from_numpy(G)
        m.gumbel.sample = lambda shape: Gt.reshape(shape)
        cap = {}
        _sel, _ret = m.select_top_k, m.return_hard_or_soft

        def sel(Nn, k, pert, **kw):
            cap["pert"], cap["k"] = pert.detach().squeeze(0).clone(), k.detach().flatten().clone()
            out = _sel(Nn, k, pert, **kw)
            cap["soft"], cap["idxs"] = out[0].detach().squeeze(0).clone(), out[2].detach().squeeze(0).clone()
            return out

        m.select_top_k = sel
        xg = x.clone().requires_grad_(True)
        out = m(xg, in_adj).to_dense()
        (out * cot).sum().backward()
        fx = {"x": x.numpy(), "deg": in_adj.to_dense().sum(-1).numpy(), "k": cap["k"].numpy(), "G": G, "cot": cot.numpy(),
              "out": out.detach().numpy(), "soft": cap["soft"].numpy(), "idxs": cap["idxs"].numpy().astype(np.int16),
              "pert": cap["pert"].numpy(), "g.x": xg.grad.numpy()}
        if not allp:
            ii = in_adj.indices().numpy().astype(np.int32)
            fx["rows"], fx["cols"], fx["adj_vals"] = ii[0], ii[1], in_adj.values().numpy()
        for k_, v in m.state_dict().items():
            fx["p." + k_] = v.detach().numpy()
        for k_, p_ in m.named_parameters():
            fx["g." + k_] = p_.grad.numpy() if p_.grad is not None else np.zeros_like(p_.detach().numpy())
        meta = dict(name=f"hard_{tag}", N=N, d=d, h=h, torch=torch.__version__, args=vars(a), k_scale=30.0 if allp else 1.0,
                    reference="dgm.py:1294-1311 return_hard_or_soft (dgg_hard=True) after select_top_k k_times_edge_prob")
        fx["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        np.savez_compressed(os.path.join(HERE, f"hard_{tag}.npz"), **fx)
        print(f"hard_{tag}: ones/row {float((out != 0).sum(-1).float().mean()):.2f}, soft>0.5/row "
              f"{float((cap['soft'] > 0.5).sum(-1).float().mean()):.2f}, |g.We| {np.abs(fx['g.node_encode_for_edges.0.weight']).max():.3e}")


def cora_model_cases():
    """SURVEY 8(c) G8 / BASELINE configs[0] in its NAMED form: the reference's own Cora tensors (cora_gcn_dgg_00.npz holds the
    inputs: features, clean and noisy edges, split) through `--model GCN_DGG` with the script's default scorer made runnable
    (u-v-deg needs extra_edge_dim=2; perturb_edge_prob False as in train_small_graphs.py:157-163), through
    GCNII_DGG with nlayers=4 and through GCNIIppi_DGG; eval mode, h = 16.  Only outputs + state_dicts are stored here."""
    z = np.load(os.path.join(HERE, "cora_gcn_dgg_00.npz"))
    N, dF = 2708, 1433
    feats = torch.zeros(N, dF)
    feats[z["feat_rows"].astype(np.int64), z["feat_cols"].astype(np.int64)] = torch.from_numpy(z["feat_vals"])
    A = torch.sparse_coo_tensor(torch.from_numpy(np.stack([z["noisy_rows"], z["noisy_cols"]]).astype(np.int64)),
                                torch.ones(len(z["noisy_rows"])), (N, N)).coalesce()
    labels, itr = torch.from_numpy(z["labels"].astype(np.int64)), torch.from_numpy(z["train_idx"])
    h, C = 16, 7
    for name, a, ctor in [
        ("cora_gcn_dgg", base_args(dgg_mode_edge_net="u-v-deg", extra_edge_dim=2, perturb_edge_prob=False, symmetric_noise=True),
         lambda a: refmodel.GCN_DGG(nfeat=dF, nlayers=2, nhidden=h, nclass=C, args=a)),
        ("cora_gcnii_dgg", base_args(perturb_edge_prob=False),
         lambda a: refmodel.GCNII_DGG(nfeat=dF, nlayers=4, nhidden=h, nclass=C, dropout=0.5, lamda=0.5, alpha=0.1, variant=False, args=a)),
        ("cora_gcniippi_dgg", base_args(perturb_edge_prob=False),
         lambda a: refmodel.GCNIIppi_DGG(nfeat=dF, nlayers=4, nhidden=h, nclass=C, dropout=0.5, lamda=0.5, alpha=0.1, variant=True, args=a)),
    ]:
        torch.manual_seed(4321)
        m = ctor(a)
        m.eval()
        # per row: the smallest relative gap between consecutive sorted scores among the ranks that carry ramp weight.  Two ranks
        # a few ulp apart are ordered by last-bit rounding; a swap exchanges two ramp weights.  The test lets
        # only such rows (and the rows that aggregate them) deviate.
        gaps = []
        for dg in m.dggs:
            _sel = dg.select_top_k

            def sel(Nn, k, pert, _sel=_sel, **kw):
                sv = torch.sort(pert.detach().squeeze(0), dim=-1, descending=True).values[:, :64]
                live = (torch.arange(64)[None, :] < (k.detach().reshape(-1, 1) + 9.5)) & (sv > 0)
                g_ = (sv[:, :-1] - sv[:, 1:]) / sv[:, :-1].clamp(min=1e-30)
                g_ = torch.where(live[:, :-1] & live[:, 1:], g_, torch.ones_like(g_))
                gaps.append(g_.min(1).values)
                return _sel(Nn, k, pert, **kw)

            dg.select_top_k = sel
        with torch.no_grad():
            out = m(feats, A)
        logp = out[0] if isinstance(out, tuple) else out
        fx = {"out": logp.numpy(), "tie_gap": torch.stack(gaps).min(0).values.numpy()}
        if name == "cora_gcn_dgg":
            fx["loss"] = np.float32(torch.nn.functional.nll_loss(logp[itr], labels[itr]).item())
            un = out[1].to_dense()
            fx["unnorm_rowsum"] = un.sum(-1).numpy()
            fx["unnorm_nnz"] = (un != 0).sum(-1).numpy().astype(np.int32)
        for k_, v in m.state_dict().items():
            fx["p." + k_] = v.detach().numpy()
        meta = dict(name=name, N=N, d=dF, h=h, C=C, torch=torch.__version__, args=vars(a), inputs="cora_gcn_dgg_00.npz",
                    reference="model.py GCN_DGG 1183-1311 / GCNII_DGG 649-740 (nlayers=4) / GCNIIppi_DGG 887-965 on utils.load_citation('cora')")
        fx["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **fx)
        print(name, "ok", tuple(logp.shape), float(logp.abs().max()))


def gat_cases():
    """GAT_DGG_00 (model.py:323-403) with its dense [N,N] attention; in_adj carries noisy edges that edge_index lacks."""
    refmodel.remove_self_loops = lambda ei: (ei[:, ei[0] != ei[1]], None)                     # torch_geometric.utils stand-ins
    refmodel.add_self_loops = lambda ei, num_nodes=None: (torch.cat([ei, torch.arange(num_nodes).repeat(2, 1)], 1), None)
    N, d, h, C = 140, 20, 8, 4
    gen = torch.Generator().manual_seed(37)
    A = random_graph(N, 8, gen).to_dense() - torch.eye(N)
    ei = A.to_sparse().coalesce().indices()
    extra = (torch.rand(N, N, generator=gen) < 0.01).float() * (1 - torch.eye(N))
    in_adj = ((A + extra) > 0).float().to_sparse().coalesce()
    x = torch.randn(N, d, generator=gen)
    a = base_args()
    torch.manual_seed(777)
    m = refmodel.GAT_DGG_00(nfeat=d, nhidden=h, nclass=C, args=a, nhead=3)
    with torch.no_grad():
        for p_ in m.parameters():
            if p_.ndim == 1 and p_.numel() in (h, C):
                p_.add_(0.1 * torch.randn(p_.shape, generator=gen))     # biases are zero-initialised: make them matter
    m.eval()
    logp, unnorm, x_dgg = m(x, in_adj=in_adj, edge_index=ei)
    cot = torch.from_numpy(grid_normal(81, (N, C)))
    (logp * cot).sum().backward()
    fx = {"x": x.numpy(), "rows": in_adj.indices()[0].numpy().astype(np.int32), "cols": in_adj.indices()[1].numpy().astype(np.int32),
          "adj_vals": in_adj.values().numpy(), "ei": ei.numpy().astype(np.int32), "cot": cot.numpy(), "out": logp.detach().numpy()}
    for k_, v in m.state_dict().items():
        fx["p." + k_] = v.detach().numpy()
    for k_, p_ in m.named_parameters():
        fx["g." + k_] = p_.grad.numpy() if p_.grad is not None else np.zeros_like(p_.detach().numpy())
    meta = dict(name="model_gat_dgg_00", N=N, d=d, h=h, C=C, nhead=3, torch=torch.__version__, args=vars(a))
    fx["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "model_gat_dgg_00.npz"), **fx)
    print("model_gat_dgg_00 ok", tuple(logp.shape), "keys", len(m.state_dict()))


def ablations_cases():
    """`DGG_Ablations` (dgm.py:1876-1968) with learned k and with k=int, and its wrappers GCN_DGG_Ablations (model.py:1436-1561)
    and GAT_DGG_Ablations (model.py:406-486).  The per-edge U(-1,1) rank noise (dgm.py:1932) is captured from torch.rand."""
    N, d, h, C = 150, 20, 16, 5
    gen = torch.Generator().manual_seed(41)
    A = random_graph(N, 10, gen).to_dense()
    A[:2, :] = (torch.rand(2, N, generator=gen) < 0.55).float()      # two hubs wider than the ELL width
    A = ((A + A.T) > 0).float()
    A.fill_diagonal_(1.0)
    in_adj = A.to_sparse().coalesce()
    x = torch.randn(N, d, generator=gen)
    a = base_args()
    real_rand = torch.rand
    captured = []

    def rec_rand(*args, **kw):
        kw.pop("device", None)
        r = real_rand(*args, **kw)
        captured.append(r.clone())
        return r

    def common(mod, extra):
        fx = dict(extra)
        for k_, v in mod.state_dict().items():
            fx["p." + k_] = v.detach().numpy()
        for k_, p_ in mod.named_parameters():
            fx["g." + k_] = p_.grad.numpy() if p_.grad is not None else np.zeros_like(p_.detach().numpy())
        return fx

    ii = in_adj.indices().numpy().astype(np.int32)
    for tag, kfix in [("learnk", None), ("k5", 5)]:
        torch.manual_seed(99)
        m = dgm.DGG_Ablations(in_dim=d, latent_dim=h, args=a)
        m.eval()
        xr = x.clone().requires_grad_(True)
        captured.clear()
        torch.manual_seed(1234)
        torch.rand = rec_rand
        try:
            out, xe = m(xr, in_adj, k=kfix)
        finally:
            torch.rand = real_rand
        assert len(captured) == 1
        outd = out.to_dense()
        cot = torch.from_numpy(grid_normal(91, (N, N)))
        cote = torch.from_numpy(grid_normal(92, (N, h)))
        ((outd * cot).sum() + (xe * cote).sum()).backward()
        fx = common(m, {"x": x.numpy(), "rows": ii[0], "cols": ii[1], "adj_vals": in_adj.values().numpy(),
                        "noise": (captured[0] * 2 - 1).numpy(), "out": outd.detach().numpy(), "xe": xe.detach().numpy(),
                        "cot": cot.numpy(), "cote": cote.numpy(), "g.x": xr.grad.numpy()})
        meta = dict(name="ablations_" + tag, N=N, d=d, h=h, k=kfix, torch=torch.__version__, args=vars(a),
                    reference="dgm.py:1904-1968 DGG_Ablations.forward")
        fx["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        np.savez_compressed(os.path.join(HERE, f"ablations_{tag}.npz"), **fx)
        print("ablations", tag, "ok: nnz/row max", int((outd != 0).sum(-1).max()), "min", int((outd != 0).sum(-1).min()))
    # wrappers: in_adj WITHOUT self loops (the wrappers add them)
    A2 = (A - torch.eye(N)).to_sparse().coalesce()
    refmodel.remove_self_loops = lambda ei: (ei[:, ei[0] != ei[1]], None)                     # torch_geometric.utils stand-ins
    refmodel.add_self_loops = lambda ei, num_nodes=None: (torch.cat([ei, torch.arange(num_nodes).repeat(2, 1)], 1), None)
    for name, ctor, kw in [("model_gcn_dgg_ablations", refmodel.GCN_DGG_Ablations, {}),
                           ("model_gat_dgg_ablations", refmodel.GAT_DGG_Ablations, {"nhead": 2})]:
        torch.manual_seed(4321)
        mm = ctor(nfeat=d, nlayers=2, nhidden=h, nclass=C, args=a, **kw)
        mm.eval()
        captured.clear()
        torch.manual_seed(1235)
        torch.rand = rec_rand
        try:
            if "gat" in name:
                logp, unnorm, x_dgg = mm(x, in_adj=A2, edge_index=A2.indices())
            else:
                logp, unnorm, x_dgg = mm(x, A2, epoch=1)
        finally:
            torch.rand = real_rand
        assert len(captured) == 1
        cot2 = torch.from_numpy(grid_normal(93, (N, C)))
        (logp * cot2).sum().backward()
        fx = common(mm, {"x": x.numpy(), "rows": A2.indices()[0].numpy().astype(np.int32),
                         "cols": A2.indices()[1].numpy().astype(np.int32), "adj_vals": A2.values().numpy(),
                         "noise": (captured[0] * 2 - 1).numpy(), "cot": cot2.numpy(), "out": logp.detach().numpy(),
                         "unnorm": unnorm.detach().to_dense().numpy(), "x_dgg": x_dgg.detach().numpy()})
        meta = dict(name=name, N=N, d=d, h=h, C=C, torch=torch.__version__, args=vars(a), **kw)
        fx["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **fx)
        print(name, "ok", tuple(logp.shape))


def dense_cases():
    """Dense all-pairs alternates (goldens G5/G6): DGG_LearnableK_SDD(dist_fn="metric", noise=False,
    hard in {F,T}) (dgm.py:259-351; needs k_net.args from the harness) and DGG_StraightThrough(dist_fn="metric",
    noise=False, hard in {T,F}) (dgm.py:140-182).  N=24 keeps torch.cdist on its direct-difference path (<= 25 rows),
    N=160 uses its matmul form."""
    for N, B, d, h in [(24, 3, 12, 16), (160, 2, 20, 32)]:
        gen = torch.Generator().manual_seed(1000 + N)
        x = torch.randn(B, N, d, generator=gen)
        cot = torch.from_numpy(grid_normal(300 + N, (B, N, N)))
        cotk = torch.from_numpy(grid_normal(301 + N, (B, N, 1)))
        for hard in (False, True):
            m = dgm.DGG_LearnableK_SDD(in_dim=d, latent_dim=h, k_bias=3.0, hard=hard, dist_fn="metric")
            m.k_net.args = Namespace(stochastic_k=False)
            with torch.no_grad():
                m.t.fill_(6.0)                           # softmax-projected features are close together: make distances matter
                m.k_net.k_project.weight.mul_(4.0)
            m.eval()
            xr = x.clone().requires_grad_(True)
            temp = 0.5
            adj, k = m(xr, temp, noise=False)
            ((adj * cot).sum() + (k * cotk).sum()).backward()
            fx = {"x": x.numpy(), "out": adj.detach().numpy(), "k": k.detach().numpy(), "cot": cot.numpy(), "cotk": cotk.numpy(),
                  "g.x": xr.grad.numpy()}
            for k_, v in m.state_dict().items():
                fx["p." + k_] = v.detach().numpy()
            for k_, p_ in m.named_parameters():
                fx["g." + k_] = p_.grad.numpy() if p_.grad is not None else np.zeros_like(p_.detach().numpy())
            # the same module evaluated in float64: torch.cdist's matmul form (> 25 rows) loses ~1e-4 on near-zero distances
            # in fp32, so the fp32 reference output is itself ~2e-4 away from this one on the diagonal
            m64 = copy.deepcopy(m).double()
            m64.zero_grad()
            x64 = x.double().requires_grad_(True)
            adj64, k64 = m64(x64, temp, noise=False)
            ((adj64 * cot.double()).sum() + (k64 * cotk.double()).sum()).backward()
            fx["out64"], fx["g64.x"] = adj64.detach().float().numpy(), x64.grad.float().numpy()
            for k_, p_ in m64.named_parameters():
                fx["g64." + k_] = p_.grad.float().numpy() if p_.grad is not None else np.zeros(tuple(p_.shape), np.float32)
            meta = dict(name=f"sdd_n{N}_h{int(hard)}", B=B, N=N, d=d, h=h, hard=hard, temp=temp, k_bias=3.0, torch=torch.__version__,
                        reference="dgm.py:259-351 DGG_LearnableK_SDD.forward(noise=False)")
            fx["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
            np.savez_compressed(os.path.join(HERE, f"sdd_n{N}_h{int(hard)}.npz"), **fx)
            print("sdd", N, hard, "ok: k range", float(k.min()), float(k.max()), "nnz>1e-6 per row", float((adj > 1e-6).sum(-1).float().mean()))
        for hard in (True, False):
            torch.manual_seed(5)
            m = dgm.DGG_StraightThrough(in_dim=d, latent_dim=h, k=5, hard=hard, dist_fn="metric")
            with torch.no_grad():
                m.t.fill_(0.7)
            m.eval()
            xr = x.clone().requires_grad_(True)
            temp = 0.8
            adj = m(xr, temp, noise=False)
            (adj * cot).sum().backward()
            fx = {"x": x.numpy(), "out": adj.detach().numpy(), "cot": cot.numpy(), "g.x": xr.grad.numpy()}
            for k_, v in m.state_dict().items():
                fx["p." + k_] = v.detach().numpy()
            for k_, p_ in m.named_parameters():
                fx["g." + k_] = p_.grad.numpy() if p_.grad is not None else np.zeros_like(p_.detach().numpy())
            m64 = copy.deepcopy(m).double()
            m64.zero_grad()
            x64 = x.double().requires_grad_(True)
            adj64 = m64(x64, temp, noise=False)
            (adj64 * cot.double()).sum().backward()
            fx["out64"], fx["g64.x"] = adj64.detach().float().numpy(), x64.grad.float().numpy()
            for k_, p_ in m64.named_parameters():
                fx["g64." + k_] = p_.grad.float().numpy() if p_.grad is not None else np.zeros(tuple(p_.shape), np.float32)
            meta = dict(name=f"st_n{N}_h{int(hard)}", B=B, N=N, d=d, h=h, hard=hard, temp=temp, k=5, torch=torch.__version__,
                        reference="dgm.py:140-182 DGG_StraightThrough.forward(noise=False)")
            fx["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
            np.savez_compressed(os.path.join(HERE, f"st_n{N}_h{int(hard)}.npz"), **fx)
            print("st", N, hard, "ok", float(adj.sum(-1).mean()))


def scores_cases():
    """The forward variants of the live class that return the raw edge probabilities as the adjacency: debug_step 0
    (dgm.py:1202-1209), debug_step 1 (dgm.py:1240-1246) and the k-select mode `edge_p-cdf` (dgm.py:1368-1401, which scatters the
    UNSORTED probabilities back, so its output is edge_p too)."""
    N, d, h = 96, 24, 16
    gen = torch.Generator().manual_seed(19)
    A = random_graph(N, 20, gen).to_dense()
    A[0, :] = (torch.rand(N, generator=gen) < 0.8).float()          # one row wider than the ELL width
    A[0, 0] = 1.0
    Wt = 0.5 + torch.rand(N, N, generator=gen)
    in_adj = (A * Wt).to_sparse().coalesce()
    x = torch.randn(N, d, generator=gen)
    cot = torch.from_numpy(grid_normal(161, (N, N)))
    for tag, kw in [("debug0_uvdist", dict(debug_step=0, dgg_mode_edge_net="u-v-dist", perturb_edge_prob=True)),
                    ("cdf_uvdeg", dict(dgg_mode_k_select="edge_p-cdf", dgg_mode_edge_net="u-v-deg", extra_edge_dim=2,
                                       perturb_edge_prob=False)),
                    ("debug1_uvdegdist", dict(debug_step=1, dgg_mode_edge_net="u-v-deg-dist", extra_edge_dim=3,
                                              perturb_edge_prob=False)),
                    ("cdf_edgeconv", dict(dgg_mode_k_select="edge_p-cdf", dgg_mode_edge_net="edge_conv", perturb_edge_prob=False)),
                    ("debug0_uvAuv", dict(debug_step=0, dgg_mode_edge_net="u-v-A_uv", extra_edge_dim=1, perturb_edge_prob=False))]:
        a = base_args(**kw)
        torch.manual_seed(1234)
        m = dgm.DGG_LearnableK_debug(in_dim=d, latent_dim=h, args=a)
        m.eval()
        xr = x.clone().requires_grad_(True)
        out = m(xr, in_adj).to_dense()
        (out * cot).sum().backward()
        ii = in_adj.indices().numpy().astype(np.int32)
        fx = {"x": x.numpy(), "rows": ii[0], "cols": ii[1], "adj_vals": in_adj.values().numpy(), "out": out.detach().numpy(),
              "cot": cot.numpy(), "g.x": xr.grad.numpy()}
        for k_, v in m.state_dict().items():
            fx["p." + k_] = v.detach().numpy()
        for k_, p_ in m.named_parameters():
            fx["g." + k_] = p_.grad.numpy() if p_.grad is not None else np.zeros_like(p_.detach().numpy())
        meta = dict(name="scores_" + tag, N=N, d=d, h=h, torch=torch.__version__, args=vars(a))
        fx["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        np.savez_compressed(os.path.join(HERE, f"scores_{tag}.npz"), **fx)
        print("scores", tag, "ok: nnz", int((out != 0).sum()), "of", in_adj._nnz(), "stored")


def allpairs_cases():
    N, d, h = 256, 32, 16
    gen = torch.Generator().manual_seed(8)
    x = torch.randn(N, d, generator=gen)
    c = 8 + 10 * torch.rand(N, 1, generator=gen)
    in_adj = (c / N * torch.ones(N, N)).to_sparse().coalesce()
    cot = torch.from_numpy(grid_normal(12, (N, N)))
    Gasym = grid_gumbel(31, (N, N))
    Gsym = grid_gumbel(32, (N, N))
    Gsym = np.triu(Gsym, 1) + np.triu(Gsym, 1).T
    for nz in ["none", "asym", "sym"]:
        a = base_args(perturb_edge_prob=(nz != "none"), symmetric_noise=(nz == "sym"))
        noise = None if nz == "none" else (Gasym if nz == "asym" else Gsym)
        run_dgg(f"allpairs_n256_{nz}", N, d, h, a, in_adj, x, noise, cot, 30.0, True)
    # larger, bench-like latent; noise / cotangent regenerated from seeds at test time
    N, d, h = 1024, 64, 64
    x = torch.from_numpy(grid_normal(41, (N, d)))
    c = torch.from_numpy(24 + 16 * np.random.Generator(np.random.PCG64(42)).random((N, 1))).float()
    in_adj = (c / N * torch.ones(N, N)).to_sparse().coalesce()
    G = grid_gumbel(43, (N, N))
    cotn = grid_normal(44, (N, N))
    a = base_args()
    run_dgg("allpairs_n1024_asym", N, d, h, a, in_adj, x, G, torch.from_numpy(cotn), 3.0, False,
            seed_info=dict(x_seed=41, G_seed=43, cot_seed=44, G_crc=crc(G), cot_crc=crc(cotn)))


def conv_cases():
    """normalize_adj (model.py:1205-1219) + GCNConv (580-599) + GraphConvolution/DenseGraphConvolution (14-77)."""
    N, F, H = 64, 20, 12
    gen = torch.Generator().manual_seed(9)
    A = torch.rand(N, N, generator=gen) * (torch.rand(N, N, generator=gen) < 0.2).float()
    A = A + torch.eye(N)
    A.requires_grad_(True)
    x = torch.randn(N, F, generator=gen, requires_grad=True)
    holder = refmodel.GCN_DGG.__new__(refmodel.GCN_DGG)
    norm = refmodel.GCN_DGG.normalize_adj(holder, A)
    conv = refmodel.GCNConv(F, H)
    with torch.no_grad():
        conv.W.copy_(torch.rand(F, H, generator=gen))
    out = conv(x, norm)
    cot = torch.from_numpy(grid_normal(13, (N, H)))
    (out * cot).sum().backward()
    fx = dict(A=A.detach().numpy(), x=x.detach().numpy(), W=conv.W.detach().numpy(), norm=norm.detach().numpy(),
              out=out.detach().numpy(), cot=cot.numpy(), gA=A.grad.numpy(), gx=x.grad.numpy(),
              gW=conv.W.grad.numpy())
    np.savez_compressed(os.path.join(HERE, "conv_gcn.npz"), **fx)
    print("conv_gcn ok")
    # GCNII layers
    for variant in [False, True]:
        for residual in [False, True]:
            A2 = norm.detach().clone().requires_grad_(True)
            inp = torch.randn(N, H, generator=gen, requires_grad=True)
            h0 = torch.randn(N, H, generator=gen, requires_grad=True)
            for cls, tag in [(refmodel.GraphConvolution, "sp"), (refmodel.DenseGraphConvolution, "dn")]:
                for t in (A2, inp, h0):
                    t.grad = None
                layer = cls(H, H, residual=residual, variant=variant)
                with torch.no_grad():
                    layer.weight.copy_((torch.rand(layer.weight.shape, generator=gen) - 0.5) * 0.5)
                o = layer(inp, A2, h0, 0.5, 0.1, 3)
                cot2 = torch.from_numpy(grid_normal(14, (N, H)))
                (o * cot2).sum().backward()
                fx = dict(A=A2.detach().numpy(), inp=inp.detach().numpy(), h0=h0.detach().numpy(),
                          W=layer.weight.detach().numpy(), out=o.detach().numpy(), cot=cot2.numpy(),
                          gA=A2.grad.numpy(), ginp=inp.grad.numpy(), gh0=h0.grad.numpy(),
                          gW=layer.weight.grad.numpy(), lamda=0.5, alpha=0.1, l=3)
                np.savez_compressed(os.path.join(HERE, f"conv_gcnii_{tag}_v{int(variant)}_r{int(residual)}.npz"), **fx)
    print("conv_gcnii ok")


if __name__ == "__main__":
    which = sys.argv[1:] or ["edgelist", "allpairs", "conv"]
    if "edgelist" in which:
        edgelist_cases()
    if "allpairs" in which:
        allpairs_cases()
    if "conv" in which:
        conv_cases()
    if "edgemlp" in which:
        edgemlp_cases()
    if "dggclass" in which:
        dggclass_cases()
    if "cora" in which:
        cora_cases()
    if "gat" in which:
        gat_cases()
    if "ablations" in which:
        ablations_cases()
    if "dense" in which:
        dense_cases()
    if "scores" in which:
        scores_cases()
    if "edgelist512" in which:
        edgelist512_cases()
    if "hard" in which:
        hard_cases()
    if "cora_models" in which:
        cora_model_cases()


def model_cases():
    """End-to-end eval-mode forwards/backwards of the reference's *_DGG wrappers (model.py:1183-1311, 649-740,
    887-965) on a small synthetic graph with explicit (captured) noise."""
    N, d, h, C = 128, 20, 16, 5
    gen = torch.Generator().manual_seed(10)
    A = random_graph(N, 10, gen)                       # includes self loops; the wrappers add them again (model.py:1249)
    A = (A.to_dense() - torch.eye(N)).to_sparse().coalesce()
    x = torch.rand(N, d, generator=gen)
    # Pick a noise seed whose reference scores have no near-tie (relative gap < 4e-6) among the ranks that carry ramp
    # weight: a near-tie is ordered by last-bit rounding, which legitimately differs between the reference's torch
    # kernels and the canonical arithmetic, and at model level a swapped pair moves outputs by ~1e-3.
    def min_gap(seed):
        Gs = torch.from_numpy(grid_gumbel(seed, (N, N)))
        a = base_args()
        torch.manual_seed(4321)
        m = refmodel.GCN_DGG(nfeat=d, nlayers=2, nhidden=h, nclass=C, args=a)
        m.eval()
        cap = {}
        dg = m.dggs[0]
        dg.gumbel.sample = lambda shape: Gs.reshape(shape)
        _sel = dg.select_top_k
        dg.select_top_k = lambda Nn, k, pert, **kw: (cap.update(p=pert.detach().squeeze(0)), _sel(Nn, k, pert, **kw))[1]
        m(x, A)
        s = torch.sort(cap["p"], dim=-1, descending=True).values[:, :26]
        return float(((s[:, :-1] - s[:, 1:]) / s[:, :-1]).min())
    seed = next(sd for sd in range(51, 200) if min_gap(sd) > 4e-6)
    G = grid_gumbel(seed, (N, N))
    cot = torch.from_numpy(grid_normal(52, (N, C)))
    for name, ctor in [
        # the small-graph script's configuration that runs as shipped: --dgg_mode_edge_net u-v-deg --extra_edge_dim 2,
        # perturb_edge_prob False (train_small_graphs.py:92-96, 157-163, 184-191; SURVEY 8b')
        ("model_gcn_dgg_uvdeg", lambda a: refmodel.GCN_DGG(nfeat=d, nlayers=2, nhidden=h, nclass=C, args=a)),
        ("model_gcn_dgg", lambda a: refmodel.GCN_DGG(nfeat=d, nlayers=2, nhidden=h, nclass=C, args=a)),
        ("model_gcnii_dgg", lambda a: refmodel.GCNII_DGG(nfeat=d, nlayers=3, nhidden=h, nclass=C, dropout=0.5, lamda=0.5,
                                                         alpha=0.1, variant=False, args=a)),
        ("model_gcniippi_dgg", lambda a: refmodel.GCNIIppi_DGG(nfeat=d, nlayers=3, nhidden=h, nclass=C, dropout=0.5,
                                                               lamda=0.5, alpha=0.1, variant=True, args=a)),
    ]:
        a = base_args(dgg_mode_edge_net="u-v-deg", extra_edge_dim=2, perturb_edge_prob=False) if name.endswith("uvdeg") \
            else base_args()
        torch.manual_seed(4321)
        m = ctor(a)
        m.eval()
        Gt = torch.from_numpy(G)
        for dg in m.dggs:
            dg.gumbel.sample = lambda shape, Gt=Gt: Gt.reshape(shape)
        out = m(x, A)
        logp = out[0] if isinstance(out, tuple) else out
        (logp * cot).sum().backward()
        extra = {"unnorm": out[1].detach().to_dense().numpy()} if isinstance(out, tuple) else {}
        fx = {**extra, "x": x.numpy(), "rows": A.indices()[0].numpy().astype(np.int32), "cols": A.indices()[1].numpy().astype(np.int32),
              "adj_vals": A.values().numpy(), "G": G, "cot": cot.numpy(), "out": logp.detach().numpy()}
        for k_, v in m.state_dict().items():
            fx["p." + k_] = v.detach().numpy()
        for k_, p in m.named_parameters():
            fx["g." + k_] = p.grad.numpy() if p.grad is not None else np.zeros_like(p.detach().numpy())
        meta = dict(name=name, N=N, d=d, h=h, C=C, torch=torch.__version__, args=vars(a))
        fx["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **fx)
        print(name, "ok", tuple(logp.shape))


if __name__ == "__main__" and "models" in (sys.argv[1:] or ["models"]):
    model_cases()
