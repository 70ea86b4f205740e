"""Harness counterpart of the reference's small-graph script (train_small_graphs.py; SURVEY.md 8b'): same flags for the
parts that reach the DGG path, same seeding, features row-normalised, adjacency from the edge list -> add_noisy_edges ->
coalesced COO fp32, model looked up by class name, nll_loss on the public split.

    python -m dgg_amd.train_small_graphs --data cora --data_dir /path/to/planetoid --model GCN_DGG_00 --epochs 5

Flag DEFAULTS are the reference script's (train_small_graphs.py:20-207), with two exceptions that the reference cannot run
as shipped: `--extra_edge_dim` defaults to 2 (the default scorer `u-v-deg` concatenates two degree
features, dgm.py:1660-1662; the reference's default 0 shape-mismatches) and the script's `edge_index=` keyword is dropped
for wrappers that reject it (reference GCN_DGG.forward raises TypeError on it).  Optimiser groups follow
train_small_graphs.py:399-418 (GCNII / GCN / SAGE / GAT branches).
Only the flags the model constructors / DGG read are kept (no tensorboard, no code snapshots).  `--checkpoint FILE` saves
the best-validation state in the reference's `save_checkpoint` layout (train_small_graphs.py:210-220: args, epoch,
model_state_dict, optimizer_state_dict); `--resume FILE` loads `model_state_dict` as its `test_best` does (line 330) -- the
modules keep the reference's state_dict keys, so files written by either side load in the other.
"""
import argparse
import random
import time

import numpy as np
import torch
import torch.nn.functional as F

from . import data as D
from . import ops
from . import model as models


def str2bool(v):
    return str(v).lower() in ("yes", "true", "t", "1")


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--epochs", type=int, default=1500)
    p.add_argument("--lr", type=float, default=0.01)
    p.add_argument("--wd1", type=float, default=0.01)
    p.add_argument("--wd2", type=float, default=5e-4)
    p.add_argument("--layer", type=int, default=2)
    p.add_argument("--hidden", type=int, default=64)
    p.add_argument("--dropout", type=float, default=0.6)
    p.add_argument("--patience", type=int, default=100)
    p.add_argument("--data", default="cora")
    p.add_argument("--data_dir", required=True, help="directory holding the ind.<name>.* Planetoid files")
    p.add_argument("--alpha", type=float, default=0.1)
    p.add_argument("--lamda", type=float, default=0.5)
    p.add_argument("--variant", type=str2bool, default=False)
    p.add_argument("--model", default="GCN_DGG")                      # reference default (train_small_graphs.py:69-73)
    p.add_argument("--edge_noise_level", type=float, default=0.00014)
    # DGG flags (reference train_small_graphs.py:92-207)
    p.add_argument("--extra_edge_dim", type=int, default=2)          # reference: 0, which its default scorer cannot run with
    p.add_argument("--extra_k_dim", type=int, default=1)
    p.add_argument("--dgg_hard", type=str2bool, default=False)
    p.add_argument("--deg_mean", type=float, default=3.899)
    p.add_argument("--deg_std", type=float, default=5.288)
    p.add_argument("--n_dgg_layers", type=int, default=2)            # reference default (only GCNII-type wrappers read it)
    p.add_argument("--debug_step", type=int, default=3)
    p.add_argument("--symmetric_noise", type=str2bool, default=True)  # reference default (train_small_graphs.py:153-156)
    p.add_argument("--perturb_edge_prob", type=str2bool, default=False)
    p.add_argument("--stochastic_k", type=str2bool, default=False)
    p.add_argument("--dgg_adj_input", default="input_adj")
    p.add_argument("--dgg_mode_edge_net", default="u-v-deg",
                   choices=["u-v-dist", "u-v-A_uv", "u-v-deg", "edge_conv", "A_uv", "u-v-deg-dist"])
    p.add_argument("--dgg_mode_k_net", default="x", choices=["pass", "learn_normalized_degree", "input_deg", "gcn-x-deg", "x"])
    p.add_argument("--dgg_mode_k_select", default="k_times_edge_prob", choices=["edge_p-cdf", "k_only", "k_times_edge_prob"])
    p.add_argument("--cache_dir", default=None, help="keep the parsed data set here (npz) and reuse it while the source files are unchanged")
    p.add_argument("--checkpoint", default=None, help="write the best-validation checkpoint here (reference save_checkpoint layout)")
    p.add_argument("--resume", default=None, help="load model_state_dict from a checkpoint before training")
    return p


def save_checkpoint(fn, args, epoch, model, optimizer, lr_scheduler=None):
    """reference train_small_graphs.py:210-220"""
    torch.save({"args": args.__dict__, "epoch": epoch, "model_state_dict": model.state_dict(),
                "optimizer_state_dict": optimizer.state_dict()}, fn)


def make_adjacency(d, noise_level, device):
    """edge list -> (+ noisy edges, utils.py:92-110) -> coalesced sparse COO fp32 (train_small_graphs.py:251-255)"""
    N = d["x"].shape[0]
    rows, cols = d["rows"], d["cols"]
    vals = np.ones(rows.shape[0], np.float32)
    if noise_level > 0.0:
        rows, cols, vals = D.add_noisy_edges(rows, cols, N, noise_level)
    ind = torch.from_numpy(np.stack([rows, cols]).astype(np.int64))
    return torch.sparse_coo_tensor(ind, torch.from_numpy(vals), (N, N)).coalesce().to(device)


def accuracy(out, y):
    return float((out.argmax(1) == y).float().mean())


def forward(model, x, adj, **kw):
    """the wrappers differ in what their forward accepts and returns (reference model.py:1236 vs 1368): the script's call
    (features, adj, edge_index=..., epoch=..., writer=...) is tried first, then without the keywords a wrapper rejects"""
    for drop in ((), ("edge_index",), ("edge_index", "epoch", "writer")):
        try:
            out = model(x, adj, **{k: v for k, v in kw.items() if k not in drop})
            break
        except TypeError as e:
            if "unexpected keyword" not in str(e):
                raise
    return out[0] if isinstance(out, tuple) else out


def main(argv=None):
    args = build_parser().parse_args(argv)
    random.seed(args.seed)
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    torch.cuda.manual_seed(args.seed)
    device = torch.device("cuda")
    d = D.load_planetoid(args.data, args.data_dir, cache_dir=args.cache_dir)
    x = torch.from_numpy(d["x"]).to(device)
    y = torch.from_numpy(d["y"]).to(device)
    idx = {k: torch.from_numpy(d[k]).to(device) for k in ("train_idx", "val_idx", "test_idx")}
    adj = make_adjacency(d, args.edge_noise_level, device)
    edge_index = torch.from_numpy(np.stack([d["rows"], d["cols"]]).astype(np.int64)).to(device)      # data.edge_index
    model = models.__dict__[args.model](nfeat=x.shape[1], nlayers=args.layer, nhidden=args.hidden, nclass=d["num_classes"],
                                        dropout=args.dropout, lamda=args.lamda, alpha=args.alpha, variant=args.variant,
                                        args=args).to(device)
    # optimiser groups by model-name substring, as reference train_small_graphs.py:399-418
    if "GCN" in args.model and "II" in args.model:
        opt = torch.optim.Adam([{"params": model.params1, "weight_decay": args.wd1},
                                {"params": model.params2, "weight_decay": args.wd2}], lr=args.lr)
    elif "GCN" in args.model:
        opt = torch.optim.Adam([dict(params=model.params1, weight_decay=5e-4), dict(params=model.params2, weight_decay=0)],
                               lr=args.lr)                      # weight decay on the first convolution only
    elif "SAGE" in args.model:
        opt = torch.optim.Adam(model.parameters(), lr=args.lr)
    elif "GAT" in args.model:
        opt = torch.optim.Adam(model.parameters(), lr=0.005, weight_decay=5e-4)
    else:
        opt = torch.optim.Adam(model.parameters(), lr=args.lr)
    if args.resume:
        model.load_state_dict(torch.load(args.resume, map_location=device)["model_state_dict"])
    best, bad, t0 = float("inf"), 0, time.time()
    for epoch in range(args.epochs):
        model.train()
        opt.zero_grad()
        with ops.step_zero_pool(device, x.shape[0], 64, max(args.hidden, 64), list(model.parameters())):   # one fill for the step's accumulators
            out = forward(model, x, adj, edge_index=edge_index, epoch=epoch, writer=None)
            loss = F.nll_loss(out[idx["train_idx"]], y[idx["train_idx"]])
            loss.backward()
        # BEFORE the optimiser step: the learned k must have stayed inside the ELL width and the noise generator must have settled
        # every row -- otherwise this step's adjacency was wrong and its gradients must not reach the weights (one sync per epoch)
        for dg in getattr(model, "dggs", []):
            if hasattr(dg, "check_ell_bound"):
                dg.check_ell_bound()
        opt.step()
        model.eval()
        with torch.no_grad():
            out = forward(model, x, adj, edge_index=edge_index)
            lv = float(F.nll_loss(out[idx["val_idx"]], y[idx["val_idx"]]))
            print("Epoch:{:04d}".format(epoch + 1), "train", "loss:{:.3f}".format(float(loss)),
                  "| val loss:{:.3f} acc:{:.2f}".format(lv, 100 * accuracy(out[idx["val_idx"]], y[idx["val_idx"]])),
                  "| test acc:{:.2f}".format(100 * accuracy(out[idx["test_idx"]], y[idx["test_idx"]])))
        if lv < best:
            best, bad = lv, 0
            if args.checkpoint:
                save_checkpoint(args.checkpoint, args, epoch, model, opt)
        else:
            bad += 1
        if bad == args.patience:
            break
    print("Train cost: {:.4f}s".format(time.time() - t0))
    return best


if __name__ == "__main__":
    main()
