"""Graph-conv layers and *_DGG model wrappers with the reference's names, signatures and state_dict keys
(reference model.py), running on the ELL adjacency and the HIP kernels.

Layers accept either an `EllAdjacency` (fast path) or a dense [N,N] tensor whose rows have at most 64 non-zeros
(converted once).  Dropout / log_softmax / sigmoid and the two `fcs` nn.Linear layers of the GCNII wrappers are
plain torch, as in the reference: they are outside the hot path (SURVEY.md section 8b, row "Model wrappers").
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.parameter import Parameter

from . import ops
from .adjacency import AllPairs, CsrAdjacency, EllAdjacency, _cached, ell_from_dense
from .dgm import DGG, DGG_Ablations, DGG_LearnableK_debug


def _as_ell(adj):
    if isinstance(adj, (EllAdjacency, CsrAdjacency)):
        return adj
    if adj.is_sparse:
        adj = adj.to_dense()
    return ell_from_dense(adj)


class GCNConv(nn.Module):
    """relu((A x) W), W ~ U[0,1)  (reference model.py:580-599).

    On an ELL adjacency with out_channels <= in_channels the layer is evaluated as relu(A (x W)): the same value up to fp32
    reassociation (goldens: 1e-5), but the aggregation and its backward gather out_channels-wide rows instead of
    in_channels-wide ones (128 -> 64 halves the bytes of the two largest kernels of the step; Cora's 1433 -> 64: 22x)."""

    def __init__(self, in_channels, out_channels, A=None, cached=False):
        super().__init__()
        self.W = nn.Parameter(torch.rand(in_channels, out_channels, requires_grad=True))

    def forward(self, x, adj):
        adj = _as_ell(adj)
        fin, fout = self.W.shape
        if isinstance(adj, EllAdjacency) and fout <= fin and fout in ops.CONV_BWD_WIDTHS:
            return adj.matmul(ops.LinearFn.apply(x, self.W, None, ops.ACT_NONE, 1), ops.ACT_RELU)
        Ax = adj.matmul(x)
        return ops.LinearFn.apply(Ax, self.W, None, ops.ACT_RELU, 1)


class GraphConvolution(nn.Module):
    """GCNII layer (reference model.py:14-44): theta = log(lamda/l + 1); hi = A input;
    support = (1-alpha) hi + alpha h0 (variant: cat[hi, h0]); out = theta support W + (1-theta) r (+ input)."""

    def __init__(self, in_features, out_features, residual=False, variant=False):
        super().__init__()
        self.variant = variant
        self.in_features = 2 * in_features if variant else in_features
        self.out_features = out_features
        self.residual = residual
        self.weight = Parameter(torch.FloatTensor(self.in_features, self.out_features))
        self.gemm_dtype = None                     # None: fp32 MFMA kernel; torch.bfloat16: hand-written bf16 MFMA kernel (dgg_bf16.hip)
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1.0 / math.sqrt(self.out_features)
        self.weight.data.uniform_(-stdv, stdv)

    def forward(self, input, adj, h0, lamda, alpha, l):
        theta = math.log(lamda / l + 1)
        hi = _as_ell(adj).matmul(input)
        res = input if self.residual else None
        if (self.variant and self.gemm_dtype == torch.bfloat16 and hi.shape[1] % 64 == 0 and self.out_features == hi.shape[1]
                and h0.shape == hi.shape):
            # reduced-precision variant layer: cat[hi, h0] is never formed (the product takes the two halves of its A operand from
            # two matrices, h0 is packed once per stack) and the backward leaves d hi / d h0 complete from one product
            return ops.GcniiVariantBf16Fn.apply(hi, h0, self.weight, res, theta, alpha)
        # variant: the GEMM input is cat[hi, h0] and r only feeds the epilogue (folded into it); otherwise support = r
        support = torch.cat([hi, h0], 1) if self.variant else (1 - alpha) * hi + alpha * h0
        if self.gemm_dtype == torch.bfloat16 and support.shape[1] % 64 == 0:
            # reduced-precision variant (BASELINE configs[4]: "bf16 fwd+bwd"): the layer product and its autograd run on the bf16
            # matrix cores (dgg_bf16.hip: v_mfma_f32_32x32x16_bf16, fp32 accumulation), epilogue fused; everything else fp32
            # (other dtypes / contraction lengths that are not a multiple of 64: the fp32 matrix-core path below)
            if self.variant:
                return ops.GcniiBf16Fn.apply(support, self.weight, hi, h0, res, theta, alpha)
            return ops.GcniiBf16Fn.apply(support, self.weight, support, None, res, theta, alpha)
        sw = ops.LinearFn.apply(support, self.weight, None, ops.ACT_NONE, 1)              # fp32 matrix cores (HIP)
        if self.variant:
            return ops.GcniiEpilogueFn.apply(sw, hi, h0, res, theta, alpha)
        return ops.GcniiEpilogueFn.apply(sw, support, None, res, theta, alpha)


class DenseGraphConvolution(GraphConvolution):
    """Same arithmetic as GraphConvolution (reference model.py:47-77 differs only in torch.mm vs torch.spmm)."""


def _normalize_adj(A_hat):
    """D^-1/2 A D^-1/2 with row sums on both sides (reference model.py:1205-1219 and its nine copies)."""
    if isinstance(A_hat, (EllAdjacency, CsrAdjacency)):
        return A_hat.normalize()
    return _as_ell(A_hat).normalize()


def _with_self_loops(in_adj):
    """in_adj + I as coalesced sparse COO (reference model.py:1249-1251, 1264) without the dense round trip."""
    if not isinstance(in_adj, torch.Tensor):
        return in_adj          # AllPairs etc.

    def make():
        a = in_adj if in_adj.is_sparse else in_adj.to_sparse()
        N = a.shape[0]
        eye_i = torch.arange(N, device=a.device)
        eye = torch.sparse_coo_tensor(torch.stack([eye_i, eye_i]), torch.ones(N, device=a.device), (N, N))
        return (a + eye).coalesce()
    # the input graph is data: the same tensor object on every forward of a training loop (train_small_graphs.py:226).  The sum is
    # cached per object and version of its values (adjacency._cached), so the CSR conversions cached on ITS result hit as well and a
    # forward on a known graph launches nothing here (which is also what lets a whole model step be captured into a hipGraph)
    if in_adj.requires_grad or (in_adj.is_sparse and in_adj._values().requires_grad):
        return make()
    return _cached("selfloops", in_adj, make)


class GCN_DGG(nn.Module):
    """Two GCNConv layers, one DGG in front of the first (reference model.py:1183-1311).  Returns
    (log_probs, unnorm_adj, None).

    `unnorm_adj` on the fused path (the default on the GPU: generator + normalize_adj + first layer as one autograd node) is
    DETACHED -- the reference returns the differentiable output of dgg_net (model.py:1290), but its scripts only log it
    (train_small_graphs.py:226-230, 268-270).  A loss term on the returned adjacency (a sparsity or degree regulariser) needs
    args.dgg_differentiable_adj = True (or args.dgg_fused_layer = False): the separate modules, whose unnorm_adj carries autograd
    into the generator."""

    def __init__(self, nfeat=32, nlayers=None, nhidden=32, nclass=10, args=None, **kwargs):
        super().__init__()
        self.convs = nn.ModuleList()
        self.conv1 = GCNConv(nfeat, nhidden)
        self.conv2 = GCNConv(nhidden, nclass)
        self.convs.append(self.conv1)
        self.convs.append(self.conv2)
        self.dgg_adj_input = args.dgg_adj_input
        self.differentiable_adj = bool(getattr(args, "dgg_differentiable_adj", False))
        self.dggs = nn.ModuleList([DGG_LearnableK_debug(in_dim=nfeat, latent_dim=nhidden, args=args)])
        self.params1 = list(self.conv1.parameters())
        self.params2 = list(self.conv2.parameters())
        self.params2.extend(list(self.dggs.parameters()))

    normalize_adj = staticmethod(_normalize_adj)

    def forward(self, x, in_adj, noise=True, epoch=None, writer=None):
        in_adj = _with_self_loops(in_adj)
        unnorm_adj = in_adj
        norm_adj = None
        for i, conv in enumerate(self.convs):
            fused = None
            if i < len(self.dggs):
                src = in_adj if self.dgg_adj_input == "input_adj" else unnorm_adj
                if writer is None and isinstance(conv, GCNConv) and isinstance(src, (torch.Tensor, AllPairs)) and x.is_cuda and not self.differentiable_adj:
                    # generator + normalize_adj + this layer as one autograd node (one pass over x for the three projections and
                    # one for their weight gradients); None when the configuration is outside its coverage
                    fused = self.dggs[i].forward_conv(x, src, conv.W, want_norm=True)
                if fused is not None:
                    x, unnorm_adj, norm_adj = fused
                else:
                    unnorm_adj = self.dgg_net(x, i, src, writer, epoch)
                    norm_adj = _normalize_adj(unnorm_adj)
            if fused is None:
                x = conv(x, norm_adj)
            if i < len(self.convs) - 1:
                x = F.dropout(x, training=self.training)
            if writer is not None:
                writer.add_histogram("gcn_conv{}_dist".format(i + 1), x, epoch)
        return F.log_softmax(x, dim=-1), unnorm_adj, None

    def dgg_net(self, x, i, unnorm_adj, writer, epoch):
        return self.dggs[i](x=x, in_adj=unnorm_adj, noise=False, writer=writer, epoch=epoch)


class DenseGraphConv(nn.Module):
    """`torch_geometric.nn.DenseGraphConv` as used by SAGE_DGG (reference model.py:84-85, 128-129; README pins
    torch_geometric 2.1.0, which is not vendored and not installed here: PARITY UNPINNED -- this is a restatement of the
    published layer, checked only against a dense torch evaluation of the same formula):
        out = lin_rel(aggr_j adj_ij x_j) + lin_root(x),  aggr "add": adj @ x;  "mean": (adj @ x) / clamp(sum_j adj_ij, min=1)
    with lin_rel = Linear(in, out, bias), lin_root = Linear(in, out, bias=False) (state_dict keys lin_rel.*, lin_root.weight).
    The aggregation runs on the sparse adjacency (ELL / CSR SpMM), the two projections on the MFMA linear kernel."""

    def __init__(self, in_channels, out_channels, aggr="add", bias=True):
        super().__init__()
        assert aggr in ("add", "mean"), "aggr 'max' needs a dense masked maximum: not on the HIP path"
        self.in_channels, self.out_channels, self.aggr = in_channels, out_channels, aggr
        self.lin_rel = nn.Linear(in_channels, out_channels, bias=bias)
        self.lin_root = nn.Linear(in_channels, out_channels, bias=False)

    def forward(self, x, adj, mask=None):
        assert mask is None, "node masks belong to the batched dense layout, which this path does not use"
        adj = _as_ell(adj)
        out = adj.matmul(x)
        if self.aggr == "mean":
            v = adj.values()
            if isinstance(adj, CsrAdjacency):    # differentiable row sums of the CSR values
                rs = torch.zeros(adj.shape[0], device=v.device, dtype=v.dtype).index_add(0, adj.erow.long(), v)
            else:
                rs = v.sum(1)
            out = out / rs.clamp(min=1).unsqueeze(1)
        return ops.LinearFn.apply(out, self.lin_rel.weight, self.lin_rel.bias, ops.ACT_NONE, 0) + \
            ops.LinearFn.apply(x, self.lin_root.weight, None, ops.ACT_NONE, 0)


class SAGE_DGG(nn.Module):
    """Two DenseGraphConv(mean) layers, one DGG in front of the first (reference model.py:122-193).  Returns log_probs."""

    def __init__(self, nfeat=32, nlayers=None, nhidden=32, nclass=10, args=None, **kwargs):
        super().__init__()
        self.convs = nn.ModuleList([DenseGraphConv(nfeat, nhidden, aggr="mean"), DenseGraphConv(nhidden, nclass, aggr="mean")])
        self.dgg_adj_input = args.dgg_adj_input
        self.dggs = nn.ModuleList([DGG_LearnableK_debug(in_dim=nfeat, latent_dim=nhidden, args=args)])

    normalize_adj = staticmethod(_normalize_adj)

    def dgg_net(self, x, i, unnorm_adj, writer, epoch):
        return self.dggs[i](x=x, in_adj=unnorm_adj, noise=False, writer=writer, epoch=epoch)

    def forward(self, x, in_adj, noise=True, epoch=None, writer=None, **kwargs):
        in_adj = _with_self_loops(in_adj)
        unnorm_adj = in_adj
        norm_adj = None
        for i, conv in enumerate(self.convs):
            if i < len(self.dggs):
                src = in_adj if self.dgg_adj_input == "input_adj" else unnorm_adj
                unnorm_adj = self.dgg_net(x, i, src, writer, epoch)
                norm_adj = _normalize_adj(unnorm_adj)
            x = conv(x, norm_adj)
            if i < len(self.convs) - 1:
                x = F.dropout(torch.relu(x), p=0.5, training=self.training)
        return F.log_softmax(x, dim=-1)


class GATConv_DGG(nn.Module):
    """Attention layer of the GAT_DGG variants (reference model.py:534-577), evaluated WITHOUT the dense [N,N] matrices.
    The reference sets logit e_ij = leaky_relu(a^T [h_i, h_j]) on the entries of `edge_list`, -1e20 elsewhere, and multiplies
    by the dense learned adjacency (model.py:564-567).  Consequences that this implementation reproduces exactly:
      * listed edge that the learned adjacency also holds:  logit e_ij * A_ij
      * adjacency entry that is NOT in edge_list (e.g. a noisy edge): logit -1e20 * A_ij -> weight 0
      * every other pair: -1e20 * 0 = -0, i.e. ALL non-neighbours attend with the weight of logit 0
    so out_i = sum_{u in row i} att_u h_j + bg_i * (sum_all h - sum_{u in row i} h_j) with the row softmax taken over the
    explicit logits plus N - cnt_i zeros (dgg_csr_bg_softmax_fwd).
    Dropout (training): on the input and on h as in the reference; F.dropout(attention) (model.py:570) masks every one of the N x N
    pairs, the non-listed ones included -- here with ONE counter-based pair mask (ops.pair_keep / ops.MaskedDenseSumFn: same law as
    torch's generator, another realisation; no N x N tensor):
        out_i = q sum_{u in row i} m_u att_u h_j + q bg_i (sum_j m_ij h_j - sum_{u in row i} m_u h_j),   q = 1 / (1 - p).
    Output widths beyond 64 keep the background at its expectation (`exact_attention_dropout = False` forces that form)."""
    exact_attention_dropout = True

    def __init__(self, in_features, out_features, dropout, alpha, bias=True):
        super().__init__()
        self.dropout, self.in_features, self.out_features, self.alpha = dropout, in_features, out_features, alpha
        self.weight = nn.Parameter(torch.FloatTensor(in_features, out_features))
        self.a = nn.Parameter(torch.zeros(size=(2 * out_features, 1)))
        if bias:
            self.bias = nn.Parameter(torch.FloatTensor(out_features))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.xavier_uniform_(self.weight.data, gain=1.414)
        if self.bias is not None:
            self.bias.data.fill_(0)
        nn.init.xavier_uniform_(self.a.data, gain=1.414)

    @staticmethod
    def union_pattern(edge_list, adj):
        """CSR pattern of edge_list UNION the adjacency's pattern; per entry: is it listed, and where is its adjacency value"""
        N = adj.shape[0]
        k1 = torch.unique(edge_list[0].long() * N + edge_list[1].long())
        k2 = adj.erow.long() * N + adj.col.long()                # coalesced: sorted, unique
        ku = torch.unique(torch.cat([k1, k2]))
        p1 = torch.searchsorted(k1, ku).clamp(max=k1.numel() - 1)
        p2 = torch.searchsorted(k2, ku).clamp(max=k2.numel() - 1)
        listed = k1[p1] == ku
        apos = torch.where(k2[p2] == ku, p2, torch.full_like(p2, -1))
        rows, cols = ku // N, (ku % N).to(torch.int32)
        rowptr = torch._convert_indices_from_coo_to_csr(rows, N, out_int32=False)
        return rowptr, cols.contiguous(), rows, listed, apos

    def forward(self, x, edge_list, adj, pattern=None):
        N, Fo = x.shape[0], self.out_features
        x = F.dropout(x, self.dropout, training=self.training)
        h = ops.LinearFn.apply(x, self.weight, None, ops.ACT_NONE, 1)
        rowptr, cols, rows, listed, apos = pattern if pattern is not None else self.union_pattern(edge_list, adj)
        s = ops.LinearFn.apply(h, self.a.reshape(2, Fo), None, ops.ACT_NONE, 0)             # [N,2]: a1.h_i, a2.h_j
        e = F.leaky_relu(s[rows, 0] + s[cols.long(), 1], negative_slope=self.alpha)
        w = torch.where(apos >= 0, adj.values()[apos.clamp(min=0)], torch.zeros_like(e))
        L = torch.where(listed, e * w, -1e20 * w)
        att, bg = ops.CsrBgSoftmaxFn.apply(L, rowptr)
        if self.training and self.dropout > 0 and self.exact_attention_dropout and Fo <= 64:
            h = F.dropout(h, self.dropout, training=True)
            seed = tuple(int(v) for v in torch.randint(0, 2 ** 31 - 1, (2,)).tolist())      # (torch's CPU generator: no device sync)
            h_prime = self.attend_dropped(h, att, bg, rowptr, cols, rows, self.dropout, seed)
        else:
            att = F.dropout(att, self.dropout, training=self.training)
            h = F.dropout(h, self.dropout, training=self.training)
            h_prime = ops.CsrSpmmFn.apply(att - bg[rows], rowptr, cols, h) + bg.unsqueeze(1) * h.sum(0, keepdim=True)
        return h_prime + self.bias if self.bias is not None else h_prime

    @staticmethod
    def attend_dropped(h, att, bg, rowptr, cols, rows, p, seed):
        """dropout(attention) @ h with the dense attention = att on the listed pairs, bg_i everywhere else, and one pair mask over all N^2"""
        q = 1.0 / (1.0 - p)
        m = ops.pair_keep(rows.to(torch.int32), cols, p, seed)
        listed = ops.CsrSpmmFn.apply(q * m * (att - bg[rows]), rowptr, cols, h)
        return listed + (q * bg).unsqueeze(1) * ops.MaskedDenseSumFn.apply(h, p, seed)


class GAT_DGG_00(nn.Module):
    """8 attention heads + output head on the encoded features of a `DGG` generator (reference model.py:323-403).  Returns
    (log_probs, unnorm_adj, x_dgg).  `edge_index` [2,E] is the ORIGINAL edge list (self loops are replaced as by
    torch_geometric's remove_self_loops / add_self_loops, model.py:386-387); default: the stored entries of in_adj."""

    def __init__(self, nfeat=32, nlayers=None, nhidden=32, nclass=10, args=None, nhead=8, nhead_out=1, alpha=0.2, dropout=0.6,
                 **kwargs):
        super().__init__()
        self.attentions = [GATConv_DGG(nhidden, nhidden, dropout=dropout, alpha=alpha) for _ in range(nhead)]
        self.out_atts = [GATConv_DGG(nhidden * nhead, nclass, dropout=dropout, alpha=alpha) for _ in range(nhead_out)]
        self.dgg = DGG(in_dim=nfeat, latent_dim=nhidden, args=args)
        for i, attention in enumerate(self.attentions):
            self.add_module("attention_{}".format(i), attention)
        for i, attention in enumerate(self.out_atts):
            self.add_module("out_att{}".format(i), attention)

    normalize_adj = staticmethod(_normalize_adj)

    def forward(self, x, in_adj=None, edge_index=None, epoch=None, writer=None):
        N = x.size(0)
        if edge_index is None:
            edge_index = in_adj.coalesce().indices()
        edge_index = edge_index[:, edge_index[0] != edge_index[1]]
        loops = torch.arange(N, device=edge_index.device, dtype=edge_index.dtype)
        edge_index = torch.cat([edge_index, torch.stack([loops, loops])], 1)
        in_adj = _with_self_loops(in_adj)
        unnorm_adj, x_dgg = self.dgg(x=x, adj=in_adj)
        pattern = GATConv_DGG.union_pattern(edge_index, unnorm_adj)          # shared by every head
        x = torch.cat([att(x_dgg, edge_index, unnorm_adj, pattern) for att in self.attentions], dim=1)
        x = F.elu(x)
        x = torch.sum(torch.stack([att(x, edge_index, unnorm_adj, pattern) for att in self.out_atts]), dim=0) / len(self.out_atts)
        return F.log_softmax(x, dim=1), unnorm_adj, x_dgg


class SAGE_DGG_00(nn.Module):
    """Two DenseGraphConv(mean) layers on the encoded features of a `DGG` generator (reference model.py:196-283; the PyG
    layer is restated, see DenseGraphConv).  Returns (log_probs, unnorm_adj, x_dgg)."""

    def __init__(self, nfeat=32, nlayers=None, nhidden=32, nclass=10, args=None, **kwargs):
        super().__init__()
        self.convs = nn.ModuleList([DenseGraphConv(nhidden, nhidden, aggr="mean"), DenseGraphConv(nhidden, nclass, aggr="mean")])
        self.dgg_adj_input = args.dgg_adj_input
        self.dggs = nn.ModuleList([DGG(in_dim=nfeat, latent_dim=nhidden, args=args)])

    normalize_adj = staticmethod(_normalize_adj)

    def dgg_net(self, x, i, unnorm_adj, writer, epoch):
        return self.dggs[i](x=x, adj=unnorm_adj, noise=False, writer=writer, epoch=epoch)

    def forward(self, x, in_adj, noise=True, epoch=None, writer=None, **kwargs):
        in_adj = _with_self_loops(in_adj)
        unnorm_adj = in_adj
        norm_adj = x_dgg = None
        for i, conv in enumerate(self.convs):
            if i < len(self.dggs):
                src = in_adj if self.dgg_adj_input == "input_adj" else unnorm_adj
                unnorm_adj, x_dgg = self.dgg_net(x, i, src, writer, epoch)
                norm_adj = _normalize_adj(unnorm_adj)
                x = x_dgg
            x = conv(x, norm_adj)
            if i < len(self.convs) - 1:
                x = F.dropout(torch.relu(x), p=0.5, training=self.training)
        return F.log_softmax(x, dim=-1), unnorm_adj, x_dgg


class GCN_DGG_00(nn.Module):
    """Two GCNConv layers on the encoded features of a `DGG` generator (reference model.py:1314-1433).  Returns
    (log_probs, unnorm_adj, x_dgg)."""

    def __init__(self, nfeat=32, nlayers=None, nhidden=32, nclass=10, args=None, **kwargs):
        super().__init__()
        self.convs = nn.ModuleList()
        self.conv1 = GCNConv(nhidden, nhidden)
        self.conv2 = GCNConv(nhidden, nclass)
        self.convs.append(self.conv1)
        self.convs.append(self.conv2)
        self.dgg_adj_input = args.dgg_adj_input
        self.dggs = nn.ModuleList([DGG(in_dim=nfeat, latent_dim=nhidden, args=args)])
        self.params1 = list(self.conv1.parameters())
        self.params2 = list(self.conv2.parameters())
        self.params2.extend(list(self.dggs.parameters()))

    normalize_adj = staticmethod(_normalize_adj)

    def forward(self, x, in_adj, noise=True, epoch=None, writer=None, **kwargs):
        in_adj = _with_self_loops(in_adj)
        unnorm_adj = in_adj
        norm_adj = x_dgg = None
        for i, conv in enumerate(self.convs):
            if i < len(self.dggs):
                src = in_adj if self.dgg_adj_input == "input_adj" else unnorm_adj
                unnorm_adj, x_dgg = self.dgg_net(x, i, src, writer, epoch)
                norm_adj = _normalize_adj(unnorm_adj)
                x = x_dgg
            x = conv(x + x_dgg, norm_adj)
            if i < len(self.convs) - 1:
                x = F.dropout(x, training=self.training)
            if writer is not None:
                writer.add_histogram("gcn_conv{}_dist".format(i + 1), x, epoch)
        return F.log_softmax(x, dim=-1), unnorm_adj, x_dgg

    def dgg_net(self, x, i, unnorm_adj, writer, epoch):
        return self.dggs[i](x=x, adj=unnorm_adj, noise=False, writer=writer, epoch=epoch)


class GCN_DGG_Ablations(GCN_DGG_00):
    """GCN_DGG_00 on a `DGG_Ablations` generator (reference model.py:1436-1561): same two-layer body, the generator is called
    with k=None (learned degree), fresh U(-1,1) rank noise on every call."""

    def __init__(self, nfeat=32, nlayers=None, nhidden=32, nclass=10, args=None, **kwargs):
        super().__init__(nfeat, nlayers, nhidden, nclass, args, **kwargs)
        self.dggs = nn.ModuleList([DGG_Ablations(in_dim=nfeat, latent_dim=nhidden, args=args)])
        self.params2 = list(self.conv2.parameters())
        self.params2.extend(list(self.dggs.parameters()))

    def dgg_net(self, x, i, unnorm_adj, writer, epoch):
        return self.dggs[i](x=x, adj=unnorm_adj, k=None, writer=writer, epoch=epoch)


class GAT_DGG_Ablations(GAT_DGG_00):
    """GAT_DGG_00 on a `DGG_Ablations` generator (reference model.py:406-486)."""

    def __init__(self, nfeat=32, nlayers=None, nhidden=32, nclass=10, args=None, nhead=8, nhead_out=1, alpha=0.2, dropout=0.6,
                 **kwargs):
        super().__init__(nfeat, nlayers, nhidden, nclass, args, nhead, nhead_out, alpha, dropout, **kwargs)
        self.dgg = DGG_Ablations(in_dim=nfeat, latent_dim=nhidden, args=args)


class GCNII_DGG(nn.Module):
    """GCNII stack with DGG-generated adjacency for the first n_dgg_layers layers (reference model.py:649-740).
    The DGG sees the raw (dropped-out) input features, not the hidden state (model.py:720)."""

    _conv_cls = DenseGraphConvolution
    _residual = False

    def __init__(self, nfeat, nlayers, nhidden, nclass, dropout, lamda, alpha, variant, args):
        super().__init__()
        self.convs = nn.ModuleList()
        for _ in range(nlayers):
            if self._residual:
                self.convs.append(self._conv_cls(nhidden, nhidden, variant=variant, residual=True))
            else:
                self.convs.append(self._conv_cls(nhidden, nhidden, variant=variant))
        self.fcs = nn.ModuleList()
        self.fcs.append(nn.Linear(nfeat, nhidden))
        self.fcs.append(nn.Linear(nhidden, nclass))
        self.dgg_adj_input = args.dgg_adj_input
        self.dggs = nn.ModuleList()
        for _ in range(args.n_dgg_layers):
            self.dggs.append(DGG_LearnableK_debug(in_dim=nfeat, latent_dim=nhidden, args=args))
        self.params1 = list(self.convs.parameters())
        self.params1.extend(list(self.dggs.parameters()))
        self.params2 = list(self.fcs.parameters())
        self.act_fn = nn.ReLU()
        self.dropout = dropout
        self.alpha = alpha
        self.lamda = lamda

    normalize_adj = staticmethod(_normalize_adj)

    def _body(self, x, in_adj, epoch, writer):
        _layers = []
        x = F.dropout(x, self.dropout, training=self.training)
        layer_inner = self.act_fn(self.fcs[0](x))
        _layers.append(layer_inner)
        in_adj = _with_self_loops(in_adj)
        unnorm_adj = in_adj
        norm_adj = None
        if self._stack_ok(layer_inner):
            # ONE adjacency for every layer (a single generator in front of the first one), variant layers on the bf16 matrix cores:
            # the whole loop below -- dropout, aggregation, layer, ReLU, and the dropout in front of the output layer -- as one
            # autograd node (ops.GcniiStackBf16Fn)
            unnorm_adj = self.dgg_net(x, 0, in_adj, writer, epoch)
            norm_adj = _normalize_adj(unnorm_adj)
            if isinstance(norm_adj, EllAdjacency):
                seed = torch.randint(0, 2 ** 31 - 1, (2,))        # (CPU generator: reproducible under torch.manual_seed, no sync)
                p_ = float(self.dropout) if self.training else 0.0
                y = ops.GcniiStackBf16Fn.apply(layer_inner, norm_adj.values(), norm_adj.idx, norm_adj.part, norm_adj.k is not None,
                                               self._residual, p_, float(self.lamda), float(self.alpha), (int(seed[0]), int(seed[1])),
                                               ops.backward_will_follow(layer_inner, norm_adj.values(), *[con.weight for con in self.convs]),
                                               *[con.weight for con in self.convs])
                return self.fcs[-1](y), unnorm_adj
            # (rows wider than the list: a CSR adjacency -- the layers one by one, below)
            for i, con in enumerate(self.convs):
                layer_inner = F.dropout(layer_inner, self.dropout, training=self.training)
                layer_inner = self.act_fn(con(layer_inner, norm_adj, _layers[0], self.lamda, self.alpha, i + 1))
            layer_inner = F.dropout(layer_inner, self.dropout, training=self.training)
            return self.fcs[-1](layer_inner), unnorm_adj
        for i, con in enumerate(self.convs):
            if i < len(self.dggs):
                src = in_adj if self.dgg_adj_input == "input_adj" else unnorm_adj
                unnorm_adj = self.dgg_net(x, i, src, writer, epoch)
                norm_adj = _normalize_adj(unnorm_adj)
            elif norm_adj is None:
                norm_adj = _normalize_adj(unnorm_adj)
            layer_inner = F.dropout(layer_inner, self.dropout, training=self.training)
            layer_inner = self.act_fn(con(layer_inner, norm_adj, _layers[0], self.lamda, self.alpha, i + 1))
        layer_inner = F.dropout(layer_inner, self.dropout, training=self.training)
        return self.fcs[-1](layer_inner), unnorm_adj

    def _stack_ok(self, h0):
        """the fused stack covers: one generator, every layer a variant GCNII layer of the same square width (a multiple of 256) with
        the bf16 product switched on, activations on the GPU; `self.fused_stack = False` keeps the layers one by one"""
        if not getattr(self, "fused_stack", True) or len(self.dggs) != 1 or not h0.is_cuda or h0.dtype != torch.float32 or len(self.convs) == 0:
            return False
        Fw = h0.shape[1]
        return Fw % 256 == 0 and all(isinstance(c, GraphConvolution) and c.variant and c.gemm_dtype == torch.bfloat16 and
                                      tuple(c.weight.shape) == (2 * Fw, Fw) and c.residual == self._residual for c in self.convs) and \
            isinstance(self.act_fn, nn.ReLU)

    def forward(self, x, in_adj, epoch=None, writer=None):
        out, _ = self._body(x, in_adj, epoch, writer)
        return F.log_softmax(out, dim=1)

    def dgg_net(self, x, i, unnorm_adj, writer, epoch):
        return self.dggs[i](x=x, in_adj=unnorm_adj, noise=self.training, writer=writer, epoch=epoch)


class GCNIIppi_DGG(GCNII_DGG):
    """PPI variant (reference model.py:887-965): GraphConvolution(residual=True) layers, sigmoid output."""

    _conv_cls = GraphConvolution
    _residual = True

    def forward(self, x, in_adj, writer=None, epoch=None):
        out, _ = self._body(x, in_adj, epoch, writer)
        return torch.sigmoid(out)
