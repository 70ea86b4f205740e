"""Tensor-level wrappers of the C-ABI kernels and the custom autograd ops built on them.

torch is used for device memory, streams and autograd bookkeeping only: every FLOP of the hot path runs in
libdgg_hip.so.  Naming follows the reference (dgm.py / model.py) and include/dgg_hip.h.
"""
import collections
import math
import os
import ctypes as C

import torch

from . import _lib

NOISE_NONE, NOISE_EXPLICIT, NOISE_HASH, NOISE_HASH_SYM, NOISE_RANKED, NOISE_RANKED_SYM = 0, 1, 2, 3, 4, 5
RSYM_WIDTHS = (8, 16, 32, 64, 128)     # latent widths of the ranked symmetric generator's kernels
ACT_NONE, ACT_LEAKY, ACT_RELU = 0, 1, 2
MODE_K_TIMES_EDGE_PROB, MODE_K_ONLY = 0, 1
MODE_HARD_ST = 3          # softk_fwd only: value (ramp - score*ramp) + score*ramp, gradient of MODE_K_TIMES_EDGE_PROB
DEFAULT_K = 64
T_DIST = -0.05  # reference dgm.py:1618

# Measurement hook (bench.py): when PROBE is a dict, the wrappers of the four gather kernels record a pair of events on the
# launch stream around their C-ABI call, so that per-kernel durations are taken INSIDE a running step (same cache state as
# the timed region) rather than from stand-alone launches.  None: no events, no overhead.
PROBE = None



# ---- pooled zero-initialised buffers ------------------------------------------------------------------------------------
# The backward needs ~10 zeroed accumulators (weight gradients, da, dxp, ...).  Inside `with zero_pool(device, nfloats):`
# they are carved out of ONE buffer zeroed by ONE fill kernel instead of one fill each (9 fills, ~45 us per step at N = 100k).
_POOL = None


class zero_pool:
    def __init__(self, device, nfloats):
        self.buf = torch.zeros((int(nfloats),), device=device, dtype=torch.float32)
        self.off = 0

    def __enter__(self):
        global _POOL
        self.prev, _POOL = _POOL, self
        return self

    def __exit__(self, *exc):
        global _POOL
        _POOL = self.prev


def step_zero_pool(device, nodes, width=64, feat=64, params=()):
    """zero_pool sized for one forward + backward of a generator + graph-convolution step on `nodes` nodes (ELL width `width`,
    conv width `feat`): the ~10 zero-initialised accumulators of the step then cost ONE fill launch instead of one each.  A pool
    that turns out too small is not an error: the remaining buffers are zero-filled one by one as without a pool."""
    nparam = sum(int(p_.numel()) for p_ in params)
    return zero_pool(device, 4 * nodes * width + 3 * nodes * max(feat, width) + 8 * nodes + 3 * nparam + (1 << 16))


def _zeros(shape, device):
    n = 1
    for s_ in (shape if isinstance(shape, (tuple, list, torch.Size)) else (shape,)):
        n *= int(s_)
    pool = _POOL
    if pool is not None and pool.buf.device == torch.device(device) and n > 0:
        n_al = (n + 63) // 64 * 64                           # 256-byte aligned slices
        if pool.off + n_al <= pool.buf.numel():
            v = pool.buf[pool.off:pool.off + n].view(shape)
            pool.off += n_al
            return v
    return torch.zeros(shape, device=device, dtype=torch.float32)


def _probe_begin():
    if PROBE is None:
        return None
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def _probe_end(name, e0):
    if e0 is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        PROBE.setdefault(name, []).append((e0, e1))


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def backward_will_follow(*tensors):
    """True when autograd will record the op about to be applied to `tensors` (grad mode on and some input requires grad).  It has to
    be evaluated by the CALLER of a torch.autograd.Function: inside Function.forward grad mode is always off, and ctx.needs_input_grad
    reflects the inputs' requires_grad flags whatever the grad mode (it is (True, ...) under torch.no_grad())."""
    return torch.is_grad_enabled() and any(torch.is_tensor(t) and t.requires_grad for t in tensors)


_KEEP = collections.deque(maxlen=64)


def _chk(t, dtype=torch.float32):
    assert t.is_cuda, "dgg ops run on the GPU only (no CPU fallback)"
    assert t.dtype == dtype, f"expected {dtype}, got {t.dtype}"
    if t.is_contiguous():
        return t
    # A contiguous copy made here is usually consumed as `_ptr(_chk(x))`: it must outlive the enqueue of the kernel that
    # reads it (once enqueued, the stream-ordered allocator makes reuse safe), so the last few copies are kept alive.
    c = t.contiguous()
    _KEEP.append(c)
    return c


# ------------------------------------------------------------------------------------------------------------
# raw kernels
# ------------------------------------------------------------------------------------------------------------
def linear_fwd(x, W, b=None, act=ACT_NONE, w_layout=0):
    x, W = _chk(x), _chk(W)
    N, d = x.shape
    out = W.shape[0] if w_layout == 0 else W.shape[1]
    assert (W.shape[1] if w_layout == 0 else W.shape[0]) == d
    y = torch.empty((N, out), device=x.device, dtype=torch.float32)
    bb = _chk(b) if b is not None else None
    pe = _probe_begin()
    _lib.check(_lib.lib().dgg_linear_fwd(_ptr(x), N, d, _ptr(W), _ptr(bb), out, w_layout, act, _ptr(y), _stream()), "linear_fwd")
    _probe_end("linear_fwd", pe)
    return y


def linear_bwd(x, W, y, dy, act=ACT_NONE, w_layout=0, need_dx=True, need_db=True):
    x, W, dy = _chk(x), _chk(W), _chk(dy)
    N, d = x.shape
    out = W.shape[0] if w_layout == 0 else W.shape[1]
    dx = torch.empty_like(x) if need_dx else None
    zz = _zeros((W.numel() + (out if need_db else 0),), x.device)                                     # one fill for both
    dW = zz[:W.numel()].view(W.shape)
    db = zz[W.numel():] if need_db else None
    ws = torch.empty((int(_lib.lib().dgg_linear_bwd_ws_floats(N, d, out)),), device=x.device, dtype=torch.float32)
    yy = _chk(y) if act != ACT_NONE else None
    pe = _probe_begin()
    _lib.check(_lib.lib().dgg_linear_bwd(_ptr(x), N, d, _ptr(W), out, w_layout, act, _ptr(yy), _ptr(dy), _ptr(dx), _ptr(dW),
                                         _ptr(db), _ptr(ws), _stream()), "linear_bwd")
    _probe_end("linear_bwd", pe)
    return dx, dW, db


def _ptr_array(ts):
    return (C.c_void_p * len(ts))(*[None if t is None else t.data_ptr() for t in ts])


def _int_array(vs):
    return (C.c_int * len(vs))(*[int(v) for v in vs])


def linear_fwd_multi(x, layers):
    """Several layers on ONE input, x read once: layers = [(W, b or None, act, w_layout), ...] -> [y_s].  Same fmaf chains as
    one linear_fwd per layer (bit-identical); falls back to those calls for shapes the fused kernel does not tile."""
    x = _chk(x)
    N, d = x.shape
    outs = [(W.shape[0] if lay == 0 else W.shape[1]) for W, _, _, lay in layers]
    if len(layers) > 1 and sum(outs) > 256 and not any(o % 32 for o in outs) and max(outs) <= 256:
        # wider than one fused pass (latent 128: xp 128 + H 64 + xk 128): consecutive groups of at most 256 columns, one pass each
        res, grp, tot = [], [], 0
        for lay_, o in zip(layers, outs):
            if grp and tot + o > 256:
                res += linear_fwd_multi(x, grp)
                grp, tot = [], 0
            grp.append(lay_)
            tot += o
        return res + linear_fwd_multi(x, grp)
    if len(layers) > 8 or any(o % 32 for o in outs) or sum(outs) > 256 or sum(outs) == 224:
        return [linear_fwd(x, W, b, act, lay) for W, b, act, lay in layers]
    # nn.Linear layout [out, d] stacked by rows (a [d, out] weight enters transposed); missing biases are zeros
    # (one launch stacks the weights: torch.cat x 2 + a fill + the transposes of the [d, out] weights were five)
    Wcat = torch.empty((sum(outs), d), device=x.device, dtype=torch.float32)
    bcat = torch.empty((sum(outs),), device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().dgg_linear_pack_weights(len(layers), _ptr_array([_chk(W) for W, *_ in layers]),
                                                  _ptr_array([None if b is None else _chk(b) for _, b, _, _ in layers]), _int_array(outs),
                                                  _int_array([lay for *_, lay in layers]), d, _ptr(Wcat), _ptr(bcat), _stream()),
               "linear_pack_weights")
    ys = [torch.empty((N, o), device=x.device, dtype=torch.float32) for o in outs]
    _lib.check(_lib.lib().dgg_linear_fwd_multi(_ptr(x), N, d, _ptr(Wcat), _ptr(bcat), len(layers), _int_array(outs),
                                               _int_array([a for _, _, a, _ in layers]), _ptr_array(ys), _stream()), "linear_fwd_multi")
    return ys


def linear_bwd_multi(x, layers):
    """Weight (and bias) gradients of several layers that share the input x, in ONE pass over x:
    layers = [(W, y or None, dy, act, w_layout, need_db), ...] -> [(dW_s, db_s or None)].  The activation derivative of layer s
    is applied to dy_s on the operand load (y_s = the layer's forward output)."""
    x = _chk(x)
    N, d = x.shape
    outs = [(W.shape[0] if lay == 0 else W.shape[1]) for W, _, _, _, lay, _ in layers]
    if len(layers) > 1 and sum(outs) > 256 and d <= 128 and not any(o % 32 for o in outs) and max(outs) <= 256:
        res, grp, tot = [], [], 0                            # (as linear_fwd_multi: groups of at most 256 columns)
        for lay_, o in zip(layers, outs):
            if grp and tot + o > 256:
                res += linear_bwd_multi(x, grp)
                grp, tot = [], 0
            grp.append(lay_)
            tot += o
        return res + linear_bwd_multi(x, grp)
    # inputs wider than 128 columns (Pubmed: 500): the 128-column tiles of the wide kernel -- widths in multiples of 64, d % 4 == 0
    wide_in = d > 128 and d % 4 == 0 and not any(o % 64 for o in outs) and x.data_ptr() % 16 == 0
    if len(layers) > 8 or any(o % 32 for o in outs) or sum(outs) > 256 or (d > 128 and not wide_in):
        res = []
        for W, y, dy, act, lay, need_db in layers:
            _, dW, db = linear_bwd(x, W, y, dy, act, lay, need_dx=False, need_db=need_db)
            res.append((dW, db))
        return res
    nW = [int(W.numel()) for W, *_ in layers]
    zz = _zeros((sum(nW) + sum(o for o, l_ in zip(outs, layers) if l_[5]),), x.device)                 # one fill for all outputs
    dWs, dbs, o_ = [], [], 0
    for (W, *_), n_ in zip(layers, nW):
        dWs.append(zz[o_:o_ + n_].view(W.shape))
        o_ += n_
    for (_, _, _, _, _, need_db), o in zip(layers, outs):
        dbs.append(zz[o_:o_ + o] if need_db else None)
        o_ += o if need_db else 0
    ws = torch.empty((int(_lib.lib().dgg_gemm_tn_multi_ws_floats(N, sum(outs), d)),), device=x.device, dtype=torch.float32)
    dys = [_chk(l_[2]) for l_ in layers]
    ysv = [(_chk(l_[1]) if (l_[1] is not None and l_[3] != ACT_NONE) else None) for l_ in layers]
    _lib.check(_lib.lib().dgg_gemm_tn_multi(len(layers), _ptr_array(dys), _int_array(outs), _ptr_array(ysv),
                                            _int_array([l_[3] for l_ in layers]), _ptr(x), N, d, _ptr_array(dWs),
                                            _int_array([0 if l_[4] == 0 else 1 for l_ in layers]), _ptr_array(dbs), _ptr(ws), _stream()),
               "gemm_tn_multi")
    return list(zip(dWs, dbs))


def gemm_tn(A, B, colsum=False):
    """A[N,M1]^T B[N,M2] -> [M1,M2] (+ column sums of A)."""
    A, B = _chk(A), _chk(B)
    N, M1 = A.shape
    M2 = B.shape[1]
    zz = _zeros((M1 * M2 + (M1 if colsum else 0),), A.device)                                         # one fill for both
    Cm = zz[:M1 * M2].view(M1, M2)
    cs = zz[M1 * M2:] if colsum else None
    ws = torch.empty((int(_lib.lib().dgg_gemm_tn_ws_floats(N, M1, M2)),), device=A.device, dtype=torch.float32)
    _lib.check(_lib.lib().dgg_gemm_tn_acc(_ptr(A), _ptr(B), N, M1, M2, _ptr(Cm), 0, _ptr(cs), _ptr(ws), _stream()), "gemm_tn_acc")
    return (Cm, cs) if colsum else Cm


def gemm_tn_pairs(pairs):
    """[(A_p [N,M1_p], B_p [N,M2_p]), ...] -> [(A_p^T B_p, column sums of A_p)] in one launch (+ one reduce per product); falls
    back to one gemm_tn per product for shapes the batched kernel does not cover"""
    As, Bs = [_chk(a) for a, _ in pairs], [_chk(b) for _, b in pairs]
    N = As[0].shape[0]
    M1s, M2s = [a.shape[1] for a in As], [b.shape[1] for b in Bs]
    tot = sum((m + 31) // 32 * 32 for m in M1s)
    if len(pairs) > 8 or tot > 256 or max(M2s) > 128:
        return [gemm_tn(a, b, colsum=True) for a, b in zip(As, Bs)]
    zz = _zeros((sum(m1 * m2 + m1 for m1, m2 in zip(M1s, M2s)),), As[0].device)
    Cs, css, o = [], [], 0
    for m1, m2 in zip(M1s, M2s):
        Cs.append(zz[o:o + m1 * m2].view(m1, m2))
        css.append(zz[o + m1 * m2:o + m1 * m2 + m1])
        o += m1 * m2 + m1
    ws = torch.empty((int(_lib.lib().dgg_gemm_tn_multi_ws_floats(N, tot, 128)),), device=As[0].device, dtype=torch.float32)
    _lib.check(_lib.lib().dgg_gemm_tn_pairs(len(pairs), _ptr_array(As), _int_array(M1s), _ptr_array(Bs), _int_array(M2s), N, _ptr_array(Cs),
                                            _ptr_array(css), _ptr(ws), _stream()), "gemm_tn_pairs")
    return list(zip(Cs, css))


_DEGSTAT_CACHE = {}


def degree_stats(deg):
    """(mean, unbiased std) of the prior degrees as a device tensor [2].  The prior degrees are an INPUT statistic of a static graph:
    the result is cached per tensor object and version (three launches per forward otherwise)."""
    import weakref
    key = id(deg)
    ent = _DEGSTAT_CACHE.get(key)
    if ent is not None and ent[0]() is deg and ent[1] == deg._version:
        return ent[2]
    degc = _chk(deg)
    out = torch.empty((2,), device=degc.device, dtype=torch.float32)
    ws = torch.empty((int(_lib.lib().dgg_degree_stats_ws_bytes()),), device=degc.device, dtype=torch.uint8)
    _lib.check(_lib.lib().dgg_degree_stats(_ptr(degc), degc.shape[0], _ptr(out), _ptr(ws), _stream()), "degree_stats")
    if not torch.cuda.is_current_stream_capturing() and not deg.requires_grad:
        for k_ in [k_ for k_, v in _DEGSTAT_CACHE.items() if v[0]() is None]:
            del _DEGSTAT_CACHE[k_]
        _DEGSTAT_CACHE[key] = (weakref.ref(deg), deg._version, out)
    return out


def allpairs_topk(xp, K=DEFAULT_K, t=T_DIST, noise_mode=NOISE_NONE, G=None, seed=(0, 0), rows=None, algo=0,
                  return_ws=False, k_limit=None, status=None):
    """k_limit: learned k of the rows (optional): ranks that the soft top-k ramp zeroes exactly come back as idx = -1.
    status (dict, optional): receives "rsym_err", a 1-element int32 DEVICE tensor that is non-zero when the ranked symmetric
    generator (noise_mode 5) could not settle every row inside its workspace, and "rsym_tier3", the number of rows that took its
    dense tier (no synchronisation here; the caller checks them).  status["sym_fallback"] = True on entry: the error word IS read back
    (one synchronisation, never inside a capture) and a failed forward is redone under NOISE_HASH_SYM, status["rsym_fell_back"] = True."""
    xp = _chk(xp)
    N, h = xp.shape
    r0, r1 = (0, N) if rows is None else rows
    idx = torch.empty((r1 - r0, K), device=xp.device, dtype=torch.int32)
    val = torch.empty((r1 - r0, K), device=xp.device, dtype=torch.float32)
    ldG = 0
    if G is not None:
        G = _chk(G)
        assert G.shape[-1] == N and G.shape[0] == N, "explicit noise must be [N, N]"
        ldG = N
    ws, ws_bytes = None, 0
    if algo != 1 or noise_mode == NOISE_RANKED_SYM:      # (the ranked symmetric generator has no workspace-free form, whatever `algo` says)
        ws_bytes = int(_lib.lib().dgg_allpairs_workspace_bytes(N, h, noise_mode, K))
        if ws_bytes:
            ws = torch.empty((ws_bytes,), device=xp.device, dtype=torch.uint8)
    pe = _probe_begin()
    _lib.check(_lib.lib().dgg_allpairs_topk(_ptr(xp), N, h, r0, r1, t, noise_mode, _ptr(G), ldG, seed[0], seed[1], K,
                                            _ptr(idx), _ptr(val), _ptr(None if k_limit is None else _chk(k_limit)), algo, _ptr(ws),
                                            ws_bytes, _stream()), "allpairs_topk")
    _probe_end("allpairs_topk", pe)
    if status is not None and noise_mode == NOISE_RANKED_SYM and ws is not None:
        off = int(_lib.lib().dgg_allpairs_rsym_ctl_offset_bytes(r1 - r0, N))
        ctl = ws[off:off + 32].view(torch.int32)
        if status.get("sym_fallback") and not torch.cuda.is_current_stream_capturing() and bool(ctl[4].item()):
            # The ranked symmetric generator could not settle every row inside its workspace (more rows far from everything else than
            # its dense tier holds: a property of the DATA -- latents a few Adam steps into training already do it at N = 100 000): the
            # rows concerned came back empty.  On request the SAME forward is evaluated again under the symmetric per-pair hash
            # generator (the same law, another realisation; exact on any data) and the caller is told to stay with it.
            status["rsym_fell_back"] = True
            status["rsym_err"] = None
            return allpairs_topk(xp, K, t, NOISE_HASH_SYM, None, seed, rows, 0, return_ws, k_limit, None)
        status["rsym_err"] = ctl[4:5].clone()
        status["rsym_tier3"] = ctl[3:4].clone()       # rows that needed the dense tier (each one costs a full walk of every owner's sequence)
        # how far below the tier-1 threshold tier 2 had to walk (log-score units; -inf: no row failed): every owner walks that deep
        nm = ctl[6:7]
        need = torch.where(nm >= 0, nm, nm ^ 0x7fffffff).view(torch.float32)
        status["rsym_depth"] = ctl[2:3].view(torch.float32) - need
    if return_ws:      # diagnostics: the guess-and-verify control block is ws[:16] = (msum f32, nfail i32, gmin0 f32)
        return idx, val, ws
    return idx, val


def ranked_probe(xp, k_limit=None, t=T_DIST, seed=(0, 0), rows=None, stride=1, max_blocks=0, lpub=None):
    """Walk statistics of the ranked search (NOISE_RANKED) on every `stride`-th row, each walk cut after `max_blocks` blocks of 64
    ranks (0: none): dict(rows, blocks_per_row, gathered_per_row, scored_per_row, budget_hit_frac, max_blocks).  One
    synchronisation; the search's depth is a property of the data (dgg_allpairs_ranked_probe)."""
    xp = _chk(xp)
    N, h = xp.shape
    r0, r1 = (0, N) if rows is None else rows
    cnt = torch.zeros((6,), device=xp.device, dtype=torch.int64)
    if torch.is_tensor(seed):
        sd = seed.cpu()
        seed = (int(sd[0]) & 0xFFFFFFFF, int(sd[1]) & 0xFFFFFFFF)
    _lib.check(_lib.lib().dgg_allpairs_ranked_probe(_ptr(xp), N, h, r0, r1, t, seed[0], seed[1], _ptr(None if k_limit is None else _chk(k_limit)),
                                                    int(stride), int(max_blocks), _ptr(cnt), _ptr(None if lpub is None else _chk(lpub)), _stream()),
               "allpairs_ranked_probe")
    c = [int(v) for v in cnt.cpu()]
    n = max(c[0], 1)
    return {"rows": c[0], "blocks_per_row": c[1] / n, "gathered_per_row": c[2] / n, "scored_per_row": c[3] / n,
            "budget_hit_frac": c[4] / n, "max_blocks": c[5], "stride": int(stride), "block_budget": int(max_blocks)}


# Cost model of the ranked search (DGG_LearnableK_debug._asym_generator_now, bench.py): ~1.44 ns of chip time per visited block of 64
# ranks (0.23 ms for 1.6 blocks per row at N = 100 000); its slowest row ~3 us per block on ONE wavefront.
RANKED_NS_PER_BLOCK, RANKED_US_PER_SERIAL_BLOCK = 1.44, 3.0


def ranked_cost_estimate(probe, N, rows=None):
    """Estimated time (us) of the ranked search from a (sampled, budgeted) ranked_probe: mean blocks per row over all rows, a row
    that hit the budget taken to the middle of the rest of its N / 0.76 / 64 blocks; at least one serial deep walk if any did."""
    rows = N if rows is None else rows
    full_walk_blocks = N / 0.76 / 64.0
    mean_blocks = probe["blocks_per_row"] + probe["budget_hit_frac"] * max(full_walk_blocks - probe["block_budget"], 0.0) * 0.5
    us = rows * mean_blocks * RANKED_NS_PER_BLOCK * 1e-3
    if probe["budget_hit_frac"] > 0:
        us = max(us, 0.5 * full_walk_blocks * RANKED_US_PER_SERIAL_BLOCK)
    return us


def rsym_status(ws, N, rows=None):
    """diagnostics of the ranked symmetric path (noise_mode 5; synchronises): rows redone by tier 2 / tier 3, the error flag, the
    guessed threshold, and (under DGG_RSYM_STATS=1) emitted pairs and scored candidates of tier 1"""
    off = int(_lib.lib().dgg_allpairs_rsym_ctl_offset_bytes(N if rows is None else rows, N))
    blk = ws[off:off + 64].cpu()
    i32, f32, i64 = blk.view(torch.int32), blk.view(torch.float32), blk.view(torch.int64)
    return dict(tier2_rows=int(i32[1]), gmin0=float(f32[2]), tier3_rows=int(i32[3]), err=int(i32[4]), emitted=int(i64[4]), delivered=int(i64[5]), tier2_delivered=int(i64[6]))


def fast_path_failed_rows(ws, N, h, rows=None, stats=False):
    """diagnostics of the unperturbed sweep (noise_mode 0, N >= 8192): rows of the last call that failed the verification of
    their guessed radius and were redone by the exhaustive fallback (synchronises); stats=True also returns the candidate
    sums (phase-A hits, those inside the tight radius, phase-B hits; collected only under DGG_SWEEP_STATS=1)"""
    off = int(_lib.lib().dgg_allpairs_sweep_ctl_offset_bytes(N if rows is None else rows, N, h))
    blk = ws[off:off + 64].cpu()
    nfail = int(blk[0:4].view(torch.int32).item())
    if not stats:
        return nfail
    # + the failed rows by reason (phase-A overflow, phase-B overflow, > FCAP candidates, < L candidates, list not full, score below the
    # normal range, last distance outside the radius)
    return nfail, [int(v) for v in blk[8:32].view(torch.int64)] + [int(v) for v in blk[32:60].view(torch.int32)]


def allpairs_topk_softk(xp, k, mode=MODE_K_TIMES_EDGE_PROB, t=T_DIST, seed=(0, 0), rows=None, lpub=None):
    """ranked-noise all-pairs top-64 with the first-k ramp fused (allpairs_topk(k_limit=k) + softk_fwd in one launch)
    -> idx, val, w [rows,64], rs [rows].  lpub (optional, [rows]): rowmin_logp_bound's upper bounds of log p over a row's other nodes --
    the walk stops on G + lpub[i] instead of the distance-free G + 1e-8 (same result, fewer ranks on spread latents)."""
    xp, k = _chk(xp), _chk(k)
    N, h = xp.shape
    r0, r1 = (0, N) if rows is None else rows
    idx = torch.empty((r1 - r0, 64), device=xp.device, dtype=torch.int32)
    val = torch.empty((r1 - r0, 64), device=xp.device, dtype=torch.float32)
    w = torch.empty((r1 - r0, 64), device=xp.device, dtype=torch.float32)
    rs = torch.empty((r1 - r0,), device=xp.device, dtype=torch.float32)
    pe = _probe_begin()
    dseed = None
    if isinstance(seed, torch.Tensor):       # device seed [2] (int32 / uint32 bits): one captured graph, fresh noise per replay
        assert seed.is_cuda and seed.numel() == 2 and seed.element_size() == 4
        dseed, seed = seed, (0, 0)
    _lib.check(_lib.lib().dgg_allpairs_topk_ranked_softk_lp(_ptr(xp), N, h, r0, r1, t, seed[0], seed[1], _ptr(dseed), _ptr(None if lpub is None else _chk(lpub)),
                                                            _ptr(k), mode, _ptr(idx), _ptr(val), _ptr(w), _ptr(rs), _stream()), "allpairs_topk_ranked_softk_lp")
    _probe_end("allpairs_topk", pe)
    return idx, val, w, rs


def rowmin_logp_bound(xp, t=T_DIST, rows=None):
    """-> lpub [rows]: for every row i a rigorous upper bound of log p_ij = log(exp(t ||xp_i - xp_j||) + 1e-8) over all j != i, from an
    fp16-MFMA lower bound of the distance to the row's nearest other node (include/dgg_hip.h, dgg_allpairs_rowmin_bound: one N^2 sweep on
    the matrix cores, ~1 ms at N = 100 000).  The searches take it in place of the distance-free bound of a pair's log-score."""
    xp = _chk(xp)
    N, h = xp.shape
    r0, r1 = (0, N) if rows is None else rows
    out = torch.empty((r1 - r0,), device=xp.device, dtype=torch.float32)
    nb = int(_lib.lib().dgg_allpairs_rowmin_ws_bytes(r1 - r0, N, h))
    ws = torch.empty((nb,), device=xp.device, dtype=torch.uint8)
    pe = _probe_begin()
    _lib.check(_lib.lib().dgg_allpairs_rowmin_bound(_ptr(xp), N, h, r0, r1, float(t), _ptr(out), _ptr(ws), nb, _stream()), "allpairs_rowmin_bound")
    _probe_end("rowmin_bound", pe)
    return out


# ---- rows wider than 64 ranks: chunked rows ------------------------------------------------------------------------------------------
CHUNK_MAXM = 32          # chunks of 64 ranks per row that the ranked search holds in REGISTER lists (2048 ranks); wider rows and the other
                         # noise generators go through threshold buffers in memory (dgg_allpairs_topk_anywide): any width
CHUNK_MAXM_ANY = 1 << 20
ANYWIDE_HASH_BOUND = os.environ.get("DGG_ANYWIDE_HASH_BOUND", "1") != "0"     # (0: the hash generators' wide rows on the distance-free filter)


class ChunkCapacityError(RuntimeError):
    """a learned degree needs more ranks than the caller allowed for the chunked rows (64 * maxm), or is not a number"""


class ChunkLayout:
    """Chunked rows (include/dgg_hip.h, dgg_chunk_layout): node i owns the chunks [cptr[i], cptr[i+1]) of the [chunks,64] arrays, rank r
    of its row is entry r % 64 of chunk r / 64.  cptr int32 [rows+1], cnode int32 [chunks] (node of every chunk), meta int32 [4] on the
    device = {chunks needed, widest row in chunks, flags, 0}; `chunks` = rows of the arrays (>= chunks in use), `maxm` = list count the
    search is launched with (>= widest row)."""

    def __init__(self, cptr, cnode, meta, chunks, maxm, rows):
        self.cptr, self.cnode, self.meta, self.chunks, self.maxm, self.rows = cptr, cnode, meta, int(chunks), int(maxm), int(rows)

    @property
    def wide(self):
        return self.chunks != self.rows

    def ranks(self):
        """rank of every entry inside its row, int64 [chunks,64] (tests / densification)"""
        first = self.cptr[:-1].long()[self.cnode.long()]
        c = torch.arange(self.chunks, device=self.cptr.device)
        return ((c - first) * 64)[:, None] + torch.arange(64, device=self.cptr.device)[None, :]


def chunk_maxm_for(ncols):
    """chunks per row that hold EVERY column of a graph of `ncols` nodes (+ one: the ramp's margin): a row never needs more"""
    return int(min(CHUNK_MAXM_ANY, (int(ncols) + 63) // 64 + 1))


def chunk_layout(k, maxm=None, ccap=None, ncols=None, sticky=None):
    """Layout of the chunked rows for the learned degrees k [rows].  ccap=None: ONE host synchronisation (the chunk count sizes the
    arrays) -> ChunkLayout with exactly the chunks in use, and `maxm` = the widest row.  ccap given (a fixed capacity, e.g. under
    hipGraph capture): no synchronisation; meta[2] carries the flags of this call, `sticky` (int32[1] device tensor, optional) collects
    them over calls -- an overflowing layout is CUT to the capacity on the device (no consumer leaves the arrays).
    maxm=None: rows of any width -- up to every column of the graph (`ncols`, default = rows; a learned degree beyond that keeps them
    all, as the reference's dense row does); maxm given: raises ChunkCapacityError when a row needs more than 64 * maxm ranks.  A learned
    degree that is NaN always raises (synchronising form) / sets flag 4."""
    k = _chk(k)
    rows = k.shape[0]
    dev = k.device
    limit = maxm is not None
    if maxm is None:
        maxm = chunk_maxm_for(rows if ncols is None else ncols)
    meta = torch.empty((4 + 384,), device=dev, dtype=torch.int32)          # {chunks, widest row, flags, 0} + scratch of the two-pass scan
    cptr = torch.empty((rows + 1,), device=dev, dtype=torch.int32)
    if ccap is not None:
        cnode = torch.empty((int(ccap),), device=dev, dtype=torch.int32)
        _lib.check(_lib.lib().dgg_chunk_layout(_ptr(k), rows, int(maxm), int(ccap), _ptr(cptr), _ptr(cnode), _ptr(meta), _ptr(sticky), _stream()),
                   "chunk_layout")
        return ChunkLayout(cptr, cnode, meta, ccap, maxm, rows)
    cap = rows + rows // 4 + 64
    while True:
        cnode = torch.empty((cap,), device=dev, dtype=torch.int32)
        _lib.check(_lib.lib().dgg_chunk_layout(_ptr(k), rows, int(maxm), cap, _ptr(cptr), _ptr(cnode), _ptr(meta), None, _stream()), "chunk_layout")
        total, widest, flags, _ = (int(v) for v in meta[:4].cpu())
        if flags & 4:
            raise ChunkCapacityError("chunk_layout: a learned degree is NaN")
        if (flags & 1) and limit:
            raise ChunkCapacityError(f"chunk_layout: a learned degree needs more than {64 * maxm} ranks (k + 9.5 > {64 * maxm}): beyond the "
                                     "capacity the caller allowed")
        if not (flags & 2):
            return ChunkLayout(cptr, cnode[:total], meta, total, max(widest, 1), rows)
        cap = total


def allpairs_topk_wide(xp, k, layout, mode=MODE_K_TIMES_EDGE_PROB, t=T_DIST, seed=(0, 0), rows=None, ramp=True, noise_mode=None, lpub=None):
    """All-pairs top-L_i on chunked rows (rows wider than 64 ranks, ANY width): -> idx, val, w [chunks,64], rs [rows] (w / rs None with
    ramp=False).  `layout` = chunk_layout(k of these rows).  noise_mode: NOISE_RANKED (default: the register-list search for the rows
    of up to 32 chunks + the threshold-buffer walk for the wider ones), NOISE_NONE / NOISE_HASH / NOISE_HASH_SYM (threshold buffers,
    every row; include/dgg_hip.h, dgg_allpairs_topk_anywide).  NOISE_RANKED_SYM has no wide-row form of its own: callers evaluate wide
    rows under NOISE_HASH_SYM (the same law, another realisation).  lpub ([rows], optional): rowmin_logp_bound's upper bounds of log p
    over a row's other nodes, for the stop tests of the ranked search; the per-pair hash generators compute it themselves (their integer
    filter admits 1 / E[p^(1/0.3)] times fewer candidates with it: 24 -> ~8 ms at N = 100 000, k ~ 130)."""
    xp, k = _chk(xp), _chk(k)
    N, h = xp.shape
    r0, r1 = (0, N) if rows is None else rows
    assert layout.rows == r1 - r0
    nm = NOISE_RANKED if noise_mode is None else int(noise_mode)
    assert nm in (NOISE_NONE, NOISE_HASH, NOISE_HASH_SYM, NOISE_RANKED), f"chunked rows: noise_mode {nm} has no wide-row evaluator"
    C_ = layout.chunks
    idx = torch.empty((C_, 64), device=xp.device, dtype=torch.int32)
    val = torch.empty((C_, 64), device=xp.device, dtype=torch.float32)
    w = torch.empty((C_, 64), device=xp.device, dtype=torch.float32) if ramp else None
    rs = torch.empty((r1 - r0,), device=xp.device, dtype=torch.float32) if ramp else None
    dseed = None
    if isinstance(seed, torch.Tensor):
        assert seed.is_cuda and seed.numel() == 2 and seed.element_size() == 4
        dseed, seed = seed, (0, 0)
    if lpub is None and nm in (NOISE_HASH, NOISE_HASH_SYM) and ANYWIDE_HASH_BOUND:
        lpub = rowmin_logp_bound(xp, t, rows=(r0, r1))
    lpub = None if lpub is None else _chk(lpub)
    pe = _probe_begin()
    if nm == NOISE_RANKED:
        _lib.check(_lib.lib().dgg_allpairs_topk_ranked_wide(_ptr(xp), N, h, r0, r1, t, seed[0], seed[1], _ptr(dseed), _ptr(k), mode, layout.maxm,
                                                            _ptr(layout.cptr), C_, _ptr(idx), _ptr(val), _ptr(w), _ptr(rs), _ptr(lpub), _stream()),
                   "allpairs_topk_ranked_wide")
    if nm != NOISE_RANKED or layout.maxm > CHUNK_MAXM:
        nb = int(_lib.lib().dgg_allpairs_anywide_ws_bytes(C_, r1 - r0, N, h))
        ws = torch.empty((nb,), device=xp.device, dtype=torch.uint8)
        _lib.check(_lib.lib().dgg_allpairs_topk_anywide(_ptr(xp), N, h, r0, r1, t, nm, seed[0], seed[1], _ptr(dseed), _ptr(k), mode, layout.maxm,
                                                        CHUNK_MAXM if nm == NOISE_RANKED else 0, _ptr(layout.cptr), C_, _ptr(idx), _ptr(val),
                                                        _ptr(w), _ptr(rs), _ptr(lpub), _ptr(ws), nb, _stream()), "allpairs_topk_anywide")
    _probe_end("allpairs_topk", pe)
    return idx, val, w, rs


def edgelist_topk(xp, rowptr, col, K=DEFAULT_K, t=T_DIST, noise_mode=NOISE_NONE, G=None, seed=(0, 0)):
    xp = _chk(xp)
    N, h = xp.shape
    rowptr, col = _chk(rowptr, torch.int64), _chk(col, torch.int32)
    idx = torch.empty((N, K), device=xp.device, dtype=torch.int32)
    val = torch.empty((N, K), device=xp.device, dtype=torch.float32)
    ldG = 0
    if G is not None:
        G = _chk(G)
        ldG = N
    _lib.check(_lib.lib().dgg_edgelist_topk(_ptr(xp), N, h, _ptr(rowptr), _ptr(col), t, noise_mode, _ptr(G), ldG, seed[0],
                                            seed[1], K, _ptr(idx), _ptr(val), _stream()), "edgelist_topk")
    return idx, val


def edgelist_topk_softk(xp, rowptr, col, k, mode=MODE_K_TIMES_EDGE_PROB, K=DEFAULT_K, t=T_DIST, noise_mode=NOISE_NONE, G=None, seed=(0, 0),
                        overflow=None):
    """edgelist_topk + softk_fwd in one launch (same bits): -> idx, val, w, rs; None when the latent width is not 16 / 32 / 64 / 128.
    overflow (int32[1] device tensor, optional): ORed with 1 when a row with more than K candidates has k + 8.5 > K."""
    xp = _chk(xp)
    N, h = xp.shape
    if h not in (16, 32, 64, 128):
        return None
    rowptr, col, k = _chk(rowptr, torch.int64), _chk(col, torch.int32), _chk(k)
    idx = torch.empty((N, K), device=xp.device, dtype=torch.int32)
    val = torch.empty((N, K), device=xp.device, dtype=torch.float32)
    w = torch.empty((N, K), device=xp.device, dtype=torch.float32)
    rs = torch.empty((N,), device=xp.device, dtype=torch.float32)
    ldG = 0
    if G is not None:
        G = _chk(G)
        ldG = N
    pe = _probe_begin()
    _lib.check(_lib.lib().dgg_edgelist_topk_softk(_ptr(xp), N, h, _ptr(rowptr), _ptr(col), t, noise_mode, _ptr(G), ldG, seed[0], seed[1],
                                                  K, _ptr(k), mode, _ptr(idx), _ptr(val), _ptr(w), _ptr(rs), _ptr(overflow), _stream()), "edgelist_topk_softk")
    _probe_end("edgelist_topk", pe)
    return idx, val, w, rs


def literal_hard_fwd(xp, cand, idx, w, t=T_DIST, noise_mode=NOISE_NONE, G=None, seed=(0, 0), threshold=0.5):
    """the debug class's literal dgg_hard output (dgm.py:1294-1311) from the soft ELL adjacency (idx, w): -> hidx, hval, hsrc [N,K];
    cand = None (all pairs) or (rowptr, col); N <= 8192"""
    xp, idx, w = _chk(xp), _chk(idx, torch.int32), _chk(w)
    N, h = xp.shape
    K = idx.shape[1]
    rowptr = col = None
    if cand is not None:
        rowptr, col = _chk(cand[0], torch.int64), _chk(cand[1], torch.int32)
    ldG = 0
    if G is not None:
        G = _chk(G)
        ldG = N
    hidx = torch.empty((N, K), device=xp.device, dtype=torch.int32)
    hval = torch.empty((N, K), device=xp.device, dtype=torch.float32)
    hsrc = torch.empty((N, K), device=xp.device, dtype=torch.int32)
    _lib.check(_lib.lib().dgg_literal_hard_fwd(_ptr(xp), N, h, _ptr(rowptr), _ptr(col), float(t), noise_mode, _ptr(G), ldG, seed[0], seed[1],
                                               _ptr(idx), _ptr(w), K, float(threshold), _ptr(hidx), _ptr(hval), _ptr(hsrc), _stream()),
               "literal_hard_fwd")
    return hidx, hval, hsrc


class LiteralHardFn(torch.autograd.Function):
    """soft ELL weights -> values of the literal hard adjacency; the cotangent of a one goes to the soft weight AT ITS COLUMN
    (the reference's `(hard - soft).detach() + soft` followed by `.to_sparse()`: only stored entries carry gradient)"""

    @staticmethod
    def forward(ctx, w, idx, xp, cand, t, noise_mode, G, seed):
        hidx, hval, hsrc = literal_hard_fwd(xp, cand, idx, w.detach(), t, noise_mode, G, seed)
        ctx.save_for_backward(hsrc)
        ctx.mark_non_differentiable(hidx)
        return hval, hidx

    @staticmethod
    def backward(ctx, g, _):
        (hsrc,) = ctx.saved_tensors
        g = _chk(g.contiguous())
        N, K = g.shape
        dw = torch.empty_like(g)
        _lib.check(_lib.lib().dgg_literal_hard_bwd(_ptr(hsrc), _ptr(g), N, K, _ptr(dw), _stream()), "literal_hard_bwd")
        return dw, None, None, None, None, None, None, None


def edge_mlp_fwd(AB, xp, erow, col, deg, ex_in, ex_mode, t_ex, wdu, wdv, wex, b1, w2, b2, act=ACT_LEAKY):
    """edge-MLP scorer on the candidate edges (dgm.py:1628-1725) -> p_edge [E], ex [E] (the per-edge extra used)"""
    AB, xp = _chk(AB), _chk(xp)
    N, h = xp.shape
    hw = AB.shape[1] // 2
    erow, col = _chk(erow, torch.int32), _chk(col, torch.int32)
    E = col.shape[0]
    p_edge = torch.empty((E,), device=xp.device, dtype=torch.float32)
    ex_out = torch.empty((E,), device=xp.device, dtype=torch.float32) if ex_mode else None
    if E == 0:                                           # no candidate edge at all (empty tensors have no data pointer)
        return p_edge, ex_out
    o = lambda t_: None if t_ is None else _chk(t_)  # noqa: E731
    _lib.check(_lib.lib().dgg_edge_mlp_fwd(_ptr(AB), _ptr(xp), N, h, hw, _ptr(erow), _ptr(col), E, _ptr(o(deg)), _ptr(o(ex_in)), ex_mode,
                                           float(t_ex), _ptr(o(wdu)), _ptr(o(wdv)), _ptr(o(wex)), _ptr(_chk(b1)), _ptr(_chk(w2)),
                                           _ptr(_chk(b2)), act, _ptr(p_edge), _ptr(ex_out), _stream()), "edge_mlp_fwd")
    return p_edge, ex_out


def edgelist_topk_p(p_edge, N, rowptr, col, K=DEFAULT_K, noise_mode=NOISE_NONE, G=None, seed=(0, 0)):
    """perturbation + per-row top-K on given edge probabilities -> idx, val, eid [N,K]"""
    p_edge = _chk(p_edge)
    rowptr, col = _chk(rowptr, torch.int64), _chk(col, torch.int32)
    idx = torch.empty((N, K), device=p_edge.device, dtype=torch.int32)
    val = torch.empty((N, K), device=p_edge.device, dtype=torch.float32)
    eid = torch.empty((N, K), device=p_edge.device, dtype=torch.int32)
    ldG = 0
    if G is not None:
        G = _chk(G)
        ldG = G.shape[-1]
    _lib.check(_lib.lib().dgg_edgelist_topk_p(_ptr(p_edge), N, _ptr(rowptr), _ptr(col), noise_mode, _ptr(G), ldG, seed[0], seed[1], K,
                                              _ptr(idx), _ptr(val), _ptr(eid), _stream()), "edgelist_topk_p")
    return idx, val, eid


def edge_mlp_bwd(AB, idx, eid, val, dval, deg, ex, wdu, wdv, wex, b1, w2, b2, act=ACT_LEAKY, perturb=False, need_dex=False,
                 rowptr=None, partp=None, w=None, nrec_max=0):
    """-> dAB [N,2hw], dpar [5hw+1] = [dwdu|dwdv|dwex|db1|dw2|db2], dex (shape of dval) or None.
    ELL adjacency: idx/eid/val/dval [N,K]; CSR-valued adjacency: rowptr given, idx = col [E], val/dval [E], eid None.
    partp (+ w, the weights it was built from, and nrec_max >= its number of records, e.g. the number of candidate edges): the
    neighbour-side sums without float atomics (dgg_edge_mlp_bwd_partp)"""
    AB = _chk(AB)
    N = AB.shape[0]
    K = idx.shape[1] if rowptr is None else 0
    hw = AB.shape[1] // 2
    zz = _zeros((N * 2 * hw + 5 * hw + 1,), AB.device)
    dAB, dpar = zz[:N * 2 * hw].view(N, 2 * hw), zz[N * 2 * hw:]
    dex = torch.empty(tuple(dval.shape), device=AB.device, dtype=torch.float32) if need_dex else None
    o = lambda t_: None if t_ is None else _chk(t_)  # noqa: E731
    if (EMLP_BWD_PARTP and partp is not None and w is not None and nrec_max > 0 and rowptr is None and hw % 4 == 0 and partp.layout is None
            and partp.rows == N and partp_has_map(N)):
        dz = torch.empty((int(nrec_max) * hw,), device=AB.device, dtype=torch.float32)
        _lib.check(_lib.lib().dgg_edge_mlp_bwd_partp(_ptr(AB), N, hw, _ptr(idx), _ptr(eid), _ptr(_chk(val)), _ptr(_chk(dval)), _ptr(_chk(w)), K,
                                                     _ptr(o(deg)), _ptr(o(ex)), _ptr(o(wdu)), _ptr(o(wdv)), _ptr(o(wex)), _ptr(_chk(b1)),
                                                     _ptr(_chk(w2)), _ptr(_chk(b2)), act, int(perturb), _ptr(partp.ws), partp.ncols, _ptr(dz),
                                                     int(nrec_max), _ptr(dAB), _ptr(dpar), _ptr(dex), _stream()), "edge_mlp_bwd_partp")
        return dAB, dpar, dex
    _lib.check(_lib.lib().dgg_edge_mlp_bwd(_ptr(AB), N, hw, _ptr(rowptr), _ptr(idx), _ptr(eid), _ptr(_chk(val)), _ptr(_chk(dval)), K, _ptr(o(deg)),
                                           _ptr(o(ex)), _ptr(o(wdu)), _ptr(o(wdv)), _ptr(o(wex)), _ptr(_chk(b1)), _ptr(_chk(w2)),
                                           _ptr(_chk(b2)), act, int(perturb), _ptr(dAB), _ptr(dpar), _ptr(dex), _stream()),
               "edge_mlp_bwd")
    return dAB, dpar, dex


# ---- CSR-valued adjacency (variable row length): the `DGG` class and the *_DGG_00 wrappers ----------------------------
class CsrSoftkFn(torch.autograd.Function):
    """select_top_k (dgm.py:1402-1435) on the CSR pattern: (p [E], k [N]) -> w [E]; rows of any width"""

    @staticmethod
    def forward(ctx, p, k, rowptr, col, noise_mode, G, seed, mode):
        p, k = _chk(p), _chk(k)
        N = rowptr.shape[0] - 1
        w, pp = torch.empty_like(p), torch.empty_like(p)
        pos = torch.empty(p.shape, device=p.device, dtype=torch.int32)
        ldG = 0
        if G is not None:
            G = _chk(G)
            ldG = G.shape[-1]
        if p.numel():
            _lib.check(_lib.lib().dgg_csr_softk_fwd(_ptr(p), _ptr(rowptr), _ptr(col), N, _ptr(k), noise_mode, _ptr(G), ldG, seed[0], seed[1], mode,
                                                    _ptr(w), _ptr(pp), _ptr(pos), _stream()), "csr_softk_fwd")
        ctx.save_for_backward(p, pp, k, pos, rowptr)
        ctx.cfg = (noise_mode != NOISE_NONE, mode)
        return w

    @staticmethod
    def backward(ctx, g):
        p, pp, k, pos, rowptr = ctx.saved_tensors
        perturb, mode = ctx.cfg
        N = rowptr.shape[0] - 1
        dp, dk = torch.empty_like(p), torch.zeros_like(k)
        if p.numel():
            _lib.check(_lib.lib().dgg_csr_softk_bwd(_ptr(p), _ptr(pp), _ptr(rowptr), N, _ptr(k), _ptr(pos), int(perturb), mode, _ptr(_chk(g.contiguous())),
                                                    _ptr(dp), _ptr(dk), _stream()), "csr_softk_bwd")
        return dp, dk, None, None, None, None, None, None


def csr_rank_ramp_fwd(p, rowptr, col, w, b):
    """dgm.py:1791-1812 -> out [E], S [N], k [N], pos [E] (int32)"""
    p = _chk(p)
    N = rowptr.shape[0] - 1
    out = torch.empty_like(p)
    S = torch.empty((N,), device=p.device, dtype=torch.float32)
    k = torch.empty((N,), device=p.device, dtype=torch.float32)
    pos = torch.empty(p.shape, device=p.device, dtype=torch.int32)
    _lib.check(_lib.lib().dgg_csr_rank_ramp_fwd(_ptr(p), _ptr(rowptr), _ptr(col), N, _ptr(_chk(w)), _ptr(_chk(b)), _ptr(out), _ptr(S),
                                                _ptr(k), _ptr(pos), _stream()), "csr_rank_ramp_fwd")
    return out, S, k, pos


def csr_rank_ramp_bwd(p, rowptr, w, b, S, k, pos, g):
    dp = torch.empty_like(p)
    dkz = torch.empty_like(S)
    _lib.check(_lib.lib().dgg_csr_rank_ramp_bwd(_ptr(_chk(p)), _ptr(rowptr), S.shape[0], _ptr(_chk(w)), _ptr(_chk(b)), _ptr(S), _ptr(k),
                                                _ptr(pos), _ptr(_chk(g)), _ptr(dp), _ptr(dkz), _stream()), "csr_rank_ramp_bwd")
    return dp, dkz


def csr_noisy_sigmoid_fwd(p, noise):
    """dgm.py:1930-1933: sigmoid(p + noise) per stored edge"""
    out = torch.empty_like(p)
    _lib.check(_lib.lib().dgg_csr_noisy_sigmoid_fwd(_ptr(_chk(p)), _ptr(_chk(noise)), p.shape[0], _ptr(out), _stream()),
               "csr_noisy_sigmoid_fwd")
    return out


def csr_noisy_sigmoid_bwd(out, g):
    dp = torch.empty_like(out)
    _lib.check(_lib.lib().dgg_csr_noisy_sigmoid_bwd(_ptr(_chk(out)), _ptr(_chk(g)), out.shape[0], _ptr(dp), _stream()),
               "csr_noisy_sigmoid_bwd")
    return dp


def csr_rank_cut_fwd(p, rowptr, col, kcut):
    """dgm.py:1940-1942 -> out [E] (p on the kcut best entries of each row, 0 elsewhere), pos [E] (int32)"""
    p = _chk(p)
    out = torch.empty_like(p)
    pos = torch.empty(p.shape, device=p.device, dtype=torch.int32)
    _lib.check(_lib.lib().dgg_csr_rank_cut_fwd(_ptr(p), _ptr(rowptr), _ptr(col), rowptr.shape[0] - 1, int(kcut), _ptr(out), _ptr(pos),
                                               _stream()), "csr_rank_cut_fwd")
    return out, pos


def csr_rank_cut_bwd(pos, g, kcut):
    dp = torch.empty_like(g)
    _lib.check(_lib.lib().dgg_csr_rank_cut_bwd(_ptr(pos), _ptr(_chk(g)), g.shape[0], int(kcut), _ptr(dp), _stream()), "csr_rank_cut_bwd")
    return dp


# ---- dense all-pairs alternates (dgg_dense.hip) ------------------------------------------------------------------------
RAMP_SDD, RAMP_TOPK = 0, 1


def dense_rows_fwd(xq, t, temp, ramp, k=None, kfix=0, hs_start=2.0, interval=7.0, hard=False):
    """xq [B,N,h], t device scalar -> out, y [B,N,N], pos [B,N,N] int32 (dgm.py:273-346 / 157-176 + 83-98)"""
    xq = _chk(xq)
    B, N, h = xq.shape
    out = torch.empty((B, N, N), device=xq.device, dtype=torch.float32)
    y = torch.empty_like(out)
    pos = torch.empty((B, N, N), device=xq.device, dtype=torch.int32)
    _lib.check(_lib.lib().dgg_dense_rows_fwd(_ptr(xq), B, N, h, _ptr(_chk(t)), float(temp), int(ramp), _ptr(_chk(k)) if k is not None else None,
                                             int(kfix), float(hs_start), float(interval), int(bool(hard)), _ptr(out), _ptr(y), _ptr(pos),
                                             _stream()), "dense_rows_fwd")
    return out, y, pos


def dense_rows_bwd(xq, t, temp, ramp, k, hs_start, interval, y, pos, g):
    """-> Cm [B,N,N], dk [B,N] or None, dt_rows [B*N]"""
    xq = _chk(xq)
    B, N, h = xq.shape
    Cm = torch.empty((B, N, N), device=xq.device, dtype=torch.float32)
    dk = torch.empty((B, N), device=xq.device, dtype=torch.float32) if ramp == RAMP_SDD else None
    dt_rows = torch.empty((B * N,), device=xq.device, dtype=torch.float32)
    _lib.check(_lib.lib().dgg_dense_rows_bwd(_ptr(xq), B, N, h, _ptr(_chk(t)), float(temp), int(ramp), _ptr(_chk(k)) if k is not None else None,
                                             float(hs_start), float(interval), _ptr(y), _ptr(pos), _ptr(_chk(g)), _ptr(Cm),
                                             _ptr(dk) if dk is not None else None, _ptr(dt_rows), _stream()), "dense_rows_bwd")
    return Cm, dk, dt_rows


def dense_pairs_dx(xq, Cm):
    xq = _chk(xq)
    B, N, h = xq.shape
    dx = torch.empty_like(xq)
    _lib.check(_lib.lib().dgg_dense_pairs_dx(_ptr(xq), B, N, h, _ptr(Cm), _ptr(dx), _stream()), "dense_pairs_dx")
    return dx


def feat_softmax_fwd(z):
    z = _chk(z)
    out = torch.empty_like(z)
    _lib.check(_lib.lib().dgg_feat_softmax_fwd(_ptr(z), z.numel() // z.shape[-1], z.shape[-1], _ptr(out), _stream()), "feat_softmax_fwd")
    return out


def feat_softmax_bwd(out, g):
    dz = torch.empty_like(out)
    _lib.check(_lib.lib().dgg_feat_softmax_bwd(_ptr(_chk(out)), _ptr(_chk(g)), out.numel() // out.shape[-1], out.shape[-1], _ptr(dz),
                                               _stream()), "feat_softmax_bwd")
    return dz


class FeatSoftmaxFn(torch.autograd.Function):
    """nn.Softmax(dim=-1) over the last dimension"""

    @staticmethod
    def forward(ctx, z):
        out = feat_softmax_fwd(z)
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g):
        return feat_softmax_bwd(ctx.saved_tensors[0], g.contiguous())


class DenseRowsFn(torch.autograd.Function):
    """[B,N,h] features (+ k for the SDD ramp) -> dense [B,N,N] adjacency; backward to the features, t and k"""

    @staticmethod
    def forward(ctx, xq, t, k, temp, ramp, kfix, hs_start, interval, hard):
        xq = xq.contiguous()
        out, y, pos = dense_rows_fwd(xq, t, temp, ramp, k, kfix, hs_start, interval, hard)
        ctx.cfg = (temp, ramp, hs_start, interval)
        ctx.save_for_backward(xq, t, k if k is not None else t, y, pos)
        return out

    @staticmethod
    def backward(ctx, g):
        xq, t, k, y, pos = ctx.saved_tensors
        temp, ramp, hs_start, interval = ctx.cfg
        Cm, dk, dt_rows = dense_rows_bwd(xq, t, temp, ramp, k if ramp == RAMP_SDD else None, hs_start, interval, y, pos, g.contiguous())
        return dense_pairs_dx(xq, Cm), dt_rows.sum().reshape(t.shape), dk, None, None, None, None, None, None


def csr_uvdist_fwd(xp, rowptr, col, t=T_DIST):
    """dgm.py:1613-1627 on the stored entries: p_e = exp(t ||xp_u - xp_v||) -> [E]"""
    xp = _chk(xp)
    p = torch.empty((col.shape[0],), device=xp.device, dtype=torch.float32)
    _lib.check(_lib.lib().dgg_csr_uvdist_fwd(_ptr(xp), _ptr(rowptr), _ptr(col), xp.shape[0], xp.shape[1], float(t), _ptr(p), _stream()),
               "csr_uvdist_fwd")
    return p


def csr_uvdist_bwd(xp, rowptr, col, p, dp, t=T_DIST):
    xp = _chk(xp)
    dxp = _zeros(tuple(xp.shape), xp.device)
    _lib.check(_lib.lib().dgg_csr_uvdist_bwd(_ptr(xp), _ptr(rowptr), _ptr(col), xp.shape[0], xp.shape[1], float(t), _ptr(_chk(p)),
                                             _ptr(_chk(dp)), _ptr(dxp), _stream()), "csr_uvdist_bwd")
    return dxp


def csr_row_sum(vals, rowptr):
    N = rowptr.shape[0] - 1
    rs = torch.empty((N,), device=vals.device, dtype=torch.float32)
    _lib.check(_lib.lib().dgg_csr_row_sum(_ptr(_chk(vals)), _ptr(rowptr), N, _ptr(rs), _stream()), "csr_row_sum")
    return rs


def csr_normalize_fwd(rowptr, col, w, rs):
    ahat = torch.empty_like(w)
    _lib.check(_lib.lib().dgg_csr_normalize_fwd(_ptr(rowptr), _ptr(col), _ptr(_chk(w)), _ptr(_chk(rs)), rs.shape[0], _ptr(ahat),
                                                _stream()), "csr_normalize_fwd")
    return ahat


def csr_norm_bwd(rowptr, col, w, rs, dA):
    da = _zeros(tuple(rs.shape), rs.device)
    dw = torch.empty_like(w)
    _lib.check(_lib.lib().dgg_csr_norm_bwd(_ptr(rowptr), _ptr(col), _ptr(_chk(w)), _ptr(_chk(rs)), _ptr(_chk(dA)), rs.shape[0], _ptr(da),
                                           _ptr(dw), _stream()), "csr_norm_bwd")
    return dw


def csr_spmm_fwd(rowptr, col, a, X):
    X = _chk(X)
    N, F = rowptr.shape[0] - 1, X.shape[1]
    Y = torch.empty((N, F), device=X.device, dtype=torch.float32)
    _lib.check(_lib.lib().dgg_csr_spmm_fwd(_ptr(rowptr), _ptr(col), _ptr(_chk(a)), _ptr(X), N, F, _ptr(Y), _stream()), "csr_spmm_fwd")
    return Y


def csr_spmm_bwd(rowptr, col, a, X, dY, need_dx=True):
    X, dY = _chk(X), _chk(dY)
    N, F = rowptr.shape[0] - 1, X.shape[1]
    dA = torch.empty_like(a)
    dX = _zeros(tuple(X.shape), X.device) if need_dx else None
    _lib.check(_lib.lib().dgg_csr_spmm_bwd(_ptr(rowptr), _ptr(col), _ptr(_chk(a)), _ptr(X), _ptr(dY), N, F, _ptr(dA), _ptr(dX),
                                           _stream()), "csr_spmm_bwd")
    return dA, dX


def select_scores(scores, K=DEFAULT_K):
    scores = _chk(scores)
    R, N = scores.shape
    idx = torch.empty((R, K), device=scores.device, dtype=torch.int32)
    val = torch.empty((R, K), device=scores.device, dtype=torch.float32)
    _lib.check(_lib.lib().dgg_select_scores(_ptr(scores), R, N, K, _ptr(idx), _ptr(val), _stream()), "select_scores")
    return idx, val


def softk_fwd(idx, val, k, mode=MODE_K_TIMES_EDGE_PROB):
    N, K = idx.shape
    w = torch.empty((N, K), device=idx.device, dtype=torch.float32)
    rs = torch.empty((N,), device=idx.device, dtype=torch.float32)
    _lib.check(_lib.lib().dgg_softk_fwd(_ptr(idx), _ptr(_chk(val)), _ptr(_chk(k)), N, K, mode, _ptr(w), _ptr(rs), _stream()), "softk_fwd")
    return w, rs


def normalize_fwd(idx, w, rs, row0=0):
    N, K = idx.shape
    ahat = torch.empty((N, K), device=idx.device, dtype=torch.float32)
    _lib.check(_lib.lib().dgg_ell_normalize_fwd(_ptr(idx), _ptr(_chk(w)), _ptr(_chk(rs)), N, K, row0, _ptr(ahat), _stream()), "ell_normalize_fwd")
    return ahat


def spmm_fwd(idx, ahat, X, act=ACT_NONE, layout=None):
    """Y = act(A X); act = ACT_RELU is GCNConv's activation when the aggregation runs after the projection.
    layout (ChunkLayout): idx / ahat are the [chunks,64] arrays of chunked rows -> Y [layout.rows, F]"""
    N, K = idx.shape
    X = _chk(X)
    F = X.shape[1]
    ahat = _chk(ahat)
    pe = _probe_begin()
    if layout is not None and layout.wide:
        assert K == 64 and N == layout.chunks
        Y = torch.empty((layout.rows, F), device=idx.device, dtype=torch.float32)
        _lib.check(_lib.lib().dgg_ell_spmm_act_fwd_chunked(_ptr(idx), _ptr(ahat), _ptr(X), layout.rows, _ptr(layout.cptr), F, act, _ptr(Y), _stream()),
                   "ell_spmm_act_fwd_chunked")
    else:
        Y = torch.empty((N, F), device=idx.device, dtype=torch.float32)
        _lib.check(_lib.lib().dgg_ell_spmm_act_fwd(_ptr(idx), _ptr(ahat), _ptr(X), N, K, F, act, _ptr(Y), _stream()), "ell_spmm_act_fwd")
    _probe_end("spmm_fwd", pe)
    return Y


def act_bwd(y, dy, act):
    """dy * act'(y), elementwise"""
    y, dy = _chk(y), _chk(dy)
    dp = torch.empty_like(dy)
    _lib.check(_lib.lib().dgg_act_bwd(_ptr(y), _ptr(dy), y.numel(), act, _ptr(dp), _stream()), "act_bwd")
    return dp


CONV_BWD_WIDTHS = (16, 32, 64, 128)


def conv_bwd_cols(idx, ahat, H, G, part, rs=None, want_da=False):
    """Backward of Z = A H through the partition with ONE gathered row of G per active entry: -> dA [rows,K] (zero outside
    the partition), dH [ncols,F], da [ncols] (neighbour-side sums of the normalisation backward; None unless want_da).
    None when the kernel does not cover the shape."""
    N, K = idx.shape
    H, G = _chk(H), _chk(G)
    F = H.shape[1]
    if part is None or F not in CONV_BWD_WIDTHS or H.data_ptr() % 16 or G.data_ptr() % 16:
        return None
    ncols = H.shape[0]
    zz = _zeros((N * K + ncols * F + (ncols if want_da else 0),), H.device)      # one fill for the three accumulators
    dA = zz[:N * K].view(N, K)
    dH = zz[N * K:N * K + ncols * F].view(ncols, F)
    da = zz[N * K + ncols * F:] if want_da else None
    pe = _probe_begin()
    _lib.check(_lib.lib().dgg_ell_conv_bwd_part(_ptr(G), _ptr(H), _ptr(_chk(ahat)), N, K, F, _ptr(part), ncols,
                                                _ptr(_chk(rs)) if want_da else None, _ptr(dA), _ptr(dH), _ptr(da), _stream()),
               "ell_conv_bwd_part")
    _probe_end("conv_bwd", pe)
    return dA, dH, da


def spmm_bwd(idx, ahat, X, dY, need_dx=True, skip_zero=False, part=None, part_cols=None):
    """dA = <dY_i, X_j> and (optionally) dX = A^T dY.  With a partition of the pattern (part_build) and F a multiple of 64,
    dX goes through the destination-ordered records (runs reduced in registers) instead of entry-wise float atomics."""
    N, K = idx.shape
    X, dY = _chk(X), _chk(dY)
    F = X.shape[1]
    dA = torch.empty((N, K), device=idx.device, dtype=torch.float32)
    # the partition must have been built over the same destination set as X's rows (part_cols; default: a square block)
    if need_dx and part is not None and F % 64 == 0 and dY.data_ptr() % 16 == 0 and X.shape[0] == (N if part_cols is None else part_cols):
        ahat = _chk(ahat)
        _lib.check(_lib.lib().dgg_ell_spmm_bwd(_ptr(idx), _ptr(ahat), _ptr(X), _ptr(dY), N, K, F, int(skip_zero), _ptr(dA), _ptr(None),
                                               _stream()), "ell_spmm_bwd")
        dX = _zeros(tuple(X.shape), X.device)
        _lib.check(_lib.lib().dgg_ell_spmm_t_part(_ptr(ahat), _ptr(dY), N, K, F, _ptr(part), X.shape[0], _ptr(dX), _stream()),
                   "ell_spmm_t_part")
        return dA, dX
    dX = _zeros(tuple(X.shape), X.device) if need_dx else None
    _lib.check(_lib.lib().dgg_ell_spmm_bwd(_ptr(idx), _ptr(_chk(ahat)), _ptr(X), _ptr(dY), N, K, F, int(skip_zero), _ptr(dA), _ptr(dX), _stream()), "ell_spmm_bwd")
    return dA, dX


def sddmm_norm(idx, ahat, w, rs, X, dY, row0, part, skip_zero=True, cols=True):
    """dA = <dY_i, X_j> and da (phase 1 of the normalisation backward) in one pass over the rows; None when the fused
    kernel does not cover the shape (then: spmm_bwd + norm_bwd_da)."""
    N, K = idx.shape
    X, dY = _chk(X), _chk(dY)
    F = X.shape[1]
    if part is None or F not in (128, 256) or X.data_ptr() % 16 or dY.data_ptr() % 16:
        return None
    dA = torch.empty((N, K), device=idx.device, dtype=torch.float32)
    coef = torch.empty((N, K), device=idx.device, dtype=torch.float32)
    da = _zeros(tuple(rs.shape), rs.device)
    ahat, w, rs = _chk(ahat), _chk(w), _chk(rs)
    pe = _probe_begin()
    _lib.check(_lib.lib().dgg_ell_sddmm_norm_part(_ptr(idx), _ptr(ahat), _ptr(w), _ptr(rs), _ptr(X), _ptr(dY), N, K, F,
                                                  row0, int(skip_zero), _ptr(part), rs.shape[0], _ptr(coef), _ptr(dA), _ptr(da),
                                                  _stream()), "ell_sddmm_norm_part")
    _probe_end("spmm_bwd", pe)
    if cols:     # neighbour-side sums (cols=False: diagnostics that time the fused kernel alone)
        _lib.check(_lib.lib().dgg_norm_da_cols_part(_ptr(part), N, K, rs.shape[0], _ptr(coef), _ptr(da), _stream()), "norm_da_cols_part")
    return dA, da


def part_build(idx, w, ncols):
    """Bucket-partition of the active ELL entries by destination node (dgg_scatter.hip), built once per forward and
    reused by the column-side terms of the backward.  Returns None when the partitioned path does not apply."""
    N, K = idx.shape
    nbytes = int(_lib.lib().dgg_part_ws_bytes(N, K, ncols))
    if nbytes == 0:
        return None
    ws = torch.empty((nbytes,), device=idx.device, dtype=torch.uint8)
    _lib.check(_lib.lib().dgg_part_build(_ptr(idx), _ptr(_chk(w)), N, K, ncols, _ptr(ws), _stream()), "part_build")
    return ws


class PartP:
    """payload partition (dgg_partp_build): workspace + the shape it was built for"""

    def __init__(self, ws, rows, K, ncols, layout=None):
        self.ws, self.rows, self.K, self.ncols = ws, rows, K, ncols
        self.layout = layout                 # ChunkLayout when the block is the [chunks,64] arrays of chunked rows (rows = chunks)


def partp_records(partp):
    """(nodeptr int32 [ncols+1], records int32 [nrec,4] = (row*64 + r, j, bits of w rs_i^-1/2, bits of the score)) of a built payload
    partition, read back from its workspace (dgg_partp_describe); tests and tools only (one synchronisation)."""
    out = (C.c_int64 * 6)()
    _lib.check(_lib.lib().dgg_partp_describe(partp.rows, partp.K, partp.ncols, out), "partp_describe")
    ws = partp.ws
    nodeptr = ws[out[1]: out[1] + 4 * (partp.ncols + 1)].view(torch.int32)
    nrec = int(nodeptr[-1].item())
    recs = ws[out[2]: out[2] + 16 * nrec].view(torch.int32).reshape(nrec, 4)
    return nodeptr, recs


def partp_build(idx, w, val, rs_rows, ncols, rs_all=None, phase=0, layout=None):
    """Payload partition of the active ELL entries by destination node: records carry w * rs_i^-1/2 and the score, there is no
    slot map.  Returns None when it does not apply.  rs_all [ncols] (row sums of every node): normalize_adj is fused and the
    result is (partition, ahat [rows,K]).  phase=1: count + scan + fill only (ahat complete); finish with partp_sort(), possibly on
    another stream -- the sorted records are read by the backward's column kernels only.
    layout (ChunkLayout, wide): idx / w / val are [chunks,64], rs_rows [layout.rows] per NODE; the sorted records then carry the
    source node of their chunk in place of the destination (dgg_partp_build_chunked)."""
    N, K = idx.shape
    nbytes = int(_lib.lib().dgg_partp_ws_bytes(N, K, ncols))
    if nbytes == 0:
        return None
    ws = torch.empty((nbytes,), device=idx.device, dtype=torch.uint8)
    if layout is not None and layout.wide:
        assert K == 64 and N == layout.chunks and rs_rows.shape[0] == layout.rows
        ahat = torch.empty((N, K), device=idx.device, dtype=torch.float32) if rs_all is not None else None
        _lib.check(_lib.lib().dgg_partp_build_chunked(_ptr(idx), _ptr(_chk(w)), _ptr(_chk(val)), _ptr(_chk(rs_rows)), N, _ptr(layout.cnode), ncols,
                                                      _ptr(None if rs_all is None else _chk(rs_all)), _ptr(ahat), _ptr(ws), int(phase), _stream()),
                   "partp_build_chunked")
        part = PartP(ws, N, K, ncols, layout)
        part.args = (idx, w, val, rs_rows, rs_all, ahat) if phase == 1 else None
        return part if rs_all is None else (part, ahat)
    if rs_all is None:
        assert phase == 0
        _lib.check(_lib.lib().dgg_partp_build(_ptr(idx), _ptr(_chk(w)), _ptr(_chk(val)), _ptr(_chk(rs_rows)), N, K, ncols, _ptr(ws), _stream()),
                   "partp_build")
        return PartP(ws, N, K, ncols)
    ahat = torch.empty((N, K), device=idx.device, dtype=torch.float32)
    _lib.check(_lib.lib().dgg_partp_build_phase(_ptr(idx), _ptr(_chk(w)), _ptr(_chk(val)), _ptr(_chk(rs_rows)), N, K, ncols, _ptr(_chk(rs_all)),
                                                _ptr(ahat), _ptr(ws), int(phase), _stream()), "partp_build_norm")
    part = PartP(ws, N, K, ncols)
    part.args = (idx, w, val, rs_rows, rs_all, ahat) if phase == 1 else None
    return part, ahat


def partp_sort(part):
    """second part of partp_build(phase=1): the per-bucket sort, on the CURRENT stream"""
    idx, w, val, rs_rows, rs_all, ahat = part.args
    if part.layout is not None:
        _lib.check(_lib.lib().dgg_partp_build_chunked(_ptr(idx), _ptr(w), _ptr(val), _ptr(rs_rows), part.rows, _ptr(part.layout.cnode), part.ncols,
                                                      _ptr(rs_all), _ptr(ahat), _ptr(part.ws), 2, _stream()), "partp_sort")
        part.args = None
        return
    _lib.check(_lib.lib().dgg_partp_build_phase(_ptr(idx), _ptr(w), _ptr(val), _ptr(rs_rows), part.rows, part.K, part.ncols, _ptr(rs_all),
                                                _ptr(ahat), _ptr(part.ws), 2, _stream()), "partp_sort")
    part.args = None


def partp_gather(partp, dA):
    """row-major dA [rows,64] (chunked rows: [chunks,64]) -> the same values in the record order of the payload partition [rows*64]"""
    dA = _chk(dA)
    assert tuple(dA.shape) == (partp.rows, 64) and partp.K == 64
    dA_rec = torch.empty((partp.rows * 64,), device=dA.device, dtype=torch.float32)
    _lib.check(_lib.lib().dgg_partp_gather_rec(_ptr(dA), partp.rows, partp.ncols, _ptr(partp.ws), _ptr(dA_rec), _stream()), "partp_gather_rec")
    return dA_rec


def conv_bwd_cols_p(idx, H, G, partp, rs, zero_dA=True, dA_ext=None, want_dA=True):
    """conv_bwd_cols on a payload partition -> dA [rows,K], dA_rec [rows*K], dH [ncols,F], da [ncols] (neighbour side); None when
    the kernel does not cover the shape.  zero_dA=False: entries outside the partition are left UNINITIALISED -- for a consumer that
    masks them itself (softk_edge_bwd_p with ahat_rows does).  dA_ext [rows,K] (optional): cotangent of the normalised adjacency from its
    other consumers, added per record (dA / dA_rec / da then hold the totals).  want_dA=False: the row-major dA is not written (returned
    as None): softk_edge_bwd_p, given dA=None, reads dA_rec through the partition's slot -> record map"""
    N, K = idx.shape
    H, G = _chk(H), _chk(G)
    F = H.shape[1]
    if partp is None or F not in CONV_BWD_WIDTHS or H.data_ptr() % 16 or G.data_ptr() % 16 or H.shape[0] != partp.ncols:
        return None
    ncols = H.shape[0]
    dA = None
    if want_dA:
        dA = _zeros((N, K), H.device) if zero_dA else torch.empty((N, K), device=H.device, dtype=torch.float32)   # entries outside the partition stay 0
    # one wavefront per destination node owns dH_j / da_j: plain stores for every node, no zero fill
    # (an EMPTY row shard launches nothing: the outputs the other ranks reduce must then be zeros, not uninitialised memory)
    alloc = torch.zeros if N == 0 else torch.empty
    dH = alloc((ncols, F), device=H.device, dtype=torch.float32)
    da = alloc((ncols,), device=H.device, dtype=torch.float32)
    dA_rec = torch.empty((N * K,), device=H.device, dtype=torch.float32)
    pe = _probe_begin()
    if dA_ext is not None:
        dA_ext = _chk(dA_ext.contiguous())
        assert tuple(dA_ext.shape) == (N, K)
    if partp.layout is not None:                 # chunked rows: G [layout.rows, F] is read at the source node of a record's chunk
        assert want_dA and G.shape[0] == partp.layout.rows
        _lib.check(_lib.lib().dgg_ell_conv_bwd_partp_chunked(_ptr(G), _ptr(H), N, F, _ptr(partp.ws), ncols, _ptr(_chk(rs)), _ptr(dA_ext), _ptr(dA),
                                                             _ptr(dA_rec), _ptr(dH), _ptr(da), _stream()), "ell_conv_bwd_partp_chunked")
        _probe_end("conv_bwd", pe)
        return dA, dA_rec, dH, da
    _lib.check(_lib.lib().dgg_ell_conv_bwd_partp_ext(_ptr(G), _ptr(H), N, K, F, _ptr(partp.ws), ncols, _ptr(_chk(rs)), _ptr(dA_ext), _ptr(dA),
                                                     _ptr(dA_rec), _ptr(dH), _ptr(da), _stream()), "ell_conv_bwd_partp")
    _probe_end("conv_bwd", pe)
    return dA, dA_rec, dH, da


def softk_edge_bwd_p(xp, idx, val, k, dA, dA_rec, rs, da, row0, t, perturb, mode, normalized, partp, ahat_rows=None, out_act=ACT_NONE,
                     phase=0, state=None):
    """softk_edge_bwd on a payload partition -> dxp [Nglobal,h], dk [N]; None when it does not apply.  out_act=ACT_LEAKY (mode 0):
    dxp comes back multiplied by LeakyReLU'(xp), the gradient of the pre-activation of the layer that produced xp.
    phase=1: the row kernel only -> (dxp, dk, state): dk is complete (the k-net backward can start, e.g. on another stream);
    phase=2 with that `state`: the per-destination kernel completes dxp.  dA=None (normalised form with ahat_rows): the row kernel
    takes an entry's cotangent from dA_rec through the partition's slot -> record map (conv_bwd_cols_p(want_dA=False))."""
    xp = _chk(xp)
    Ng, h = xp.shape
    N, K = idx.shape
    if partp is None or h not in (16, 32, 64, 128) or mode not in (MODE_K_TIMES_EDGE_PROB, MODE_K_ONLY) or Ng != partp.ncols:
        return None
    lay = partp.layout
    nrows = N if lay is None else lay.rows
    if phase == 2:
        dxp, dk, rowinfo = state
    else:
        # mode 0: every row of dxp is written (own rows by the row kernel, the others by the per-node kernel): no zero fill
        dxp = torch.empty_like(xp) if (mode == MODE_K_TIMES_EDGE_PROB and N > 0) else _zeros(tuple(xp.shape), xp.device)   # (N == 0: see conv_bwd_cols_p)
        rowinfo = torch.empty((N, 4), device=xp.device, dtype=torch.float32)
        dk = torch.empty((nrows,), device=xp.device, dtype=torch.float32)
    pe = _probe_begin()
    if lay is not None:                          # chunked rows: one wavefront per node walks its chunks; rowinfo per chunk
        assert dA is not None and K == 64
        _lib.check(_lib.lib().dgg_softk_edge_bwd_partp_chunked(_ptr(xp), nrows, _ptr(lay.cptr), N, h, _ptr(idx), _ptr(_chk(val)), _ptr(_chk(k)), _ptr(rs),
                                                               _ptr(_chk(dA)), _ptr(dA_rec), _ptr(da), _ptr(None if ahat_rows is None else _chk(ahat_rows)),
                                                               row0, t, int(perturb), mode, int(normalized), _ptr(partp.ws), Ng, _ptr(rowinfo), _ptr(dk),
                                                               _ptr(dxp), int(out_act), int(phase), _stream()), "softk_edge_bwd_partp_chunked")
        _probe_end("edge_bwd" if phase == 0 else ("edge_bwd_rows" if phase == 1 else "edge_bwd_node"), pe)
        if phase == 1:
            return dxp, dk, (dxp, dk, rowinfo)
        return dxp, dk
    _lib.check(_lib.lib().dgg_softk_edge_bwd_partp_phase(_ptr(xp), N, h, _ptr(idx), _ptr(_chk(val)), _ptr(_chk(k)), _ptr(rs), _ptr(None if dA is None else _chk(dA)),
                                                         _ptr(dA_rec), _ptr(da), _ptr(None if ahat_rows is None else _chk(ahat_rows)), K, row0, t,
                                                         int(perturb), mode, int(normalized), _ptr(partp.ws), Ng, _ptr(rowinfo), _ptr(dk),
                                                         _ptr(dxp), int(out_act), int(phase), _stream()), "softk_edge_bwd_partp")
    _probe_end("edge_bwd" if phase == 0 else ("edge_bwd_rows" if phase == 1 else "edge_bwd_node"), pe)
    if phase == 1:
        return dxp, dk, (dxp, dk, rowinfo)
    return dxp, dk


def norm_bwd_da(idx, w, rs, dA, row0=0, part=None):
    N, K = idx.shape
    da = _zeros(tuple(rs.shape), rs.device)
    if part is not None:
        coef = torch.empty((N, K), device=idx.device, dtype=torch.float32)
        _lib.check(_lib.lib().dgg_norm_bwd_da_part(_ptr(idx), _ptr(_chk(w)), _ptr(_chk(rs)), _ptr(_chk(dA)), N, K, row0, _ptr(part),
                                                   rs.shape[0], _ptr(coef), _ptr(da), _stream()), "norm_bwd_da_part")
        return da
    _lib.check(_lib.lib().dgg_norm_bwd_da(_ptr(idx), _ptr(_chk(w)), _ptr(_chk(rs)), _ptr(_chk(dA)), N, K, row0, _ptr(da), _stream()), "norm_bwd_da")
    return da


def softk_bwd(idx, val, k, dA, rs=None, da=None, row0=0, mode=MODE_K_TIMES_EDGE_PROB, normalized=False, ahat_rows=None):
    """ahat_rows [N,K] (normalized only): `da` holds the neighbour-side sums (conv_bwd_cols_p); the row side is formed inside"""
    N, K = idx.shape
    dval = torch.empty((N, K), device=idx.device, dtype=torch.float32)
    dk = torch.empty((N,), device=idx.device, dtype=torch.float32) if mode != 2 else None
    val = _chk(val) if val is not None else None
    k = _chk(k) if k is not None else None
    if ahat_rows is not None:
        assert normalized
        _lib.check(_lib.lib().dgg_softk_bwd_rows(_ptr(idx), _ptr(val), _ptr(k), _ptr(_chk(rs)), _ptr(_chk(dA)), _ptr(_chk(da)), _ptr(_chk(ahat_rows)),
                                                 N, K, row0, mode, _ptr(dval), _ptr(dk), _stream()), "softk_bwd_rows")
        return dval, dk
    _lib.check(_lib.lib().dgg_softk_bwd(_ptr(idx), _ptr(val), _ptr(k), _ptr(rs), _ptr(_chk(dA)), _ptr(da), N, K, row0, mode,
                                        int(normalized), _ptr(dval), _ptr(dk), _stream()), "softk_bwd")
    return dval, dk


def softk_edge_bwd(xp, idx, val, k, dA, rs=None, da=None, row0=0, t=T_DIST, perturb=False, mode=MODE_K_TIMES_EDGE_PROB, normalized=False,
                   part=None, want_dval=False, ahat_rows=None):
    """softk_bwd + edge_bwd in one call when the destination-ordered path applies (d loss / d score never leaves the registers
    of the row kernel) -> dxp [Nglobal,h], dk [N], dval [N,K] or None; None when it does not apply (then: the two calls)."""
    xp = _chk(xp)
    Ng, h = xp.shape
    N, K = idx.shape
    if part is None or h not in (16, 32, 64, 128) or mode not in (MODE_K_TIMES_EDGE_PROB, MODE_K_ONLY):
        return None
    dxp = _zeros(tuple(xp.shape), xp.device)
    coef = torch.empty(N * K + Ng, device=xp.device, dtype=torch.float32)
    dk = torch.empty((N,), device=xp.device, dtype=torch.float32)
    dval = torch.empty((N, K), device=xp.device, dtype=torch.float32) if want_dval else None
    val, k, dA = _chk(val), _chk(k), _chk(dA)
    pe = _probe_begin()
    # ahat_rows: `da` holds the neighbour-side sums only (conv_bwd_cols); the row side is formed inside the kernel
    _lib.check(_lib.lib().dgg_softk_edge_bwd_part(_ptr(xp), N, h, _ptr(idx), _ptr(val), _ptr(k), _ptr(rs), _ptr(dA), _ptr(da),
                                                  _ptr(None if ahat_rows is None else _chk(ahat_rows)), K, row0, t,
                                                  int(perturb), mode, int(normalized), _ptr(part), Ng, _ptr(coef), _ptr(dval), _ptr(dk),
                                                  _ptr(dxp), _stream()), "softk_edge_bwd_part")
    _probe_end("edge_bwd", pe)
    return dxp, dk, dval


_ONES64 = {}


def _ones64(n, device):
    key = (n, str(device))
    if key not in _ONES64:
        _ONES64.clear()
        _ONES64[key] = torch.ones((n, 64), device=device, dtype=torch.float32)
    return _ONES64[key]


WIDE_EDGE_BWD_PART = __import__("os").environ.get("DGG_WIDE_EDGE_BWD_PART", "1") != "0"
EMLP_BWD_PARTP = __import__("os").environ.get("DGG_EMLP_BWD_PARTP", "1") != "0"   # edge-MLP backward: neighbour sums through the payload partition
WIDE_EDGE_BWD_SLICED = __import__("os").environ.get("DGG_WIDE_EDGE_BWD_SLICED", "1") != "0"   # its row pass one 256-feature slice at a time


def edge_bwd(xp, idx, val, dval, row0=0, t=T_DIST, perturb=False, part=None):
    xp = _chk(xp)
    Ng, h = xp.shape
    N, K = idx.shape
    dxp = _zeros(tuple(xp.shape), xp.device)
    if part is not None and h in (16, 32, 64, 128):
        coef = torch.empty(N * K + Ng, device=xp.device, dtype=torch.float32)   # per-record coefficients + column sums
        val, dval = _chk(val), _chk(dval)
        pe = _probe_begin()
        _lib.check(_lib.lib().dgg_edge_bwd_part(_ptr(xp), N, h, _ptr(idx), _ptr(val), _ptr(dval), K, row0, t, int(perturb),
                                                _ptr(part), Ng, _ptr(coef), _ptr(dxp), _stream()), "edge_bwd_part")
        _probe_end("edge_bwd", pe)
        return dxp
    if part is not None and WIDE_EDGE_BWD_PART and 128 < h <= 2048 and h % 64 == 0 and row0 == 0 and N == Ng:
        # wide latents (PPI: 2048) without float atomics: row pass (own side stored, coefficients dd kept), then the neighbour side as a
        # transposed aggregation of xp with the coefficients through the destination-ordered partition
        own = torch.empty((N, h), device=xp.device, dtype=torch.float32)
        dd = torch.empty((N, K), device=xp.device, dtype=torch.float32)
        if h % 256 == 0 and WIDE_EDGE_BWD_SLICED:               # one 256-feature slice of the gathered rows at a time (L2-resident)
            ws = torch.empty((int(_lib.lib().dgg_edge_bwd_wide_rows_ws_floats(N, K, h)),), device=xp.device, dtype=torch.float32)
            _lib.check(_lib.lib().dgg_edge_bwd_wide_rows_sliced(_ptr(xp), N, h, _ptr(idx), _ptr(_chk(val)), _ptr(_chk(dval)), K, row0, t, int(perturb),
                                                                _ptr(ws), _ptr(own), _ptr(dd), _stream()), "edge_bwd_wide_rows_sliced")
        else:
            _lib.check(_lib.lib().dgg_edge_bwd_wide_rows(_ptr(xp), N, h, _ptr(idx), _ptr(_chk(val)), _ptr(_chk(dval)), K, row0, t, int(perturb),
                                                         _ptr(own), _ptr(dd), _stream()), "edge_bwd_wide_rows")
        _lib.check(_lib.lib().dgg_ell_spmm_t_part(_ptr(dd), _ptr(xp), N, K, h, _ptr(part), Ng, _ptr(dxp), _stream()), "ell_spmm_t_part")
        # column sums of the coefficients through the same partition (a 64-wide aggregation of ones: torch's index_add_ took 1.1 ms here)
        cs64 = _zeros((Ng, 64), xp.device)
        _lib.check(_lib.lib().dgg_ell_spmm_t_part(_ptr(dd), _ptr(_ones64(N, xp.device)), N, K, 64, _ptr(part), Ng, _ptr(cs64), _stream()),
                   "ell_spmm_t_part")
        return torch.addcmul(own.sub_(dxp), cs64[:, :1], xp)
    _lib.check(_lib.lib().dgg_edge_bwd(_ptr(xp), N, h, _ptr(idx), _ptr(_chk(val)), _ptr(_chk(dval)), K, row0, t, int(perturb), _ptr(dxp),
                                       _stream()), "edge_bwd")
    return dxp


# latent dims from here on run the k-net as MFMA GEMMs (bit-identical to the thread-per-node kernels, which hold a node's
# whole row in registers and slow down steeply beyond 64 features: 0.57 ms vs 0.2 ms per step at h = 128, N = 100k)
KNET_WIDE_FROM = int(__import__("os").environ.get("DGG_KNET_WIDE_FROM", "128"))


def _lin_bf16(x, W, b, act):
    """act(x W^T + b) with the product on the bf16 matrix cores (operands rounded to bf16, fp32 accumulation); W [out, d] as stored"""
    y = gemm_nt_bf16(pack_bf16(x), pack_bf16(W))
    if b is not None:
        y.add_(b)
    if act == ACT_LEAKY:
        y = torch.where(y > 0, y, 0.01 * y)
    return y


def _knet_wide_fwd(xk, deg, mu_sd, W1, b1, Wmu, bmu, Wp, bp, bf16=False):
    """wide latents: the three layers as MFMA GEMMs on feat = [xk | nd] (same fmaf chains as the fused kernels).  bf16 (the module's
    `gemm_dtype = torch.bfloat16`, BASELINE configs[4]): the two wide products -- k_embed 2049 -> 1024 and k_mu 1024 -> 512 at latent
    2048, 12 GFLOP per PPI graph and three times that with the backward -- on the bf16 matrix cores instead of the fp32 ones."""
    N, h = xk.shape
    feat = torch.empty((N, h + 1), device=xk.device, dtype=torch.float32)
    _lib.check(_lib.lib().dgg_knet_feat(_ptr(xk), _ptr(_chk(deg)), _ptr(mu_sd), N, h, _ptr(feat), _stream()), "knet_feat")
    if bf16:
        z = _lin_bf16(feat, W1, b1, ACT_LEAKY)
        m = _lin_bf16(z, Wmu, bmu, ACT_NONE)
    else:
        z = linear_fwd(feat, W1, b1, ACT_LEAKY)
        m = linear_fwd(z, Wmu, bmu, ACT_NONE)
    kp = linear_fwd(m, Wp.reshape(1, -1), bp, ACT_NONE)
    k = torch.empty((N,), device=xk.device, dtype=torch.float32)
    u = torch.empty((N,), device=xk.device, dtype=torch.float32)
    _lib.check(_lib.lib().dgg_knet_out_fwd(_ptr(kp), _ptr(mu_sd), N, _ptr(k), _ptr(u), _stream()), "knet_out_fwd")
    return k, z, u, feat


def _lin_bf16_bwd(x, W, dy):
    """backward of y = x W^T + b on the bf16 matrix cores -> dx = dy W, dW = dy^T x, db"""
    dyb = pack_bf16(dy)
    dx = gemm_nt_bf16(dyb, pack_bf16(W, transpose=True))                          # [N,out] x [d,out]^T
    dW = gemm_nt_bf16(pack_bf16(dy, transpose=True), pack_bf16(x, transpose=True))  # [out,N] x [d,N]^T
    return dx, dW, dy.sum(0)


def _knet_wide_bwd(h, mu_sd, W1, Wmu, bmu, Wp, z, u, feat, dk, bf16=False):
    N = z.shape[0]
    dkp = torch.empty((N, 1), device=z.device, dtype=torch.float32)
    _lib.check(_lib.lib().dgg_knet_out_bwd(_ptr(u), _ptr(_chk(dk)), _ptr(mu_sd), N, _ptr(dkp), _stream()), "knet_out_bwd")
    m = _lin_bf16(z, Wmu, bmu, ACT_NONE) if bf16 else linear_fwd(z, Wmu, bmu, ACT_NONE)       # recomputed (k_project weight gradient)
    Wp2 = Wp.reshape(1, -1)
    dm, dWp, dbp = linear_bwd(m, Wp2, None, dkp, ACT_NONE)
    if bf16:
        dz, dWmu, dbmu = _lin_bf16_bwd(z, Wmu, dm)
        dfeat, dW1, db1 = _lin_bf16_bwd(feat, W1, torch.where(z > 0, dz, 0.01 * dz))
    else:
        dz, dWmu, dbmu = linear_bwd(z, Wmu, None, dm, ACT_NONE)
        dfeat, dW1, db1 = linear_bwd(feat, W1, z, dz, ACT_LEAKY)
    return dfeat[:, :h].contiguous(), dW1, db1, dWmu, dbmu, dWp, dbp


def knet_x_fwd(xk, deg, mu_sd, W1, b1, Wmu, bmu, Wp, bp, save=True, bf16=False):
    xk = _chk(xk)
    N, h = xk.shape
    h2, h4 = W1.shape[0], Wmu.shape[0]
    if h >= KNET_WIDE_FROM:
        return _knet_wide_fwd(xk, deg, mu_sd, W1, b1, Wmu, bmu, Wp, bp, bf16=bf16 and h % 64 == 0)
    dev = xk.device
    k = torch.empty((N,), device=dev, dtype=torch.float32)
    z = torch.empty((N, h2), device=dev, dtype=torch.float32) if save else None
    u = torch.empty((N,), device=dev, dtype=torch.float32) if save else None
    feat = torch.empty((N, h + 1), device=dev, dtype=torch.float32) if save else None
    _lib.check(_lib.lib().dgg_knet_x_fwd(_ptr(xk), N, h, _ptr(_chk(deg)), _ptr(mu_sd), _ptr(_chk(W1)), _ptr(_chk(b1)), h2, _ptr(_chk(Wmu)),
                                         _ptr(_chk(bmu)), h4, _ptr(_chk(Wp)), _ptr(_chk(bp)), _ptr(k), _ptr(z), _ptr(u), _ptr(feat), _stream()),
               "knet_x_fwd")
    return k, z, u, feat


KNET_MFMA_WIDTHS = (16, 32, 64)
DA_MAP = True       # conv_bwd_cols_p(want_dA=False) + softk_edge_bwd_p(dA=None): see ShardedDGGConv._backward


def partp_has_map(rows):
    """the payload partition of a block of `rows` rows carries the slot -> record map (small blocks; DGG_DA_MAP=0/1 forces)"""
    return bool(_lib.lib().dgg_partp_has_map(int(rows)))
PREMASK = True      # softk_edge_bwd_p / knet_x_bwd_fused can return gradients of the PRE-activation (out_act): see ShardedDGGConv._premask


def knet_x_fwd_slim(xk, deg, mu_sd, W1, b1, Wmu, bmu, Wp, bp):
    """k-net forward on the matrix cores, saving only u (the fused backward re-runs layer 1): -> k [N], u [N]; same bits as knet_x_fwd"""
    xk = _chk(xk)
    N, h = xk.shape
    assert h in KNET_MFMA_WIDTHS and W1.shape[0] * 2 == h and Wmu.shape[0] * 4 == h
    k = torch.empty((N,), device=xk.device, dtype=torch.float32)
    u = torch.empty((N,), device=xk.device, dtype=torch.float32)
    _lib.check(_lib.lib().dgg_knet_x_fwd_mfma(_ptr(xk), N, h, _ptr(_chk(deg)), _ptr(mu_sd), _ptr(_chk(W1)), _ptr(_chk(b1)), _ptr(_chk(Wmu)),
                                              _ptr(_chk(bmu)), _ptr(_chk(Wp)), _ptr(_chk(bp)), _ptr(k), _ptr(u), _stream()), "knet_x_fwd_mfma")
    return k, u


def knet_x_bwd_fused(xk, deg, mu_sd, W1, b1, Wmu, bmu, Wp, u, dk, out_act=ACT_NONE):
    """one pass over xk -> dxk [N,h], dW1, db1, dWmu, dbmu, dWp ([h4]), dbp ([1])  (dgg_knet_x_bwd_reg: per-workgroup slabs + one
    reduce launch, nothing accumulated into).  out_act=ACT_LEAKY: dxk comes back multiplied by LeakyReLU'(xk)"""
    xk = _chk(xk)
    N, h = xk.shape
    h2, h4 = W1.shape[0], Wmu.shape[0]
    assert h in KNET_MFMA_WIDTHS and h2 * 2 == h and h4 * 4 == h
    dev = xk.device
    dxk = torch.empty((N, h), device=dev, dtype=torch.float32)
    out = torch.empty((h2 * (h + 1) + h2 + h4 * h2 + 2 * h4 + 1,), device=dev, dtype=torch.float32)
    o = 0
    gW1 = out[o:o + h2 * (h + 1)].view(h2, h + 1); o += h2 * (h + 1)
    gb1 = out[o:o + h2]; o += h2
    gWmu = out[o:o + h4 * h2].view(h4, h2); o += h4 * h2
    gbmu = out[o:o + h4]; o += h4
    gWp = out[o:o + h4]; o += h4
    gbp = out[o:o + 1]
    ws = torch.empty((int(_lib.lib().dgg_knet_x_bwd_ws_bytes(N, h)),), device=dev, dtype=torch.uint8)
    _lib.check(_lib.lib().dgg_knet_x_bwd_reg(_ptr(xk), N, h, _ptr(_chk(deg)), _ptr(mu_sd), _ptr(_chk(W1)), _ptr(_chk(b1)), _ptr(_chk(Wmu)),
                                             _ptr(_chk(bmu)), _ptr(_chk(Wp)), _ptr(_chk(u)), _ptr(_chk(dk)), _ptr(dxk), _ptr(gW1), _ptr(gb1),
                                             _ptr(gWmu), _ptr(gbmu), _ptr(gWp), _ptr(gbp), int(out_act), _ptr(ws), _stream()), "knet_x_bwd_reg")
    return dxk, gW1, gb1, gWmu, gbmu, gWp, gbp


def knet_x_bwd(h, mu_sd, W1, Wmu, bmu, Wp, z, u, feat, dk, bf16=False):
    """-> dxk [N,h], dW1, db1, dWmu, dbmu, dWp ([1,h4]), dbp ([1])"""
    N = z.shape[0]
    h2, h4 = W1.shape[0], Wmu.shape[0]
    if h >= KNET_WIDE_FROM:
        return _knet_wide_bwd(h, mu_sd, W1, Wmu, bmu, Wp, z, u, feat, dk, bf16=bf16 and h % 64 == 0)
    dev = z.device
    dkp = torch.empty((N, 1), device=dev, dtype=torch.float32)
    dm = torch.empty((N, h4), device=dev, dtype=torch.float32)
    dpre1 = torch.empty((N, h2), device=dev, dtype=torch.float32)
    dxk = torch.empty((N, h), device=dev, dtype=torch.float32)
    m = torch.empty((N, h4), device=dev, dtype=torch.float32)
    _lib.check(_lib.lib().dgg_knet_x_bwd_nodes(N, h, _ptr(mu_sd), _ptr(_chk(W1)), h2, _ptr(_chk(Wmu)), h4, _ptr(_chk(Wp)), _ptr(_chk(bmu)),
                                               _ptr(z), _ptr(u), _ptr(_chk(dk)), _ptr(dkp), _ptr(dm), _ptr(dpre1), _ptr(dxk), _ptr(m),
                                               _stream()), "knet_x_bwd_nodes")
    (dW1, db1), (dWmu, dbmu), (dWp, dbp) = gemm_tn_pairs([(dpre1, feat), (dm, z), (dkp, m)])    # three tiny products, one launch
    return dxk, dW1, db1, dWmu, dbmu, dWp, dbp


def knet_input_deg_fwd(deg, dmean, dstd, Wd, bd, Wmu, bmu, Wp, bp):
    deg = _chk(deg)
    N = deg.shape[0]
    k = torch.empty((N,), device=deg.device, dtype=torch.float32)
    _lib.check(_lib.lib().dgg_knet_input_deg_fwd(_ptr(deg), N, float(dmean), float(dstd), _ptr(_chk(Wd)), _ptr(_chk(bd)), _ptr(_chk(Wmu)),
                                                 _ptr(_chk(bmu)), Wmu.shape[0], _ptr(_chk(Wp)), _ptr(_chk(bp)), _ptr(k), _stream()),
               "knet_input_deg_fwd")
    return k


def knet_deg_fwd(deg, mu_sd, dmean, dstd, eps, Wd, bd, Wmu, bmu, Wp, bp):
    """degree-only k-net (dgm.py:1492-1526) -> k [N], u [N] (pre-relu).  mu_sd: device [2] or None (constants)."""
    deg = _chk(deg)
    N = deg.shape[0]
    k = torch.empty((N,), device=deg.device, dtype=torch.float32)
    u = torch.empty((N,), device=deg.device, dtype=torch.float32)
    _lib.check(_lib.lib().dgg_knet_deg_fwd(_ptr(deg), N, _ptr(mu_sd), float(dmean), float(dstd), float(eps), _ptr(_chk(Wd)),
                                           _ptr(_chk(bd)), _ptr(_chk(Wmu)), _ptr(_chk(bmu)), Wmu.shape[0], _ptr(_chk(Wp)),
                                           _ptr(_chk(bp)), _ptr(k), _ptr(u), _stream()), "knet_deg_fwd")
    return k, u


def knet_deg_bwd_sums(deg, mu_sd, dmean, dstd, eps, u, dk):
    """-> S [2] = (sum dkp, sum dkp * nd)"""
    S = torch.zeros((2,), device=deg.device, dtype=torch.float32)
    _lib.check(_lib.lib().dgg_knet_deg_bwd_sums(_ptr(_chk(deg)), deg.shape[0], _ptr(mu_sd), float(dmean), float(dstd), float(eps),
                                                _ptr(_chk(u)), _ptr(_chk(dk)), _ptr(S), _stream()), "knet_deg_bwd_sums")
    return S


# ------------------------------------------------------------------------------------------------------------
# autograd ops
# ------------------------------------------------------------------------------------------------------------
class LinearFn(torch.autograd.Function):
    """act(x W^T + b) on the fp32 matrix cores (nn.Linear+LeakyReLU dgm.py:1097-1100; GCNConv mm+relu model.py:596-598)."""

    @staticmethod
    def forward(ctx, x, W, b, act, w_layout):
        y = linear_fwd(x, W, b, act, w_layout)
        ctx.save_for_backward(x, W, y)
        ctx.act, ctx.w_layout, ctx.has_b = act, w_layout, b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W, y = ctx.saved_tensors
        dx, dW, db = linear_bwd(x, W, y, dy, ctx.act, ctx.w_layout, need_dx=ctx.needs_input_grad[0], need_db=ctx.has_b)
        return dx, dW, db, None, None


class EllNormalizeFn(torch.autograd.Function):
    """D^-1/2 A D^-1/2 with row sums on both sides (normalize_adj, model.py:1205-1219) on ELL values."""

    @staticmethod
    def forward(ctx, w, idx, rs, part=None):
        ahat = normalize_fwd(idx, w, rs)
        ctx.save_for_backward(w, idx, rs)
        ctx.part = part if (part is not None and idx.shape[0] == rs.shape[0]) else None
        return ahat

    @staticmethod
    def backward(ctx, dA):
        w, idx, rs = ctx.saved_tensors
        dA = dA.contiguous()
        da = norm_bwd_da(idx, w, rs, dA, part=ctx.part)          # neighbour-side sums through the partition when there is one
        # mode 2 = no ramp: dw = dA a_i a_j - 0.5 da_i a_i / rs_i
        dw, _ = softk_bwd(idx, None, None, dA, rs=rs, da=da, mode=2, normalized=True)
        return dw, None, None, None


class EllSpmmFn(torch.autograd.Function):
    """Y = act(A X) on the ELL adjacency (torch.mm(adj, x) model.py:594, 67 / torch.spmm model.py:34; act = ReLU when GCNConv
    aggregates the projected features, relu(A (x W)))."""

    @staticmethod
    def forward(ctx, ahat, idx, X, skip_zero=False, part=None, act=ACT_NONE, partp=None, layout=None):
        # layout (ChunkLayout): ahat / idx are the [chunks,64] arrays of chunked rows (rows wider than 64 ranks), Y [layout.rows, F]
        Y = spmm_fwd(idx, ahat, X, act, layout=layout)
        ctx.save_for_backward(ahat, idx, X, Y if act != ACT_NONE else X)
        ctx.skip_zero, ctx.part, ctx.act, ctx.partp, ctx.layout = skip_zero, part, act, partp, layout
        return Y

    @staticmethod
    def backward(ctx, dY):
        ahat, idx, X, Y = ctx.saved_tensors
        dY = dY.contiguous()
        if ctx.act != ACT_NONE:
            dY = act_bwd(Y, dY, ctx.act)
        lay = ctx.layout
        if lay is not None and lay.wide:
            if ctx.skip_zero and ctx.partp is not None and X.shape[0] == lay.rows:
                got = conv_bwd_cols_p(idx, X, dY, ctx.partp[0], ctx.partp[1], zero_dA=True)       # (chunk-aware through the partition's layout)
                if got is not None:
                    return got[0], None, (got[2] if ctx.needs_input_grad[2] else None), None, None, None, None, None
            # every chunk as a row of its own against the cotangent row of its node
            dYc = dY.index_select(0, lay.cnode.long()[:idx.shape[0]])
            dA, dX = spmm_bwd(idx, ahat, X, dYc, need_dx=ctx.needs_input_grad[2], skip_zero=ctx.skip_zero)
            return dA, None, dX, None, None, None, None, None
        if ctx.needs_input_grad[2] and ctx.skip_zero and ctx.partp is not None and X.shape[0] == idx.shape[0]:
            # the adjacency of the fused layer, read by a later layer: the same per-destination kernel as the layer's own aggregation
            # backward (the records carry the normalised values): dA by plain stores, dX owned by the destination's wavefront
            got = conv_bwd_cols_p(idx, X, dY, ctx.partp[0], ctx.partp[1], zero_dA=True)
            if got is not None:
                return got[0], None, got[2], None, None, None, None, None
        if ctx.needs_input_grad[2] and ctx.skip_zero and X.shape[0] == idx.shape[0]:
            # learned input on a DGG adjacency: SDDMM and transposed SpMM from ONE gathered cotangent row per entry
            got = conv_bwd_cols(idx, ahat, X, dY, ctx.part)
            if got is not None:
                return got[0], None, got[1], None, None, None, None, None
        dA, dX = spmm_bwd(idx, ahat, X, dY, need_dx=ctx.needs_input_grad[2], skip_zero=ctx.skip_zero, part=ctx.part)
        return dA, None, dX, None, None, None, None, None


class CsrNormalizeFn(torch.autograd.Function):
    """D^-1/2 A D^-1/2 with row sums on both sides (normalize_adj, model.py:1340-1352) on CSR values; the row sums are
    part of the graph, as in the reference."""

    @staticmethod
    def forward(ctx, w, rowptr, col):
        rs = csr_row_sum(w, rowptr)
        ahat = csr_normalize_fwd(rowptr, col, w, rs)
        ctx.save_for_backward(w, rowptr, col, rs)
        return ahat

    @staticmethod
    def backward(ctx, dA):
        w, rowptr, col, rs = ctx.saved_tensors
        return csr_norm_bwd(rowptr, col, w, rs, dA.contiguous()), None, None


class CsrSpmmFn(torch.autograd.Function):
    """Y = A X on a CSR-valued adjacency (torch.mm(adj, x), model.py:594)."""

    @staticmethod
    def forward(ctx, a, rowptr, col, X):
        Y = csr_spmm_fwd(rowptr, col, a, X)
        ctx.save_for_backward(a, rowptr, col, X)
        return Y

    @staticmethod
    def backward(ctx, dY):
        a, rowptr, col, X = ctx.saved_tensors
        dA, dX = csr_spmm_bwd(rowptr, col, a, X, dY.contiguous(), need_dx=ctx.needs_input_grad[3])
        return dA, None, None, dX


def pair_keep(erow, col, p, seed):
    """counter-based dropout mask of the dense attention matrix on the listed pairs -> float [E] of 1.0 (kept) / 0.0 (dgg_pair_keep)"""
    erow, col = _chk(erow, torch.int32), _chk(col, torch.int32)
    out = torch.empty(col.shape, device=col.device, dtype=torch.float32)
    if col.numel():
        _lib.check(_lib.lib().dgg_pair_keep(_ptr(erow), _ptr(col), col.numel(), float(p), seed[0], seed[1], _ptr(out), _stream()), "pair_keep")
    return out


class MaskedDenseSumFn(torch.autograd.Function):
    """out_i = sum_j keep(i, j) X_j over ALL N columns with the counter-based pair mask (GATConv_DGG's F.dropout(attention) on the pairs
    that are not listed: they all carry the row's background weight, model.py:564-570); backward: the transposed masked sum."""

    @staticmethod
    def forward(ctx, X, p, seed):
        X = _chk(X)
        N, F = X.shape
        out = torch.empty_like(X)
        _lib.check(_lib.lib().dgg_masked_dense_sum(_ptr(X), N, F, float(p), seed[0], seed[1], 0, _ptr(out), _stream()), "masked_dense_sum")
        ctx.cfg = (float(p), seed)
        return out

    @staticmethod
    def backward(ctx, g):
        g = _chk(g.contiguous())
        N, F = g.shape
        p, seed = ctx.cfg
        dX = torch.empty_like(g)
        _lib.check(_lib.lib().dgg_masked_dense_sum(_ptr(g), N, F, p, seed[0], seed[1], 1, _ptr(dX), _stream()), "masked_dense_sum")
        return dX, None, None


class CsrBgSoftmaxFn(torch.autograd.Function):
    """Row softmax of a dense logit matrix given by explicit logits L [E] on a CSR pattern plus N - cnt_i background
    logits of 0 per row (GATConv_DGG, model.py:565-569) -> att [E], bg [N]."""

    @staticmethod
    def forward(ctx, L, rowptr):
        L = _chk(L)
        N = rowptr.shape[0] - 1
        att = torch.empty_like(L)
        bg = torch.empty((N,), device=L.device, dtype=torch.float32)
        _lib.check(_lib.lib().dgg_csr_bg_softmax_fwd(_ptr(L), _ptr(rowptr), N, _ptr(att), _ptr(bg), _stream()), "csr_bg_softmax_fwd")
        ctx.save_for_backward(att, bg, rowptr)
        return att, bg

    @staticmethod
    def backward(ctx, datt, dbg):
        att, bg, rowptr = ctx.saved_tensors
        dL = torch.empty_like(att)
        _lib.check(_lib.lib().dgg_csr_bg_softmax_bwd(_ptr(att), _ptr(bg), _ptr(rowptr), bg.shape[0], _ptr(_chk(datt.contiguous())),
                                                     _ptr(_chk(dbg.contiguous())), _ptr(dL), _stream()), "csr_bg_softmax_bwd")
        return dL, None


# ---- bf16 matrix-core path of the GCNII layer product (dgg_bf16.hip) ------------------------------------------------------
def pack_bf16(src, transpose=False):
    """fp32 [R,C] -> bf16 [R, C64] or (transpose) [C, R64]: the row length padded with zeros to a multiple of 64"""
    src = _chk(src)
    R, Cc = src.shape
    ld = ((R if transpose else Cc) + 63) // 64 * 64
    dst = torch.empty((Cc if transpose else R, ld), device=src.device, dtype=torch.bfloat16)
    _lib.check(_lib.lib().dgg_pack_bf16(_ptr(src), R, Cc, int(transpose), _ptr(dst), ld, _stream()), "pack_bf16")
    return dst


def pack_bf16_both(src):
    """fp32 [R,C] -> (bf16 [R, C64], bf16 [C, R64]) from one read of src: a weight's two operand layouts"""
    src = _chk(src)
    R, Cc = src.shape
    ld, ldT = (Cc + 63) // 64 * 64, (R + 63) // 64 * 64
    dst = torch.empty((R, ld), device=src.device, dtype=torch.bfloat16)
    dstT = torch.empty((Cc, ldT), device=src.device, dtype=torch.bfloat16)
    _lib.check(_lib.lib().dgg_pack_bf16_both(_ptr(src), R, Cc, _ptr(dst), ld, _ptr(dstT), ldT, _stream()), "pack_bf16_both")
    return dst, dstT


def gemm_nt_bf16(A, B, scale=1.0):
    """A [M,K] bf16, B [N,K] bf16 (K contiguous, a multiple of 64) -> fp32 scale * A B^T"""
    assert A.dtype == torch.bfloat16 and B.dtype == torch.bfloat16 and A.shape[1] == B.shape[1] and A.is_contiguous() and B.is_contiguous()
    M, K = A.shape
    N = B.shape[0]
    out = torch.empty((M, N), device=A.device, dtype=torch.float32)
    pe = _probe_begin()
    _lib.check(_lib.lib().dgg_gemm_nt_bf16(_ptr(A), _ptr(B), M, N, K, float(scale), _ptr(out), _stream()), "gemm_nt_bf16")
    _probe_end("gemm_bf16_bwd", pe)
    return out


def _packed_weight(weight, transpose):
    """bf16 copy of a weight for one product.  Packed on EVERY call (~15 us at 4096 x 2048, 2 % of the layer's GEMM time): a cache
    keyed on the storage address and the autograd version counter returned stale packs for a new parameter allocated at a freed
    address and after updates made through `p.data`, which do not move the counter."""
    return pack_bf16(weight.detach(), transpose=transpose)


class GcniiBf16Fn(torch.autograd.Function):
    """GCNII layer with the product on the bf16 matrix cores and the epilogue fused (model.py:36-44):
    out = theta * (support @ weight) + (1 - theta) * r (+ inp), r = (1 - alpha) hi + alpha h0 (h0 None: r = hi).
    support [n,K] and weight [K,F] are rounded to bf16 for the product (fp32 accumulation); everything else is fp32.
    Backward: d support = theta g W^T, d weight = theta support^T g on the same kernel; d hi / d h0 / d inp elementwise."""

    @staticmethod
    def forward(ctx, support, weight, hi, h0, inp, theta, alpha):
        n, K = support.shape
        F = weight.shape[1]
        assert K % 64 == 0, "gcnii bf16: in_features must be a multiple of 64"
        S = pack_bf16(support)                                   # [n, K]
        Wt = _packed_weight(weight, True)                        # [F, K]
        hi = _chk(hi)
        h0c = _chk(h0) if h0 is not None else None
        inpc = _chk(inp) if inp is not None else None
        out = torch.empty((n, F), device=support.device, dtype=torch.float32)
        pe = _probe_begin()
        _lib.check(_lib.lib().dgg_gcnii_gemm_bf16(_ptr(S), _ptr(Wt), n, F, K, _ptr(hi), _ptr(h0c), _ptr(inpc), float(theta), float(alpha),
                                                  _ptr(out), _stream()), "gcnii_gemm_bf16")
        _probe_end("gemm_bf16_fwd", pe)
        ctx.save_for_backward(S, weight)
        ctx.theta, ctx.alpha, ctx.has_h0, ctx.has_inp = float(theta), float(alpha), h0 is not None, inp is not None
        return out

    @staticmethod
    def backward(ctx, g):
        S, weight = ctx.saved_tensors
        g = _chk(g.contiguous())
        n, F = g.shape
        K = weight.shape[0]
        dS = dW = None
        if ctx.needs_input_grad[0]:
            dS = gemm_nt_bf16(pack_bf16(g), _packed_weight(weight, False), ctx.theta)     # [n,F] x [K,F]^T
        if ctx.needs_input_grad[1]:
            # contraction over the n nodes: both operands transposed, n zero-padded to a multiple of 64
            St = pack_bf16(S.float(), transpose=True)            # bf16 -> fp32 is exact: the transposed operand is re-packed from it
            dW = gemm_nt_bf16(St, pack_bf16(g, transpose=True), ctx.theta)               # [K,n] x [F,n]^T
        dsw, dhi = torch.empty_like(g), torch.empty_like(g)
        dh0 = torch.empty_like(g) if ctx.has_h0 else None
        _lib.check(_lib.lib().dgg_gcnii_epilogue_bwd(_ptr(g), g.numel(), ctx.theta, ctx.alpha, _ptr(dsw), _ptr(dhi), _ptr(dh0), _stream()),
                   "gcnii_epilogue_bwd")
        return dS, dW, dhi, dh0, (g if ctx.has_inp else None), None, None


def _h0_packs(h0):
    """bf16 copies of h0 -- plain for the forward product, transposed for the weight gradient -- made ONCE per forward of a GCNII
    stack: h0 is the same tensor object in every layer (model.py:724), the packs ride on it as an attribute and die with it; the
    version counter guards against an in-place update in between"""
    ent = getattr(h0, "_dgg_bf16_packs", None)
    if ent is None or ent[0] != h0._version or ent[1] != h0.data_ptr():
        hd = h0.detach()
        ent = (h0._version, h0.data_ptr(), pack_bf16(hd), pack_bf16(hd, transpose=True))
        h0._dgg_bf16_packs = ent
    return ent[2], ent[3]


class GcniiVariantBf16Fn(torch.autograd.Function):
    """The VARIANT GCNII layer (support = cat[hi, h0], model.py:37-44) on the bf16 matrix cores without the concatenation:
    out = theta * ([hi | h0] @ weight) + (1 - theta) * ((1 - alpha) hi + alpha h0) (+ inp).
    Forward: the A operand of the product is split (bf16(hi) packed here, bf16(h0) once per stack); backward: [d hi | d h0] in one
    product with the elementwise terms in its epilogue, d weight as two products hi^T g and h0^T g written into the two row blocks
    of the gradient.  Same operand rounding as GcniiBf16Fn on cat[hi, h0]."""

    @staticmethod
    def forward(ctx, hi, h0, weight, inp, theta, alpha):
        n, F = hi.shape
        assert weight.shape[0] == 2 * F and F % 64 == 0
        hi, h0c = _chk(hi), _chk(h0)
        inpc = _chk(inp) if inp is not None else None
        S1 = pack_bf16(hi)
        S2, _ = _h0_packs(h0)
        Wt = _packed_weight(weight, True)                        # [Fout, 2F]
        Fo = weight.shape[1]
        out = torch.empty((n, Fo), device=hi.device, dtype=torch.float32)
        pe = _probe_begin()
        _lib.check(_lib.lib().dgg_gcnii_gemm_bf16_split(_ptr(S1), _ptr(S2), _ptr(Wt), n, Fo, 2 * F, F, _ptr(hi), _ptr(h0c), _ptr(inpc),
                                                        float(theta), float(alpha), _ptr(out), _stream()), "gcnii_gemm_bf16_split")
        _probe_end("gemm_bf16_fwd", pe)
        ctx.save_for_backward(hi, h0, weight)
        ctx.theta, ctx.alpha, ctx.has_inp = float(theta), float(alpha), inp is not None
        return out

    @staticmethod
    def backward(ctx, g):
        hi, h0, weight = ctx.saved_tensors
        g = _chk(g.contiguous())
        n, Fo = g.shape
        F = hi.shape[1]
        assert Fo == F, "the variant layer is square (nhidden -> nhidden)"
        dhi = dh0 = dW = None
        Gp = pack_bf16(g)
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            dhi, dh0 = torch.empty_like(g), torch.empty_like(g)
            pe = _probe_begin()
            _lib.check(_lib.lib().dgg_gcnii_dsupport_bf16(_ptr(Gp), _ptr(_packed_weight(weight, False)), n, F, _ptr(g), ctx.theta, ctx.alpha,
                                                          _ptr(dhi), _ptr(dh0), _stream()), "gcnii_dsupport_bf16")
            _probe_end("gemm_bf16_bwd", pe)
        if ctx.needs_input_grad[2]:
            # contraction over the n nodes: operands transposed, n zero-padded to a multiple of 64
            GT = pack_bf16(g, transpose=True)                    # [F, n64]
            hiT = pack_bf16(hi, transpose=True)
            _, h0T = _h0_packs(h0)
            dW = torch.empty_like(weight)
            pe = _probe_begin()
            if F % 128 == 0:
                _lib.check(_lib.lib().dgg_gemm_nt_bf16_rows2(_ptr(hiT), _ptr(h0T), F, _ptr(GT), 2 * F, Fo, hiT.shape[1], ctx.theta, _ptr(dW),
                                                             _stream()), "gemm_nt_bf16_rows2")
            else:
                for r0_, AT in ((0, hiT), (F, h0T)):
                    _lib.check(_lib.lib().dgg_gemm_nt_bf16(_ptr(AT), _ptr(GT), F, Fo, AT.shape[1], ctx.theta,
                                                           C.c_void_p(dW.data_ptr() + 4 * r0_ * Fo), _stream()), "gemm_nt_bf16")
            _probe_end("gemm_bf16_bwd", pe)
        return dhi, dh0, dW, (g if ctx.has_inp else None), None, None


def dropout_hash(x, p, s0, s1, accumulate_into=None, bf16_copy=False):
    """x * keep / (1 - p) with the counter-based mask of the fused GCNII stack (dgg_dropout_hash); accumulate_into: += instead;
    bf16_copy: -> (out, bf16(out)) (p = 0: a plain copy with its bf16 twin)"""
    x = _chk(x)
    out = torch.empty_like(x) if accumulate_into is None else accumulate_into
    outb = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16) if bf16_copy else None
    _lib.check(_lib.lib().dgg_dropout_hash(_ptr(x), x.numel(), float(p), int(s0) & 0xFFFFFFFF, int(s1) & 0xFFFFFFFF,
                                           int(accumulate_into is not None), _ptr(out), _ptr(outb), _stream()), "dropout_hash")
    return (out, outb) if bf16_copy else out


_STACK_KEY = 0x9E3779B9
STACK_BF16_GATHERS = os.environ.get("DGG_STACK_BF16_GATHERS", "1") != "0"      # see GcniiStackBf16Fn
STACK_SPLIT_EPILOGUE = os.environ.get("DGG_STACK_SPLIT_EPILOGUE", "1") != "0"  # forward product + one elementwise pass instead of the fused epilogue


class GcniiStackBf16Fn(torch.autograd.Function):
    """A whole stack of VARIANT GCNII layers on ONE adjacency as a single autograd node, products on the bf16 matrix cores
    (reference model.py:716-731 / 942-957: `for con: x = dropout(x); x = relu(con(x, adj, h0, ...))`, then the
    dropout in front of the output layer; layer = model.py:32-44 with support = cat[hi, h0]):

        xd_0 = dropout(h0);   hi_l = A xd_{l-1};   xd_l = dropout(relu(theta_l [hi_l | h0] W_l + (1 - theta_l)((1 - alpha) hi_l + alpha h0) + xd_{l-1}))

    returns xd_L.  What the per-layer modules launch and this does not: the bf16 pack of hi (written by the aggregation itself), the
    ReLU and the dropout (in the product's epilogue; the mask is counter-based and read back off the stored activation, xd != 0), their
    two backward kernels and the two packs of the gradient (one pass: dgg_gcnii_gout_pack), the residual add (A^T d hi accumulates
    into a copy of the gradient).  Same operand rounding as
    GcniiVariantBf16Fn; the dropout draws from its own counter-based stream (seeded per forward from torch's CPU generator).
    STACK_BF16_GATHERS (default on; F a multiple of 512): the three kernels that GATHER 2048-wide rows -- the aggregation, the SDDMM
    (both read the layer's input activation) and the transposed aggregation (reads d hi) -- work on bf16 copies that the producing
    epilogues write beside the fp32 tensors: they are bound by L2 bandwidth at this width, the copies halve their bytes (fp32
    accumulation; the residual, the epilogue terms and every saved tensor stay fp32).  This rounds the AGGREGATED activations to 8
    significant bits as well -- BASELINE configs[4] is "bf16 fwd+bwd" -- where GcniiVariantBf16Fn rounds only the product operands."""

    @staticmethod
    def forward(ctx, h0, ahat, idx, part, skip_zero, residual, p, lamda, alpha, seed, train, *weights):
        # train: backward_will_follow(h0, ahat, *weights), evaluated by the caller (an eval forward packs one weight layout, not two)
        n, F = h0.shape
        L = len(weights)
        assert F % 256 == 0 and all(tuple(W.shape) == (2 * F, F) for W in weights)
        h0c, ahat = _chk(h0), _chk(ahat)
        s0, s1 = int(seed[0]) & 0xFFFFFFFF, int(seed[1]) & 0xFFFFFFFF
        K = idx.shape[1]
        S2, _ = _h0_packs(h0)
        b16 = STACK_BF16_GATHERS and F % 512 == 0
        pe = _probe_begin()
        if b16:
            xd, xdb = dropout_hash(h0c, p, s0, s1, bf16_copy=True)
        else:
            xd, xdb = (dropout_hash(h0c, p, s0, s1) if p > 0 else h0c), None
        his, xds, xdbs, wps = [], [xd], [xdb], []
        train = bool(train)                                      # a backward will follow: it re-uses this forward's weight packs
        # the product operand cat[bf16(hi) | bf16(h0)] [n, 2F]: the right half once per stack, the left half by every layer's aggregation
        if STACK_SPLIT_EPILOGUE:
            hib, ldh = torch.empty((n, 2 * F), device=h0.device, dtype=torch.bfloat16), 2 * F
            hib[:, F:].copy_(S2)
            sw = torch.empty((n, F), device=h0.device, dtype=torch.float32)
        else:
            hib, ldh = torch.empty((n, F), device=h0.device, dtype=torch.bfloat16), F
        for l, W in enumerate(weights, 1):
            theta = math.log(lamda / l + 1)
            hi = torch.empty((n, F), device=h0.device, dtype=torch.float32)
            if b16:
                _lib.check(_lib.lib().dgg_ell_spmm_fwd_b16(_ptr(idx), _ptr(ahat), _ptr(xdb), n, K, F, _ptr(hi), _ptr(hib), ldh, _stream()), "ell_spmm_fwd_b16")
            else:
                _lib.check(_lib.lib().dgg_ell_spmm_fwd_bf16(_ptr(idx), _ptr(ahat), _ptr(xd), n, K, F, _ptr(hi), _ptr(hib), ldh, _stream()), "ell_spmm_fwd_bf16")
            if train:                                            # both layouts from one read; the plain one is d support's operand
                Wp, Wt = pack_bf16_both(W.detach())
                wps.append(Wp)
            else:
                Wt = _packed_weight(W, True)                     # [F, 2F]
            out = torch.empty((n, F), device=h0.device, dtype=torch.float32)
            outb = torch.empty((n, F), device=h0.device, dtype=torch.bfloat16) if (b16 and l < L) else None      # (the last output is gathered by no one)
            sl = (s1 ^ (_STACK_KEY * l)) & 0xFFFFFFFF
            if STACK_SPLIT_EPILOGUE:       # plain product, then ONE elementwise pass (faster than the product with the epilogue inside)
                _lib.check(_lib.lib().dgg_gemm_nt_bf16(_ptr(hib), _ptr(Wt), n, F, 2 * F, 1.0, _ptr(sw), _stream()), "gemm_nt_bf16")
                _lib.check(_lib.lib().dgg_gcnii_stack_epilogue(_ptr(sw), _ptr(hi), _ptr(h0c), _ptr(xd if residual else None), n * F, float(theta),
                                                               float(alpha), float(p), s0, sl, _ptr(out), _ptr(outb), _stream()), "gcnii_stack_epilogue")
            else:
                _lib.check(_lib.lib().dgg_gcnii_gemm_bf16_split_act(_ptr(hib), _ptr(S2), _ptr(Wt), n, F, 2 * F, F, _ptr(hi), _ptr(h0c),
                                                                    _ptr(xd if residual else None), float(theta), float(alpha), 1, float(p), s0,
                                                                    sl, _ptr(out), _ptr(outb), _stream()), "gcnii_gemm_bf16_split_act")
            his.append(hi)
            xds.append(out)
            xdbs.append(outb)
            xd, xdb = out, outb
        _probe_end("gcnii_stack_fwd", pe)
        ctx.save_for_backward(h0, ahat, idx, *weights, *his, *xds)
        ctx.xdbs = xdbs[:L] if b16 else None                     # bf16 copies of xd_0 .. xd_{L-1} (plain tensors, no autograd edge)
        ctx.wps = wps if train else None                         # bf16(W_l) as stored: made with this forward's W_l, which autograd's version check on the saved weights keeps unchanged until the backward
        ctx.cfg = (L, part, bool(skip_zero), bool(residual), float(p), float(lamda), float(alpha), s0, s1)
        return xd

    @staticmethod
    def backward(ctx, g):
        L, part, skip_zero, residual, p, lamda, alpha, s0, s1 = ctx.cfg
        sv = ctx.saved_tensors
        h0, ahat, idx = sv[0], sv[1], sv[2]
        weights, his, xds = sv[3:3 + L], sv[3 + L:3 + 2 * L], sv[3 + 2 * L:]
        n, F = h0.shape
        K = idx.shape[1]
        n64 = (n + 63) // 64 * 64
        dev = h0.device
        if part is None:
            part = part_build(idx, ahat, n)
        _, h0T = _h0_packs(h0)
        scale = 1.0 / (1.0 - p)
        b16 = ctx.xdbs is not None
        gx = _chk(g.contiguous())
        dh0 = torch.empty((n, F), device=dev, dtype=torch.float32)
        dA, sd_ws = None, None
        dWs = [None] * L
        pe = _probe_begin()
        for l in range(L, 0, -1):
            theta = math.log(lamda / l + 1)
            W, hi, xd_l, xd_prev = weights[l - 1], his[l - 1], xds[l], xds[l - 1]
            gout = torch.empty((n, F), device=dev, dtype=torch.float32)
            Gp = torch.empty((n, F), device=dev, dtype=torch.bfloat16)
            GT = torch.empty((F, n64), device=dev, dtype=torch.bfloat16)
            gnext = torch.empty((n, F), device=dev, dtype=torch.float32) if residual else None       # becomes g + A^T d hi below
            _lib.check(_lib.lib().dgg_gcnii_gout_pack(_ptr(gx), _ptr(xd_l), float(scale), n, F, _ptr(gout), _ptr(Gp), _ptr(GT), n64, _ptr(gnext),
                                                      _stream()), "gcnii_gout_pack")
            dhi = None if b16 else torch.empty((n, F), device=dev, dtype=torch.float32)          # (the bf16 gathers read only the copy)
            dhib = torch.empty((n, F), device=dev, dtype=torch.bfloat16) if b16 else None
            Wp = ctx.wps[l - 1] if ctx.wps is not None else _packed_weight(W, False)
            _lib.check(_lib.lib().dgg_gcnii_dsupport_bf16_b(_ptr(Gp), _ptr(Wp), n, F, _ptr(gout), float(theta), float(alpha),
                                                            _ptr(dhi), _ptr(dh0), _ptr(dhib), int(l != L), _stream()), "gcnii_dsupport_bf16")   # d h0: summed over the layers in the epilogue
            if ctx.needs_input_grad[11 + l - 1]:
                hiT = pack_bf16(hi, transpose=True)
                dW = torch.empty_like(W)
                _lib.check(_lib.lib().dgg_gemm_nt_bf16_rows2(_ptr(hiT), _ptr(h0T), F, _ptr(GT), 2 * F, F, n64, float(theta), _ptr(dW), _stream()),
                           "gemm_nt_bf16_rows2")
                dWs[l - 1] = dW
            gx = gnext if residual else torch.zeros_like(gout)              # + g through the residual; A^T d hi accumulates into it
            if b16:                                                          # d A: the slices' sums pile up over the layers, ONE slice reduction after the loop
                if sd_ws is None:
                    sd_ws = torch.empty((int(_lib.lib().dgg_ell_sddmm_b16_ws_floats(n, K, F)),), device=dev, dtype=torch.float32)
                _lib.check(_lib.lib().dgg_ell_sddmm_b16_sliced(_ptr(idx), _ptr(ahat), _ptr(ctx.xdbs[l - 1]), _ptr(dhib), n, K, F, int(skip_zero), _ptr(sd_ws),
                                                               None, int(l != L), _stream()), "ell_sddmm_b16_sliced")
                _lib.check(_lib.lib().dgg_ell_spmm_t_part_b16(_ptr(ahat), _ptr(dhib), n, K, F, _ptr(part), n, _ptr(gx), _stream()), "ell_spmm_t_part_b16")
            else:
                dA_l = torch.empty((n, K), device=dev, dtype=torch.float32)
                _lib.check(_lib.lib().dgg_ell_spmm_bwd(_ptr(idx), _ptr(ahat), _ptr(xd_prev), _ptr(dhi), n, K, F, int(skip_zero), _ptr(dA_l), _ptr(None),
                                                       _stream()), "ell_spmm_bwd")
                _lib.check(_lib.lib().dgg_ell_spmm_t_part(_ptr(ahat), _ptr(dhi), n, K, F, _ptr(part), n, _ptr(gx), _stream()), "ell_spmm_t_part")
                dA = dA_l if dA is None else dA.add_(dA_l)
        if sd_ws is not None:
            dA = torch.empty((n, K), device=dev, dtype=torch.float32)
            _lib.check(_lib.lib().dgg_ell_sddmm_slices_sum(_ptr(sd_ws), n, K, F, _ptr(dA), 0, _stream()), "ell_sddmm_slices_sum")
        if p > 0:
            dropout_hash(gx, p, s0, s1, accumulate_into=dh0)     # back through xd_0 = dropout(h0): the same mask
        else:
            dh0.add_(gx)
        _probe_end("gcnii_stack_bwd", pe)
        return (dh0, dA, None, None, None, None, None, None, None, None, None) + tuple(dWs)


class GcniiEpilogueFn(torch.autograd.Function):
    """out = theta * sw + (1 - theta) * ((1 - alpha) * hi + alpha * h0) (+ inp)  (GraphConvolution.forward, model.py:36-44);
    h0 None: r = hi (the non-variant layer, whose support is r itself)."""

    @staticmethod
    def forward(ctx, sw, hi, h0, inp, theta, alpha):
        sw, hi = _chk(sw), _chk(hi)
        h0 = _chk(h0) if h0 is not None else None
        inp = _chk(inp) if inp is not None else None
        out = torch.empty_like(sw)
        _lib.check(_lib.lib().dgg_gcnii_epilogue_fwd(_ptr(sw), _ptr(hi), _ptr(h0), _ptr(inp), sw.numel(), float(theta), float(alpha),
                                                     _ptr(out), _stream()), "gcnii_epilogue_fwd")
        ctx.theta, ctx.alpha, ctx.has_h0, ctx.has_inp = float(theta), float(alpha), h0 is not None, inp is not None
        return out

    @staticmethod
    def backward(ctx, g):
        g = _chk(g.contiguous())
        dsw, dhi = torch.empty_like(g), torch.empty_like(g)
        dh0 = torch.empty_like(g) if ctx.has_h0 else None
        _lib.check(_lib.lib().dgg_gcnii_epilogue_bwd(_ptr(g), g.numel(), ctx.theta, ctx.alpha, _ptr(dsw), _ptr(dhi), _ptr(dh0),
                                                     _stream()), "gcnii_epilogue_bwd")
        return dsw, dhi, dh0, (g if ctx.has_inp else None), None, None
