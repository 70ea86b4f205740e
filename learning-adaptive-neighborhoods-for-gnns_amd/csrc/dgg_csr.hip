// dgg_csr.hip -- adjacency with the sparsity of in_adj (CSR, variable row length) and the tail of the `DGG` class.
//
// The `DGG` module "for ICLR" (reference dgm.py:1730-1815, used by GCN_DGG_00 / SAGE_DGG_00 / GAT_DGG_00,
// model.py:1314-1433) keeps EVERY candidate edge with weight rank * (ramp + 1), so its output is as wide as the widest
// row of in_adj (168 on Cora) and does not fit the 64-wide ELL of the DGG_LearnableK_debug path.  This file holds
//   rank_ramp fwd/bwd   S_i = sum_j rank_ij, k_i = leaky(S_i w + b), position in the row sorted by (rank desc, column asc),
//                       out = rank * ((1 - 0.5 (1 + tanh(pos - k))) + 1)                       dgm.py:1791-1812
//   row_sum / normalize D^-1/2 A D^-1/2 with row sums on both sides                            model.py:1340-1352
//   spmm fwd/bwd        torch.mm(adj, x) and its autograd                                      model.py:594
// One wavefront per row, entries walked in chunks of 64 (lane = entry) or sequentially (lane = feature).
#include "dgg_common.h"
#include "dgg_api_internal.h"

using namespace dgg;

namespace {

constexpr int WPB = 4;

// canonical row reduction: entry e of the row goes to slot e mod 64 (sequential adds), then the xor butterfly
__device__ __forceinline__ float row_sum_strided(const float *__restrict__ v, int64_t e0, int64_t e1, int lane) {
    float s = 0.0f;
    for (int64_t e = e0 + lane; e < e1; e += 64) s = __fadd_rn(s, v[e]);
    return wave_sum_butterfly(s);
}

__global__ __launch_bounds__(WPB * 64) void csr_row_sum_kernel(const float *__restrict__ vals, const int64_t *__restrict__ rowptr,
                                                              int64_t N, float *__restrict__ rs) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * WPB + dgg::wave_id();
    if (i >= N) return;
    const float s = row_sum_strided(vals, rowptr[i], rowptr[i + 1], lane);
    if (lane == 0) rs[i] = s;
}

__global__ __launch_bounds__(WPB * 64) void csr_normalize_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                                const float *__restrict__ w, const float *__restrict__ rs,
                                                                int64_t N, float *__restrict__ ahat) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * WPB + dgg::wave_id();
    if (i >= N) return;
    const float ai = __fdiv_rn(1.0f, c_sqrt(rs[i]));
    for (int64_t e = rowptr[i] + lane; e < rowptr[i + 1]; e += 64)
        ahat[e] = __fmul_rn(__fmul_rn(ai, w[e]), __fdiv_rn(1.0f, c_sqrt(rs[col[e]])));
}

// Y_i[c] = sum_e a_e X[col_e][c], e ascending (fmaf chain); lanes = features, grid.y walks blocks of 64 features
__global__ __launch_bounds__(WPB * 64) void csr_spmm_fwd_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                               const float *__restrict__ a, const float *__restrict__ X, int64_t N,
                                                               int F, float *__restrict__ Y) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * WPB + dgg::wave_id();
    if (i >= N) return;
    const int c = blockIdx.y * 64 + lane;
    float acc = 0.0f;
    const int64_t e0 = rowptr[i], e1 = rowptr[i + 1];
    for (int64_t eb = e0; eb < e1; eb += 64) {                   // entries of a chunk are loaded one per lane, then broadcast
        const int64_t e = eb + lane;
        const int32_t jl = e < e1 ? col[e] : 0;
        const float al = e < e1 ? a[e] : 0.0f;
        const int n = e1 - eb < 64 ? (int)(e1 - eb) : 64;
        for (int r = 0; r < n; r++) {
            const int32_t j = bcast(jl, r);
            const float av = bcast(al, r);
            if (c < F) acc = __fmaf_rn(av, X[(int64_t)j * F + c], acc);
        }
    }
    if (c < F) Y[i * F + c] = acc;
}

// dA_e = <dY_i, X_j>;  dX_j += a_e dY_i (float atomics, optional)
__global__ __launch_bounds__(WPB * 64) void csr_spmm_bwd_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                               const float *__restrict__ a, const float *__restrict__ X,
                                                               const float *__restrict__ dY, int64_t N, int F,
                                                               float *__restrict__ dA, float *__restrict__ dX) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * WPB + dgg::wave_id();
    if (i >= N) return;
    for (int64_t e = rowptr[i]; e < rowptr[i + 1]; e++) {
        const int64_t j = col[e];
        const float av = a[e];
        float part = 0.0f;
        for (int c = lane; c < F; c += 64) {
            const float g = dY[i * F + c];
            part = fmaf(g, X[j * F + c], part);
            if (dX && av != 0.0f) atomicAdd(dX + j * F + c, av * g);
        }
        part = wave_sum_dpp(part, lane);
        if (lane == 0) dA[e] = part;
    }
}

// normalisation backward, phase 1: da_i += sum_e g_e a_j (row, plain add by one lane), da_j += g_e a_i (atomics); g = dA w
__global__ __launch_bounds__(WPB * 64) void csr_norm_bwd_da_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                                  const float *__restrict__ w, const float *__restrict__ rs,
                                                                  const float *__restrict__ dA, int64_t N, float *__restrict__ da) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * WPB + dgg::wave_id();
    if (i >= N) return;
    const float ai = 1.0f / sqrtf(rs[i]);
    float rowpart = 0.0f;
    for (int64_t e = rowptr[i] + lane; e < rowptr[i + 1]; e += 64) {
        const float g = dA[e] * w[e];
        if (g != 0.0f) {
            const int32_t j = col[e];
            rowpart += g * (1.0f / sqrtf(rs[j]));
            atomicAdd(da + j, g * ai);
        }
    }
    rowpart = wave_sum_dpp(rowpart, lane);
    if (lane == 0 && rowpart != 0.0f) atomicAdd(da + i, rowpart);
}
// phase 2: dw_e = dA_e a_i a_j + drs_i, drs_i = -0.5 da_i a_i / rs_i
__global__ __launch_bounds__(WPB * 64) void csr_norm_bwd_dw_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                                  const float *__restrict__ rs, const float *__restrict__ dA,
                                                                  const float *__restrict__ da, int64_t N, float *__restrict__ dw) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * WPB + dgg::wave_id();
    if (i >= N) return;
    const float ai = 1.0f / sqrtf(rs[i]);
    const float drs = -0.5f * da[i] * ai / rs[i];
    for (int64_t e = rowptr[i] + lane; e < rowptr[i + 1]; e += 64) dw[e] = dA[e] * ai * (1.0f / sqrtf(rs[col[e]])) + drs;
}

// ---- `DGG.forward` tail (dgm.py:1791-1812) ----------------------------------------------------------------------------
// position of an entry = number of entries of its row that sort before it under (rank desc, column asc): every lane
// counts for its own entries while the row streams through in chunks of 64 broadcast from registers
__global__ __launch_bounds__(WPB * 64) void csr_rank_ramp_fwd_kernel(const float *__restrict__ p, const int64_t *__restrict__ rowptr,
                                                                    const int32_t *__restrict__ col, int64_t N,
                                                                    const float *__restrict__ wb, const float *__restrict__ bb,
                                                                    float *__restrict__ out, float *__restrict__ S,
                                                                    float *__restrict__ k, int32_t *__restrict__ pos) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * WPB + dgg::wave_id();
    if (i >= N) return;
    const int64_t e0 = rowptr[i], e1 = rowptr[i + 1];
    const float s = row_sum_strided(p, e0, e1, lane);
    const float z = __fadd_rn(__fmul_rn(s, wb[0]), bb[0]);        // degree_decoder: Linear(1,1) + LeakyReLU
    const float ki = z > 0.0f ? z : __fmul_rn(0.01f, z);
    if (lane == 0) { S[i] = s; k[i] = ki; }
    for (int64_t mb = e0; mb < e1; mb += 64) {                   // my entry of this chunk
        const int64_t me = mb + lane;
        const bool have = me < e1;
        const uint64_t mykey = have ? make_key(p[me], col[me]) : 0ull;
        int cnt = 0;
        for (int64_t qb = e0; qb < e1; qb += 64) {
            const int64_t q = qb + lane;
            const uint64_t qk = q < e1 ? make_key(p[q], col[q]) : 0ull;
            const int n = e1 - qb < 64 ? (int)(e1 - qb) : 64;
            for (int r = 0; r < n; r++) cnt += shfl_u64(qk, r) > mykey ? 1 : 0;
        }
        if (have) {
            pos[me] = cnt;
            const float f = __fadd_rn(c_ramp((float)cnt, ki), 1.0f);
            out[me] = __fmul_rn(p[me], f);
        }
    }
}

// g = d out -> dp (direct + through S -> k), dkz_i = d loss / d (S_i w + b)
__global__ __launch_bounds__(WPB * 64) void csr_rank_ramp_bwd_kernel(const float *__restrict__ p, const int64_t *__restrict__ rowptr,
                                                                    int64_t N, const float *__restrict__ wb,
                                                                    const float *__restrict__ bb, const float *__restrict__ S,
                                                                    const float *__restrict__ k, const int32_t *__restrict__ pos,
                                                                    const float *__restrict__ g, float *__restrict__ dp,
                                                                    float *__restrict__ dkz) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * WPB + dgg::wave_id();
    if (i >= N) return;
    const int64_t e0 = rowptr[i], e1 = rowptr[i + 1];
    const float ki = k[i], w = wb[0];
    float dk = 0.0f;
    for (int64_t e = e0 + lane; e < e1; e += 64) {
        const float th = c_tanh((float)pos[e] - ki);
        dk += g[e] * p[e] * 0.5f * (1.0f - th * th);
    }
    dk = wave_sum_butterfly(dk);
    const float z = S[i] * w + bb[0];
    const float dz = z > 0.0f ? dk : 0.01f * dk;
    if (lane == 0) dkz[i] = dz;
    for (int64_t e = e0 + lane; e < e1; e += 64) {
        const float th = c_tanh((float)pos[e] - ki);
        const float f = 1.0f - 0.5f * (1.0f + th) + 1.0f;
        dp[e] = g[e] * f + dz * w;
    }
}

// ---- `DGG_LearnableK_debug.select_top_k` on rows of ANY width (dgm.py:1402-1435) ---------------------------------------------
// The 64-wide ELL of the fast path is exact only while ceil(k_i + 8.5) <= 64 (or the row has no more candidates than that).  For
// graphs whose rows are wider and whose learned degrees may exceed the bound, the same computation runs on the CSR pattern of
// in_adj: perturbation (dgm.py:1213-1229), position of every candidate in its row's sort (counting; ties: lower column first),
// ramp 1 - 0.5 (1 + tanh(pos - k_i)), w = p' * ramp (k_times_edge_prob) or ramp (k_only).  One wavefront per row.
__global__ __launch_bounds__(WPB * 64) void csr_softk_fwd_kernel(const float *__restrict__ p, const int64_t *__restrict__ rowptr,
                                                                const int32_t *__restrict__ col, int64_t N, const float *__restrict__ k,
                                                                int noise_mode, const float *__restrict__ G, int64_t ldG, uint32_t s0,
                                                                uint32_t s1, int mode, float *__restrict__ w, float *__restrict__ pp,
                                                                int32_t *__restrict__ pos) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * WPB + dgg::wave_id();
    if (i >= N) return;
    const int64_t e0 = rowptr[i], e1 = rowptr[i + 1];
    const float ki = k[i];
    auto perturbed = [&](int64_t e) {
        const float pe = p[e];
        if (noise_mode == 0) return pe;
        const int32_t j = col[e];
        const float g = noise_mode == 1 ? G[i * ldG + j] : pair_noise(s0, s1, (uint32_t)i, (uint32_t)j, noise_mode == 3);
        return c_exp(__fadd_rn(c_log(__fadd_rn(pe, 1e-8f)), g));
    };
    for (int64_t e = e0 + lane; e < e1; e += 64) pp[e] = perturbed(e);
    // (same wavefront wrote pp: the loads below are ordered behind the stores)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int64_t mb = e0; mb < e1; mb += 64) {
        const int64_t me = mb + lane;
        const bool have = me < e1;
        const uint64_t mykey = have ? make_key(pp[me], col[me]) : 0ull;
        int cnt = 0;
        for (int64_t qb = e0; qb < e1; qb += 64) {
            const int64_t q = qb + lane;
            const uint64_t qk = q < e1 ? make_key(pp[q], col[q]) : 0ull;
            const int n = e1 - qb < 64 ? (int)(e1 - qb) : 64;
            for (int r = 0; r < n; r++) cnt += shfl_u64(qk, r) > mykey ? 1 : 0;
        }
        if (have) {
            pos[me] = cnt;
            const float f = c_ramp((float)cnt, ki);
            w[me] = mode == 0 ? __fmul_rn(pp[me], f) : f;
        }
    }
}

// g = d loss / d w -> dp [E] (through the perturbation: d p' / d p = p' / (p + 1e-8)), dk [N]
__global__ __launch_bounds__(WPB * 64) void csr_softk_bwd_kernel(const float *__restrict__ p, const float *__restrict__ pp,
                                                                const int64_t *__restrict__ rowptr, int64_t N, const float *__restrict__ k,
                                                                const int32_t *__restrict__ pos, int perturb, int mode,
                                                                const float *__restrict__ g, float *__restrict__ dp, float *__restrict__ dk) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * WPB + dgg::wave_id();
    if (i >= N) return;
    const float ki = k[i];
    float acc = 0.0f;
    for (int64_t e = rowptr[i] + lane; e < rowptr[i + 1]; e += 64) {
        const float th = c_tanh((float)pos[e] - ki);
        const float f = 1.0f - 0.5f * (1.0f + th), dfdk = 0.5f * (1.0f - th * th);
        float dpp = 0.0f;
        if (mode == 0) { dpp = g[e] * f; acc += g[e] * pp[e] * dfdk; }
        else acc += g[e] * dfdk;
        dp[e] = perturb ? dpp * pp[e] / (p[e] + 1e-8f) : dpp;
    }
    acc = wave_sum_butterfly(acc);
    if (lane == 0) dk[i] = acc;
}

// ---- `DGG_Ablations.forward` (dgm.py:1927-1962): noisy second sigmoid and the fixed-k truncation -----------------------
// edge_rank = sigmoid(sigmoid(score) + noise), noise ~ U(-1,1) per stored edge (dgm.py:1930-1933)
__global__ __launch_bounds__(256) void csr_noisy_sigmoid_fwd_kernel(const float *__restrict__ p, const float *__restrict__ noise,
                                                                  int64_t E, float *__restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e < E) out[e] = 1.0f / (1.0f + c_exp(-__fadd_rn(p[e], noise[e])));
}
__global__ __launch_bounds__(256) void csr_noisy_sigmoid_bwd_kernel(const float *__restrict__ out, const float *__restrict__ g,
                                                                  int64_t E, float *__restrict__ dp) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e < E) dp[e] = g[e] * out[e] * (1.0f - out[e]);
}
// srt_edge_rank[:, k:] = 0 (dgm.py:1940-1942): an entry survives iff fewer than kcut entries of its row sort before it
__global__ __launch_bounds__(WPB * 64) void csr_rank_cut_fwd_kernel(const float *__restrict__ p, const int64_t *__restrict__ rowptr,
                                                                   const int32_t *__restrict__ col, int64_t N, int kcut,
                                                                   float *__restrict__ out, int32_t *__restrict__ pos) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * WPB + dgg::wave_id();
    if (i >= N) return;
    const int64_t e0 = rowptr[i], e1 = rowptr[i + 1];
    for (int64_t mb = e0; mb < e1; mb += 64) {
        const int64_t me = mb + lane;
        const bool have = me < e1;
        const uint64_t mykey = have ? make_key(p[me], col[me]) : 0ull;
        int cnt = 0;
        for (int64_t qb = e0; qb < e1; qb += 64) {
            const int64_t q = qb + lane;
            const uint64_t qk = q < e1 ? make_key(p[q], col[q]) : 0ull;
            const int n = e1 - qb < 64 ? (int)(e1 - qb) : 64;
            for (int r = 0; r < n; r++) cnt += shfl_u64(qk, r) > mykey ? 1 : 0;
        }
        if (have) {
            pos[me] = cnt;
            out[me] = cnt < kcut ? p[me] : 0.0f;
        }
    }
}
__global__ __launch_bounds__(256) void csr_rank_cut_bwd_kernel(const int32_t *__restrict__ pos, const float *__restrict__ g, int64_t E,
                                                             int kcut, float *__restrict__ dp) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e < E) dp[e] = pos[e] < kcut ? g[e] : 0.0f;
}

// ---- raw edge probabilities as the adjacency (debug_step 0/1 and k-select mode edge_p-cdf of DGG_LearnableK_debug) --------
// dgm.py:1202-1209, 1240-1246: the forward returns edge_p itself; dgm.py:1368-1401: `edge_p-cdf` scatters the UNSORTED
// probabilities back (src = s_edge_p), so its output is edge_p as well.  u-v-dist scorer on the stored entries of in_adj
// (dgm.py:1613-1627): p_e = exp(t ||xp_u - xp_v||), lane = entry, canonical distance.
__global__ __launch_bounds__(WPB * 64) void csr_uvdist_fwd_kernel(const float *__restrict__ xp, const int64_t *__restrict__ rowptr,
                                                                 const int32_t *__restrict__ col, int64_t N, int h, float t,
                                                                 float *__restrict__ p) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * WPB + dgg::wave_id();
    if (i >= N) return;
    const float *xi = xp + i * h;
    for (int64_t e = rowptr[i] + lane; e < rowptr[i + 1]; e += 64) {
        const float d = c_sqrt(pair_d2_thread(xi, xp + (int64_t)col[e] * h, h));
        p[e] = c_exp(__fmul_rn(t, d));
    }
}
// dp -> dxp (accumulated, caller zeroes): lane = feature, entries of the row in sequence; own row by plain accumulation in
// registers, neighbour rows by float atomics (edge lists are small: the 100k-node path never comes here)
__global__ __launch_bounds__(WPB * 64) void csr_uvdist_bwd_kernel(const float *__restrict__ xp, const int64_t *__restrict__ rowptr,
                                                                 const int32_t *__restrict__ col, int64_t N, int h, float t,
                                                                 const float *__restrict__ p, const float *__restrict__ dp,
                                                                 float *__restrict__ dxp) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * WPB + dgg::wave_id();
    if (i >= N) return;
    const float *xi = xp + i * h;
    for (int64_t e = rowptr[i]; e < rowptr[i + 1]; e++) {
        const float g = dp[e] * p[e] * t;                          // d loss / d dist
        if (g == 0.0f) continue;
        const int64_t j = col[e];
        const float *xj = xp + j * h;
        float d2 = 0.0f;
        for (int c = lane; c < h; c += 64) { const float df = xi[c] - xj[c]; d2 += df * df; }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) d2 += __shfl_xor(d2, off, 64);
        if (!(d2 > 0.0f)) continue;                                // zero distance: torch's norm backward yields 0
        const float coef = g / sqrtf(d2);
        for (int c = lane; c < h; c += 64) {
            const float v = coef * (xi[c] - xj[c]);
            atomicAdd(dxp + i * h + c, v);
            atomicAdd(dxp + j * h + c, -v);
        }
    }
}

// ---- GATConv_DGG (reference model.py:534-577): row softmax with a uniform background ---------------------------------
// The reference builds a dense [N,N] logit matrix: e_ij on the entries of edge_index, -1e20 elsewhere, multiplied by the
// dense learned adjacency.  Every pair that is in neither list gets logit -1e20 * 0 = -0, i.e. exp(0) = 1 in the softmax:
// all non-neighbours attend with the same weight.  So a row is `cnt` explicit logits plus (N - cnt) background entries of
// logit 0:  M = max(max_u L_u, 0), Z = sum_u exp(L_u - M) + (N - cnt) exp(-M), att_u = exp(L_u - M) / Z, bg = exp(-M) / Z
// (bg = 0 and M = max_u L_u when the row has no background entry).
__global__ __launch_bounds__(WPB * 64) void bg_softmax_fwd_kernel(const float *__restrict__ L, const int64_t *__restrict__ rowptr,
                                                                 int64_t N, float *__restrict__ att, float *__restrict__ bg) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * WPB + dgg::wave_id();
    if (i >= N) return;
    const int64_t e0 = rowptr[i], e1 = rowptr[i + 1];
    const float nbg = (float)(N - (e1 - e0));
    float m = nbg > 0.0f ? 0.0f : -INFINITY;
    for (int64_t e = e0 + lane; e < e1; e += 64) m = fmaxf(m, L[e]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    float z = 0.0f;
    for (int64_t e = e0 + lane; e < e1; e += 64) z += __expf(L[e] - m);
    z = wave_sum_butterfly(z);
    const float eb = nbg > 0.0f ? __expf(-m) : 0.0f;
    z += nbg * eb;
    const float iz = 1.0f / z;
    for (int64_t e = e0 + lane; e < e1; e += 64) att[e] = __expf(L[e] - m) * iz;
    if (lane == 0) bg[i] = eb * iz;
}
// dL_u = att_u (datt_u - S_i),  S_i = sum_v att_v datt_v + bg_i dbg_i
__global__ __launch_bounds__(WPB * 64) void bg_softmax_bwd_kernel(const float *__restrict__ att, const float *__restrict__ bg,
                                                                 const int64_t *__restrict__ rowptr, int64_t N,
                                                                 const float *__restrict__ datt, const float *__restrict__ dbg,
                                                                 float *__restrict__ dL) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * WPB + dgg::wave_id();
    if (i >= N) return;
    const int64_t e0 = rowptr[i], e1 = rowptr[i + 1];
    float s = 0.0f;
    for (int64_t e = e0 + lane; e < e1; e += 64) s = fmaf(att[e], datt[e], s);
    s = wave_sum_butterfly(s) + bg[i] * dbg[i];
    for (int64_t e = e0 + lane; e < e1; e += 64) dL[e] = att[e] * (datt[e] - s);
}

// ---- GATConv_DGG, training mode: dropout of the DENSE attention matrix (reference model.py:570, F.dropout(attention)) -------------
// Every pair (i, j) of the N x N matrix is kept with probability 1 - p, the non-listed pairs (weight bg_i each) included.  The mask
// is counter-based -- pair (i, j) is kept iff drop_keep(s0, s1 ^ 0x9E3779B9 (i + 1), j) -- so nothing N x N is stored: the forward
// sums the kept rows of h per node, the backward regenerates the mask transposed.
__device__ __forceinline__ bool pair_keep(uint32_t s0, uint32_t s1, uint32_t i, uint32_t j, uint32_t thr24) {
    return drop_keep(s0, s1 ^ (0x9E3779B9u * (i + 1u)), j, thr24);
}
// out_r = sum_s keep(r, s) X_s (transpose 0) or sum_s keep(s, r) X_s (transpose 1); X, out [N,F], F <= FP <= 64 (FP lanes per
// column, 64 / FP columns per wave-instruction); one wavefront per output row
template <int FP>
__global__ __launch_bounds__(WPB * 64) void masked_dense_sum_kernel(const float *__restrict__ X, int64_t N, int F, uint32_t thr24,
                                                                   uint32_t s0, uint32_t s1, int transpose, float *__restrict__ out) {
    constexpr int CPI = 64 / FP;
    const int lane = threadIdx.x & 63, f = lane % FP, sub = lane / FP;
    const int64_t r = (int64_t)blockIdx.x * WPB + dgg::wave_id();
    if (r >= N) return;
    float acc = 0.0f;
    for (int64_t s = sub; s < N; s += CPI) {
        const bool keep = transpose ? pair_keep(s0, s1, (uint32_t)s, (uint32_t)r, thr24) : pair_keep(s0, s1, (uint32_t)r, (uint32_t)s, thr24);
        if (keep && f < F) acc += X[s * F + f];
    }
#pragma unroll
    for (int off = FP; off < 64; off <<= 1) acc += __shfl_xor(acc, off, 64);
    if (sub == 0 && f < F) out[r * F + f] = acc;
}
__global__ void pair_keep_kernel(const int32_t *__restrict__ erow, const int32_t *__restrict__ col, int64_t E, uint32_t thr24, uint32_t s0,
                                 uint32_t s1, float *__restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < E) out[e] = pair_keep(s0, s1, (uint32_t)erow[e], (uint32_t)col[e], thr24) ? 1.0f : 0.0f;
}

}  // namespace

extern "C" {

int dgg_csr_bg_softmax_fwd(const float *L, const int64_t *rowptr, int64_t N, float *att, float *bg, void *stream) {
    if (N == 0) return 0;
    hipLaunchKernelGGL(bg_softmax_fwd_kernel, dim3((unsigned)((N + WPB - 1) / WPB)), dim3(WPB * 64), 0, (hipStream_t)stream, L, rowptr, N,
                       att, bg);
    return dgg_check_launch("csr_bg_softmax_fwd");
}
int dgg_csr_bg_softmax_bwd(const float *att, const float *bg, const int64_t *rowptr, int64_t N, const float *datt, const float *dbg,
                           float *dL, void *stream) {
    if (N == 0) return 0;
    hipLaunchKernelGGL(bg_softmax_bwd_kernel, dim3((unsigned)((N + WPB - 1) / WPB)), dim3(WPB * 64), 0, (hipStream_t)stream, att, bg,
                       rowptr, N, datt, dbg, dL);
    return dgg_check_launch("csr_bg_softmax_bwd");
}

int dgg_masked_dense_sum(const float *X, int64_t N, int F, float p, uint32_t s0, uint32_t s1, int transpose, float *out, void *stream) {
    if (F < 1 || F > 64 || !(p >= 0.0f && p < 1.0f) || N >= ((int64_t)1 << 32))
        return dgg_set_error(DGG_ERR_UNSUPPORTED, "masked_dense_sum: F in [1,64], p in [0,1), N < 2^32");
    if (N == 0) return 0;
    const uint32_t thr = (uint32_t)(p * 16777216.0f);
    const dim3 grid((unsigned)((N + WPB - 1) / WPB)), blk(WPB * 64);
    hipStream_t st = (hipStream_t)stream;
    if (F <= 8) hipLaunchKernelGGL(masked_dense_sum_kernel<8>, grid, blk, 0, st, X, N, F, thr, s0, s1, transpose, out);
    else if (F <= 16) hipLaunchKernelGGL(masked_dense_sum_kernel<16>, grid, blk, 0, st, X, N, F, thr, s0, s1, transpose, out);
    else if (F <= 32) hipLaunchKernelGGL(masked_dense_sum_kernel<32>, grid, blk, 0, st, X, N, F, thr, s0, s1, transpose, out);
    else hipLaunchKernelGGL(masked_dense_sum_kernel<64>, grid, blk, 0, st, X, N, F, thr, s0, s1, transpose, out);
    return dgg_check_launch("masked_dense_sum");
}

int dgg_pair_keep(const int32_t *erow, const int32_t *col, int64_t E, float p, uint32_t s0, uint32_t s1, float *out, void *stream) {
    if (!(p >= 0.0f && p < 1.0f)) return dgg_set_error(DGG_ERR_ARG, "pair_keep: p in [0,1)");
    if (E == 0) return 0;
    hipLaunchKernelGGL(pair_keep_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, (hipStream_t)stream, erow, col, E,
                       (uint32_t)(p * 16777216.0f), s0, s1, out);
    return dgg_check_launch("pair_keep");
}

int dgg_csr_row_sum(const float *vals, const int64_t *rowptr, int64_t N, float *rs, void *stream) {
    if (N == 0) return 0;
    hipLaunchKernelGGL(csr_row_sum_kernel, dim3((unsigned)((N + WPB - 1) / WPB)), dim3(WPB * 64), 0, (hipStream_t)stream, vals, rowptr, N, rs);
    return dgg_check_launch("csr_row_sum");
}

int dgg_csr_normalize_fwd(const int64_t *rowptr, const int32_t *col, const float *w, const float *rs, int64_t N, float *ahat,
                          void *stream) {
    if (N == 0) return 0;
    hipLaunchKernelGGL(csr_normalize_kernel, dim3((unsigned)((N + WPB - 1) / WPB)), dim3(WPB * 64), 0, (hipStream_t)stream, rowptr, col, w,
                       rs, N, ahat);
    return dgg_check_launch("csr_normalize_fwd");
}

int dgg_csr_spmm_fwd(const int64_t *rowptr, const int32_t *col, const float *a, const float *X, int64_t N, int F, float *Y,
                     void *stream) {
    if (N == 0 || F == 0) return 0;
    hipLaunchKernelGGL(csr_spmm_fwd_kernel, dim3((unsigned)((N + WPB - 1) / WPB), (unsigned)((F + 63) / 64)), dim3(WPB * 64), 0,
                       (hipStream_t)stream, rowptr, col, a, X, N, F, Y);
    return dgg_check_launch("csr_spmm_fwd");
}

// dA [E] overwritten; dX (nullable, [N,F]) accumulated into (caller zeroes)
int dgg_csr_spmm_bwd(const int64_t *rowptr, const int32_t *col, const float *a, const float *X, const float *dY, int64_t N, int F,
                     float *dA, float *dX, void *stream) {
    if (N == 0) return 0;
    hipLaunchKernelGGL(csr_spmm_bwd_kernel, dim3((unsigned)((N + WPB - 1) / WPB)), dim3(WPB * 64), 0, (hipStream_t)stream, rowptr, col, a, X,
                       dY, N, F, dA, dX);
    return dgg_check_launch("csr_spmm_bwd");
}

// da_ws [N] zeroed by the caller; dw [E] overwritten
int dgg_csr_norm_bwd(const int64_t *rowptr, const int32_t *col, const float *w, const float *rs, const float *dA, int64_t N,
                     float *da_ws, float *dw, void *stream) {
    if (N == 0) return 0;
    const dim3 grid((unsigned)((N + WPB - 1) / WPB));
    hipLaunchKernelGGL(csr_norm_bwd_da_kernel, grid, dim3(WPB * 64), 0, (hipStream_t)stream, rowptr, col, w, rs, dA, N, da_ws);
    hipLaunchKernelGGL(csr_norm_bwd_dw_kernel, grid, dim3(WPB * 64), 0, (hipStream_t)stream, rowptr, col, rs, dA, da_ws, N, dw);
    return dgg_check_launch("csr_norm_bwd");
}

// select_top_k on the CSR pattern (rows of any width; dgm.py:1402-1435): p [E] edge probabilities, k [N] learned degrees;
// noise_mode 0 none / 1 explicit G [N, ldG] / 2 hash / 3 symmetric hash; mode 0 k_times_edge_prob, 1 k_only.
// -> w [E], pp [E] (perturbed probabilities), pos [E] (rank of the entry in its row)
int dgg_csr_softk_fwd(const float *p, const int64_t *rowptr, const int32_t *col, int64_t N, const float *k, int noise_mode, const float *G,
                      int64_t ldG, uint32_t s0, uint32_t s1, int mode, float *w, float *pp, int32_t *pos, void *stream) {
    if (noise_mode < 0 || noise_mode > 3 || (noise_mode == 1 && !G)) return dgg_set_error(DGG_ERR_ARG, "csr_softk_fwd: bad noise arguments");
    if (mode != 0 && mode != 1) return dgg_set_error(DGG_ERR_ARG, "csr_softk_fwd: mode must be 0 (k_times_edge_prob) or 1 (k_only)");
    if (N == 0) return 0;
    hipLaunchKernelGGL(csr_softk_fwd_kernel, dim3((unsigned)((N + WPB - 1) / WPB)), dim3(WPB * 64), 0, (hipStream_t)stream, p, rowptr, col, N, k,
                       noise_mode, G, ldG, s0, s1, mode, w, pp, pos);
    return dgg_check_launch("csr_softk_fwd");
}
int dgg_csr_softk_bwd(const float *p, const float *pp, const int64_t *rowptr, int64_t N, const float *k, const int32_t *pos, int perturb,
                      int mode, const float *g, float *dp, float *dk, void *stream) {
    if (N == 0) return 0;
    hipLaunchKernelGGL(csr_softk_bwd_kernel, dim3((unsigned)((N + WPB - 1) / WPB)), dim3(WPB * 64), 0, (hipStream_t)stream, p, pp, rowptr, N, k,
                       pos, perturb, mode, g, dp, dk);
    return dgg_check_launch("csr_softk_bwd");
}

int dgg_csr_rank_ramp_fwd(const float *p, const int64_t *rowptr, const int32_t *col, int64_t N, const float *w, const float *b,
                          float *out, float *S, float *k, int32_t *pos, void *stream) {
    if (N == 0) return 0;
    hipLaunchKernelGGL(csr_rank_ramp_fwd_kernel, dim3((unsigned)((N + WPB - 1) / WPB)), dim3(WPB * 64), 0, (hipStream_t)stream, p, rowptr,
                       col, N, w, b, out, S, k, pos);
    return dgg_check_launch("csr_rank_ramp_fwd");
}

int dgg_csr_rank_ramp_bwd(const float *p, const int64_t *rowptr, int64_t N, const float *w, const float *b, const float *S,
                          const float *k, const int32_t *pos, const float *g, float *dp, float *dkz, void *stream) {
    if (N == 0) return 0;
    hipLaunchKernelGGL(csr_rank_ramp_bwd_kernel, dim3((unsigned)((N + WPB - 1) / WPB)), dim3(WPB * 64), 0, (hipStream_t)stream, p, rowptr,
                       N, w, b, S, k, pos, g, dp, dkz);
    return dgg_check_launch("csr_rank_ramp_bwd");
}

int dgg_csr_noisy_sigmoid_fwd(const float *p, const float *noise, int64_t E, float *out, void *stream) {
    if (E == 0) return 0;
    hipLaunchKernelGGL(csr_noisy_sigmoid_fwd_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, noise, E, out);
    return dgg_check_launch("csr_noisy_sigmoid_fwd");
}

int dgg_csr_noisy_sigmoid_bwd(const float *out, const float *g, int64_t E, float *dp, void *stream) {
    if (E == 0) return 0;
    hipLaunchKernelGGL(csr_noisy_sigmoid_bwd_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, g, E, dp);
    return dgg_check_launch("csr_noisy_sigmoid_bwd");
}

int dgg_csr_rank_cut_fwd(const float *p, const int64_t *rowptr, const int32_t *col, int64_t N, int kcut, float *out, int32_t *pos,
                         void *stream) {
    if (N == 0) return 0;
    if (kcut < 0) return dgg_set_error(DGG_ERR_ARG, "csr_rank_cut_fwd: kcut must be >= 0");
    hipLaunchKernelGGL(csr_rank_cut_fwd_kernel, dim3((unsigned)((N + WPB - 1) / WPB)), dim3(WPB * 64), 0, (hipStream_t)stream, p, rowptr,
                       col, N, kcut, out, pos);
    return dgg_check_launch("csr_rank_cut_fwd");
}

int dgg_csr_rank_cut_bwd(const int32_t *pos, const float *g, int64_t E, int kcut, float *dp, void *stream) {
    if (E == 0) return 0;
    hipLaunchKernelGGL(csr_rank_cut_bwd_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pos, g, E, kcut, dp);
    return dgg_check_launch("csr_rank_cut_bwd");
}

int dgg_csr_uvdist_fwd(const float *xp, const int64_t *rowptr, const int32_t *col, int64_t N, int h, float t, float *p, void *stream) {
    if (N == 0) return 0;
    if (h < 1) return dgg_set_error(DGG_ERR_ARG, "csr_uvdist_fwd: h must be positive");
    hipLaunchKernelGGL(csr_uvdist_fwd_kernel, dim3((unsigned)((N + WPB - 1) / WPB)), dim3(WPB * 64), 0, (hipStream_t)stream, xp, rowptr, col,
                       N, h, t, p);
    return dgg_check_launch("csr_uvdist_fwd");
}

int dgg_csr_uvdist_bwd(const float *xp, const int64_t *rowptr, const int32_t *col, int64_t N, int h, float t, const float *p,
                       const float *dp, float *dxp, void *stream) {
    if (N == 0) return 0;
    hipLaunchKernelGGL(csr_uvdist_bwd_kernel, dim3((unsigned)((N + WPB - 1) / WPB)), dim3(WPB * 64), 0, (hipStream_t)stream, xp, rowptr, col,
                       N, h, t, p, dp, dxp);
    return dgg_check_launch("csr_uvdist_bwd");
}

}  // extern "C"
