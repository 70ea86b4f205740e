// dgg_ell.hip -- everything after selection, on the fixed-width ELL adjacency (idx int32 [N,K], K <= 64).
//
// Replaces the dense [N,N] tails of the reference:
//   smooth first-k ramp * score, unsort         reference dgm.py:1410-1420 (k_times_edge_prob), 1427-1434 (k_only)
//   normalize_adj  D^-1/2 A D^-1/2 (row sums)   reference model.py:1205-1219
//   torch.mm(adj, x) / torch.spmm(adj, input)   reference model.py:594, 67, 34
// and the autograd of all of it (SDDMM, transposed SpMM, ramp/normalisation chain, score -> features).
// One wavefront owns one row: ELL entry r lives in lane r, feature vectors are spread over lanes, row
// reductions are 64-lane butterflies.
#include "dgg_common.h"
#include "dgg_api_internal.h"

using namespace dgg;

namespace {

constexpr int WPB = 4;   // wavefronts (rows) per workgroup

__device__ __forceinline__ float inv_sqrt_c(float rs) { return __fdiv_rn(1.0f, c_sqrt(rs)); }

// w = score * ramp (mode 0), ramp (mode 1) or the straight-through value (ramp - score * ramp) + score * ramp (mode 3: forward
// value of `(hard - soft).detach() + soft` with hard = the ramp mask, dgm.py:343-346); rs = row sum (butterfly order)
__global__ __launch_bounds__(WPB * 64) void softk_fwd_kernel(const int32_t *__restrict__ idx, const float *__restrict__ val,
                                                            const float *__restrict__ k, int64_t N, int K, int mode,
                                                            float *__restrict__ w, float *__restrict__ rs) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * WPB + dgg::wave_id();
    if (i >= N) return;
    float wv = 0.0f;
    if (lane < K) {
        float f = c_ramp((float)lane, k[i]);
        float v = f;
        if (mode == 0 || mode == 3) {
            const float a = __fmul_rn(val[i * K + lane], f);
            v = mode == 0 ? a : __fadd_rn(__fadd_rn(f, -a), a);
        }
        wv = idx[i * K + lane] >= 0 ? v : 0.0f;
        w[i * K + lane] = wv;
    }
    float s = wave_sum_butterfly(wv);
    if (lane == 0) rs[i] = s;
}

// ahat_ir = (a_i * w_ir) * a_j,  a = 1/sqrt(rs);  rs is indexed by GLOBAL node id, local row i is node row0+i
__global__ void normalize_fwd_kernel(const int32_t *__restrict__ idx, const float *__restrict__ w,
                                     const float *__restrict__ rs, int64_t N, int K, int64_t row0,
                                     float *__restrict__ ahat) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N * K) return;
    int64_t i = e / K;
    int32_t j = idx[e];
    float out = 0.0f;
    if (j >= 0) {
        float ai = inv_sqrt_c(rs[row0 + i]);
        float aj = inv_sqrt_c(rs[j]);
        out = __fmul_rn(__fmul_rn(ai, w[e]), aj);
    }
    ahat[e] = out;
}

// Y_i[c] = sum_r ahat_ir * X[idx_ir][c], r ascending (fmaf chain).  grid.y walks feature blocks of 64*VEC.
template <int VEC>
__global__ __launch_bounds__(WPB * 64) void spmm_fwd_kernel(const int32_t *__restrict__ idx, const float *__restrict__ ahat,
                                                           const float *__restrict__ X, int64_t N, int K, int F,
                                                           float *__restrict__ Y, __bf16 *__restrict__ Yb = nullptr, int64_t ldyb = 0,
                                                           const int32_t *__restrict__ cptr = nullptr) {
    // (cptr != NULL: chunked rows, K = 64 -- row i is the chunks [cptr[i], cptr[i+1]) of idx / ahat, walked in rank order)
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * WPB + dgg::wave_id();
    if (i >= N) return;
    const int c0 = (blockIdx.y * 64 + lane) * VEC;
    const int64_t cb = cptr ? (int64_t)cptr[i] : i, ce = cptr ? (int64_t)cptr[i + 1] : i + 1;
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; v++) acc[v] = 0.0f;
    for (int64_t ch = cb; ch < ce; ch++) {
    int32_t jl = lane < K ? idx[ch * K + lane] : -1;
    float al = lane < K ? ahat[ch * K + lane] : 0.0f;
    for (int r = 0; r < K; r++) {
        int32_t j = bcast(jl, r);
        float a = bcast(al, r);
        if (j < 0 || a == 0.0f) continue;            // wave-uniform; fmaf(0, x, acc) == acc
        if (c0 < F) {
            const float *xr = X + (int64_t)j * F + c0;
            if (VEC == 4) {
                float4 xv = *reinterpret_cast<const float4 *>(xr);
                acc[0] = __fmaf_rn(a, xv.x, acc[0]); acc[1 % VEC] = __fmaf_rn(a, xv.y, acc[1 % VEC]);
                acc[2 % VEC] = __fmaf_rn(a, xv.z, acc[2 % VEC]); acc[3 % VEC] = __fmaf_rn(a, xv.w, acc[3 % VEC]);
            } else if (VEC == 2) {
                float2 xv = *reinterpret_cast<const float2 *>(xr);
                acc[0] = __fmaf_rn(a, xv.x, acc[0]); acc[1 % VEC] = __fmaf_rn(a, xv.y, acc[1 % VEC]);
            } else {
                acc[0] = __fmaf_rn(a, xr[0], acc[0]);
            }
        }
    }
    }
    if (c0 < F) {
#pragma unroll
        for (int v = 0; v < VEC; v++) Y[i * F + c0 + v] = acc[v];
        if (Yb) {                                    // the bf16 copy the layer product reads (GCNII stack: no separate pack pass)
#pragma unroll
            for (int v = 0; v < VEC; v++) Yb[i * ldyb + c0 + v] = (__bf16)acc[v];
        }
    }
}

// bf16 helpers: two values per 32-bit word, low half first
__device__ __forceinline__ float bf_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

// The aggregation of the fused GCNII stack on a bf16 COPY of the activation (half the gathered bytes: the stack's gather kernels
// are bound by L2 bandwidth at F = 2048): lane l owns 8 consecutive features (one 16-byte load per neighbour), grid.y walks blocks
// of 512 features; fp32 accumulation in entry order; writes Y fp32 and, if asked, bf16(Y).
__global__ __launch_bounds__(WPB * 64) void spmm_fwd_b16_kernel(const int32_t *__restrict__ idx, const float *__restrict__ ahat,
                                                               const uint4 *__restrict__ Xb, int64_t N, int K, int F,
                                                               float *__restrict__ Y, __bf16 *__restrict__ Yb, int64_t ldyb) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * WPB + dgg::wave_id();
    if (i >= N) return;
    const int c8 = blockIdx.y * 64 + lane;                        // index of this lane's group of 8 features
    const int F8 = F / 8;
    int32_t jl = lane < K ? idx[i * K + lane] : -1;
    float al = lane < K ? ahat[i * K + lane] : 0.0f;
    float acc[8];
#pragma unroll
    for (int v = 0; v < 8; v++) acc[v] = 0.0f;
    for (int r = 0; r < K; r++) {
        const int32_t j = bcast(jl, r);
        const float a = bcast(al, r);
        if (j < 0 || a == 0.0f) continue;                         // wave-uniform
        if (c8 < F8) {
            const uint4 xv = Xb[(int64_t)j * F8 + c8];
            acc[0] = __fmaf_rn(a, bf_lo(xv.x), acc[0]); acc[1] = __fmaf_rn(a, bf_hi(xv.x), acc[1]);
            acc[2] = __fmaf_rn(a, bf_lo(xv.y), acc[2]); acc[3] = __fmaf_rn(a, bf_hi(xv.y), acc[3]);
            acc[4] = __fmaf_rn(a, bf_lo(xv.z), acc[4]); acc[5] = __fmaf_rn(a, bf_hi(xv.z), acc[5]);
            acc[6] = __fmaf_rn(a, bf_lo(xv.w), acc[6]); acc[7] = __fmaf_rn(a, bf_hi(xv.w), acc[7]);
        }
    }
    if (c8 < F8) {
        float4 *yo = reinterpret_cast<float4 *>(Y + i * F + 8 * c8);
        yo[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
        yo[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
        if (Yb) {
#pragma unroll
            for (int v = 0; v < 8; v++) Yb[i * ldyb + 8 * c8 + v] = (__bf16)acc[v];
        }
    }
}

// sddmm_wide_kernel on bf16 copies of both operands (X gathered, dY the row's own cotangent), fp32 accumulation
__global__ __launch_bounds__(WPB * 64) void sddmm_wide_b16_kernel(const int32_t *__restrict__ idx, const float *__restrict__ ahat,
                                                                 const uint4 *__restrict__ Xb, const uint4 *__restrict__ dYb,
                                                                 int64_t N, int K, int F, int skip_zero, float *__restrict__ dA) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * WPB + dgg::wave_id();
    if (i >= N) return;
    const int32_t jl = lane < K ? idx[i * K + lane] : -1;
    const float al = lane < K ? ahat[i * K + lane] : 0.0f;
    const int F8 = F / 8;
    const uint4 *gy = dYb + i * F8;
    float mine = 0.0f;
    constexpr int NQ = 4;
    for (int r0 = 0; r0 < K; r0 += NQ) {
        const uint4 *xr[NQ];
        bool v[NQ];
        bool any = false;
        float part[NQ];
#pragma unroll
        for (int u = 0; u < NQ; u++) {
            const int r = r0 + u < K ? r0 + u : K - 1;
            const int32_t j = bcast(jl, r);
            v[u] = r0 + u < K && j >= 0 && !(skip_zero && bcast(al, r) == 0.0f);
            xr[u] = Xb + (int64_t)(j < 0 ? 0 : j) * F8;
            part[u] = 0.0f;
            any = any || v[u];
        }
        if (!any) continue;                                      // wave-uniform
        for (int c = lane; c < F8; c += 64) {
            const uint4 g = gy[c];
            const float g0 = bf_lo(g.x), g1 = bf_hi(g.x), g2 = bf_lo(g.y), g3 = bf_hi(g.y), g4 = bf_lo(g.z), g5 = bf_hi(g.z), g6 = bf_lo(g.w),
                        g7 = bf_hi(g.w);
#pragma unroll
            for (int u = 0; u < NQ; u++) {
                if (v[u]) {
                    const uint4 xv = xr[u][c];
                    float p_ = part[u];
                    p_ = fmaf(g0, bf_lo(xv.x), p_); p_ = fmaf(g1, bf_hi(xv.x), p_); p_ = fmaf(g2, bf_lo(xv.y), p_); p_ = fmaf(g3, bf_hi(xv.y), p_);
                    p_ = fmaf(g4, bf_lo(xv.z), p_); p_ = fmaf(g5, bf_hi(xv.z), p_); p_ = fmaf(g6, bf_lo(xv.w), p_); p_ = fmaf(g7, bf_hi(xv.w), p_);
                    part[u] = p_;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < NQ; u++) {
            const float tot = wave_sum_dpp(part[u], lane);
            if (lane == r0 + u) mine = tot;
        }
    }
    if (lane < K) dA[i * K + lane] = mine;
}

// The SDDMM of the stack one 512-feature SLICE at a time (grid.y; x runs fastest, so the whole chip works on one slice before the
// next): a slice of the gathered activation is 1 KB per node -- 1.8 MB for a PPI graph, resident in an XCD's 4 MB L2 -- where whole
// 4 KB rows (7.3 MB) are not: the unsliced kernel gathered at 3.4 TB/s.  One 16-byte load per lane and neighbour; the slice's partial
// dot products go to part[slice][N*K] and sddmm_slices_sum adds the slices in slice order (deterministic, no float atomics).
__global__ __launch_bounds__(WPB * 64) void sddmm_slice_b16_kernel(const int32_t *__restrict__ idx, const float *__restrict__ ahat,
                                                                  const uint4 *__restrict__ Xb, const uint4 *__restrict__ dYb,
                                                                  int64_t N, int K, int F, int skip_zero, float *__restrict__ part,
                                                                  int accumulate) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * WPB + dgg::wave_id();
    if (i >= N) return;
    const int32_t jl = lane < K ? idx[i * K + lane] : -1;
    const float al = lane < K ? ahat[i * K + lane] : 0.0f;
    const int F8 = F / 8, c = blockIdx.y * 64 + lane;
    const uint4 g = dYb[i * F8 + c];
    const float g0 = bf_lo(g.x), g1 = bf_hi(g.x), g2 = bf_lo(g.y), g3 = bf_hi(g.y), g4 = bf_lo(g.z), g5 = bf_hi(g.z), g6 = bf_lo(g.w),
                g7 = bf_hi(g.w);
    float mine = 0.0f;
    constexpr int NQ = 8;
    for (int r0 = 0; r0 < K; r0 += NQ) {
        uint4 xv[NQ];
        bool v[NQ];
        bool any = false;
#pragma unroll
        for (int u = 0; u < NQ; u++) {
            const int r = r0 + u < K ? r0 + u : K - 1;
            const int32_t j = bcast(jl, r);
            v[u] = r0 + u < K && j >= 0 && !(skip_zero && bcast(al, r) == 0.0f);
            any = any || v[u];
            xv[u] = make_uint4(0, 0, 0, 0);
            if (v[u]) xv[u] = Xb[(int64_t)j * F8 + c];             // (wave-uniform predicate)
        }
        if (!any) continue;                                      // wave-uniform
#pragma unroll
        for (int u = 0; u < NQ; u++) {
            float p_ = 0.0f;
            p_ = fmaf(g0, bf_lo(xv[u].x), p_); p_ = fmaf(g1, bf_hi(xv[u].x), p_); p_ = fmaf(g2, bf_lo(xv[u].y), p_); p_ = fmaf(g3, bf_hi(xv[u].y), p_);
            p_ = fmaf(g4, bf_lo(xv[u].z), p_); p_ = fmaf(g5, bf_hi(xv[u].z), p_); p_ = fmaf(g6, bf_lo(xv[u].w), p_); p_ = fmaf(g7, bf_hi(xv[u].w), p_);
            const float tot = wave_sum_dpp(p_, lane);
            if (lane == r0 + u) mine = tot;
        }
    }
    if (lane < K) {                                              // (accumulate: this wavefront is the only writer of its (slice, row) words)
        float *o = part + (int64_t)blockIdx.y * N * K + i * K + lane;
        *o = accumulate ? *o + mine : mine;
    }
}
// dA[e] (+)= sum over the slices, in slice order
__global__ void sddmm_slices_sum(const float *__restrict__ part, int64_t n, int S, int accumulate, float *__restrict__ dA) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float s_ = part[e];
    for (int q = 1; q < S; q++) s_ += part[(int64_t)q * n + e];
    dA[e] = accumulate ? dA[e] + s_ : s_;
}

// Same aggregation for NARROW feature rows (F = 16, 32, 64: the projected features H = X W of a GCNConv whose output is
// narrower than its input, aggregated after the projection).  F/4 lanes per gathered row (16-byte loads), 256/F rows per
// wave-instruction and NBT such batches in flight, so that one load instruction still moves 1 KiB; the 256/F partial sums
// of a feature are combined by an xor butterfly at the end (order: entry r goes to slot r mod (256/F), slots summed pairwise).
// ACT 2 applies the ReLU of GCNConv (model.py:598) in the epilogue.
// CHUNKED (rows wider than 64 ranks, K = 64): row i is the chunks [cptr[i], cptr[i+1]) of idx / ahat, walked in rank order.
template <int F, bool CHUNKED = false>
__global__ __launch_bounds__(WPB * 64) void spmm_fwd_narrow(const int32_t *__restrict__ idx, const float *__restrict__ ahat,
                                                           const float *__restrict__ X, int64_t N, int K, int act,
                                                           float *__restrict__ Y, const int32_t *__restrict__ cptr = nullptr) {
    constexpr int LPR = F / 4, NPI = 64 / LPR, NBT = 4;
    const int lane = threadIdx.x & 63, c4 = lane % LPR, slot = lane / LPR;
    const int64_t i = (int64_t)blockIdx.x * WPB + dgg::wave_id();
    if (i >= N) return;
    const int64_t cb = CHUNKED ? (int64_t)cptr[i] : i, ce = CHUNKED ? (int64_t)cptr[i + 1] : i + 1;
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (int64_t ch = cb; ch < ce; ch++) {
    const int32_t jl = lane < K ? idx[ch * K + lane] : -1;
    const float al = lane < K ? ahat[ch * K + lane] : 0.0f;
    for (int r0 = 0; r0 < K; r0 += NBT * NPI) {
        int32_t j[NBT];
        float a[NBT];
        bool any = false;
#pragma unroll
        for (int b = 0; b < NBT; b++) {
            const int r = r0 + b * NPI + slot;
            const int rr = r < 64 ? r : 63;
            j[b] = __shfl(jl, rr, 64);
            a[b] = __shfl(al, rr, 64);
            if (r >= K || j[b] < 0) a[b] = 0.0f;
            any = any || a[b] != 0.0f;
        }
        if (__ballot(any) == 0ull) continue;                     // wave-uniform
        float4 xv[NBT];
#pragma unroll
        for (int b = 0; b < NBT; b++)                            // unconditional gathers (inactive: row 0, weight 0)
            xv[b] = *reinterpret_cast<const float4 *>(X + (int64_t)(a[b] != 0.0f ? j[b] : 0) * F + 4 * c4);
#pragma unroll
        for (int b = 0; b < NBT; b++) {
            acc.x = fmaf(a[b], xv[b].x, acc.x); acc.y = fmaf(a[b], xv[b].y, acc.y);
            acc.z = fmaf(a[b], xv[b].z, acc.z); acc.w = fmaf(a[b], xv[b].w, acc.w);
        }
    }
    }
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) {
        acc.x += __shfl_xor(acc.x, off, 64); acc.y += __shfl_xor(acc.y, off, 64);
        acc.z += __shfl_xor(acc.z, off, 64); acc.w += __shfl_xor(acc.w, off, 64);
    }
    if (slot == 0) {
        if (act == 2) { acc.x = fmaxf(acc.x, 0.0f); acc.y = fmaxf(acc.y, 0.0f); acc.z = fmaxf(acc.z, 0.0f); acc.w = fmaxf(acc.w, 0.0f); }
        *reinterpret_cast<float4 *>(Y + i * F + 4 * c4) = acc;
    }
}

// y = act(y) in place (wide rows, where the aggregation kernels have no fused epilogue)
__global__ void act_inplace_kernel(float *__restrict__ y, int64_t n, int act) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n && act == 2) y[e] = fmaxf(y[e], 0.0f);
}

// dA_ir = <dY_i, X_j>;  dX_j += ahat_ir * dY_i (fp32 atomics, optional).  One wavefront per row, features on lanes.
__global__ __launch_bounds__(WPB * 64) void spmm_bwd_kernel(const int32_t *__restrict__ idx, const float *__restrict__ ahat,
                                                           const float *__restrict__ X, const float *__restrict__ dY,
                                                           int64_t N, int K, int F, int skip_zero,
                                                           float *__restrict__ dA, float *__restrict__ dX) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * WPB + dgg::wave_id();
    if (i >= N) return;
    int32_t jl = lane < K ? idx[i * K + lane] : -1;
    float al = lane < K ? ahat[i * K + lane] : 0.0f;
    float mine = 0.0f;
    // the row's cotangent stays in registers (F <= 256) instead of being re-read for every neighbour
    const bool small = F <= 256;
    float g0 = 0.0f, g1 = 0.0f, g2 = 0.0f, g3 = 0.0f;
    if (small) {
        if (lane < F) g0 = dY[i * F + lane];
        if (lane + 64 < F) g1 = dY[i * F + lane + 64];
        if (lane + 128 < F) g2 = dY[i * F + lane + 128];
        if (lane + 192 < F) g3 = dY[i * F + lane + 192];
    }
    for (int r = 0; r < K; r++) {
        int32_t j = bcast(jl, r);
        float a = bcast(al, r);
        if (j < 0 || (skip_zero && a == 0.0f)) continue;       // wave-uniform
        float part = 0.0f;
        const float *xr = X + (int64_t)j * F;
        if (small) {
            if (lane < F) part = fmaf(g0, xr[lane], part);
            if (lane + 64 < F) part = fmaf(g1, xr[lane + 64], part);
            if (lane + 128 < F) part = fmaf(g2, xr[lane + 128], part);
            if (lane + 192 < F) part = fmaf(g3, xr[lane + 192], part);
            if (dX && a != 0.0f) {
                if (lane < F) atomicAdd(dX + (int64_t)j * F + lane, a * g0);
                if (lane + 64 < F) atomicAdd(dX + (int64_t)j * F + lane + 64, a * g1);
                if (lane + 128 < F) atomicAdd(dX + (int64_t)j * F + lane + 128, a * g2);
                if (lane + 192 < F) atomicAdd(dX + (int64_t)j * F + lane + 192, a * g3);
            }
        } else {
            for (int c = lane; c < F; c += 64) {
                float g = dY[i * F + c];
                part = fmaf(g, xr[c], part);
                if (dX && a != 0.0f) atomicAdd(dX + (int64_t)j * F + c, a * g);
            }
        }
        part = wave_sum_dpp(part, lane);
        if (lane == r) mine = part;
    }
    if (lane < K) dA[i * K + lane] = mine;
}

// The same for a handful of features (F <= 8: the class scores of a model's LAST graph convolution, model.py:1290): entries on
// lanes -- each lane forms its entry's dot product and its F atomics itself; no per-entry wavefront reduction, no serial walk of
// the 64 slots (38 -> 9 us on the Pubmed shape with F = 3).
template <int FMAX>
__global__ __launch_bounds__(WPB * 64) void spmm_bwd_tiny_kernel(const int32_t *__restrict__ idx, const float *__restrict__ ahat,
                                                                const float *__restrict__ X, const float *__restrict__ dY,
                                                                int64_t N, int K, int F, int skip_zero,
                                                                float *__restrict__ dA, float *__restrict__ dX) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * WPB + dgg::wave_id();
    if (i >= N) return;
    float g[FMAX];
#pragma unroll
    for (int q = 0; q < FMAX; q++) g[q] = q < F ? dY[i * F + q] : 0.0f;      // (one address for the whole wavefront)
    for (int r = lane; r < K; r += 64) {
        const int32_t j = idx[i * K + r];
        const float a = ahat[i * K + r];
        float dot = 0.0f;
        if (j >= 0 && !(skip_zero && a == 0.0f)) {
#pragma unroll
            for (int q = 0; q < FMAX; q++) {
                if (q < F) {
                    dot = fmaf(g[q], X[(int64_t)j * F + q], dot);
                    if (dX && a != 0.0f) atomicAdd(dX + (int64_t)j * F + q, a * g[q]);
                }
            }
        }
        dA[i * K + r] = dot;
    }
}

// SDDMM only (no dX) for wide rows (F > 256, a multiple of 4; GCNII layers of the PPI configuration: 2048): the wavefront
// walks the row's cotangent ONCE per batch of four neighbours -- 16-byte loads, four gathered rows in flight -- instead of
// once per neighbour.
__global__ __launch_bounds__(WPB * 64) void sddmm_wide_kernel(const int32_t *__restrict__ idx, const float *__restrict__ ahat,
                                                             const float *__restrict__ X, const float *__restrict__ dY,
                                                             int64_t N, int K, int F, int skip_zero, float *__restrict__ dA) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * WPB + dgg::wave_id();
    if (i >= N) return;
    const int32_t jl = lane < K ? idx[i * K + lane] : -1;
    const float al = lane < K ? ahat[i * K + lane] : 0.0f;
    const float4 *gy = reinterpret_cast<const float4 *>(dY + i * F);
    const int F4 = F / 4;
    float mine = 0.0f;
    constexpr int NQ = 4;
    for (int r0 = 0; r0 < K; r0 += NQ) {
        const float4 *xr[NQ];
        bool v[NQ];
        bool any = false;
        float part[NQ];
#pragma unroll
        for (int u = 0; u < NQ; u++) {
            const int r = r0 + u < K ? r0 + u : K - 1;
            const int32_t j = bcast(jl, r);
            v[u] = r0 + u < K && j >= 0 && !(skip_zero && bcast(al, r) == 0.0f);
            xr[u] = reinterpret_cast<const float4 *>(X + (int64_t)(j < 0 ? 0 : j) * F);
            part[u] = 0.0f;
            any = any || v[u];
        }
        if (!any) continue;                                      // wave-uniform
        for (int c = lane; c < F4; c += 64) {
            const float4 g = gy[c];
#pragma unroll
            for (int u = 0; u < NQ; u++) {
                if (v[u]) {
                    const float4 xv = xr[u][c];
                    part[u] = fmaf(g.x, xv.x, part[u]); part[u] = fmaf(g.y, xv.y, part[u]);
                    part[u] = fmaf(g.z, xv.z, part[u]); part[u] = fmaf(g.w, xv.w, part[u]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < NQ; u++) {
            const float tot = wave_sum_dpp(part[u], lane);
            if (lane == r0 + u) mine = tot;
        }
    }
    if (lane < K) dA[i * K + lane] = mine;
}

// SDDMM only (no dX), F = 128*NV4: TWO neighbours per wave-instruction.  Each 32-lane half owns one neighbour and each
// lane 4*NV4 features (16-byte loads: a 128-feature row is one 512-byte coalesced segment per half), so the per-edge
// reduction is a 5-step DPP butterfly inside the half and the instruction count per edge halves.
// NORM: also the row-side half of the normalisation backward, which needs nothing but dA, ahat and w of the row:
//   da_i = sum_r dA_ir w_ir a_j = (sum_r dA_ir ahat_ir) / a_i   (plain store),
//   coef_ir = dA_ir w_ir a_i  written in the destination-ordered record order of the partition (dgg_scatter.hip),
// which removes a separate pass over idx / w / dA and its 4-byte gathers of rs[j].
template <int NV4, bool NORM>
__global__ __launch_bounds__(WPB * 64) void sddmm_pair_kernel(const int32_t *__restrict__ idx, const float *__restrict__ ahat,
                                                             const float *__restrict__ X, const float *__restrict__ dY,
                                                             int64_t N, int K, int skip_zero, float *__restrict__ dA,
                                                             const float *__restrict__ w, const float *__restrict__ rs,
                                                             int64_t row0, const int *__restrict__ slotmap,
                                                             float *__restrict__ coef, float *__restrict__ da) {
    constexpr int F = 128 * NV4;
    const int lane = threadIdx.x & 63, sub = lane & 31, hh = lane >> 5;
    const int64_t i = (int64_t)blockIdx.x * WPB + dgg::wave_id();
    if (i >= N) return;
    int32_t jl = lane < K ? idx[i * K + lane] : -1;
    float al = lane < K ? ahat[i * K + lane] : 0.0f;
    float4 g[NV4];
#pragma unroll
    for (int q = 0; q < NV4; q++) g[q] = *reinterpret_cast<const float4 *>(dY + i * F + q * 128 + sub * 4);
    float mine = 0.0f;
    for (int r = 0; r < K; r += 2) {
        const int32_t j0 = bcast(jl, r), j1 = bcast(jl, r + 1 < 64 ? r + 1 : 63);
        const float a0 = bcast(al, r), a1 = bcast(al, r + 1 < 64 ? r + 1 : 63);
        const bool v0 = j0 >= 0 && !(skip_zero && a0 == 0.0f);
        const bool v1 = r + 1 < K && j1 >= 0 && !(skip_zero && a1 == 0.0f);
        if (!v0 && !v1) continue;                                // wave-uniform
        const int32_t j = hh ? j1 : j0;
        const bool v = hh ? v1 : v0;
        float part = 0.0f;
        if (v) {
            const float *xr = X + (int64_t)j * F + sub * 4;
#pragma unroll
            for (int q = 0; q < NV4; q++) {
                float4 xv = *reinterpret_cast<const float4 *>(xr + q * 128);
                part = fmaf(g[q].x, xv.x, part); part = fmaf(g[q].y, xv.y, part);
                part = fmaf(g[q].z, xv.z, part); part = fmaf(g[q].w, xv.w, part);
            }
        }
        part += __uint_as_float(xor_shfl<16>(__float_as_uint(part), lane));
        part += __uint_as_float(xor_shfl<8>(__float_as_uint(part), lane));
        part += __uint_as_float(xor_shfl<4>(__float_as_uint(part), lane));
        part += __uint_as_float(xor_shfl<2>(__float_as_uint(part), lane));
        part += __uint_as_float(xor_shfl<1>(__float_as_uint(part), lane));
        const float t0 = bcast(part, 0), t1 = bcast(part, 32);
        if (lane == r) mine = t0;
        if (lane == r + 1) mine = t1;
    }
    if (lane < K) dA[i * K + lane] = mine;
    if (NORM) {
        const float ri = rs[row0 + i];
        const float ai = 1.0f / sqrtf(ri);
        float rowpart = mine * al;                               // lanes >= K hold 0
        rowpart = wave_sum_dpp(rowpart, lane);
        if (lane == 0) da[row0 + i] = rowpart * sqrtf(ri);
        if (lane < K) {
            const int sl = slotmap[i * K + lane];
            if (sl >= 0) coef[sl] = mine * w[i * K + lane] * ai;
        }
    }
}

// normalisation backward, phase 1: da[i] += sum_r dA_ir w_ir a_j ;  da[j] += dA_ir w_ir a_i   (da zeroed by caller)
__global__ __launch_bounds__(WPB * 64) void norm_bwd_da_kernel(const int32_t *__restrict__ idx, const float *__restrict__ w,
                                                              const float *__restrict__ rs, const float *__restrict__ dA,
                                                              int64_t N, int K, int64_t row0, float *__restrict__ da) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * WPB + dgg::wave_id();
    if (i >= N) return;
    float rowpart = 0.0f;
    if (lane < K) {
        int32_t j = idx[i * K + lane];
        if (j >= 0) {
            float g = dA[i * K + lane] * w[i * K + lane];
            if (g != 0.0f) {
                float ai = inv_sqrt_c(rs[row0 + i]), aj = inv_sqrt_c(rs[j]);
                rowpart = g * aj;
                atomicAdd(da + j, g * ai);
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) rowpart += __shfl_xor(rowpart, off, 64);
    if (lane == 0 && rowpart != 0.0f) atomicAdd(da + row0 + i, rowpart);
}

// phase 2: dw = dA a_i a_j + drs_i (normalized) or dw = dA (not normalized);  dval = dw * f ; dk = sum dw * s * f'
__global__ __launch_bounds__(WPB * 64) void softk_bwd_kernel(const int32_t *__restrict__ idx, const float *__restrict__ val,
                                                            const float *__restrict__ k, const float *__restrict__ rs,
                                                            const float *__restrict__ dA, const float *__restrict__ da,
                                                            int64_t N, int K, int64_t row0, int mode, int normalized,
                                                            float *__restrict__ dval, float *__restrict__ dk,
                                                            const float *__restrict__ ahat_rows) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * WPB + dgg::wave_id();
    if (i >= N) return;
    float skp = 0.0f;
    // ahat_rows: `da` carries the neighbour-side sums only (the column kernels of the partitioned backward); the row side
    // sum_r dA_ir w_ir a_j = rs_i^1/2 sum_r dA_ir ahat_ir is formed here (lane r owns entry r)
    float darow = 0.0f;
    if (normalized && ahat_rows) {
        for (int r = lane; r < K; r += 64) darow = fmaf(dA[i * K + r], ahat_rows[i * K + r], darow);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) darow += __shfl_xor(darow, off, 64);
        darow *= sqrtf(rs[row0 + i]);
    }
    {
        // every load is unconditional (clamped lane / neighbour): predicated loads would be issued one dependent branch at a time
        const int lc = lane < K ? lane : K - 1;
        const int64_t e = i * K + lc;
        const int32_t j = idx[e];
        float dw = dA[e];
        const float v = mode == 0 ? val[e] : 0.0f;          // kernel-uniform branch (val may be NULL in the other modes)
        const bool live = lane < K && j >= 0;
        if (normalized) {
            const float rsi = rs[row0 + i];
            const float ai = inv_sqrt_c(rsi), aj = inv_sqrt_c(rs[j >= 0 ? j : row0 + i]);
            const float drs = -0.5f * (da[row0 + i] + darow) * ai / rsi;
            dw = dw * ai * aj + drs;
        }
        float dv = dw;                                // mode 2: no ramp, plain normalisation backward (dval = dw, dk = 0)
        if (mode != 2) {
            const float th = c_tanh((float)lane - k[i]);
            const float f = 1.0f - 0.5f * (1.0f + th);
            const float dfdk = 0.5f * (1.0f - th * th);
            dv = mode == 0 ? dw * f : 0.0f;
            skp = live ? (mode == 0 ? dw * v * dfdk : dw * dfdk) : 0.0f;
        }
        if (lane < K) dval[e] = live ? dv : 0.0f;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) skp += __shfl_xor(skp, off, 64);
    if (lane == 0 && dk) dk[i] = skp;
}

// score backward: dval (wrt the stored score) -> dxp (fp32 atomics; dxp zeroed by caller).  Features on lanes.
__global__ __launch_bounds__(WPB * 64) void edge_bwd_kernel(const float *__restrict__ xp, int64_t N, int h,
                                                           const int32_t *__restrict__ idx, const float *__restrict__ val,
                                                           const float *__restrict__ dval, int K, int64_t row0, float t,
                                                           int perturb, float *__restrict__ dxp) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * WPB + dgg::wave_id();
    if (i >= N) return;
    const int64_t gi = row0 + i;
    int32_t jl = lane < K ? idx[i * K + lane] : -1;
    float gl = lane < K ? dval[i * K + lane] : 0.0f;
    float vl = lane < K ? val[i * K + lane] : 0.0f;
    const int c0 = lane, c1 = lane + 64;            // h <= 128
    float xi0 = c0 < h ? xp[gi * h + c0] : 0.0f;
    float xi1 = c1 < h ? xp[gi * h + c1] : 0.0f;
    float acc0 = 0.0f, acc1 = 0.0f;
    for (int r = 0; r < K; r++) {
        int32_t j = bcast(jl, r);
        float g = bcast(gl, r);
        if (j < 0 || g == 0.0f) continue;
        float v = bcast(vl, r);
        float d0 = c0 < h ? xi0 - xp[(int64_t)j * h + c0] : 0.0f;
        float d1 = c1 < h ? xi1 - xp[(int64_t)j * h + c1] : 0.0f;
        float d2 = d0 * d0 + d1 * d1;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) d2 += __shfl_xor(d2, off, 64);
        if (d2 == 0.0f) continue;                    // vector_norm backward at 0 is 0 (self loop)
        float dist = sqrtf(d2);
        float p = c_exp(t * dist);
        float dp = perturb ? g * v / (p + 1e-8f) : g;
        float dd = dp * t * p / dist;
        float e0 = dd * d0, e1 = dd * d1;
        acc0 += e0; acc1 += e1;
        if (c0 < h) atomicAdd(dxp + (int64_t)j * h + c0, -e0);
        if (c1 < h) atomicAdd(dxp + (int64_t)j * h + c1, -e1);
    }
    if (c0 < h) atomicAdd(dxp + gi * h + c0, acc0);
    if (c1 < h) atomicAdd(dxp + gi * h + c1, acc1);
}

// same for any latent width (the PPI configuration runs the DGG at latent_dim = hidden = 2048, train_ppi.py:44,
// model.py:907-910): features strided over the lanes, two passes over the (L1-resident) neighbour row
__global__ __launch_bounds__(WPB * 64) void edge_bwd_wide_kernel(const float *__restrict__ xp, int64_t N, int h,
                                                                const int32_t *__restrict__ idx, const float *__restrict__ val,
                                                                const float *__restrict__ dval, int K, int64_t row0, float t,
                                                                int perturb, float *__restrict__ dxp) {
    // ONE WORKGROUP PER ROW (small graphs, wide latents): the row's entries are dealt to the four wavefronts in groups of EQ
    const int lane = threadIdx.x & 63, wave = dgg::wave_id();
    const int64_t i = blockIdx.x;
    const int64_t gi = row0 + i;
    const float *xi = xp + gi * h;
    int32_t jl = lane < K ? idx[i * K + lane] : -1;
    float gl = lane < K ? dval[i * K + lane] : 0.0f;
    float vl = lane < K ? val[i * K + lane] : 0.0f;
    constexpr int EQ = 4;                                        // entries in flight per pass over the features
    for (int r0 = wave * EQ; r0 < K; r0 += WPB * EQ) {
        int32_t j[EQ];
        float g[EQ], v[EQ], d2[EQ], dd[EQ];
        const float *xj[EQ];
        bool any = false;
#pragma unroll
        for (int u = 0; u < EQ; u++) {
            const int r = r0 + u < K ? r0 + u : K - 1;
            j[u] = bcast(jl, r);
            g[u] = r0 + u < K ? bcast(gl, r) : 0.0f;
            v[u] = bcast(vl, r);
            if (j[u] < 0) g[u] = 0.0f;
            xj[u] = xp + (int64_t)(j[u] < 0 ? 0 : j[u]) * h;
            d2[u] = 0.0f;
            any = any || g[u] != 0.0f;
        }
        if (!any) continue;                                      // wave-uniform
        for (int c = lane; c < h; c += 64) {
            const float xv = xi[c];
#pragma unroll
            for (int u = 0; u < EQ; u++) { const float d = xv - xj[u][c]; d2[u] = fmaf(d, d, d2[u]); }
        }
#pragma unroll
        for (int u = 0; u < EQ; u++) {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) d2[u] += __shfl_xor(d2[u], off, 64);
            dd[u] = 0.0f;
            if (g[u] != 0.0f && d2[u] != 0.0f) {
                const float dist = sqrtf(d2[u]);
                const float p = c_exp(t * dist);
                const float dp = perturb ? g[u] * v[u] / (p + 1e-8f) : g[u];
                dd[u] = dp * t * p / dist;
            }
        }
        for (int c = lane; c < h; c += 64) {
            const float xv = xi[c];
            float own = 0.0f;
#pragma unroll
            for (int u = 0; u < EQ; u++) {
                if (dd[u] != 0.0f) {
                    const float e = dd[u] * (xv - xj[u][c]);
                    own += e;
                    atomicAdd(dxp + (int64_t)j[u] * h + c, -e);
                }
            }
            if (own != 0.0f) atomicAdd(dxp + gi * h + c, own);
        }
    }
}

// ROW pass of the wide score backward without float atomics (h <= 2048, a multiple of 64): the row's own side
//   own_i = sum_r dd_ir (xp_i - xp_j)  accumulates per wavefront in LDS and is STORED (dxp_rows [N,h], overwritten);
// the coefficient dd_ir = (d loss / d dist_ir) / dist_ir of every entry goes to dd [N,K] for the transposed pass
//   dxp_j -= sum_i dd_ij (xp_i - xp_j)  =  -(DD^T xp)_j + (sum_i dd_ij) xp_j      (dgg_ell_spmm_t_part with a = dd, dY = xp).
// 98 M atomics per 2 400-node graph at latent 2048 (PPI configuration) were 482 us; the two gather passes take ~1/3 of that.
constexpr int EBW_HMAX = 2048;
__global__ __launch_bounds__(WPB * 64) void edge_bwd_wide_rows_kernel(const float *__restrict__ xp, int64_t N, int h,
                                                                     const int32_t *__restrict__ idx, const float *__restrict__ val,
                                                                     const float *__restrict__ dval, int K, int64_t row0, float t,
                                                                     int perturb, float *__restrict__ dxp_rows, float *__restrict__ dd_out) {
    __shared__ float own_s[WPB][EBW_HMAX];
    const int lane = threadIdx.x & 63, wave = dgg::wave_id();
    const int64_t i = blockIdx.x;
    const int64_t gi = row0 + i;
    const float *xi = xp + gi * h;
    int32_t jl = lane < K ? idx[i * K + lane] : -1;
    float gl = lane < K ? dval[i * K + lane] : 0.0f;
    float vl = lane < K ? val[i * K + lane] : 0.0f;
    for (int c = lane; c < h; c += 64) own_s[wave][c] = 0.0f;
    constexpr int EQ = 4;                                        // entries in flight per pass over the features
    for (int r0 = wave * EQ; r0 < K; r0 += WPB * EQ) {
        int32_t j[EQ];
        float g[EQ], v[EQ], d2[EQ], dd[EQ];
        const float *xj[EQ];
        bool any = false;
#pragma unroll
        for (int u = 0; u < EQ; u++) {
            const int r = r0 + u < K ? r0 + u : K - 1;
            j[u] = bcast(jl, r);
            g[u] = r0 + u < K ? bcast(gl, r) : 0.0f;
            v[u] = bcast(vl, r);
            if (j[u] < 0) g[u] = 0.0f;
            xj[u] = xp + (int64_t)(j[u] < 0 ? 0 : j[u]) * h;
            d2[u] = 0.0f;
            any = any || g[u] != 0.0f;
        }
        if (!any) {                                              // wave-uniform
            if (lane < EQ && r0 + lane < K) dd_out[i * K + r0 + lane] = 0.0f;
            continue;
        }
        for (int c = lane; c < h; c += 64) {
            const float xv = xi[c];
#pragma unroll
            for (int u = 0; u < EQ; u++) { const float d = xv - xj[u][c]; d2[u] = fmaf(d, d, d2[u]); }
        }
#pragma unroll
        for (int u = 0; u < EQ; u++) {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) d2[u] += __shfl_xor(d2[u], off, 64);
            dd[u] = 0.0f;
            if (g[u] != 0.0f && d2[u] != 0.0f) {
                const float dist = sqrtf(d2[u]);
                const float p = c_exp(t * dist);
                const float dp = perturb ? g[u] * v[u] / (p + 1e-8f) : g[u];
                dd[u] = dp * t * p / dist;
            }
            if (lane == u && r0 + u < K) dd_out[i * K + r0 + u] = dd[u];
        }
        for (int c = lane; c < h; c += 64) {
            const float xv = xi[c];
            float own = 0.0f;
#pragma unroll
            for (int u = 0; u < EQ; u++)
                if (dd[u] != 0.0f) own += dd[u] * (xv - xj[u][c]);
            own_s[wave][c] += own;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < h; c += WPB * 64) {
        float tot = own_s[0][c];
#pragma unroll
        for (int w = 1; w < WPB; w++) tot += own_s[w][c];
        dxp_rows[i * h + c] = tot;
    }
}

// The same row pass one 256-feature SLICE at a time (h a multiple of 256; grid.y = slice, x fastest: the whole chip works on one slice
// before the next, and a slice of xp -- 1 KB per node, 1.8 MB for a PPI graph -- stays in an XCD's L2 where 8 KB rows do not: the
// unsliced kernel gathers its 2 x 0.43 GB at 5 TB/s).  Pass 1: the slice's share of every entry's squared distance -> part[slice][N*K].
// Pass 2: a wavefront sums its row's shares (slice order), forms dd, and accumulates its 256 features of own_i in registers.
__global__ __launch_bounds__(WPB * 64) void edge_bwd_wide_d2_slice(const float *__restrict__ xp, int64_t N, int h, const int32_t *__restrict__ idx,
                                                                  const float *__restrict__ dval, int K, int64_t row0,
                                                                  float *__restrict__ part) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * WPB + dgg::wave_id();
    if (i >= N) return;
    const int c4 = blockIdx.y * 64 + lane;                       // this lane's float4 of the row
    const int32_t jl = lane < K ? idx[i * K + lane] : -1;
    const float gl = lane < K ? dval[i * K + lane] : 0.0f;
    const float4 xi = reinterpret_cast<const float4 *>(xp + (row0 + i) * h)[c4];
    float mine = 0.0f;
    constexpr int NQ = 8;
    for (int r0 = 0; r0 < K; r0 += NQ) {
        float4 xj[NQ];
        bool v[NQ];
        bool any = false;
#pragma unroll
        for (int u = 0; u < NQ; u++) {
            const int r = r0 + u < K ? r0 + u : K - 1;
            const int32_t j = bcast(jl, r);
            v[u] = r0 + u < K && j >= 0 && bcast(gl, r) != 0.0f;  // (an entry without a gradient needs no distance: dd = 0)
            any = any || v[u];
            xj[u] = xi;
            if (v[u]) xj[u] = reinterpret_cast<const float4 *>(xp + (int64_t)j * h)[c4];   // (wave-uniform predicate)
        }
        if (!any) continue;                                      // wave-uniform
#pragma unroll
        for (int u = 0; u < NQ; u++) {
            const float dx = xi.x - xj[u].x, dy = xi.y - xj[u].y, dz = xi.z - xj[u].z, dw = xi.w - xj[u].w;
            const float p_ = fmaf(dw, dw, fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
            const float tot = wave_sum_dpp(p_, lane);
            if (lane == r0 + u) mine = tot;
        }
    }
    if (lane < K) part[(int64_t)blockIdx.y * N * K + i * K + lane] = mine;
}
__global__ __launch_bounds__(WPB * 64) void edge_bwd_wide_own_slice(const float *__restrict__ xp, int64_t N, int h, const int32_t *__restrict__ idx,
                                                                   const float *__restrict__ val, const float *__restrict__ dval, int K,
                                                                   int64_t row0, float t, int perturb, const float *__restrict__ part, int S,
                                                                   float *__restrict__ dxp_rows, float *__restrict__ dd_out) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * WPB + dgg::wave_id();
    if (i >= N) return;
    const int c4 = blockIdx.y * 64 + lane;                       // this lane's float4 of the row
    const int32_t jl = lane < K ? idx[i * K + lane] : -1;
    float ddl = 0.0f;                                            // lane r: the coefficient of entry r
    if (lane < K && jl >= 0) {
        const float g = dval[i * K + lane];
        float d2 = 0.0f;
        for (int s_ = 0; s_ < S; s_++) d2 += part[(int64_t)s_ * N * K + i * K + lane];
        if (g != 0.0f && d2 != 0.0f) {
            const float dist = sqrtf(d2);
            const float p = c_exp(t * dist);
            const float dp = perturb ? g * val[i * K + lane] / (p + 1e-8f) : g;
            ddl = dp * t * p / dist;
        }
    }
    if (blockIdx.y == 0 && lane < K) dd_out[i * K + lane] = ddl;
    const float4 xi = reinterpret_cast<const float4 *>(xp + (row0 + i) * h)[c4];
    float4 own = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    constexpr int NQ = 8;
    for (int r0 = 0; r0 < K; r0 += NQ) {
        float4 xj[NQ];
        float dd[NQ];
        bool any = false;
#pragma unroll
        for (int u = 0; u < NQ; u++) {
            const int r = r0 + u < K ? r0 + u : K - 1;
            const int32_t j = bcast(jl, r);
            dd[u] = r0 + u < K ? bcast(ddl, r) : 0.0f;
            any = any || dd[u] != 0.0f;
            xj[u] = xi;
            if (dd[u] != 0.0f) xj[u] = reinterpret_cast<const float4 *>(xp + (int64_t)j * h)[c4];   // (wave-uniform predicate)
        }
        if (!any) continue;                                      // wave-uniform
#pragma unroll
        for (int u = 0; u < NQ; u++) {
            own.x = fmaf(dd[u], xi.x - xj[u].x, own.x); own.y = fmaf(dd[u], xi.y - xj[u].y, own.y);
            own.z = fmaf(dd[u], xi.z - xj[u].z, own.z); own.w = fmaf(dd[u], xi.w - xj[u].w, own.w);
        }
    }
    reinterpret_cast<float4 *>(dxp_rows + i * h)[c4] = own;
}

// GCNII layer epilogue (reference model.py:36-44 / 69-77): out = theta * (support W) + (1 - theta) * r (+ input), with
// r = (1 - alpha) * hi + alpha * h0 folded in (h0 == NULL: r = hi, the non-variant layer whose support IS r).  One pass
// instead of five elementwise launches; the backward is three scaled copies of the cotangent.
__global__ void gcnii_epilogue_fwd_kernel(const float *__restrict__ sw, const float *__restrict__ hi, const float *__restrict__ h0,
                                          const float *__restrict__ inp, int64_t n, float theta, float alpha, float *__restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const float r = h0 ? (1.0f - alpha) * hi[e] + alpha * h0[e] : hi[e];
    float o = theta * sw[e] + (1.0f - theta) * r;
    if (inp) o += inp[e];
    out[e] = o;
}
// the same with the layer's ReLU, the next layer's counter-based dropout and a bf16 copy of the result (fused GCNII stack): as a
// SEPARATE pass after the plain product it measured faster than inside the product's epilogue (91 + 17 us against 133 at n = 2250:
// the product kernel with a fused epilogue waits for its operand loads before every K-step's MFMAs)
__global__ void gcnii_stack_epilogue_kernel(const float4 *__restrict__ sw, const float4 *__restrict__ hi, const float4 *__restrict__ h0,
                                            const float4 *__restrict__ inp, int64_t n4, float theta, float alpha, uint32_t thr24, float scale,
                                            uint32_t s0, uint32_t s1, float4 *__restrict__ out, uint2 *__restrict__ outb) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n4) return;
    const float4 a = sw[e], b = hi[e], c = h0[e];
    const float4 d = inp ? inp[e] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const float omt = 1.0f - theta, oma = 1.0f - alpha;
    float o[4] = {theta * a.x + omt * (oma * b.x + alpha * c.x) + d.x, theta * a.y + omt * (oma * b.y + alpha * c.y) + d.y,
                  theta * a.z + omt * (oma * b.z + alpha * c.z) + d.z, theta * a.w + omt * (oma * b.w + alpha * c.w) + d.w};
#pragma unroll
    for (int q = 0; q < 4; q++) {
        o[q] = o[q] > 0.0f ? o[q] : 0.0f;
        if (thr24) o[q] = drop_keep(s0, s1, (uint32_t)(4 * e + q), thr24) ? o[q] * scale : 0.0f;
    }
    out[e] = make_float4(o[0], o[1], o[2], o[3]);
    if (outb) {
        const __bf16 h_[4] = {(__bf16)o[0], (__bf16)o[1], (__bf16)o[2], (__bf16)o[3]};
        outb[e] = *reinterpret_cast<const uint2 *>(h_);
    }
}
__global__ void gcnii_epilogue_bwd_kernel(const float *__restrict__ g, int64_t n, float theta, float alpha, float *__restrict__ dsw,
                                          float *__restrict__ dhi, float *__restrict__ dh0) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const float gv = g[e];
    dsw[e] = theta * gv;
    if (dh0) { dhi[e] = (1.0f - theta) * (1.0f - alpha) * gv; dh0[e] = (1.0f - theta) * alpha * gv; }
    else dhi[e] = (1.0f - theta) * gv;
}

// out = x * keep(e) / (1 - p) (accumulate: out += ...): the dropout of the fused GCNII stack on a tensor that no product produces
// (the stack's input h0, and -- with the same seeds -- the gradient that flows back through it)
__global__ void dropout_hash_kernel(const float *__restrict__ x, int64_t n, uint32_t thr24, float scale, uint32_t s0, uint32_t s1,
                                    int accumulate, float *__restrict__ out, __bf16 *__restrict__ outb) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float v = (thr24 == 0 || drop_keep(s0, s1, (uint32_t)e, thr24)) ? x[e] * scale : 0.0f;
    if (accumulate) v += out[e];
    out[e] = v;
    if (outb) outb[e] = (__bf16)v;                               // (the copy the stack's gather kernels read)
}

inline unsigned rows_grid(int64_t N) { return (unsigned)((N + WPB - 1) / WPB); }

}  // namespace

extern "C" {

int dgg_ell_spmm_act_fwd(const int32_t *idx, const float *ahat, const float *X, int64_t N, int K, int F, int act, float *Y,
                         void *stream);

int dgg_dropout_hash(const float *x, int64_t n, float p, uint32_t s0, uint32_t s1, int accumulate, float *out, void *outb, void *stream) {
    if (!(p >= 0.0f && p < 1.0f) || n >= ((int64_t)1 << 32)) return dgg_set_error(DGG_ERR_ARG, "dropout_hash: p in [0,1), n < 2^32");
    if (n == 0) return 0;
    const uint32_t thr = (uint32_t)(p * 16777216.0f);
    hipLaunchKernelGGL(dropout_hash_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, n, thr,
                       1.0f / (1.0f - p), s0, s1, accumulate, out, reinterpret_cast<__bf16 *>(outb));
    return dgg_check_launch("dropout_hash");
}

int dgg_softk_fwd(const int32_t *idx, const float *val, const float *k, int64_t N, int K, int mode, float *w, float *rs,
                  void *stream) {
    if (K < 1 || K > 64) return dgg_set_error(DGG_ERR_UNSUPPORTED, "ELL width K must be in [1,64]");
    if (mode != 0 && mode != 1 && mode != 3) return dgg_set_error(DGG_ERR_ARG, "softk_fwd: mode must be 0 (k_times), 1 (k_only) or 3 (hard)");
    if (N == 0) return 0;
    hipLaunchKernelGGL(softk_fwd_kernel, dim3(rows_grid(N)), dim3(WPB * 64), 0, (hipStream_t)stream, idx, val, k, N, K,
                       mode, w, rs);
    return dgg_check_launch("softk_fwd");
}

int dgg_ell_normalize_fwd(const int32_t *idx, const float *w, const float *rs, int64_t N, int K, int64_t row0,
                          float *ahat, void *stream) {
    if (N == 0) return 0;
    int64_t n = N * K;
    hipLaunchKernelGGL(normalize_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, idx, w,
                       rs, N, K, row0, ahat);
    return dgg_check_launch("ell_normalize_fwd");
}

// Y = A X and, in the same pass, Yb = bf16(Y) [N, F] (F a multiple of 256, 16-byte aligned rows): the aggregation of a GCNII layer
// on the bf16 matrix cores hands its product operand over without a pack pass (model.py:34 + the operand rounding of dgg_bf16.hip)
int dgg_ell_spmm_fwd_bf16(const int32_t *idx, const float *ahat, const float *X, int64_t N, int K, int F, float *Y, void *Yb, int64_t ldyb,
                          void *stream) {
    if (K < 1 || K > 64) return dgg_set_error(DGG_ERR_UNSUPPORTED, "ELL width K must be in [1,64]");
    if (F % 256 != 0 || ((uintptr_t)X % 16) || ((uintptr_t)Y % 16) || !Yb || ldyb < F)
        return dgg_set_error(DGG_ERR_UNSUPPORTED, "ell_spmm_fwd_bf16: F must be a multiple of 256, rows 16-byte aligned, Yb given with ldyb >= F");
    if (N == 0) return 0;
    dim3 grid(rows_grid(N), (unsigned)(F / 256));
    hipLaunchKernelGGL(spmm_fwd_kernel<4>, grid, dim3(WPB * 64), 0, (hipStream_t)stream, idx, ahat, X, N, K, F, Y,
                       reinterpret_cast<__bf16 *>(Yb), ldyb);
    return dgg_check_launch("ell_spmm_fwd_bf16");
}

// dgg_ell_spmm_fwd_bf16 gathering a bf16 COPY of X (Xb [N, F] bf16, F a multiple of 512, 16-byte aligned rows): half the gathered bytes
int dgg_ell_spmm_fwd_b16(const int32_t *idx, const float *ahat, const void *Xb, int64_t N, int K, int F, float *Y, void *Yb, int64_t ldyb,
                         void *stream) {
    if (K < 1 || K > 64) return dgg_set_error(DGG_ERR_UNSUPPORTED, "ELL width K must be in [1,64]");
    if (F % 512 != 0 || ((uintptr_t)Xb % 16) || ((uintptr_t)Y % 16)) return dgg_set_error(DGG_ERR_UNSUPPORTED, "ell_spmm_fwd_b16: F must be a multiple of 512, rows 16-byte aligned");
    if (N == 0) return 0;
    dim3 grid(rows_grid(N), (unsigned)(F / 512));
    hipLaunchKernelGGL(spmm_fwd_b16_kernel, grid, dim3(WPB * 64), 0, (hipStream_t)stream, idx, ahat, reinterpret_cast<const uint4 *>(Xb), N, K, F,
                       Y, reinterpret_cast<__bf16 *>(Yb), ldyb);
    return dgg_check_launch("ell_spmm_fwd_b16");
}
// dA = <dY_i, X_j> on bf16 copies of both operands (Xb, dYb [N, F] bf16, F a multiple of 8; fp32 accumulation): the SDDMM of the stack
int dgg_ell_sddmm_b16(const int32_t *idx, const float *ahat, const void *Xb, const void *dYb, int64_t N, int K, int F, int skip_zero, float *dA,
                      void *stream) {
    if (K < 1 || K > 64) return dgg_set_error(DGG_ERR_UNSUPPORTED, "ELL width K must be in [1,64]");
    if (F % 8 != 0 || ((uintptr_t)Xb % 16) || ((uintptr_t)dYb % 16)) return dgg_set_error(DGG_ERR_UNSUPPORTED, "ell_sddmm_b16: F must be a multiple of 8, rows 16-byte aligned");
    if (N == 0) return 0;
    hipLaunchKernelGGL(sddmm_wide_b16_kernel, dim3(rows_grid(N)), dim3(WPB * 64), 0, (hipStream_t)stream, idx, ahat, reinterpret_cast<const uint4 *>(Xb),
                       reinterpret_cast<const uint4 *>(dYb), N, K, F, skip_zero, dA);
    return dgg_check_launch("ell_sddmm_b16");
}

// dgg_ell_sddmm_b16 one 512-feature slice at a time (F a multiple of 512): dA (+)= <dY_i, X_j>; ws: dgg_ell_sddmm_b16_ws_floats(N, K, F)
// floats of scratch (the slices' partial sums); accumulate: dA += (the stack sums dA over its layers) instead of =.
// dA == NULL: the slice sums stay in ws -- overwritten, or ADDED to what ws holds (accumulate) -- and dgg_ell_sddmm_slices_sum adds the
// slices once, after the last of a series of calls (the GCNII stack: nine layers, one sum).
size_t dgg_ell_sddmm_b16_ws_floats(int64_t N, int K, int F) { return (size_t)(F / 512) * (size_t)N * (size_t)K; }
int dgg_ell_sddmm_slices_sum(const float *ws, int64_t N, int K, int F, float *dA, int accumulate, void *stream) {
    if (F % 512 != 0 || !ws || !dA) return dgg_set_error(DGG_ERR_ARG, "ell_sddmm_slices_sum: F a multiple of 512, workspace and dA required");
    const int64_t n = N * K;
    if (n == 0) return 0;
    hipLaunchKernelGGL(sddmm_slices_sum, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ws, n, F / 512, accumulate, dA);
    return dgg_check_launch("ell_sddmm_slices_sum");
}
int dgg_ell_sddmm_b16_sliced(const int32_t *idx, const float *ahat, const void *Xb, const void *dYb, int64_t N, int K, int F, int skip_zero,
                             float *ws, float *dA, int accumulate, void *stream) {
    if (K < 1 || K > 64) return dgg_set_error(DGG_ERR_UNSUPPORTED, "ELL width K must be in [1,64]");
    if (F % 512 != 0 || ((uintptr_t)Xb % 16) || ((uintptr_t)dYb % 16)) return dgg_set_error(DGG_ERR_UNSUPPORTED, "ell_sddmm_b16_sliced: F must be a multiple of 512, rows 16-byte aligned");
    if (!ws) return dgg_set_error(DGG_ERR_ARG, "ell_sddmm_b16_sliced: the workspace is required");
    if (N == 0) return 0;
    const int S = F / 512;
    hipLaunchKernelGGL(sddmm_slice_b16_kernel, dim3(rows_grid(N), (unsigned)S), dim3(WPB * 64), 0, (hipStream_t)stream, idx, ahat,
                       reinterpret_cast<const uint4 *>(Xb), reinterpret_cast<const uint4 *>(dYb), N, K, F, skip_zero, ws, dA ? 0 : accumulate);
    if (dA) return dgg_ell_sddmm_slices_sum(ws, N, K, F, dA, accumulate, stream);
    return dgg_check_launch("ell_sddmm_b16_sliced");
}

int dgg_ell_spmm_fwd(const int32_t *idx, const float *ahat, const float *X, int64_t N, int K, int F, float *Y,
                     void *stream) {
    return dgg_ell_spmm_act_fwd(idx, ahat, X, N, K, F, 0, Y, stream);
}

// Y = act(A X), act 0 (none) or 2 (ReLU: GCNConv's activation when the aggregation runs AFTER the projection)
int dgg_ell_spmm_act_fwd(const int32_t *idx, const float *ahat, const float *X, int64_t N, int K, int F, int act, float *Y,
                         void *stream) {
    if (K < 1 || K > 64) return dgg_set_error(DGG_ERR_UNSUPPORTED, "ELL width K must be in [1,64]");
    if (act != 0 && act != 2) return dgg_set_error(DGG_ERR_ARG, "ell_spmm_act_fwd: act must be 0 (none) or 2 (ReLU)");
    if (N == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    bool al16 = ((uintptr_t)X % 16 == 0) && ((uintptr_t)Y % 16 == 0), al8 = ((uintptr_t)X % 8 == 0);
    if (al16 && (F == 16 || F == 32 || F == 64)) {
        if (F == 64) hipLaunchKernelGGL(spmm_fwd_narrow<64>, dim3(rows_grid(N)), dim3(WPB * 64), 0, st, idx, ahat, X, N, K, act, Y);
        else if (F == 32) hipLaunchKernelGGL(spmm_fwd_narrow<32>, dim3(rows_grid(N)), dim3(WPB * 64), 0, st, idx, ahat, X, N, K, act, Y);
        else hipLaunchKernelGGL(spmm_fwd_narrow<16>, dim3(rows_grid(N)), dim3(WPB * 64), 0, st, idx, ahat, X, N, K, act, Y);
        return dgg_check_launch("ell_spmm_fwd");
    }
    if (F % 4 == 0 && F >= 256 && al16) {
        dim3 grid(rows_grid(N), (unsigned)((F + 255) / 256));
        hipLaunchKernelGGL(spmm_fwd_kernel<4>, grid, dim3(WPB * 64), 0, st, idx, ahat, X, N, K, F, Y);
    } else if (F % 2 == 0 && F >= 128 && al8) {
        dim3 grid(rows_grid(N), (unsigned)((F + 127) / 128));
        hipLaunchKernelGGL(spmm_fwd_kernel<2>, grid, dim3(WPB * 64), 0, st, idx, ahat, X, N, K, F, Y);
    } else {
        dim3 grid(rows_grid(N), (unsigned)((F + 63) / 64));
        hipLaunchKernelGGL(spmm_fwd_kernel<1>, grid, dim3(WPB * 64), 0, st, idx, ahat, X, N, K, F, Y);
    }
    if (act != 0) {
        const int64_t n = N * F;
        hipLaunchKernelGGL(act_inplace_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, Y, n, act);
    }
    return dgg_check_launch("ell_spmm_fwd");
}

// dgg_ell_spmm_act_fwd on chunked rows (rows wider than 64 ranks, dgg_chunk_layout): idx / ahat [chunks,64], row i = the chunks
// [cptr[i], cptr[i+1]), Y [rows,F]
int dgg_ell_spmm_act_fwd_chunked(const int32_t *idx, const float *ahat, const float *X, int64_t rows, const int32_t *cptr, int F, int act, float *Y,
                                 void *stream) {
    if (act != 0 && act != 2) return dgg_set_error(DGG_ERR_ARG, "ell_spmm_act_fwd_chunked: act must be 0 (none) or 2 (ReLU)");
    if (!cptr) return dgg_set_error(DGG_ERR_ARG, "ell_spmm_act_fwd_chunked: cptr is required");
    if (rows == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int K = 64;
    bool al16 = ((uintptr_t)X % 16 == 0) && ((uintptr_t)Y % 16 == 0), al8 = ((uintptr_t)X % 8 == 0);
    if (al16 && (F == 16 || F == 32 || F == 64)) {
        if (F == 64) hipLaunchKernelGGL((spmm_fwd_narrow<64, true>), dim3(rows_grid(rows)), dim3(WPB * 64), 0, st, idx, ahat, X, rows, K, act, Y, cptr);
        else if (F == 32) hipLaunchKernelGGL((spmm_fwd_narrow<32, true>), dim3(rows_grid(rows)), dim3(WPB * 64), 0, st, idx, ahat, X, rows, K, act, Y, cptr);
        else hipLaunchKernelGGL((spmm_fwd_narrow<16, true>), dim3(rows_grid(rows)), dim3(WPB * 64), 0, st, idx, ahat, X, rows, K, act, Y, cptr);
        return dgg_check_launch("ell_spmm_fwd_chunked");
    }
    if (F % 4 == 0 && F >= 256 && al16) {
        dim3 grid(rows_grid(rows), (unsigned)((F + 255) / 256));
        hipLaunchKernelGGL(spmm_fwd_kernel<4>, grid, dim3(WPB * 64), 0, st, idx, ahat, X, rows, K, F, Y, (__bf16 *)nullptr, (int64_t)0, cptr);
    } else if (F % 2 == 0 && F >= 128 && al8) {
        dim3 grid(rows_grid(rows), (unsigned)((F + 127) / 128));
        hipLaunchKernelGGL(spmm_fwd_kernel<2>, grid, dim3(WPB * 64), 0, st, idx, ahat, X, rows, K, F, Y, (__bf16 *)nullptr, (int64_t)0, cptr);
    } else {
        dim3 grid(rows_grid(rows), (unsigned)((F + 63) / 64));
        hipLaunchKernelGGL(spmm_fwd_kernel<1>, grid, dim3(WPB * 64), 0, st, idx, ahat, X, rows, K, F, Y, (__bf16 *)nullptr, (int64_t)0, cptr);
    }
    if (act != 0) {
        const int64_t n = rows * F;
        hipLaunchKernelGGL(act_inplace_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, Y, n, act);
    }
    return dgg_check_launch("ell_spmm_fwd_chunked");
}

// dA [N,K] overwritten; dX (nullable, [Nglobal,F]) accumulated into with atomics
int dgg_ell_spmm_bwd(const int32_t *idx, const float *ahat, const float *X, const float *dY, int64_t N, int K, int F,
                     int skip_zero, float *dA, float *dX, void *stream) {
    if (K < 1 || K > 64) return dgg_set_error(DGG_ERR_UNSUPPORTED, "ELL width K must be in [1,64]");
    if (N == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const bool al16 = (reinterpret_cast<uintptr_t>(X) % 16 == 0) && (reinterpret_cast<uintptr_t>(dY) % 16 == 0);
    if (!dX && al16 && F == 128)
        hipLaunchKernelGGL((sddmm_pair_kernel<1, false>), dim3(rows_grid(N)), dim3(WPB * 64), 0, st, idx, ahat, X, dY, N, K, skip_zero, dA,
                           nullptr, nullptr, 0, nullptr, nullptr, nullptr);
    else if (!dX && al16 && F == 256)
        hipLaunchKernelGGL((sddmm_pair_kernel<2, false>), dim3(rows_grid(N)), dim3(WPB * 64), 0, st, idx, ahat, X, dY, N, K, skip_zero, dA,
                           nullptr, nullptr, 0, nullptr, nullptr, nullptr);
    else if (!dX && al16 && F > 256 && F % 4 == 0)
        hipLaunchKernelGGL(sddmm_wide_kernel, dim3(rows_grid(N)), dim3(WPB * 64), 0, st, idx, ahat, X, dY, N, K, F, skip_zero, dA);
    else if (F <= 8)
        hipLaunchKernelGGL(spmm_bwd_tiny_kernel<8>, dim3(rows_grid(N)), dim3(WPB * 64), 0, st, idx, ahat, X, dY, N, K, F, skip_zero, dA, dX);
    else
        hipLaunchKernelGGL(spmm_bwd_kernel, dim3(rows_grid(N)), dim3(WPB * 64), 0, st, idx, ahat, X, dY, N, K, F, skip_zero, dA, dX);
    return dgg_check_launch("ell_spmm_bwd");
}

// SDDMM (dA = <dY_i, X_j>, no dX) fused with the ROW side of the normalisation backward's phase 1 (one launch): dA [rows,K],
// da[row0 + i] (da [ncols] zeroed by the caller) and the per-entry coefficients coef_ws [rows*K] in partition order; follow
// with dgg_norm_da_cols_part for the neighbour-side sums.  DGG_ERR_UNSUPPORTED when the fused
// kernel does not cover the shape (F not in {128, 256} or unaligned rows): use dgg_ell_spmm_bwd + dgg_norm_bwd_da_part.
int dgg_ell_sddmm_norm_part(const int32_t *idx, const float *ahat, const float *w, const float *rs, const float *X,
                            const float *dY, int64_t rows, int K, int F, int64_t row0, int skip_zero, const void *part_ws,
                            int64_t ncols, float *coef_ws, float *dA, float *da, void *stream) {
    if (K < 1 || K > 64) return dgg_set_error(DGG_ERR_UNSUPPORTED, "ELL width K must be in [1,64]");
    const bool al16 = (reinterpret_cast<uintptr_t>(X) % 16 == 0) && (reinterpret_cast<uintptr_t>(dY) % 16 == 0);
    const int *slotmap = dgg_part_slotmap(part_ws, rows, K, ncols);
    if (!al16 || (F != 128 && F != 256) || !slotmap)
        return dgg_set_error(DGG_ERR_UNSUPPORTED, "sddmm_norm_part: feature width must be 128 or 256 (16-byte aligned rows)");
    if (rows == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (F == 128)
        hipLaunchKernelGGL((sddmm_pair_kernel<1, true>), dim3(rows_grid(rows)), dim3(WPB * 64), 0, st, idx, ahat, X, dY, rows, K,
                           skip_zero, dA, w, rs, row0, slotmap, coef_ws, da);
    else
        hipLaunchKernelGGL((sddmm_pair_kernel<2, true>), dim3(rows_grid(rows)), dim3(WPB * 64), 0, st, idx, ahat, X, dY, rows, K,
                           skip_zero, dA, w, rs, row0, slotmap, coef_ws, da);
    return dgg_check_launch("ell_sddmm_norm_part");
}

// second half of the normalisation backward's phase 1: da_j += column sums of the coefficients written by
// dgg_ell_sddmm_norm_part (or dgg_norm_bwd_da_part's row pass), through the partition
int dgg_norm_da_cols_part(const void *part_ws, int64_t rows, int K, int64_t ncols, const float *coef_ws, float *da, void *stream) {
    if (rows == 0) return 0;
    if (!dgg_part_slotmap(part_ws, rows, K, ncols)) return dgg_set_error(DGG_ERR_ARG, "norm_da_cols_part: no partition");
    return dgg_norm_da_cols_impl(part_ws, rows, K, ncols, coef_ws, da, (hipStream_t)stream);
}

int dgg_gcnii_epilogue_fwd(const float *sw, const float *hi, const float *h0, const float *inp, int64_t n, float theta, float alpha,
                           float *out, void *stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(gcnii_epilogue_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, sw, hi, h0, inp, n,
                       theta, alpha, out);
    return dgg_check_launch("gcnii_epilogue_fwd");
}
// out = dropout(relu(theta sw + (1 - theta)((1 - alpha) hi + alpha h0) (+ inp))) and bf16(out) (outb nullable); n a multiple of 4,
// 16-byte aligned operands; the mask as dgg_gcnii_gemm_bf16_split_act (element e kept iff hash24(s0, s1, e) >= drop_p 2^24)
int dgg_gcnii_stack_epilogue(const float *sw, const float *hi, const float *h0, const float *inp, int64_t n, float theta, float alpha,
                             float drop_p, uint32_t s0, uint32_t s1, float *out, void *outb, void *stream) {
    if (n % 4 != 0 || !(drop_p >= 0.0f && drop_p < 1.0f) || n >= ((int64_t)1 << 32) || !hi || !h0)
        return dgg_set_error(DGG_ERR_ARG, "gcnii_stack_epilogue: n a multiple of 4 below 2^32, drop_p in [0,1), hi and h0 given");
    if (n == 0) return 0;
    hipLaunchKernelGGL(gcnii_stack_epilogue_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4 *>(sw), reinterpret_cast<const float4 *>(hi), reinterpret_cast<const float4 *>(h0),
                       reinterpret_cast<const float4 *>(inp), n / 4, theta, alpha, (uint32_t)(drop_p * 16777216.0f), 1.0f / (1.0f - drop_p), s0, s1,
                       reinterpret_cast<float4 *>(out), reinterpret_cast<uint2 *>(outb));
    return dgg_check_launch("gcnii_stack_epilogue");
}
int dgg_gcnii_epilogue_bwd(const float *g, int64_t n, float theta, float alpha, float *dsw, float *dhi, float *dh0, void *stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(gcnii_epilogue_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, n, theta,
                       alpha, dsw, dhi, dh0);
    return dgg_check_launch("gcnii_epilogue_bwd");
}

int dgg_norm_bwd_da(const int32_t *idx, const float *w, const float *rs, const float *dA, int64_t N, int K, int64_t row0,
                    float *da, void *stream) {
    if (N == 0) return 0;
    hipLaunchKernelGGL(norm_bwd_da_kernel, dim3(rows_grid(N)), dim3(WPB * 64), 0, (hipStream_t)stream, idx, w, rs, dA, N, K,
                       row0, da);
    return dgg_check_launch("norm_bwd_da");
}

int dgg_softk_bwd(const int32_t *idx, const float *val, const float *k, const float *rs, const float *dA, const float *da,
                  int64_t N, int K, int64_t row0, int mode, int normalized, float *dval, float *dk, void *stream) {
    if (N == 0) return 0;
    hipLaunchKernelGGL(softk_bwd_kernel, dim3(rows_grid(N)), dim3(WPB * 64), 0, (hipStream_t)stream, idx, val, k, rs, dA, da,
                       N, K, row0, mode, normalized, dval, dk, (const float *)nullptr);
    return dgg_check_launch("softk_bwd");
}

int dgg_softk_bwd_rows(const int32_t *idx, const float *val, const float *k, const float *rs, const float *dA, const float *da_cols,
                       const float *ahat_rows, int64_t N, int K, int64_t row0, int mode, float *dval, float *dk, void *stream) {
    if (!rs || !da_cols || !ahat_rows) return dgg_set_error(DGG_ERR_ARG, "softk_bwd_rows: rs, da_cols and ahat_rows are required");
    if (N == 0) return 0;
    hipLaunchKernelGGL(softk_bwd_kernel, dim3(rows_grid(N)), dim3(WPB * 64), 0, (hipStream_t)stream, idx, val, k, rs, dA, da_cols,
                       N, K, row0, mode, 1, dval, dk, ahat_rows);
    return dgg_check_launch("softk_bwd_rows");
}

int dgg_edge_bwd_wide_rows(const float *xp, int64_t N, int h, const int32_t *idx, const float *val, const float *dval, int K,
                           int64_t row0, float t, int perturb, float *dxp_rows, float *dd, void *stream) {
    if (h % 64 != 0 || h > EBW_HMAX || K < 1 || K > 64)
        return dgg_set_error(DGG_ERR_UNSUPPORTED, "edge_bwd_wide_rows: latent_dim a multiple of 64 up to 2048, K in [1,64]");
    if (N == 0) return 0;
    hipLaunchKernelGGL(edge_bwd_wide_rows_kernel, dim3((unsigned)N), dim3(WPB * 64), 0, (hipStream_t)stream, xp, N, h, idx, val, dval, K,
                       row0, t, perturb, dxp_rows, dd);
    return dgg_check_launch("edge_bwd_wide_rows");
}

// dgg_edge_bwd_wide_rows one 256-feature slice at a time (h a multiple of 256, xp rows 16-byte aligned); ws: dgg_edge_bwd_wide_rows_ws_floats
// floats (the slices' shares of the squared distances)
size_t dgg_edge_bwd_wide_rows_ws_floats(int64_t N, int K, int h) { return (size_t)(h / 256) * (size_t)N * (size_t)K; }
int dgg_edge_bwd_wide_rows_sliced(const float *xp, int64_t N, int h, const int32_t *idx, const float *val, const float *dval, int K,
                                  int64_t row0, float t, int perturb, float *ws, float *dxp_rows, float *dd, void *stream) {
    if (h % 256 != 0 || K < 1 || K > 64 || ((uintptr_t)xp % 16) || ((uintptr_t)dxp_rows % 16))
        return dgg_set_error(DGG_ERR_UNSUPPORTED, "edge_bwd_wide_rows_sliced: latent_dim a multiple of 256, K in [1,64], 16-byte aligned rows");
    if (!ws || !dd || !dxp_rows) return dgg_set_error(DGG_ERR_ARG, "edge_bwd_wide_rows_sliced: workspace, dd and dxp_rows are required");
    if (N == 0) return 0;
    const int S = h / 256;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(edge_bwd_wide_d2_slice, dim3(rows_grid(N), (unsigned)S), dim3(WPB * 64), 0, st, xp, N, h, idx, dval, K, row0, ws);
    hipLaunchKernelGGL(edge_bwd_wide_own_slice, dim3(rows_grid(N), (unsigned)S), dim3(WPB * 64), 0, st, xp, N, h, idx, val, dval, K, row0, t, perturb,
                       ws, S, dxp_rows, dd);
    return dgg_check_launch("edge_bwd_wide_rows_sliced");
}

int dgg_edge_bwd(const float *xp, int64_t N, int h, const int32_t *idx, const float *val, const float *dval, int K,
                 int64_t row0, float t, int perturb, float *dxp, void *stream) {
    if (N == 0) return 0;
    if (h > 128)
        hipLaunchKernelGGL(edge_bwd_wide_kernel, dim3((unsigned)N), dim3(WPB * 64), 0, (hipStream_t)stream, xp, N, h, idx, val,
                           dval, K, row0, t, perturb, dxp);
    else
        hipLaunchKernelGGL(edge_bwd_kernel, dim3(rows_grid(N)), dim3(WPB * 64), 0, (hipStream_t)stream, xp, N, h, idx, val, dval,
                           K, row0, t, perturb, dxp);
    return dgg_check_launch("edge_bwd");
}

}  // extern "C"
