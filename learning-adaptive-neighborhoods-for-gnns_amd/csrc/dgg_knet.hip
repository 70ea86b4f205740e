// dgg_knet.hip -- the learned-degree estimator (per-node MLP that outputs k_i).
//
// Replaces k_estimate_net of the live class (reference dgm.py:1472-1586) and LearnableKEncoder.forward
// (dgm.py:2051-2063, deterministic branch):
//   mode "x":          nd = (deg - mean) / (std + 1e-5);  z = leaky(k_embed([xk, nd]));  m = k_mu(z);
//                      kp = k_project(m);  k = relu(kp * std + mean) + 1
//   mode "input_deg":  nd from the constants deg_mean/deg_std;  in3 = input_degree_project(nd);
//                      m = k_mu(in3);  kp = k_project(m);  k = relu(kp * deg_std + deg_mean) + 1
// The reference obtains deg by densifying in_adj (dgm.py:1568); here deg is a length-N vector (CSR row sums, or
// the prior degree in all-pairs mode).  Every output unit is an fmaf chain over the input units in ascending order (the
// order of the CPU oracle).  latent_dim in {16,32,64,128} with the reference's h/2, h/4 widths: one THREAD per node, the
// node's activations in registers, the (wave-uniform) weights through scalar loads -- full-rate v_fma, no cross-lane
// traffic.  Other shapes: one wavefront per node, lane o computes output unit o.
#include "dgg_common.h"
#include <algorithm>
#include "dgg_api_internal.h"

using namespace dgg;

namespace dggk {

// mu_sd[0] = mean(deg), mu_sd[1] = unbiased std(deg); double accumulation.  Two passes (mean, then squared deviations)
// over DS_BLOCKS workgroups each, partial sums combined in a fixed order (deterministic), then a one-thread finish.
constexpr int DS_BLOCKS = 256;
__device__ __forceinline__ double ds_block_sum(double v, double *red) {
    const int tid = threadIdx.x;
    red[tid] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    const double r = red[0];
    __syncthreads();
    return r;
}
__device__ __forceinline__ double ds_total(const double *part, double *red) {   // same order in every workgroup
    return ds_block_sum(part[threadIdx.x], red);
}
template <int PASS>
__global__ __launch_bounds__(256) void degree_stats_pass(const float *__restrict__ deg, int64_t N, double *__restrict__ part,
                                                         float *__restrict__ mu_sd) {
    __shared__ double red[256];
    double m = 0.0;
    if (PASS >= 1) m = ds_total(part, red) / (double)N;
    if (PASS == 2) {
        const double v = ds_total(part + DS_BLOCKS, red);
        if (threadIdx.x == 0) { mu_sd[0] = (float)m; mu_sd[1] = (float)sqrt(v / (double)(N - 1)); }
        return;
    }
    double s = 0.0;
    const int64_t stride = (int64_t)gridDim.x * 256 * 4;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < N; i += stride) {
        float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (i + 3 < N && ((reinterpret_cast<uintptr_t>(deg) & 15) == 0)) {
            const float4 q = *reinterpret_cast<const float4 *>(deg + i);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
            for (int u = 0; u < 4; u++) { const double d = (double)v[u] - m; s += PASS == 0 ? d : d * d; }
        } else {
            for (int u = 0; u < 4 && i + u < N; u++) { const double d = (double)deg[i + u] - m; s += PASS == 0 ? d : d * d; }
        }
    }
    s = ds_block_sum(s, red);
    if (threadIdx.x == 0) part[PASS * DS_BLOCKS + blockIdx.x] = s;
}

constexpr int WPB = 4;

// LDS layout: W1 [h2][h+1] | b1 [h2] | Wmu [h4][h2] | bmu [h4] | Wp [h4] | bp
__global__ __launch_bounds__(WPB * 64) void knet_x_fwd_kernel(
    const float *__restrict__ xk, int64_t N, int h, const float *__restrict__ deg, const float *__restrict__ mu_sd,
    const float *__restrict__ W1, const float *__restrict__ b1, int h2, const float *__restrict__ Wmu,
    const float *__restrict__ bmu, int h4, const float *__restrict__ Wp, const float *__restrict__ bp,
    float *__restrict__ k, float *__restrict__ z_save, float *__restrict__ u_save, float *__restrict__ feat_save) {
    extern __shared__ float sm[];
    float *sW1 = sm, *sWmu = sW1 + h2 * (h + 1);
    for (int e = threadIdx.x; e < h2 * (h + 1); e += blockDim.x) sW1[e] = W1[e];
    for (int e = threadIdx.x; e < h4 * h2; e += blockDim.x) sWmu[e] = Wmu[e];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const float mu = mu_sd[0], sd = mu_sd[1];
    for (int64_t i = (int64_t)blockIdx.x * WPB + dgg::wave_id(); i < N; i += (int64_t)gridDim.x * WPB) {
        float nd = __fdiv_rn(__fadd_rn(deg[i], -mu), __fadd_rn(sd, 1e-5f));
        float x0 = lane < h ? xk[i * h + lane] : 0.0f;
        float x1 = lane + 64 < h ? xk[i * h + lane + 64] : 0.0f;
        if (feat_save) {
            if (lane < h) feat_save[i * (h + 1) + lane] = x0;
            if (lane + 64 < h) feat_save[i * (h + 1) + lane + 64] = x1;
            if (lane == 0) feat_save[i * (h + 1) + h] = nd;
        }
        // z_o, o = lane < h2
        float acc = 0.0f;
        const float *wrow = sW1 + (lane < h2 ? lane : 0) * (h + 1);
        for (int c = 0; c < h; c++) {
            float xv = c < 64 ? bcast(x0, c) : bcast(x1, c - 64);
            acc = __fmaf_rn(xv, wrow[c], acc);
        }
        acc = __fmaf_rn(nd, wrow[h], acc);
        acc = __fadd_rn(acc, lane < h2 ? b1[lane] : 0.0f);
        float z = acc > 0.0f ? acc : __fmul_rn(0.01f, acc);
        if (z_save && lane < h2) z_save[i * h2 + lane] = z;
        // m_o, o = lane < h4
        float am = 0.0f;
        const float *mrow = sWmu + (lane < h4 ? lane : 0) * h2;
        for (int c = 0; c < h2; c++) am = __fmaf_rn(bcast(z, c), mrow[c], am);
        float m = __fadd_rn(am, lane < h4 ? bmu[lane] : 0.0f);
        // kp
        float ak = 0.0f;
        for (int c = 0; c < h4; c++) ak = __fmaf_rn(bcast(m, c), Wp[c], ak);
        float kp = __fadd_rn(ak, bp[0]);
        float u = __fadd_rn(__fmul_rn(kp, sd), mu);
        if (lane == 0) {
            k[i] = __fadd_rn(u > 0.0f ? u : 0.0f, 1.0f);
            if (u_save) u_save[i] = u;
        }
    }
}

// per-node backward: dk -> dkp [N], dm [N,h4], dpre1 [N,h2], dxk [N,h]; m is recomputed from z
__global__ __launch_bounds__(WPB * 64) void knet_x_bwd_kernel(
    int64_t N, int h, const float *__restrict__ mu_sd, const float *__restrict__ W1, int h2,
    const float *__restrict__ Wmu, int h4, const float *__restrict__ Wp, const float *__restrict__ z,
    const float *__restrict__ u, const float *__restrict__ dk, float *__restrict__ dkp_out, float *__restrict__ dm_out,
    float *__restrict__ dpre1_out, float *__restrict__ dxk, float *__restrict__ m_out, const float *__restrict__ bmu) {
    extern __shared__ float sm[];
    float *sW1 = sm, *sWmu = sW1 + h2 * (h + 1);
    for (int e = threadIdx.x; e < h2 * (h + 1); e += blockDim.x) sW1[e] = W1[e];
    for (int e = threadIdx.x; e < h4 * h2; e += blockDim.x) sWmu[e] = Wmu[e];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const float sd = mu_sd[1];
    for (int64_t i = (int64_t)blockIdx.x * WPB + dgg::wave_id(); i < N; i += (int64_t)gridDim.x * WPB) {
        float dkp = u[i] > 0.0f ? dk[i] * sd : 0.0f;
        float zl = lane < h2 ? z[i * h2 + lane] : 0.0f;
        // recompute m (needed by the k_project weight gradient)
        float am = 0.0f;
        const float *mrow = sWmu + (lane < h4 ? lane : 0) * h2;
        for (int c = 0; c < h2; c++) am = __fmaf_rn(bcast(zl, c), mrow[c], am);
        if (lane < h4) m_out[i * h4 + lane] = __fadd_rn(am, bmu[lane]);
        float dm = lane < h4 ? dkp * Wp[lane] : 0.0f;
        if (lane < h4) dm_out[i * h4 + lane] = dm;
        // dz_c = sum_o dm_o Wmu[o][c], c = lane < h2
        float dz = 0.0f;
        for (int o = 0; o < h4; o++) dz = fmaf(bcast(dm, o), sWmu[o * h2 + (lane < h2 ? lane : 0)], dz);
        float dp1 = lane < h2 ? (zl > 0.0f ? dz : 0.01f * dz) : 0.0f;
        if (lane < h2) dpre1_out[i * h2 + lane] = dp1;
        // dxk_c = sum_o dp1_o W1[o][c]
        float a0 = 0.0f, a1 = 0.0f;
        for (int o = 0; o < h2; o++) {
            float g = bcast(dp1, o);
            if (lane < h) a0 = fmaf(g, sW1[o * (h + 1) + lane], a0);
            if (lane + 64 < h) a1 = fmaf(g, sW1[o * (h + 1) + lane + 64], a1);
        }
        if (lane < h) dxk[i * h + lane] = a0;
        if (lane + 64 < h) dxk[i * h + lane + 64] = a1;
        if (lane == 0) dkp_out[i] = dkp;
    }
}

// ---- thread-per-node variants (H = latent_dim, H2 = H/2, H4 = H/4) ---------------------------------------------------
template <int H>
__global__ __launch_bounds__(64) void knet_x_fwd_tpn(
    const float *__restrict__ xk, int64_t N, const float *__restrict__ deg, const float *__restrict__ mu_sd,
    const float *__restrict__ W1, const float *__restrict__ b1, const float *__restrict__ Wmu,
    const float *__restrict__ bmu, const float *__restrict__ Wp, const float *__restrict__ bp,
    float *__restrict__ k, float *__restrict__ z_save, float *__restrict__ u_save, float *__restrict__ feat_save) {
    constexpr int H2 = H / 2, H4 = H / 4;
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= N) return;
    const float mu = mu_sd[0], sd = mu_sd[1];
    const float nd = __fdiv_rn(__fadd_rn(deg[i], -mu), __fadd_rn(sd, 1e-5f));
    float x[H];
#pragma unroll
    for (int c = 0; c < H; c += 4) {
        const float4 v = *reinterpret_cast<const float4 *>(xk + i * H + c);
        x[c] = v.x; x[c + 1] = v.y; x[c + 2] = v.z; x[c + 3] = v.w;
    }
    if (feat_save) {
        float *f = feat_save + i * (H + 1);
#pragma unroll
        for (int c = 0; c < H; c++) f[c] = x[c];
        f[H] = nd;
    }
    float z[H2];
#pragma unroll
    for (int o = 0; o < H2; o++) {
        float acc = 0.0f;
#pragma unroll
        for (int c = 0; c < H; c++) acc = __fmaf_rn(x[c], W1[o * (H + 1) + c], acc);
        acc = __fmaf_rn(nd, W1[o * (H + 1) + H], acc);
        acc = __fadd_rn(acc, b1[o]);
        z[o] = acc > 0.0f ? acc : __fmul_rn(0.01f, acc);
    }
    if (z_save) {
#pragma unroll
        for (int o = 0; o < H2; o += 4)
            *reinterpret_cast<float4 *>(z_save + i * H2 + o) = make_float4(z[o], z[o + 1], z[o + 2], z[o + 3]);
    }
    float ak = 0.0f;
#pragma unroll
    for (int o = 0; o < H4; o++) {
        float am = 0.0f;
#pragma unroll
        for (int c = 0; c < H2; c++) am = __fmaf_rn(z[c], Wmu[o * H2 + c], am);
        const float m = __fadd_rn(am, bmu[o]);
        ak = __fmaf_rn(m, Wp[o], ak);
    }
    const float kp = __fadd_rn(ak, bp[0]);
    const float u = __fadd_rn(__fmul_rn(kp, sd), mu);
    k[i] = __fadd_rn(u > 0.0f ? u : 0.0f, 1.0f);
    if (u_save) u_save[i] = u;
}

template <int H>
__global__ __launch_bounds__(64) void knet_x_bwd_tpn(
    int64_t N, const float *__restrict__ mu_sd, const float *__restrict__ W1, const float *__restrict__ Wmu,
    const float *__restrict__ Wp, const float *__restrict__ z, const float *__restrict__ u, const float *__restrict__ dk,
    float *__restrict__ dkp_out, float *__restrict__ dm_out, float *__restrict__ dpre1_out, float *__restrict__ dxk,
    float *__restrict__ m_out, const float *__restrict__ bmu) {
    constexpr int H2 = H / 2, H4 = H / 4;
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= N) return;
    const float sd = mu_sd[1];
    const float dkp = u[i] > 0.0f ? dk[i] * sd : 0.0f;
    dkp_out[i] = dkp;
    float zl[H2];
#pragma unroll
    for (int c = 0; c < H2; c += 4) {
        const float4 v = *reinterpret_cast<const float4 *>(z + i * H2 + c);
        zl[c] = v.x; zl[c + 1] = v.y; zl[c + 2] = v.z; zl[c + 3] = v.w;
    }
    float mm[H4], dm[H4];
#pragma unroll
    for (int o = 0; o < H4; o++) {                               // m recomputed (k_project weight gradient)
        float am = 0.0f;
#pragma unroll
        for (int c = 0; c < H2; c++) am = __fmaf_rn(zl[c], Wmu[o * H2 + c], am);
        mm[o] = __fadd_rn(am, bmu[o]);
        dm[o] = dkp * Wp[o];
    }
#pragma unroll
    for (int o = 0; o < H4; o += 4) {
        *reinterpret_cast<float4 *>(m_out + i * H4 + o) = make_float4(mm[o], mm[o + 1], mm[o + 2], mm[o + 3]);
        *reinterpret_cast<float4 *>(dm_out + i * H4 + o) = make_float4(dm[o], dm[o + 1], dm[o + 2], dm[o + 3]);
    }
    float dp1[H2];
#pragma unroll
    for (int c = 0; c < H2; c++) {
        float dz = 0.0f;
#pragma unroll
        for (int o = 0; o < H4; o++) dz = fmaf(dm[o], Wmu[o * H2 + c], dz);
        dp1[c] = zl[c] > 0.0f ? dz : 0.01f * dz;
    }
#pragma unroll
    for (int c = 0; c < H2; c += 4)
        *reinterpret_cast<float4 *>(dpre1_out + i * H2 + c) = make_float4(dp1[c], dp1[c + 1], dp1[c + 2], dp1[c + 3]);
#pragma unroll
    for (int c = 0; c < H; c += 4) {
        float a[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int o = 0; o < H2; o++) {
#pragma unroll
            for (int q = 0; q < 4; q++) a[q] = fmaf(dp1[o], W1[o * (H + 1) + c + q], a[q]);
        }
        *reinterpret_cast<float4 *>(dxk + i * H + c) = make_float4(a[0], a[1], a[2], a[3]);
    }
}

// ---- the k-net (mode "x") on the fp32 MATRIX cores ---------------------------------------------------------------------------
// k_embed (Linear(h+1, h/2) + LeakyReLU) -> k_mu (Linear(h/2, h/4)) -> k_project (Linear(h/4, 1)) is a small MLP over N nodes: the
// thread-per-node kernels above run it at 9 % of the fp32 vector peak (52 + 45 us per step at N = 100k) and hand z, [xk | nd], dkp,
// dm, dpre1 and m through HBM to three weight-gradient GEMMs (+ 45 us, 160 MB in all).  Here a wavefront owns 32 nodes at a time
// and every layer is a chain of v_mfma_f32_32x32x2_f32 (an exact k-ordered fmaf chain, so k comes out bit-identical):
//   D[o][n] += W[o][c] * act[n][c]:  first operand = the layer's weights in operand order (lane (li, hh) of step s holds
//   W[li][2s + hh]; the bias rides as one more contraction step against a column of ones); second operand = the activations of node
//   li; an accumulator register r of lane (li, hh) is output feature (r & 3) + 8 (r >> 2) + 4 hh of node li.
// Backward (knet_x_bwd_reg): layer 1 is re-run from xk (the forward saves only u); dz and d feat are MFMA chains whose second
// operands are computed on the fly / ARE the previous chain's accumulator registers (the contraction index is walked in the
// accumulator's own order); the weight gradients are contractions over the 32 nodes of the block, accumulated in registers over all
// the blocks of a wavefront:
//   G1 [h/2 x (h+2)]   = sum_n dpre1_n (x) [xk_n | nd_n | 1]   -> dW1 and (ones column) db1
//   G2 [(h/4+1) x h/2] = sum_n [dm_n ; dkp_n] (x) z_n          -> dWmu and v = sum_n dkp_n z_n
//   dbp = S0 = sum dkp,  dbmu = Wp S0,  dWp = Wmu v + bmu S0   (dm_n = dkp_n Wp, m_n = Wmu z_n + bmu; knet_bwd_reduce)
typedef float kf32x16 __attribute__((ext_vector_type(16)));
// Round 4: the forward with the node rows in REGISTERS.  Round 3 staged every 64-node group through a 26 KB LDS tile per wavefront (one
// workgroup per CU, one wavefront per SIMD, nothing to hide the load -> LDS -> MFMA chain behind: MFMA-busy 0.11, 29 us for a
// 33 MB stream).  Here a wavefront owns ONE block of 32 nodes:
//  * layer-1 operand (node li, k = 2s + half): lane (li, half) loads the contiguous half row [half*H/2, (half+1)*H/2) with 16-byte
//    loads; one v_permlane32_swap per register pair turns (reg 2t, reg 2t+1) into the operands of steps t and H/4 + t
//    (linear_fwd_reg's trick, dgg_linear.hip);
//  * layer-2 operand (node li, k = 2s + half) out of the layer-1 ACCUMULATOR: lane (li, half) holds outputs o = (r&3) + 8(r>>2) +
//    4 half; after one swap per register pair (r, r+1), r even, the lower half holds the even outputs o(r), o(r) + 4 and the
//    upper half the odd ones o(r) + 1, o(r) + 5 -- exactly what the k-ordered chain of the next layer wants from each half.
// Same k-ordered fmaf chains as the thread-per-node kernels: the same bits.  One block per wavefront, eight wavefronts per workgroup
// (the weights reach the lanes through 13 KB of LDS, staged once per workgroup).  Measured at N = 100 000 (us): 18.7 against 29 for the
// LDS-staged persistent form of round 3; of those, ~5 are the MFMA chain (not overlapped: every wavefront walks the same phases at the
// same time), ~3 the row loads, ~5 launch + staging; a coalesced-load + LDS-transpose variant measured the same (20.1).
template <int H>
__global__ __launch_bounds__(512) void knet_x_fwd_reg(const float *__restrict__ xk, int64_t N, const float *__restrict__ deg,
                                                      const float *__restrict__ mu_sd, const float *__restrict__ W1,
                                                      const float *__restrict__ b1, const float *__restrict__ Wmu,
                                                      const float *__restrict__ bmu, const float *__restrict__ Wp,
                                                      const float *__restrict__ bp, float *__restrict__ k, float *__restrict__ u_save) {
    constexpr int H2 = H / 2, H4 = H / 4, XR = H / 2, S1 = H / 2 + 1, S2 = H2 / 2 + 1;
    // the weights in operand order, once per workgroup of eight wavefronts
    __shared__ float wsh[(S1 + S2) * 64];
    const int lane = threadIdx.x & 63, wave = dgg::wave_id(), li = lane & 31, hh = lane >> 5;
    constexpr int WAVES = 8;
    const int64_t blk = (int64_t)blockIdx.x * WAVES + wave;
    const int64_t n = blk * 32 + li;
    const bool valid = n < N;
    const int64_t nc = valid ? n : N - 1;
    float xc[XR];
    {                                                            // (in flight across the staging and its barrier)
        const float4 *p = reinterpret_cast<const float4 *>(xk + nc * H + hh * (H / 2));
#pragma unroll
        for (int q = 0; q < XR / 4; q++) {
            const float4 v = p[q];
            xc[4 * q] = v.x; xc[4 * q + 1] = v.y; xc[4 * q + 2] = v.z; xc[4 * q + 3] = v.w;
        }
    }
    const float dg = deg[nc];
    const float mu = mu_sd[0], sd = mu_sd[1];
    for (int e = threadIdx.x; e < (S1 + S2) * 64; e += WAVES * 64) {
        const int l = e & 63, sI = e >> 6, o = l & 31, c = 2 * (sI < S1 ? sI : sI - S1) + (l >> 5);
        float v;
        if (sI < S1) v = o < H2 ? (c <= H ? W1[o * (H + 1) + c] : b1[o]) : 0.0f;
        else v = o < H4 ? (c < H2 ? Wmu[o * H2 + c] : (c == H2 ? bmu[o] : 0.0f)) : 0.0f;
        wsh[e] = v;
    }
    __syncthreads();
    if (blk * 32 >= N) return;
    float w1[S1], w2[S2];
#pragma unroll
    for (int sI = 0; sI < S1; sI++) w1[sI] = wsh[sI * 64 + lane];
#pragma unroll
    for (int sI = 0; sI < S2; sI++) w2[sI] = wsh[(S1 + sI) * 64 + lane];
    float wp[H4];
#pragma unroll
    for (int o = 0; o < H4; o++) wp[o] = Wp[o];
    const float bpv = bp[0];
#pragma unroll
    for (int t = 0; t < XR / 2; t++) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(xc[2 * t]), __float_as_uint(xc[2 * t + 1]), false, false);
        xc[2 * t] = __uint_as_float(r[0]);                       // operand of step t        (k = 2t + half)
        xc[2 * t + 1] = __uint_as_float(r[1]);                   // operand of step H/4 + t  (k = H/2 + 2t + half)
    }
    const float nd = __fdiv_rn(__fadd_rn(dg, -mu), __fadd_rn(sd, 1e-5f));
    kf32x16 a1;
#pragma unroll
    for (int r = 0; r < 16; r++) a1[r] = 0.0f;
#pragma unroll
    for (int sI = 0; sI < S1 - 1; sI++)
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[sI], sI < H / 4 ? xc[2 * sI] : xc[2 * (sI - H / 4) + 1], a1, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[S1 - 1], hh == 0 ? nd : 1.0f, a1, 0, 0, 0);      // k = H: nd, k = H + 1: the bias
    float zs[16];
#pragma unroll
    for (int r = 0; r < 16; r++) zs[r] = a1[r] > 0.0f ? a1[r] : __fmul_rn(0.01f, a1[r]);
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        const auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(zs[r]), __float_as_uint(zs[r + 1]), false, false);
        zs[r] = __uint_as_float(q[0]);                           // lower half: o(r)      upper half: o(r) + 1
        zs[r + 1] = __uint_as_float(q[1]);                       // lower half: o(r) + 4  upper half: o(r) + 5
    }
    kf32x16 a2;
#pragma unroll
    for (int r = 0; r < 16; r++) a2[r] = 0.0f;
#pragma unroll
    for (int sI = 0; sI < S2 - 1; sI++) {
        const int o = 2 * sI;                                    // (the lower half's k; the upper half's k = o + 1 sits in the same register)
        const int pr = 4 * (o >> 3) + (o & 2);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(w2[sI], (o & 4) ? zs[pr + 1] : zs[pr], a2, 0, 0, 0);
    }
    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(w2[S2 - 1], hh == 0 ? 1.0f : 0.0f, a2, 0, 0, 0);     // k = H2: the bias
    // k_project: one lane per node gathers the h/4 values of m (its own rows and the other half's) and runs the ascending chain
    float m[16];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const float other = __uint_as_float(dgg::xor_shfl<32>(__float_as_uint(a2[r]), lane));
        const int a_mine = (r & 3) + 8 * (r >> 2) + 4 * hh, a_oth = (r & 3) + 8 * (r >> 2) + 4 * (1 - hh);
        m[a_mine & 15] = a2[r];
        m[a_oth & 15] = other;
    }
    float ak = 0.0f;
#pragma unroll
    for (int o = 0; o < H4; o++) ak = __fmaf_rn(m[o], wp[o], ak);
    const float kp = __fadd_rn(ak, bpv);
    const float u = __fadd_rn(__fmul_rn(kp, sd), mu);
    if (hh == 0 && valid) {
        k[n] = __fadd_rn(u > 0.0f ? u : 0.0f, 1.0f);
        if (u_save) u_save[n] = u;
    }
}

// The backward with the layer-1 operands in registers (as knet_x_fwd_reg), 32-node blocks, two wavefronts per SIMD, and the
// weight-gradient partials leaving the kernel as PLAIN stores (one slab per workgroup; round 3: 2.2k same-address float atomics from
// each of 256 workgroups); knet_bwd_reduce sums the slabs and runs the parameter-sized tail (dbp, dbmu, dWp).
// Measured at N = 100 000 (us): 32.2 + 7.5 against 67.6 + 4.8 for round 3's LDS-staged, one-wavefront-per-SIMD form.
// LDS per wavefront: the two 32 x 33 transposition tiles (dpre1 / z, node-major: first operands of G1 / G2), the block's rows
// [32][68] (second operand of G1, read by column) and nd / dkp of its 32 nodes; W1 in accumulator order once per workgroup.
template <int H>
struct KnetBwd {
    static constexpr int H2 = H / 2, H4 = H / 4, NB = (H + 2 + 31) / 32, NBX = H / 32 > 0 ? H / 32 : 1;
    static constexpr int XS = H + 4;                              // row stride of the X tile (16-byte aligned rows, conflict-free b128 writes)
    static constexpr int WAVE_F = 2 * 32 * 33 + 32 * XS + 64;     // floats of LDS per wavefront
    static constexpr int W1B_F = NBX * 16 * 64;                   // W1 in accumulator order, shared by the workgroup
    static constexpr int NREG = (NB + 1) * 16;                    // accumulator registers of a wavefront's partial sums (G1 blocks, G2)
    static constexpr int SLAB_F = (NREG + 1) * 64;                // + the per-lane partial of S0
};

template <int H>
__global__ __launch_bounds__(256, 2) void knet_x_bwd_reg(const float *__restrict__ xk, int64_t N, const float *__restrict__ deg,
                                                         const float *__restrict__ mu_sd, const float *__restrict__ W1,
                                                         const float *__restrict__ b1, const float *__restrict__ Wmu,
                                                         const float *__restrict__ Wp, const float *__restrict__ u,
                                                         const float *__restrict__ dk, float *__restrict__ dxk, float *__restrict__ slab,
                                                         int out_act) {
    using KB = KnetBwd<H>;
    constexpr int H2 = KB::H2, H4 = KB::H4, NB = KB::NB, NBX = KB::NBX, XS = KB::XS, XR = H / 2, S1 = H / 2 + 1;
    constexpr int KS3 = (H4 + 1) / 2;                             // dz: contraction over the h/4 outputs of k_mu
    constexpr int RED_F = 4 * KB::SLAB_F;
    constexpr int LDS_F = (4 * KB::WAVE_F + KB::W1B_F) > RED_F ? (4 * KB::WAVE_F + KB::W1B_F) : RED_F;
    __shared__ __attribute__((aligned(16))) float sm[LDS_F];
    const int tid = threadIdx.x, lane = tid & 63, wave = dgg::wave_id(), li = lane & 31, hh = lane >> 5;
    float *t1 = sm + wave * KB::WAVE_F, *t2 = t1 + 32 * 33, *xt = t2 + 32 * 33, *ndl = xt + 32 * XS, *dkl = ndl + 32;
    float *w1b = sm + 4 * KB::WAVE_F;                             // [cb][sI][lane] = W1[o(sI, half)][cb*32 + li]
    for (int e = tid; e < KB::W1B_F; e += 256) {
        const int l = e & 63, sI = (e >> 6) & 15, cb = e >> 10;
        const int o = (sI & 3) + 8 * (sI >> 2) + 4 * (l >> 5), c = cb * 32 + (l & 31);
        w1b[e] = (o < H2 && c < H) ? W1[o * (H + 1) + c] : 0.0f;
    }
    const float mu = mu_sd[0], sd = mu_sd[1];
    // layer 1 (forward order) and Wmu^T in operand order, staged once per workgroup in the wavefronts' tile space (read back into
    // registers before the first block overwrites it)
    float *stg = sm;                                              // [S1 + KS3][64]
    for (int e = tid; e < (S1 + KS3) * 64; e += 256) {
        const int l = e & 63, sI = e >> 6, o = l & 31, hq = l >> 5;
        float v;
        if (sI < S1) { const int c = 2 * sI + hq; v = o < H2 ? (c <= H ? W1[o * (H + 1) + c] : b1[o]) : 0.0f; }
        else { const int a = 2 * (sI - S1) + hq; v = (o < H2 && a < H4) ? Wmu[a * H2 + o] : 0.0f; }
        stg[e] = v;
    }
    __syncthreads();
    float w1[S1];
#pragma unroll
    for (int sI = 0; sI < S1; sI++) w1[sI] = stg[sI * 64 + lane];
    float wmt[KS3], wp2[KS3];                                     // dz[o][n] += Wmu[a][o] * dm[n][a]: first operand Wmu^T, second dkp * Wp[a]
#pragma unroll
    for (int sI = 0; sI < KS3; sI++) {
        const int a = 2 * sI + hh;
        wmt[sI] = stg[(S1 + sI) * 64 + lane];
        wp2[sI] = a < H4 ? Wp[a] : 0.0f;
    }
    const float fa = li < H4 ? Wp[li] : (li == H4 ? 1.0f : 0.0f);   // G2 first operand = fa * dkp[n]: rows dm (a < h/4) and dkp (a = h/4)
    __syncthreads();                                              // staging space is tile space from here on
    kf32x16 g1[NB], g2;
#pragma unroll
    for (int a = 0; a < NB; a++)
#pragma unroll
        for (int r = 0; r < 16; r++) g1[a][r] = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; r++) g2[r] = 0.0f;
    float s0 = 0.0f;
    __syncthreads();                                              // w1b
    const int64_t nblk = (N + 31) / 32;
    for (int64_t blk = (int64_t)blockIdx.x * 4 + wave; blk < nblk; blk += (int64_t)gridDim.x * 4) {
        const int64_t n = blk * 32 + li;
        const bool valid = n < N;
        const int64_t nc = valid ? n : N - 1;
        float xc[XR];
        {
            const float4 *p = reinterpret_cast<const float4 *>(xk + nc * H + hh * (H / 2));
#pragma unroll
            for (int q = 0; q < XR / 4; q++) {
                const float4 v = p[q];
                xc[4 * q] = v.x; xc[4 * q + 1] = v.y; xc[4 * q + 2] = v.z; xc[4 * q + 3] = v.w;
            }
        }
        const float dg = deg[nc], uu = u[nc], dkv = dk[nc];
        // the block's rows, row-major, for the column reads of G1 (before the operand swap: the half row is contiguous)
#pragma unroll
        for (int q = 0; q < XR / 4; q++)
            *reinterpret_cast<float4 *>(xt + li * XS + hh * (H / 2) + 4 * q) = make_float4(xc[4 * q], xc[4 * q + 1], xc[4 * q + 2], xc[4 * q + 3]);
        const float nd = __fdiv_rn(__fadd_rn(dg, -mu), __fadd_rn(sd, 1e-5f));
        const float dkp = (valid && uu > 0.0f) ? dkv * sd : 0.0f;
        if (hh == 0) { s0 += dkp; ndl[li] = nd; dkl[li] = dkp; }
#pragma unroll
        for (int t = 0; t < XR / 2; t++) {
            const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(xc[2 * t]), __float_as_uint(xc[2 * t + 1]), false, false);
            xc[2 * t] = __uint_as_float(r[0]);
            xc[2 * t + 1] = __uint_as_float(r[1]);
        }
        kf32x16 a1;
#pragma unroll
        for (int r = 0; r < 16; r++) a1[r] = 0.0f;
#pragma unroll
        for (int sI = 0; sI < S1 - 1; sI++)
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[sI], sI < H / 4 ? xc[2 * sI] : xc[2 * (sI - H / 4) + 1], a1, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[S1 - 1], hh == 0 ? nd : 1.0f, a1, 0, 0, 0);
        kf32x16 dz;
#pragma unroll
        for (int r = 0; r < 16; r++) dz[r] = 0.0f;
#pragma unroll
        for (int sI = 0; sI < KS3; sI++) dz = __builtin_amdgcn_mfma_f32_32x32x2f32(wmt[sI], dkp * wp2[sI], dz, 0, 0, 0);
        kf32x16 dp1;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float z = a1[r] > 0.0f ? a1[r] : 0.01f * a1[r];
            dp1[r] = a1[r] > 0.0f ? dz[r] : 0.01f * dz[r];
            const int o = (r & 3) + 8 * (r >> 2) + 4 * hh;
            t1[li * 33 + o] = o < H2 ? dp1[r] : 0.0f;             // dpre1, node-major: first operand of G1
            t2[li * 33 + o] = o < H2 ? z : 0.0f;                  // z, node-major: second operand of G2
        }
        // dxk = W1[:, :h]^T dpre1: the contraction runs over the accumulator registers themselves
#pragma unroll
        for (int cb = 0; cb < NBX; cb++) {
            kf32x16 df;
#pragma unroll
            for (int r = 0; r < 16; r++) df[r] = 0.0f;
            float wv[16];
#pragma unroll
            for (int sI = 0; sI < 16; sI++) wv[sI] = w1b[(cb * 16 + sI) * 64 + lane];
#pragma unroll
            for (int sI = 0; sI < 16; sI++) df = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[sI], dp1[sI], df, 0, 0, 0);
            if (valid) {
#pragma unroll
                for (int q4 = 0; q4 < 4; q4++) {
                    const int c = cb * 32 + 8 * q4 + 4 * hh;
                    if (c < H) {
                        float4 v = make_float4(df[4 * q4], df[4 * q4 + 1], df[4 * q4 + 2], df[4 * q4 + 3]);
                        if (out_act == 1) {                      // d loss / d (pre-activation of xk): LeakyReLU'(xk), the row is in the tile
                            const float4 xv = *reinterpret_cast<const float4 *>(xt + li * XS + c);
                            v.x *= xv.x > 0.0f ? 1.0f : 0.01f; v.y *= xv.y > 0.0f ? 1.0f : 0.01f;
                            v.z *= xv.z > 0.0f ? 1.0f : 0.01f; v.w *= xv.w > 0.0f ? 1.0f : 0.01f;
                        }
                        *reinterpret_cast<float4 *>(dxk + n * H + c) = v;
                    }
                }
            }
        }
        // G1 [o][c] += sum_n dpre1[n][o] * [xk | nd | 1][n][c]: contraction over the block's 32 nodes (2 per step)
#pragma unroll
        for (int s8 = 0; s8 < 16; s8 += 8) {                      // 8 steps' operands in flight, then their 8 * NB MFMAs
            float av[8], bv[8][NB];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int nn = 2 * (s8 + q) + hh;
                av[q] = t1[nn * 33 + li];
#pragma unroll
                for (int a = 0; a < NB; a++) {
                    const int c = a * 32 + li;
                    bv[q][a] = (a + 1) * 32 <= H ? xt[nn * XS + c] : (c < H ? xt[nn * XS + c] : (c == H ? ndl[nn] : (c == H + 1 ? 1.0f : 0.0f)));
                }
            }
#pragma unroll
            for (int q = 0; q < 8; q++)
#pragma unroll
                for (int a = 0; a < NB; a++) g1[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q][a], g1[a], 0, 0, 0);
        }
        // G2 [(dm ; dkp)][c] += sum_n (fa * dkp[n]) * z[n][c]
        {
            float av[16], bv[16];
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int nn = 2 * q + hh;
                av[q] = fa * dkl[nn];
                bv[q] = t2[nn * 33 + li];
            }
#pragma unroll
            for (int q = 0; q < 16; q++) g2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q], g2, 0, 0, 0);
        }
    }
    // the workgroup's four wavefronts summed through LDS, one slab of plain stores per workgroup
    __syncthreads();
    float *red = sm;                                               // [4][NREG + 1][64]
#pragma unroll
    for (int a = 0; a < NB; a++)
#pragma unroll
        for (int r = 0; r < 16; r++) red[(wave * (KB::NREG + 1) + a * 16 + r) * 64 + lane] = g1[a][r];
#pragma unroll
    for (int r = 0; r < 16; r++) red[(wave * (KB::NREG + 1) + NB * 16 + r) * 64 + lane] = g2[r];
    red[(wave * (KB::NREG + 1) + KB::NREG) * 64 + lane] = s0;
    __syncthreads();
    float *out = slab + (int64_t)blockIdx.x * KB::SLAB_F;
    for (int e = tid; e < KB::SLAB_F; e += 256)
        out[e] = (red[e] + red[KB::SLAB_F + e]) + (red[2 * KB::SLAB_F + e] + red[3 * KB::SLAB_F + e]);
}

// sums the workgroup slabs of knet_x_bwd_reg: block q < NREG reduces accumulator register q (64 values) over all slabs and writes
// them to their places in gW1 / gb1 / gWmu; the block of G2's register 8 (row h/4: v = sum_n dkp_n z_n) also sums S0 and runs the
// parameter-sized tail dbp = S0, dbmu = Wp S0, dWp = Wmu v + bmu S0
template <int H>
__global__ __launch_bounds__(1024) void knet_bwd_reduce(const float *__restrict__ slab, int nslab, const float *__restrict__ Wmu,
                                                        const float *__restrict__ bmu, const float *__restrict__ Wp, float *__restrict__ gW1,
                                                        float *__restrict__ gb1, float *__restrict__ gWmu, float *__restrict__ gbmu,
                                                        float *__restrict__ gWp, float *__restrict__ gbp) {
    using KB = KnetBwd<H>;
    constexpr int H2 = KB::H2, H4 = KB::H4, NB = KB::NB;
    __shared__ float part[16][64], vsh[64], s0sh;
    const int tid = threadIdx.x, lane = tid & 63, pw = tid >> 6, li = lane & 31, hh = lane >> 5;
    const int q = blockIdx.x, a = q >> 4, r = q & 15;
    auto column = [&](int reg) {                                 // 16 wavefronts x 8 loads in flight over the slabs
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; j++) acc[j] = 0.0f;
        for (int s0 = pw; s0 < nslab; s0 += 128) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int sI = s0 + 16 * j;
                acc[j] += sI < nslab ? slab[(int64_t)sI * KB::SLAB_F + reg * 64 + lane] : 0.0f;
            }
        }
        part[pw][lane] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
        __syncthreads();
        float v = 0.0f;
#pragma unroll
        for (int j = 0; j < 16; j++) v += part[j][lane];
        __syncthreads();
        return v;
    };
    const float v = column(q);
    const int row = (r & 3) + 8 * (r >> 2) + 4 * hh, col = (a < NB ? a * 32 : 0) + li;
    if (pw == 0) {
        if (a < NB) {
            if (row < H2) {
                if (col <= H) gW1[row * (H + 1) + col] = v;
                else if (col == H + 1) gb1[row] = v;
            }
        } else if (col < H2) {
            if (row < H4) gWmu[row * H2 + col] = v;
        }
    }
    // row h/4 of G2 lives in register (h/4 & 3) + 4 (h/4 >> 3) of the half (h/4 >> 2) & 1
    constexpr int VR = (H4 & 3) + 4 * (H4 >> 3), VH = (H4 >> 2) & 1;
    if (q == NB * 16 + VR) {
        if (pw == 0 && hh == VH) vsh[li] = v;
        const float sv = column(KB::NREG);
        float tot = sv;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) tot += __shfl_xor(tot, off, 64);
        if (tid == 0) s0sh = tot;
        __syncthreads();
        const float S0 = s0sh;
        if (tid == 0) gbp[0] = S0;
        if (tid < H4) {
            gbmu[tid] = Wp[tid] * S0;
            float acc = bmu[tid] * S0;
            for (int c = 0; c < H2; c++) acc = fmaf(Wmu[tid * H2 + c], vsh[c], acc);
            gWp[tid] = acc;
        }
    }
}

// degree-only modes (dgm.py:1492-1526): nd = (deg - mu) / (sd + eps); mu/sd are constants ("input_deg", eps 1e-5) or the
// batch statistics read from device memory ("learn_normalized_degree", eps 0)
__global__ void knet_deg_fwd_kernel(const float *__restrict__ deg, int64_t N, const float *__restrict__ mu_sd, float dmean,
                                    float dstd, float eps, const float *__restrict__ Wd, const float *__restrict__ bd,
                                    const float *__restrict__ Wmu, const float *__restrict__ bmu, int h4,
                                    const float *__restrict__ Wp, const float *__restrict__ bp, float *__restrict__ k,
                                    float *__restrict__ u_save) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    if (mu_sd) { dmean = mu_sd[0]; dstd = mu_sd[1]; }
    float nd = __fdiv_rn(__fadd_rn(deg[i], -dmean), __fadd_rn(dstd, eps));
    float in3[3];
    for (int o = 0; o < 3; o++) in3[o] = __fadd_rn(__fmaf_rn(nd, Wd[o], 0.0f), bd[o]);
    float ak = 0.0f;
    for (int o = 0; o < h4; o++) {
        float acc = 0.0f;
        for (int c = 0; c < 3; c++) acc = __fmaf_rn(in3[c], Wmu[o * 3 + c], acc);
        float m = __fadd_rn(acc, bmu[o]);
        ak = __fmaf_rn(m, Wp[o], ak);
    }
    float kp = __fadd_rn(ak, bp[0]);
    float u = __fadd_rn(__fmul_rn(kp, dstd), dmean);
    k[i] = __fadd_rn(u > 0.0f ? u : 0.0f, 1.0f);
    if (u_save) u_save[i] = u;
}

// backward: the net is affine in nd_i, so all parameter gradients follow from S0 = sum dkp, S1 = sum dkp * nd
// (dkp = dk * sd * [u > 0]); block partials in double, one float atomic per block and sum
__global__ __launch_bounds__(256) void knet_deg_bwd_sums_kernel(const float *__restrict__ deg, int64_t N,
                                                                const float *__restrict__ mu_sd, float dmean, float dstd,
                                                                float eps, const float *__restrict__ u,
                                                                const float *__restrict__ dk, float *__restrict__ S) {
    __shared__ double r0[256], r1[256];
    if (mu_sd) { dmean = mu_sd[0]; dstd = mu_sd[1]; }
    double s0 = 0.0, s1 = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256) {
        const double dkp = u[i] > 0.0f ? (double)dk[i] * dstd : 0.0;
        s0 += dkp;
        s1 += dkp * (((double)deg[i] - dmean) / ((double)dstd + eps));
    }
    r0[threadIdx.x] = s0; r1[threadIdx.x] = s1;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { r0[threadIdx.x] += r0[threadIdx.x + o]; r1[threadIdx.x] += r1[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { atomicAdd(S, (float)r0[0]); atomicAdd(S + 1, (float)r1[0]); }
}

// ---- building blocks of the k-net for latent widths beyond the register-resident kernels (latent_dim > 128, e.g. the
// PPI configuration's 2048): the three layers run as MFMA GEMMs (dgg_linear_fwd / dgg_linear_bwd on feat = [xk | nd]:
// the same k-ordered fmaf chains as above), with these elementwise ends
__global__ void knet_feat_kernel(const float *__restrict__ xk, const float *__restrict__ deg, const float *__restrict__ mu_sd,
                                 int64_t N, int h, float *__restrict__ feat) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N * (h + 1)) return;
    const int64_t i = e / (h + 1);
    const int c = (int)(e % (h + 1));
    feat[e] = c < h ? xk[i * h + c] : __fdiv_rn(__fadd_rn(deg[i], -mu_sd[0]), __fadd_rn(mu_sd[1], 1e-5f));
}
__global__ void knet_out_fwd_kernel(const float *__restrict__ kp, const float *__restrict__ mu_sd, int64_t N,
                                    float *__restrict__ k, float *__restrict__ u) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float uu = __fadd_rn(__fmul_rn(kp[i], mu_sd[1]), mu_sd[0]);
    k[i] = __fadd_rn(uu > 0.0f ? uu : 0.0f, 1.0f);
    u[i] = uu;
}
__global__ void knet_out_bwd_kernel(const float *__restrict__ u, const float *__restrict__ dk, const float *__restrict__ mu_sd,
                                    int64_t N, float *__restrict__ dkp) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    dkp[i] = u[i] > 0.0f ? dk[i] * mu_sd[1] : 0.0f;
}

}  // namespace dggk
using namespace dggk;

extern "C" {

size_t dgg_degree_stats_ws_bytes(void) { return 2 * DS_BLOCKS * sizeof(double); }

int dgg_degree_stats(const float *deg, int64_t N, float *mu_sd, void *ws, void *stream) {
    if (N < 2) return dgg_set_error(DGG_ERR_ARG, "degree_stats needs N >= 2");
    if (!ws) return dgg_set_error(DGG_ERR_ARG, "degree_stats needs dgg_degree_stats_ws_bytes() bytes of workspace");
    double *part = reinterpret_cast<double *>(ws);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(degree_stats_pass<0>, dim3(DS_BLOCKS), dim3(256), 0, st, deg, N, part, mu_sd);
    hipLaunchKernelGGL(degree_stats_pass<1>, dim3(DS_BLOCKS), dim3(256), 0, st, deg, N, part, mu_sd);
    hipLaunchKernelGGL(degree_stats_pass<2>, dim3(1), dim3(256), 0, st, deg, N, part, mu_sd);
    return dgg_check_launch("degree_stats");
}

// feat_save (nullable): [N, h+1] = [xk | nd], saved for the k_embed weight gradient
int dgg_knet_x_fwd(const float *xk, int64_t N, int h, const float *deg, const float *mu_sd, const float *W1,
                   const float *b1, int h2, const float *Wmu, const float *bmu, int h4, const float *Wp, const float *bp,
                   float *k, float *z_save, float *u_save, float *feat_save, void *stream) {
    if (h > 128 || h2 > 64 || h4 > 64) return dgg_set_error(DGG_ERR_UNSUPPORTED, "k-net supports latent_dim <= 128");
    if (N == 0) return 0;
    if (h2 * 2 == h && h4 * 4 == h && (h == 16 || h == 32 || h == 64 || h == 128)) {
        const unsigned nblk = (unsigned)((N + 63) / 64);
#define DGG_KNET_FWD(HH)                                                                                                   \
    hipLaunchKernelGGL(knet_x_fwd_tpn<HH>, dim3(nblk), dim3(64), 0, (hipStream_t)stream, xk, N, deg, mu_sd, W1, b1, Wmu, bmu, Wp, \
                       bp, k, z_save, u_save, feat_save)
        switch (h) {
            case 16: DGG_KNET_FWD(16); break;
            case 32: DGG_KNET_FWD(32); break;
            case 64: DGG_KNET_FWD(64); break;
            default: DGG_KNET_FWD(128); break;
        }
#undef DGG_KNET_FWD
        return dgg_check_launch("knet_x_fwd");
    }
    size_t lds = sizeof(float) * ((size_t)h2 * (h + 1) + (size_t)h4 * h2);
    unsigned blocks = (unsigned)((N + WPB - 1) / WPB < 2048 ? (N + WPB - 1) / WPB : 2048);
    hipLaunchKernelGGL(knet_x_fwd_kernel, dim3(blocks), dim3(WPB * 64), lds, (hipStream_t)stream, xk, N, h, deg, mu_sd, W1,
                       b1, h2, Wmu, bmu, h4, Wp, bp, k, z_save, u_save, feat_save);
    return dgg_check_launch("knet_x_fwd");
}

int dgg_knet_x_bwd_nodes(int64_t N, int h, const float *mu_sd, const float *W1, int h2, const float *Wmu, int h4,
                         const float *Wp, const float *bmu, const float *z, const float *u, const float *dk, float *dkp,
                         float *dm, float *dpre1, float *dxk, float *m_out, void *stream) {
    if (h > 128 || h2 > 64 || h4 > 64) return dgg_set_error(DGG_ERR_UNSUPPORTED, "k-net supports latent_dim <= 128");
    if (N == 0) return 0;
    if (h2 * 2 == h && h4 * 4 == h && (h == 16 || h == 32 || h == 64 || h == 128)) {
        const unsigned nblk = (unsigned)((N + 63) / 64);
#define DGG_KNET_BWD(HH)                                                                                                   \
    hipLaunchKernelGGL(knet_x_bwd_tpn<HH>, dim3(nblk), dim3(64), 0, (hipStream_t)stream, N, mu_sd, W1, Wmu, Wp, z, u, dk, dkp, dm, \
                       dpre1, dxk, m_out, bmu)
        switch (h) {
            case 16: DGG_KNET_BWD(16); break;
            case 32: DGG_KNET_BWD(32); break;
            case 64: DGG_KNET_BWD(64); break;
            default: DGG_KNET_BWD(128); break;
        }
#undef DGG_KNET_BWD
        return dgg_check_launch("knet_x_bwd_nodes");
    }
    size_t lds = sizeof(float) * ((size_t)h2 * (h + 1) + (size_t)h4 * h2);
    unsigned blocks = (unsigned)((N + WPB - 1) / WPB < 2048 ? (N + WPB - 1) / WPB : 2048);
    hipLaunchKernelGGL(knet_x_bwd_kernel, dim3(blocks), dim3(WPB * 64), lds, (hipStream_t)stream, N, h, mu_sd, W1, h2, Wmu,
                       h4, Wp, z, u, dk, dkp, dm, dpre1, dxk, m_out, bmu);
    return dgg_check_launch("knet_x_bwd_nodes");
}

// k-net, mode "x", on the fp32 matrix cores (latent_dim h in {16, 32, 64}): forward -> k [N] and u [N] (the pre-ReLU output, the only
// tensor the backward needs besides xk); same bits as dgg_knet_x_fwd
int dgg_knet_x_fwd_mfma(const float *xk, int64_t N, int h, const float *deg, const float *mu_sd, const float *W1, const float *b1,
                        const float *Wmu, const float *bmu, const float *Wp, const float *bp, float *k, float *u_save, void *stream) {
    if (h != 16 && h != 32 && h != 64) return dgg_set_error(DGG_ERR_UNSUPPORTED, "MFMA k-net: latent_dim in {16, 32, 64}");
    if (N == 0) return 0;
    if (reinterpret_cast<uintptr_t>(xk) % 16) return dgg_set_error(DGG_ERR_ARG, "knet_x_fwd_mfma: xk must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)((N + 255) / 256);          // one 32-node block per wavefront, eight wavefronts per workgroup
    switch (h) {
        case 16: hipLaunchKernelGGL(knet_x_fwd_reg<16>, dim3(grid), dim3(512), 0, st, xk, N, deg, mu_sd, W1, b1, Wmu, bmu, Wp, bp, k, u_save); break;
        case 32: hipLaunchKernelGGL(knet_x_fwd_reg<32>, dim3(grid), dim3(512), 0, st, xk, N, deg, mu_sd, W1, b1, Wmu, bmu, Wp, bp, k, u_save); break;
        default: hipLaunchKernelGGL(knet_x_fwd_reg<64>, dim3(grid), dim3(512), 0, st, xk, N, deg, mu_sd, W1, b1, Wmu, bmu, Wp, bp, k, u_save); break;
    }
    return dgg_check_launch("knet_x_fwd_mfma");
}

// its backward, ONE pass over xk (knet_x_bwd_reg + knet_bwd_reduce): every output is OVERWRITTEN, nothing to zero;
// ws: dgg_knet_x_bwd_ws_bytes(N, h) bytes of scratch
static int knet_bwd_grid(int64_t N) {
    const int64_t nblk = (N + 31) / 32;
    return (int)std::min<int64_t>(512, (nblk + 3) / 4);
}
size_t dgg_knet_x_bwd_ws_bytes(int64_t N, int h) {
    if (h != 16 && h != 32 && h != 64) return 0;
    const size_t slab_f = h == 16 ? KnetBwd<16>::SLAB_F : (h == 32 ? KnetBwd<32>::SLAB_F : KnetBwd<64>::SLAB_F);
    return (size_t)std::max(knet_bwd_grid(N), 1) * slab_f * sizeof(float);
}
int dgg_knet_x_bwd_reg(const float *xk, int64_t N, int h, const float *deg, const float *mu_sd, const float *W1, const float *b1,
                       const float *Wmu, const float *bmu, const float *Wp, const float *u, const float *dk, float *dxk, float *gW1,
                       float *gb1, float *gWmu, float *gbmu, float *gWp, float *gbp, int out_act, void *ws, void *stream) {
    if (h != 16 && h != 32 && h != 64) return dgg_set_error(DGG_ERR_UNSUPPORTED, "MFMA k-net: latent_dim in {16, 32, 64}");
    if (out_act != 0 && out_act != 1) return dgg_set_error(DGG_ERR_ARG, "knet_x_bwd_reg: out_act is 0 or 1 (LeakyReLU)");
    if (!ws || (reinterpret_cast<uintptr_t>(xk) % 16) || (reinterpret_cast<uintptr_t>(dxk) % 16))
        return dgg_set_error(DGG_ERR_ARG, "knet_x_bwd_reg: workspace missing or xk / dxk not 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int grid = N > 0 ? knet_bwd_grid(N) : 0;
    float *slab = reinterpret_cast<float *>(ws);
#define DGG_KNET_BWD_REG(HH)                                                                                               \
    if (grid > 0)                                                                                                          \
        hipLaunchKernelGGL(knet_x_bwd_reg<HH>, dim3((unsigned)grid), dim3(256), 0, st, xk, N, deg, mu_sd, W1, b1, Wmu, Wp, u, dk, dxk, slab, out_act); \
    hipLaunchKernelGGL(knet_bwd_reduce<HH>, dim3((unsigned)KnetBwd<HH>::NREG), dim3(1024), 0, st, slab, grid, Wmu, bmu, Wp, gW1, gb1, gWmu, gbmu, \
                       gWp, gbp)
    switch (h) {
        case 16: DGG_KNET_BWD_REG(16); break;
        case 32: DGG_KNET_BWD_REG(32); break;
        default: DGG_KNET_BWD_REG(64); break;
    }
#undef DGG_KNET_BWD_REG
    return dgg_check_launch("knet_x_bwd_reg");
}

int dgg_knet_feat(const float *xk, const float *deg, const float *mu_sd, int64_t N, int h, float *feat, void *stream) {
    if (N == 0) return 0;
    hipLaunchKernelGGL(knet_feat_kernel, dim3((unsigned)((N * (h + 1) + 255) / 256)), dim3(256), 0, (hipStream_t)stream, xk, deg, mu_sd,
                       N, h, feat);
    return dgg_check_launch("knet_feat");
}
int dgg_knet_out_fwd(const float *kp, const float *mu_sd, int64_t N, float *k, float *u, void *stream) {
    if (N == 0) return 0;
    hipLaunchKernelGGL(knet_out_fwd_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, kp, mu_sd, N, k, u);
    return dgg_check_launch("knet_out_fwd");
}
int dgg_knet_out_bwd(const float *u, const float *dk, const float *mu_sd, int64_t N, float *dkp, void *stream) {
    if (N == 0) return 0;
    hipLaunchKernelGGL(knet_out_bwd_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, u, dk, mu_sd, N, dkp);
    return dgg_check_launch("knet_out_bwd");
}

int dgg_knet_deg_fwd(const float *deg, int64_t N, const float *mu_sd, float dmean, float dstd, float eps, const float *Wd,
                     const float *bd, const float *Wmu, const float *bmu, int h4, const float *Wp, const float *bp, float *k,
                     float *u_save, void *stream) {
    if (N == 0) return 0;
    hipLaunchKernelGGL(knet_deg_fwd_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, deg, N, mu_sd,
                       dmean, dstd, eps, Wd, bd, Wmu, bmu, h4, Wp, bp, k, u_save);
    return dgg_check_launch("knet_deg_fwd");
}

int dgg_knet_input_deg_fwd(const float *deg, int64_t N, float dmean, float dstd, const float *Wd, const float *bd,
                           const float *Wmu, const float *bmu, int h4, const float *Wp, const float *bp, float *k,
                           void *stream) {
    return dgg_knet_deg_fwd(deg, N, nullptr, dmean, dstd, 1e-5f, Wd, bd, Wmu, bmu, h4, Wp, bp, k, nullptr, stream);
}

// S [2] is ACCUMULATED into (caller zeroes it)
int dgg_knet_deg_bwd_sums(const float *deg, int64_t N, const float *mu_sd, float dmean, float dstd, float eps, const float *u,
                          const float *dk, float *S, void *stream) {
    if (N == 0) return 0;
    const unsigned grid = (unsigned)((N + 255) / 256 < 256 ? (N + 255) / 256 : 256);
    hipLaunchKernelGGL(knet_deg_bwd_sums_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, deg, N, mu_sd, dmean, dstd, eps,
                       u, dk, S);
    return dgg_check_launch("knet_deg_bwd_sums");
}

}  // extern "C"
