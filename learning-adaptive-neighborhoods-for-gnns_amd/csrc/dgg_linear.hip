// dgg_linear.hip -- dense layers of the hot path on the fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Replaces the ATen GEMMs behind
//   node_encode_for_edges / node_encode_for_k   nn.Linear + LeakyReLU     reference dgm.py:1097-1100, 1123-1126
//   GCNConv                                     relu(mm(mm(adj,x), W))    reference model.py:594-598
//   GraphConvolution                            mm(support, weight)       reference model.py:41
// and their autograd.  On gfx950 an fp32-input MFMA is an exact k-ordered fmaf chain, so the forward is
// bit-identical to  acc = 0; acc = fmaf(x[c], w[c], acc) (c ascending); acc + bias; activation.
//
// Forward tile: a workgroup (4 wavefronts) owns 128 rows x 128 output columns; each wavefront owns a
// 32-row strip with four 32x32 accumulators; K is walked in blocks of 32 staged through LDS (row stride 33
// floats -> conflict-free ds_read_b32 of the one-float-per-lane MFMA operands).
#include "dgg_common.h"
#include "dgg_api_internal.h"

#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int LIN_BK(int nacc) { return 32; }   // 64 for narrow tiles measured slower (224 VGPRs: two wavefronts per SIMD)

__device__ __forceinline__ float act_apply(float v, int act) {
    if (act == 1) return v > 0.0f ? v : __fmul_rn(0.01f, v);
    if (act == 2) return v > 0.0f ? v : 0.0f;
    return v;
}

// MULTI: several layers that read the SAME input in one pass (node_encode_for_edges + node_encode_for_k + the GCNConv
// projection, dgm.py:1097-1100, 1123-1126, model.py:596): W / b are the row-concatenation of the layers' weights (layout 0),
// and every 32-column block of the concatenated output goes to its own destination with its own activation.
struct LinSegs {
    float *y[8];       // destination of column block cb (already offset to the block's first column)
    int ld[8];         // row stride of that destination
    int act[8];        // its activation
};

// y[N,out] = act(x[N,d] * W^T + b);  w_layout 0: W[out][d], 1: W[d][out].  NACC 32-column accumulators per wave
// (tile width BN = 32*NACC is matched to `out`, so narrow layers do not pay for 128 columns of MFMAs)
template <int NACC, int WAVES, bool MULTI = false>
// narrow tiles: at most 128 registers, i.e. four workgroups per CU -- with three, the 782 workgroups of a 100k-row input no longer
// fit the 768 slots of the chip in one round
__global__ __launch_bounds__(64 * WAVES, NACC <= 2 ? 4 : 2) void linear_fwd_mfma(const float *__restrict__ x, int64_t N, int d,
                                                       const float *__restrict__ W, const float *__restrict__ b,
                                                       int out, int w_layout, int act, float *__restrict__ y, LinSegs segs) {
    constexpr int BN = 32 * NACC, BM = 32 * WAVES, NT = 64 * WAVES;   // small N: fewer rows per workgroup, more workgroups
    // K-block: one block of MFMAs must last longer than an HBM round trip for the register prefetch of the next block to
    // hide it (32 steps x 2 accumulators x 64 cycles = 1.7 us); the wide tile keeps 32 (static LDS limit)
    constexpr int BK = LIN_BK(NACC), LDP = BK + 1, F4R = BK / 4;
    __shared__ float xs[BM * LDP];
    __shared__ float ws[BN * LDP];   // [j][k] (+pad)
    const int tid = threadIdx.x, lane = tid & 63, wave = dgg::wave_id();
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    f32x16 acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; a++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[a][r] = 0.0f;
    const int li = lane & 31, hh = lane >> 5;
    const bool vec4 = (d % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);   // wave-uniform
    // Register prefetch: the global loads of K-block i+1 are issued before the MFMAs of block i and written to LDS after
    // them, so that HBM latency overlaps the matrix-core work of the same workgroup.
    constexpr int XQ = BM * BK / 4 / NT;                        // float4 per thread and K-block (x tile)
    constexpr int WQ = BN * BK / NT;                            // floats per thread and K-block (W tile)
    float4 xr[XQ];
    float wr[WQ];                                                // (validity of a W element is recomputed from its indices at the LDS write)
    unsigned xm[XQ];                                             // per-element validity bits of the x tile
    // Loads are UNCONDITIONAL (clamped addresses) and their values are NOT touched here: the 0/1 masks are applied when the
    // registers are written to LDS, one K-block later.  (Multiplying by the mask right after the load -- or predicating the
    // load -- makes the compiler wait for the data BEFORE the MFMAs of the current block, i.e. no overlap at all.)
    auto load_block = [&](int k0) {
        if (vec4) {   // 16-byte loads: 8 lanes cover one 128-byte row segment
#pragma unroll
            for (int q = 0; q < XQ; q++) {
                const int e = tid + q * NT, r = e / F4R, c4 = (e % F4R) * 4;
                const int64_t gi = m0 + r;
                const bool ok = gi < N && k0 + c4 < d;
                const int64_t gic = gi < N ? gi : N - 1;
                const int kc = k0 + c4 < d ? k0 + c4 : d - 4;
                xr[q] = *reinterpret_cast<const float4 *>(x + gic * d + kc);
                xm[q] = ok ? 15u : 0u;
            }
        } else {
#pragma unroll
            for (int q = 0; q < XQ; q++) {
                const int e = tid + q * NT, r = e / F4R, c4 = (e % F4R) * 4;
                const int64_t gi = m0 + r;
                const int64_t gic = gi < N ? gi : N - 1;
                float t[4];
                unsigned bits = 0;
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int kk = k0 + c4 + u;
                    t[u] = x[gic * d + (kk < d ? kk : d - 1)];
                    bits |= (gi < N && kk < d) ? (1u << u) : 0u;
                }
                xr[q] = make_float4(t[0], t[1], t[2], t[3]);
                xm[q] = bits;
            }
        }
#pragma unroll
        for (int q = 0; q < WQ; q++) {
            const int e = tid + q * NT;
            const int j = w_layout == 0 ? e / BK : e % BN, c = w_layout == 0 ? e % BK : e / BN;
            const int gj = n0 + j, gk = k0 + c;
            const int gjc = gj < out ? gj : out - 1, gkc = gk < d ? gk : d - 1;
            wr[q] = W[w_layout == 0 ? (int64_t)gjc * d + gkc : (int64_t)gkc * out + gjc];
        }
    };
    // bias of this lane's columns, fetched before the K loop so that the epilogue has no load to wait for (a wait on a load
    // there would also wait for every earlier STORE: vmcnt counts both)
    float bj[NACC];
#pragma unroll
    for (int a = 0; a < NACC; a++) {
        const int gj = n0 + a * 32 + li;
        bj[a] = b ? b[gj < out ? gj : out - 1] : 0.0f;
    }
    load_block(0);
    for (int k0 = 0; k0 < d; k0 += BK) {
        __syncthreads();                                         // previous block's LDS reads are done
#pragma unroll
        for (int q = 0; q < XQ; q++) {
            const int e = tid + q * NT, r = e / F4R, c4 = (e % F4R) * 4;
            float *dst = xs + r * LDP + c4;
            dst[0] = (xm[q] & 1u) ? xr[q].x : 0.0f; dst[1] = (xm[q] & 2u) ? xr[q].y : 0.0f;
            dst[2] = (xm[q] & 4u) ? xr[q].z : 0.0f; dst[3] = (xm[q] & 8u) ? xr[q].w : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < WQ; q++) {
            const int e = tid + q * NT;
            const int j = w_layout == 0 ? e / BK : e % BN, c = w_layout == 0 ? e % BK : e / BN;
            ws[j * LDP + c] = (n0 + j < out && k0 + c < d) ? wr[q] : 0.0f;
        }
        __syncthreads();
        if (k0 + BK < d) load_block(k0 + BK);                    // in flight during the MFMAs below
        const float *xa = xs + (wave * 32 + li) * LDP + hh;
#pragma unroll
        for (int s = 0; s < BK / 2; s++) {
            float av = xa[2 * s];
#pragma unroll
            for (int a = 0; a < NACC; a++) {
                float bv = ws[(a * 32 + li) * LDP + 2 * s + hh];
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[a], 0, 0, 0);
            }
        }
    }
    const bool hasb = b != nullptr;
    auto finish_act = [&](float v, float bias, int actv) {
        const float vb = __fadd_rn(v, bias);
        v = hasb ? vb : v;
        const float lk = v > 0.0f ? v : __fmul_rn(0.01f, v), rl = v > 0.0f ? v : 0.0f;
        return actv == 1 ? lk : (actv == 2 ? rl : v);
    };
    auto finish = [&](float v, float bias) { return finish_act(v, bias, act); };
    if constexpr (MULTI) {
        // out is a multiple of 32 * NACC here (host-checked), so only the row edge needs predicates
        const bool full = m0 + BM <= N;
#pragma unroll
        for (int a = 0; a < NACC; a++) {
            const int cb = n0 / 32 + a;
            float *yb = segs.y[cb] + (m0 + wave * 32 + 4 * hh) * (int64_t)segs.ld[cb] + li;
            const int av = segs.act[cb];
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = (r & 3) + 8 * (r >> 2);
                if (full || m0 + wave * 32 + 4 * hh + row < N) yb[(int64_t)row * segs.ld[cb]] = finish_act(acc[a][r], bj[a], av);
            }
        }
        return;
    }
    // Interior tiles (all but the last row / column block) store without per-element predicates: a predicated store is a
    // branch, and after every branch the compiler re-waits for ALL outstanding memory operations -- earlier stores included --
    // which serialises the 16 * NACC stores of a wavefront.
    if (m0 + BM <= N && n0 + BN <= out) {
        float *yb = y + (m0 + wave * 32 + 4 * hh) * out + n0 + li;
#pragma unroll
        for (int a = 0; a < NACC; a++)
#pragma unroll
            for (int r = 0; r < 16; r++) yb[(int64_t)((r & 3) + 8 * (r >> 2)) * out + a * 32] = finish(acc[a][r], bj[a]);
        return;
    }
#pragma unroll
    for (int a = 0; a < NACC; a++) {
        const int gj = n0 + a * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
            const int64_t gi = m0 + wave * 32 + row;
            const float v = finish(acc[a][r], bj[a]);
            if (gi < N && gj < out) y[gi * out + gj] = v;
        }
    }
}

// ---- large-N forward: weights resident in LDS, rows straight from HBM into the MFMA operand registers ---------------------
// linear_fwd_mfma restages the weight tile for every 128-row tile and walks K in synchronised blocks; at N >= 10^5 the
// matrix cores are busy ~30 % of the time.  Here a workgroup (4 wavefronts, one per SIMD, the full register file each) is
// PERSISTENT: the whole weight matrix (all column blocks) is written to LDS once, in MFMA operand order, and every wavefront
// then walks 32-row blocks on its own -- no barrier after the prologue.
//  * Row operand of step s = (lane row, k = 2s + lane/32): lane (row, half) loads the CONTIGUOUS half [half*D/2, (half+1)*D/2)
//    of its row with 16-byte loads, and one v_permlane32_swap per register pair turns (reg 2t, reg 2t+1) into the operands of
//    steps t and D/4 + t -- X never touches LDS.  The next block's rows are in flight during the current block's MFMAs.
//  * Weight operands of step s + PD are read from LDS before the MFMAs of step s issue (left alone the compiler reads them right
//    before their use and waits out the LDS latency between every two MFMAs).
//  * The product is formed TRANSPOSED (weights first, rows second): a lane then holds ONE node and, per group of four accumulator
//    registers, four consecutive output columns -- the block leaves as four 16-byte stores per column block straight from the
//    accumulator, with the bias as one more contraction step (see store_block).  Round 3 computed it the other way round (one
//    column per lane) and turned every block through an LDS tile: 35 of 90 us in the epilogue.
//  * Work split: `nmain` row blocks (a multiple of the wavefront count) are dealt whole; the remaining < #wavefronts blocks are
//    dealt one (row block, column block) unit at a time -- a whole extra block on a few SIMDs would cost a full round (3125
//    blocks of a 100k-row input on 1024 SIMDs: 3 rounds + 53 blocks).
// Same k-ordered fmaf chain per output as linear_fwd_mfma: bit-identical results.
template <int D, int NACC>
__global__ __launch_bounds__(256, 1) void linear_fwd_reg(const float *__restrict__ x, int64_t N, const float *__restrict__ W,
                                                         const float *__restrict__ b, LinSegs segs, int64_t nrb) {
    constexpr int S = D / 2, XR = D / 2;                         // MFMA steps per output; operand registers per lane
    __shared__ float wsf[S * NACC * 64];                         // [step][column block][lane]
    const int tid = threadIdx.x, lane = tid & 63, wave = dgg::wave_id(), li = lane & 31, hh = lane >> 5;
    const int64_t nw = (int64_t)gridDim.x * 4, slot = (int64_t)blockIdx.x * 4 + wave;
    const int64_t nrbf = N / 32;                                 // FULL row blocks (a partial last block is a remainder unit)
    const int64_t nmain = nrbf - nrbf % nw, nfull = nmain / nw;
    const int64_t R = nrb - nmain, nunits = nfull + (slot < R * NACC ? (R * NACC - slot + nw - 1) / nw : 0);
    // unit t of this wavefront: a whole row block while t < nfull, then one (row block, column block) of the remainder
    auto unit_rb = [&](int64_t t) {
        if (t < nfull) return slot + t * nw;
        const int64_t u = slot + (t - nfull) * nw;
        return nmain + u % (R > 0 ? R : 1);
    };
    float xc[XR], xn[XR];
    auto load_rows = [&](int64_t rb, float(&dst)[XR]) {
        int64_t row = rb * 32 + li;
        row = row < N ? row : N - 1;
        const float4 *p = reinterpret_cast<const float4 *>(x + row * D + hh * (D / 2));
#pragma unroll
        for (int q = 0; q < XR / 4; q++) {
            const float4 v = p[q];
            dst[4 * q] = v.x; dst[4 * q + 1] = v.y; dst[4 * q + 2] = v.z; dst[4 * q + 3] = v.w;
        }
    };
    if (nunits > 0) load_rows(unit_rb(0), xc);                   // in flight while the weights are staged
#pragma unroll                                                    // (all loads in flight: a rolled loop pays one L2 round trip per pass)
    for (int it = 0; it < NACC * 32 * (D / 4) / 256; it++) {     // lanes along the output index: conflict-free LDS writes
        const int e = tid + it * 256;
        const int j = e % (NACC * 32), q = e / (NACC * 32);
        const float4 v = *reinterpret_cast<const float4 *>(W + (int64_t)j * D + 4 * q);
        float *dst = wsf + ((2 * q) * NACC + (j >> 5)) * 64 + (j & 31);
        dst[0] = v.x; dst[32] = v.y; dst[NACC * 64] = v.z; dst[NACC * 64 + 32] = v.w;
    }
    const bool hasb = b != nullptr;
    float bj[NACC];
    int actv[NACC];
#pragma unroll
    for (int a = 0; a < NACC; a++) {
        bj[a] = hasb ? b[a * 32 + li] : 0.0f;
        actv[a] = segs.act[a];
    }
    __syncthreads();
    auto swap_rows = [&]() {
#pragma unroll
        for (int t = 0; t < XR / 2; t++) {
            const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(xc[2 * t]), __float_as_uint(xc[2 * t + 1]), false, false);
            xc[2 * t] = __uint_as_float(r[0]);                   // operand of step t        (k = 2t + half)
            xc[2 * t + 1] = __uint_as_float(r[1]);               // operand of step D/4 + t  (k = D/2 + 2t + half)
        }
    };
    // An accumulator is computed TRANSPOSED -- first operand = the weights (lane li = output column), second = the rows (lane li = node)
    // -- so that lane (li, hh) ends up with node rb*32 + li and, in registers 4q .. 4q+3, the four CONSECUTIVE columns 8q + 4hh + (0..3)
    // of the block: the output leaves as four 16-byte stores per column block straight from the accumulator (consecutive stores
    // complete each row's 128-byte line), no LDS turn, no waits.  The bias rides as ONE more contraction step against a column of
    // ones: fma(1, b, acc) is the correctly rounded acc + b, the same bits as the separate add.  (Round 3 turned each 32 x 32 block
    // through an LDS tile behind 11 vector instructions per element -- bias, three activation selects: 35 of the kernel's 90 us.)
    auto act4 = [&](float4 v, int av) {
        if (av == 1) {                                           // LeakyReLU(0.01): max(v, 0.01 v) == v > 0 ? v : 0.01 v
            v.x = fmaxf(v.x, __fmul_rn(0.01f, v.x)); v.y = fmaxf(v.y, __fmul_rn(0.01f, v.y));
            v.z = fmaxf(v.z, __fmul_rn(0.01f, v.z)); v.w = fmaxf(v.w, __fmul_rn(0.01f, v.w));
        } else if (av == 2) {
            v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f);
        }
        return v;
    };
    auto store_block = [&](const f32x16 &acc, int a, int64_t rb) {
        const int64_t n = rb * 32 + li;
        if (n >= N) return;
        const int av = actv[a];                                  // (wave-uniform)
        float *yb = segs.y[a] + n * (int64_t)segs.ld[a] + 4 * hh;
#pragma unroll
        for (int q = 0; q < 4; q++)
            *reinterpret_cast<float4 *>(yb + 8 * q) = act4(make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]), av);
    };
    const float one_or_zero = hh == 0 ? 1.0f : 0.0f;
    // one whole row block
    auto main_block = [&](int64_t rb) {
        f32x16 acc[NACC];
#pragma unroll
        for (int a = 0; a < NACC; a++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][r] = 0.0f;
        constexpr int PD = NACC >= 4 ? 1 : (NACC >= 2 ? 2 : 4), RING = PD + 1;
        float bw[RING][NACC];
#pragma unroll
        for (int s = 0; s < PD; s++)
#pragma unroll
            for (int a = 0; a < NACC; a++) bw[s][a] = wsf[(s * NACC + a) * 64 + lane];
#pragma unroll
        for (int s = 0; s < S; s++) {
            if (s + PD < S) {
#pragma unroll
                for (int a = 0; a < NACC; a++) bw[(s + PD) % RING][a] = wsf[((s + PD) * NACC + a) * 64 + lane];
            }
            __builtin_amdgcn_sched_barrier(0);
            const float av = s < S / 2 ? xc[2 * s] : xc[2 * (s - S / 2) + 1];
#pragma unroll
            for (int a = 0; a < NACC; a++) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(bw[s % RING][a], av, acc[a], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (hasb) {
#pragma unroll
            for (int a = 0; a < NACC; a++) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(bj[a] * one_or_zero, one_or_zero, acc[a], 0, 0, 0);
        }
#pragma unroll
        for (int a = 0; a < NACC; a++) store_block(acc[a], a, rb);
    };
    int64_t t = 0;
    for (; t < nfull; t++) {
        load_rows(unit_rb(t + 1 < nunits ? t + 1 : t), xn);      // unconditional; in flight during the MFMAs below
        asm volatile("" ::: "memory");                           // the weights are re-read from LDS per block (hoisted out of this loop
                                                                 //  they would need S*NACC registers and spill)
        swap_rows();
        main_block(unit_rb(t));
#pragma unroll
        for (int q = 0; q < XR; q++) xc[q] = xn[q];
    }
    for (; t < nunits; t++) {                                    // remainder: one (row block, column block) per unit
        const int64_t rb = unit_rb(t);
        load_rows(unit_rb(t + 1 < nunits ? t + 1 : t), xn);
        asm volatile("" ::: "memory");
        swap_rows();
        const int a = (int)((slot + (t - nfull) * nw) / R);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.0f;
        const float *wa = wsf + a * 64 + lane;
#pragma unroll                                                    // (full: xc must stay in registers)
        for (int s = 0; s < S; s++) {
            const float av = s < S / 2 ? xc[2 * s] : xc[2 * (s - S / 2) + 1];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[s * NACC * 64], av, acc, 0, 0, 0);
        }
        if (hasb) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b[a * 32 + li] * one_or_zero, one_or_zero, acc, 0, 0, 0);
        store_block(acc, a, rb);
#pragma unroll
        for (int q = 0; q < XR; q++) xc[q] = xn[q];
    }
}

// dp = dy * act'(y)
__global__ void act_bwd_kernel(const float *__restrict__ y, const float *__restrict__ dy, int64_t n, int act,
                               float *__restrict__ dp) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float g = dy[i], v = y[i];
        if (act == 1) g = v > 0.0f ? g : 0.01f * g;
        else if (act == 2) g = v > 0.0f ? g : 0.0f;
        dp[i] = g;
    }
}

// Weight-gradient GEMM: C[M1,M2] += A[N,M1]^T B[N,M2] with tiny M1, M2 (<= a few hundred) and huge N.
// Stage 1 (gemm_tn_persist): a PERSISTENT grid of GT_GRID workgroups streams the node rows; every wavefront keeps a
// (32*MB) x (32*NB) block of C in registers for the whole kernel (up to 64 x 128 = the full weight of the hot path, so A
// and B are read exactly once).  The fp32 MFMA operands are one float per lane (A[n][o0+..+lane%32],
// B[n][c0+..+lane%32], n = k-step*2 + lane/32), i.e. 128-byte coalesced row segments, loaded straight from global
// memory -- no LDS, no barriers in the loop -- with register double buffering: the PF k-steps of batch i+1 are in
// flight during the MFMAs of batch i.  The four wavefronts of a workgroup are summed through LDS at the end and the
// workgroup writes ONE partial block to slab[workgroup] with plain stores -- NOT atomics: every workgroup would hit the
// same few KB of C, and same-address fp32 atomics run ~14x below the streaming atomic rate (MI355X_MICROARCH.md,
// "Global float atomics").
// Stage 2: gemm_tn_reduce sums the slab over workgroups (32-way split, 32 atomics per output) into C / colsum.
constexpr int GT_GRID = 256, GT_PF = 8;
template <int MB, int NB, int PFK, bool ACT>
__global__ __launch_bounds__(256) void gemm_tn_persist(const float *__restrict__ A, const float *__restrict__ B, int64_t N,
                                                       int M1, int M2, int M1p, int M2p, float *__restrict__ slab,
                                                       float *__restrict__ cs_slab, const float *__restrict__ Yact, int act) {
    extern __shared__ float red[];                               // [MB*NB*16*64] + [MB*32]
    const int lane = threadIdx.x & 63, wave = dgg::wave_id(), li = lane & 31, hh = lane >> 5;
    const int o0 = blockIdx.y * 32 * MB, c0 = blockIdx.z * 32 * NB;
    // Loads are UNCONDITIONAL (clamped addresses) so that the compiler can keep two batches in flight (a predicated load
    // forces s_waitcnt vmcnt(0)): columns beyond M1 / M2 land in padded slab rows / columns that the reduce ignores;
    // node rows beyond N are clamped to N-1 and their A operand is multiplied by 0.
    int64_t acol[MB], bcol[NB];
#pragma unroll
    for (int m = 0; m < MB; m++) acol[m] = o0 + m * 32 + li < M1 ? o0 + m * 32 + li : M1 - 1;
#pragma unroll
    for (int a = 0; a < NB; a++) bcol[a] = c0 + a * 32 + li < M2 ? c0 + a * 32 + li : M2 - 1;
    f32x16 acc[MB][NB];
#pragma unroll
    for (int m = 0; m < MB; m++)
#pragma unroll
        for (int a = 0; a < NB; a++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[m][a][r] = 0.0f;
    float csum[MB];
#pragma unroll
    for (int m = 0; m < MB; m++) csum[m] = 0.0f;
    constexpr int PF = PFK;                                      // k-steps per batch (fewer with the activation mask: registers)
    const int64_t step = (int64_t)gridDim.x * 4 * 2 * PF;         // node rows per sweep of the whole grid
    // Yact != NULL: A is a cotangent that still has to pass through the activation of the forward (dp = dy * act'(y)); the mask
    // is applied to the operand on the fly, which saves the separate act_bwd pass and its [N, M1] buffer
    float av[2][PF][MB], bv[2][PF][NB], rmask[2][PF], yv[2][PF][MB];
    auto load = [&](int buf, int64_t base) {
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const int64_t n = base + 2 * u + hh;
            const int64_t nc = n < N ? n : N - 1;
            rmask[buf][u] = n < N ? 1.0f : 0.0f;
#pragma unroll
            for (int m = 0; m < MB; m++) {
                av[buf][u][m] = A[nc * M1 + acol[m]];
                if (ACT) yv[buf][u][m] = Yact[nc * M1 + acol[m]];
            }
#pragma unroll
            for (int a = 0; a < NB; a++) bv[buf][u][a] = B[nc * M2 + bcol[a]];
        }
    };
    auto mma = [&](int buf) {
#pragma unroll
        for (int u = 0; u < PF; u++) {
#pragma unroll
            for (int m = 0; m < MB; m++) {
                float am = rmask[buf][u];
                if (ACT && act == 1) am = yv[buf][u][m] > 0.0f ? am : 0.01f * am;    // LeakyReLU'
                else if (ACT && act == 2) am = yv[buf][u][m] > 0.0f ? am : 0.0f;     // ReLU'
                av[buf][u][m] *= am;
                csum[m] += av[buf][u][m];
#pragma unroll
                for (int a = 0; a < NB; a++)
                    acc[m][a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[buf][u][m], bv[buf][u][a], acc[m][a], 0, 0, 0);
            }
        }
    };
    int64_t base = ((int64_t)blockIdx.x * 4 + wave) * 2 * PF;
    if (base < N) load(0, base);
    while (base < N) {                                           // two batches per trip: buffers alternate statically
        if (base + step < N) load(1, base + step);
        mma(0);
        base += step;
        if (base >= N) break;
        if (base + step < N) load(0, base + step);
        mma(1);
        base += step;
    }
    // sum the four wavefronts through LDS (wave 0 accumulates), then one plain store per element
    float *cred = red + MB * NB * 16 * 64;
    for (int w = 1; w < 4; w++) {
        if (wave == w) {
#pragma unroll
            for (int m = 0; m < MB; m++) {
#pragma unroll
                for (int a = 0; a < NB; a++)
#pragma unroll
                    for (int r = 0; r < 16; r++) red[((m * NB + a) * 16 + r) * 64 + lane] = acc[m][a][r];
                cred[m * 64 + lane] = csum[m];
            }
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int m = 0; m < MB; m++) {
#pragma unroll
                for (int a = 0; a < NB; a++)
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[m][a][r] += red[((m * NB + a) * 16 + r) * 64 + lane];
                csum[m] += cred[m * 64 + lane];
            }
        }
        __syncthreads();
    }
    if (wave != 0) return;
    float *sl = slab + (int64_t)blockIdx.x * M1p * M2p;
#pragma unroll
    for (int m = 0; m < MB; m++) {
#pragma unroll
        for (int a = 0; a < NB; a++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int go = o0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                const int gc = c0 + a * 32 + li;
                if (go < M1p && gc < M2p) sl[(int64_t)go * M2p + gc] = acc[m][a][r];
            }
        }
        if (cs_slab && blockIdx.z == 0) {
            float t = csum[m] + __uint_as_float(dgg::xor_shfl<32>(__float_as_uint(csum[m]), lane));
            if (hh == 0 && o0 + m * 32 + li < M1p) cs_slab[(int64_t)blockIdx.x * M1p + o0 + m * 32 + li] = t;
        }
    }
}

// Several weight gradients that share the streamed operand B (= the layer input X) in ONE pass:
//   C_s[M1_s, M2] += (A_s * act_s'(Y_s))^T B   for s = 0..nseg-1   (node_encode_for_edges, node_encode_for_k, GCNConv weight:
// autograd of dgm.py:1097-1100, 1123-1126 and model.py:596 -- three GEMMs over the same 51 MB of node features).
// blockIdx.x enumerates (row stream g, 32-column block yb of the concatenated A operands); the mapping keeps the blocks of one
// row stream on ONE XCD (block ids that differ by multiples of 8 share an XCD's L2) and adjacent in dispatch order, so the rows
// of B leave the fabric once and the other yb blocks hit them in L2.  Structure otherwise as gemm_tn_persist<1, NB>.
struct TnSegs {
    const float *A[8];     // per column block: operand, the forward output that carries its activation mask (or NULL),
    const float *Y[8];
    int ld[8];             // row stride of the operand (its M1),
    int o0[8];             // first column of this block inside the operand,
    int act[8];            // activation of the forward (0 none, 1 LeakyReLU, 2 ReLU)
    const float *Bp[8];    // per column block: its own streamed operand (NULL: the kernel's common B) and that operand's width
    int m2[8];
    int nyb;
};
template <int NB, int PF>
__global__ __launch_bounds__(256) void gemm_tn_multi(TnSegs segs, const float *B, int64_t N, int M2, int M2p, int G,
                                                     float *__restrict__ slab, float *__restrict__ cs_slab) {
    extern __shared__ float red[];                               // [NB*16*64] + [64]
    const int lane = threadIdx.x & 63, wave = dgg::wave_id(), li = lane & 31, hh = lane >> 5;
    const int nyb = segs.nyb;
    const int bid = blockIdx.x, tt = bid / (8 * nyb), rem = bid % (8 * nyb);
    const int yb = rem / 8, g = tt * 8 + (rem % 8);
    if (g >= G) return;
    const float *__restrict__ A = segs.A[yb];
    const float *__restrict__ Yact = segs.Y[yb];
    const int M1 = segs.ld[yb], act = Yact ? segs.act[yb] : 0;
    if (segs.Bp[yb]) { B = segs.Bp[yb]; M2 = segs.m2[yb]; }     // a pair with its own streamed operand (dgg_gemm_tn_pairs)
    const int nbu = (M2 + 31) / 32;                              // column blocks of B that exist (block-uniform): the others are skipped
    const int64_t acol = segs.o0[yb] + li < M1 ? segs.o0[yb] + li : M1 - 1;
    int64_t bcol[NB];
#pragma unroll
    for (int a = 0; a < NB; a++) bcol[a] = a * 32 + li < M2 ? a * 32 + li : M2 - 1;
    f32x16 acc[NB];
#pragma unroll
    for (int a = 0; a < NB; a++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[a][r] = 0.0f;
    float csum = 0.0f;
    const int64_t step = (int64_t)G * 4 * 2 * PF;
    float av[2][PF], bv[2][PF][NB], rmask[2][PF], yv[2][PF];
    auto load = [&](int buf, int64_t base) {
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const int64_t n = base + 2 * u + hh;
            const int64_t nc = n < N ? n : N - 1;
            rmask[buf][u] = n < N ? 1.0f : 0.0f;
            av[buf][u] = A[nc * M1 + acol];
            yv[buf][u] = 1.0f;
#pragma unroll
            for (int a = 0; a < NB; a++)
                if (a < nbu) bv[buf][u][a] = B[nc * M2 + bcol[a]];
        }
        if (act != 0) {                                          // block-uniform
#pragma unroll
            for (int u = 0; u < PF; u++) {
                const int64_t n = base + 2 * u + hh;
                yv[buf][u] = Yact[(n < N ? n : N - 1) * M1 + acol];
            }
        }
    };
    auto mma = [&](int buf) {
#pragma unroll
        for (int u = 0; u < PF; u++) {
            float am = rmask[buf][u];
            if (act == 1) am = yv[buf][u] > 0.0f ? am : 0.01f * am;
            else if (act == 2) am = yv[buf][u] > 0.0f ? am : 0.0f;
            const float a_ = av[buf][u] * am;
            csum += a_;
#pragma unroll
            for (int a = 0; a < NB; a++)
                if (a < nbu) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_, bv[buf][u][a], acc[a], 0, 0, 0);
        }
    };
    int64_t base = ((int64_t)g * 4 + wave) * 2 * PF;
    if (base < N) load(0, base);
    while (base < N) {
        if (base + step < N) load(1, base + step);
        mma(0);
        base += step;
        if (base >= N) break;
        if (base + step < N) load(0, base + step);
        mma(1);
        base += step;
    }
    float *cred = red + NB * 16 * 64;
    for (int w = 1; w < 4; w++) {
        if (wave == w) {
#pragma unroll
            for (int a = 0; a < NB; a++)
#pragma unroll
                for (int r = 0; r < 16; r++) red[(a * 16 + r) * 64 + lane] = acc[a][r];
            cred[lane] = csum;
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int a = 0; a < NB; a++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[a][r] += red[(a * 16 + r) * 64 + lane];
            csum += cred[lane];
        }
        __syncthreads();
    }
    if (wave != 0) return;
    const int M1tp = nyb * 32;
    float *sl = slab + ((int64_t)g * M1tp + yb * 32) * M2p;
#pragma unroll
    for (int a = 0; a < NB; a++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int go = (r & 3) + 8 * (r >> 2) + 4 * hh, gc = a * 32 + li;
            if (gc < M2p) sl[(int64_t)go * M2p + gc] = acc[a][r];
        }
    const float t = csum + __uint_as_float(dgg::xor_shfl<32>(__float_as_uint(csum), lane));
    if (hh == 0) cs_slab[(int64_t)g * M1tp + yb * 32 + li] = t;
}

// gemm_tn_multi for the shape of the hot path (B = the layer input, exactly 128 wide; every A operand a multiple of 64 wide):
// a wavefront owns 64 columns of A x all 128 columns of B (eight accumulators) and reads its operands with ONE 8-byte load
// (A, and the activation mask's Y) and ONE 16-byte load (B) per two rows -- 2-3 load instructions per 8 MFMAs where
// gemm_tn_multi issues 5-6 per 4, and B is re-read per 64 instead of per 32 columns of A.  That needs lane l to hold columns
// (2l, 2l+1) of A and (4l .. 4l+3) of B, i.e. accumulator (m, a) holds output rows 2i+m and columns 4j+a: a permutation that is
// undone when the partial result is written (16-byte stores).  Loads run PF row pairs ahead in a ring that is refilled slot
// by slot (every slot's next load is issued right after its MFMAs).  Partial results and reduce as gemm_tn_multi.
template <int PF, bool ACT>
__device__ __forceinline__ void tn_wide_stream(const float *__restrict__ A, const float *__restrict__ Yact, int act, const float *__restrict__ B,
                                               int64_t N, int M1, int acol, int ldb, int bcol, int hh, int64_t base, int64_t stride,
                                               f32x16 (&acc)[2][4], float (&csum)[2]) {
    float2 av[PF], yv[PF];
    float4 bv[PF];
    auto load = [&](int u, int64_t b0) {
        const int64_t n = b0 + 2 * u + hh, nc = n < N ? n : N - 1;          // unconditional, clamped
#ifdef DGG_TNW_NOLOAD
        av[u] = make_float2((float)nc, 1.0f); bv[u] = make_float4(1.0f, (float)nc, 2.0f, 3.0f);
        if (ACT) yv[u] = av[u];
#else
        av[u] = *reinterpret_cast<const float2 *>(A + nc * M1 + acol);
        if (ACT) yv[u] = *reinterpret_cast<const float2 *>(Yact + nc * M1 + acol);
        bv[u] = *reinterpret_cast<const float4 *>(B + nc * ldb + bcol);
#endif
    };
#pragma unroll
    for (int u = 0; u < PF; u++) load(u, base);
    for (; base < N; base += stride) {
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const float rm = base + 2 * u + hh < N ? 1.0f : 0.0f;
            float m0 = rm, m1 = rm;
            if (ACT) {
                if (act == 1) { m0 = yv[u].x > 0.0f ? rm : 0.01f * rm; m1 = yv[u].y > 0.0f ? rm : 0.01f * rm; }
                else { m0 = yv[u].x > 0.0f ? rm : 0.0f; m1 = yv[u].y > 0.0f ? rm : 0.0f; }
            }
            const float a0 = av[u].x * m0, a1 = av[u].y * m1;
            const float4 b = bv[u];
            load(u, base + stride);                              // this slot's next row pair: in flight for PF steps
            csum[0] += a0; csum[1] += a1;
#ifdef DGG_TNW_NOMFMA
            acc[0][0][0] = fmaf(a0, b.x, acc[0][0][0]); acc[0][1][0] = fmaf(a0, b.y, acc[0][1][0]);
            acc[0][2][0] = fmaf(a0, b.z, acc[0][2][0]); acc[0][3][0] = fmaf(a0, b.w, acc[0][3][0]);
            acc[1][0][0] = fmaf(a1, b.x, acc[1][0][0]); acc[1][1][0] = fmaf(a1, b.y, acc[1][1][0]);
            acc[1][2][0] = fmaf(a1, b.z, acc[1][2][0]); acc[1][3][0] = fmaf(a1, b.w, acc[1][3][0]);
#else
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b.x, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b.y, acc[0][1], 0, 0, 0);
            acc[0][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b.z, acc[0][2], 0, 0, 0);
            acc[0][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b.w, acc[0][3], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b.x, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b.y, acc[1][1], 0, 0, 0);
            acc[1][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b.z, acc[1][2], 0, 0, 0);
            acc[1][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b.w, acc[1][3], 0, 0, 0);
#endif
        }
    }
}

// The same stream on the bf16 matrix cores with SPLIT operands: x = hi + lo + O(2^-17 x), hi = bf16(x), lo = bf16(x - hi), and
// a b ~ hi_a hi_b + hi_a lo_b + lo_a hi_b (the dropped lo_a lo_b term is 2^-16 of the product) -- three v_mfma_f32_32x32x16_bf16 per
// 16 rows where the exact form needs eight v_mfma_f32_32x32x2_f32: 3/16 of the matrix-pipe time for a relative error of ~2e-5 per
// term (fp32 accumulation), far inside the 2e-4 the weight gradients are held to.  The products feed GRADIENTS only: nothing on the
// score path (whose bits decide the neighbour lists) goes through here.  A lane holds rows base + 8 hh + r (r = 0..7) of its two A
// and four B columns: eight 8-byte and eight 16-byte loads per 16 rows, the next block's in flight during the MFMAs.
typedef __bf16 tn_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void tn_split(const float (&x)[8], tn_bf16x8 &hi, tn_bf16x8 &lo) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
        hi[r] = (__bf16)x[r];
        lo[r] = (__bf16)(x[r] - (float)hi[r]);
    }
}
template <bool ACT>
__device__ __forceinline__ void tn_wide_stream_b3(const float *__restrict__ A, const float *__restrict__ Yact, int act, const float *__restrict__ B,
                                                  int64_t N, int M1, int acol, int ldb, int bcol, int hh, int64_t base, int64_t stride,
                                                  f32x16 (&acc)[2][4], float (&csum)[2]) {
    float2 av[8], yv[8];
    float4 bv[8];
    auto load = [&](int64_t b0) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int64_t n = b0 + 8 * hh + r, nc = n < N ? n : N - 1;      // unconditional, clamped
            av[r] = *reinterpret_cast<const float2 *>(A + nc * M1 + acol);
            if (ACT) yv[r] = *reinterpret_cast<const float2 *>(Yact + nc * M1 + acol);
            bv[r] = *reinterpret_cast<const float4 *>(B + nc * ldb + bcol);
        }
    };
    load(base);
    for (; base < N; base += stride) {
        float a0[8], a1[8], b0[8], b1[8], b2[8], b3[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const float rm = base + 8 * hh + r < N ? 1.0f : 0.0f;
            float m0 = rm, m1 = rm;
            if (ACT) {
                if (act == 1) { m0 = yv[r].x > 0.0f ? rm : 0.01f * rm; m1 = yv[r].y > 0.0f ? rm : 0.01f * rm; }
                else { m0 = yv[r].x > 0.0f ? rm : 0.0f; m1 = yv[r].y > 0.0f ? rm : 0.0f; }
            }
            a0[r] = av[r].x * m0; a1[r] = av[r].y * m1;
            b0[r] = bv[r].x; b1[r] = bv[r].y; b2[r] = bv[r].z; b3[r] = bv[r].w;
            csum[0] += a0[r]; csum[1] += a1[r];
        }
#ifdef DGG_TN_B3_EARLY
        load(base + stride);
#endif
        tn_bf16x8 ah[2], al[2], bh[4], bl[4];
        tn_split(a0, ah[0], al[0]); tn_split(a1, ah[1], al[1]);
        tn_split(b0, bh[0], bl[0]); tn_split(b1, bh[1], bl[1]); tn_split(b2, bh[2], bl[2]); tn_split(b3, bh[3], bl[3]);
#ifndef DGG_TN_B3_EARLY
        // the next block's loads go out once this block lives in its packed form (48 registers instead of 96: issued before the
        // conversions the kernel spills), in flight during the 24 MFMAs of this block and the other wavefront's whole block
        load(base + stride);
#endif
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int a = 0; a < 4; a++) {
                acc[m][a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[m], bh[a], acc[m][a], 0, 0, 0);       // (small terms first)
                acc[m][a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bl[a], acc[m][a], 0, 0, 0);
                acc[m][a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bh[a], acc[m][a], 0, 0, 0);
            }
    }
}

// B wider than 128 columns (M2 a multiple of 4; Pubmed's 500 input features): the workgroups of a row stream additionally enumerate
// the 128-column tiles of B (`ntile`); a lane whose four columns lie beyond M2 reads the last valid group instead and its output
// columns are dropped by the reduce (they sit in the padding of the M2p-wide slab, or beyond it and are not stored).
// (HASACT = false: no segment carries an activation mask -- the headline step's cotangents arrive premasked -- and the kernel is
//  compiled without that path: with it the split-bf16 stream needs 16 more registers than a wavefront has and spills)
template <int PF, bool B3 = false, bool HASACT = true>
__global__ __launch_bounds__(256, 2) void gemm_tn_wide(TnSegs segs, const float *__restrict__ B, int64_t N, int G, int M2, int M2p, int ntile,
                                                       float *__restrict__ slab, float *__restrict__ cs_slab) {
    extern __shared__ float red[];                               // [8*16*64] + [2*64]
    const int lane = threadIdx.x & 63, wave = dgg::wave_id(), li = lane & 31, hh = lane >> 5;
    const int npair = segs.nyb / 2, nblk = npair * ntile;
    const int bid = blockIdx.x, tt = bid / (8 * nblk), rem = bid % (8 * nblk);
    const int yb = 2 * ((rem / 8) % npair), tile = (rem / 8) / npair;
    const int g = tt * 8 + (rem % 8);                             // the blocks of one row stream stay on one XCD (see gemm_tn_multi)
    if (g >= G) return;
    const int col0 = tile * 128 + 4 * li, bcol = col0 + 4 <= M2 ? col0 : M2 - 4;
    const float *__restrict__ A = segs.A[yb];
    const float *__restrict__ Yact = segs.Y[yb];
    const int M1 = segs.ld[yb], act = Yact ? segs.act[yb] : 0, acol = segs.o0[yb] + 2 * li;
    f32x16 acc[2][4];
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[m][a][r] = 0.0f;
    float csum[2] = {0.0f, 0.0f};
    if constexpr (B3) {                                          // split-bf16 products: blocks of 16 rows per wavefront
        const int64_t base = ((int64_t)g * 4 + wave) * 16, stride = (int64_t)G * 4 * 16;
        if (HASACT && act != 0) tn_wide_stream_b3<true>(A, Yact, act, B, N, M1, acol, M2, bcol, hh, base, stride, acc, csum);
        else tn_wide_stream_b3<false>(A, Yact, act, B, N, M1, acol, M2, bcol, hh, base, stride, acc, csum);
    } else {
        const int64_t base = ((int64_t)g * 4 + wave) * 2 * PF, stride = (int64_t)G * 4 * 2 * PF;
        if (act != 0) tn_wide_stream<PF, true>(A, Yact, act, B, N, M1, acol, M2, bcol, hh, base, stride, acc, csum);
        else tn_wide_stream<PF, false>(A, Yact, act, B, N, M1, acol, M2, bcol, hh, base, stride, acc, csum);
    }
    float *cred = red + 8 * 16 * 64;
    for (int w = 1; w < 4; w++) {
        if (wave == w) {
#pragma unroll
            for (int q = 0; q < 8; q++)
#pragma unroll
                for (int r = 0; r < 16; r++) red[(q * 16 + r) * 64 + lane] = acc[q >> 2][q & 3][r];
            cred[lane] = csum[0];
            cred[64 + lane] = csum[1];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int q = 0; q < 8; q++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[q >> 2][q & 3][r] += red[(q * 16 + r) * 64 + lane];
            csum[0] += cred[lane];
            csum[1] += cred[64 + lane];
        }
        __syncthreads();
    }
    if (wave != 0) return;
    const int M1tp = segs.nyb * 32;
    float *sl = slab + ((int64_t)g * M1tp + yb * 32) * M2p + col0;
    if (col0 < M2p) {
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int go = (r & 3) + 8 * (r >> 2) + 4 * hh;  // accumulator row -> column 2*go + m of this A pair block
                *reinterpret_cast<float4 *>(sl + (int64_t)(2 * go + m) * M2p) = make_float4(acc[m][0][r], acc[m][1][r], acc[m][2][r], acc[m][3][r]);
            }
    }
    if (tile != 0) return;                                       // (every tile of B sees the same A: the column sums are written once)
#pragma unroll
    for (int m = 0; m < 2; m++) {
        const float t = csum[m] + __uint_as_float(dgg::xor_shfl<32>(__float_as_uint(csum[m]), lane));
        if (hh == 0) cs_slab[(int64_t)g * M1tp + yb * 32 + 2 * li + m] = t;
    }
}

// all segments of a gemm_tn_multi / gemm_tn_pairs slab in ONE launch (blockIdx.z = segment)
struct RedSegs {
    float *C[8];
    float *colsum[8];
    int M1[8], M2[8], M1p[8], rowbase[8], c_layout[8];
};
// One workgroup = 64 consecutive output elements x 16 groups of partial slabs: every thread sums its <= nchunks/16 slabs with the loads
// in flight together, the 16 groups meet in LDS, ONE thread per element adds the total to C (round 3: gridDim.y = nchunks/8 partial sums
// per element met in C through float atomics -- 516k contended atomics, 23 us of the headline step; now 5).
__global__ __launch_bounds__(1024) void gemm_tn_reduce_multi(const float *__restrict__ slab, const float *__restrict__ cs_slab, int nchunks,
                                                             int M2p, int64_t chunk_stride, int64_t cs_stride, RedSegs rs) {
    __shared__ float part[16][64];
    const int sgi = blockIdx.z;
    const int M1 = rs.M1[sgi], M2 = rs.M2[sgi], M1p = rs.M1p[sgi];
    const int lane = threadIdx.x & 63, pw = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    if (blockIdx.x * 64 >= M1p * M2p) return;                    // (block-uniform)
    auto total = [&](const float *base, int64_t stride, bool ok) {
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; j++) acc[j] = 0.0f;
        for (int k0 = pw; k0 < nchunks; k0 += 128) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int k = k0 + 16 * j;
                acc[j] += (ok && k < nchunks) ? base[(int64_t)k * stride] : 0.0f;
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
    const int o = e / M2p, c = e % M2p;
    const bool ok = e < M1p * M2p && o < M1 && c < M2;
    const float v = total(slab + (int64_t)rs.rowbase[sgi] * M2p + e, chunk_stride, ok);
    if (ok && pw == 0) {
        float *dst = rs.c_layout[sgi] == 0 ? rs.C[sgi] + (int64_t)o * M2 + c : rs.C[sgi] + (int64_t)c * M1 + o;
        *dst += v;                                               // (the only writer of this element)
    }
    if (rs.colsum[sgi] && blockIdx.x * 64 < M1) {                // the column sums of A (bias gradients): the first ceil(M1/64) blocks
        const bool okc = e < M1;
        const float cv = total(cs_slab + rs.rowbase[sgi] + e, cs_stride, okc);
        if (okc && pw == 0) rs.colsum[sgi][e] += cv;
    }
}

constexpr int GT_SPLIT = 32;
__global__ __launch_bounds__(256) void gemm_tn_reduce(const float *__restrict__ slab, const float *__restrict__ cs_slab,
                                                      int nchunks, int M1, int M2, int M1p, int M2p,
                                                      float *__restrict__ Cout, int c_layout, float *__restrict__ colsum,
                                                      int64_t chunk_stride, int64_t cs_stride) {
    const int e = blockIdx.x * 256 + threadIdx.x;                // element of the padded [M1p][M2p] block
    const int per = (nchunks + (int)gridDim.y - 1) / (int)gridDim.y;
    const int k0 = blockIdx.y * per, k1 = k0 + per < nchunks ? k0 + per : nchunks;
    if (k0 >= k1) return;
    if (e < M1p * M2p) {
        const int o = e / M2p, c = e % M2p;
        if (o < M1 && c < M2) {
            float s = 0.0f;
            for (int k = k0; k < k1; k++) s += slab[(int64_t)k * chunk_stride + e];
            float *dst = c_layout == 0 ? Cout + (int64_t)o * M2 + c : Cout + (int64_t)c * M1 + o;
            atomicAdd(dst, s);
        }
    }
    if (colsum && cs_slab && e < M1) {
        float s = 0.0f;
        for (int k = k0; k < k1; k++) s += cs_slab[(int64_t)k * cs_stride + e];
        atomicAdd(colsum + e, s);
    }
}

template <int NACC>
void launch_linear_fwd_n(const float *x, int64_t N, int d, const float *W, const float *b, int out, int w_layout, int act, float *y,
                         hipStream_t st) {
    const unsigned gy = (unsigned)((out + 32 * NACC - 1) / (32 * NACC));
    // 128 rows per workgroup.  (64- and 32-row workgroups -- more workgroups for Pubmed-size inputs -- measured slower:
    // fewer threads then stage the same weight tile per K-block.)
    hipLaunchKernelGGL((linear_fwd_mfma<NACC, 4>), dim3((unsigned)((N + 127) / 128), gy), dim3(256), 0, st, x, N, d, W, b, out, w_layout, act, y, LinSegs{});
}

int launch_linear_fwd(const float *x, int64_t N, int d, const float *W, const float *b, int out, int w_layout, int act,
                      float *y, hipStream_t st) {
    // (wide layers on few rows -- the latent-2048 k-net of a 2 000-node PPI graph: 18 row tiles x 8 column tiles of 128 = 144
    //  workgroups for 256 CUs -- take 64-column tiles: twice the workgroups, x re-read from L2; DGG_LIN_NACC1=<blocks> forces a shape)
    static const int nacc_env = [] { const char *e = getenv("DGG_LIN_NACC1"); return e ? atoi(e) : 0; }();
    const int64_t wg4 = ((N + 127) / 128) * ((out + 127) / 128);
    if (out <= 32) launch_linear_fwd_n<1>(x, N, d, W, b, out, w_layout, act, y, st);
    else if (out <= 64 || nacc_env == 2 || (nacc_env == 0 && wg4 < 256)) launch_linear_fwd_n<2>(x, N, d, W, b, out, w_layout, act, y, st);
    else launch_linear_fwd_n<4>(x, N, d, W, b, out, w_layout, act, y, st);
    return dgg_check_launch("linear_fwd");
}

template <int NACC>
void launch_linear_fwd_multi_n(const float *x, int64_t N, int d, const float *W, const float *b, int out, const LinSegs &segs, hipStream_t st) {
    // 128 rows per workgroup.  (64-row workgroups for inputs that give fewer 128-row tiles than CUs -- Pubmed: 155 -- measured slower
    // for the fused projections as well: 0.770 against 0.697 ms per Pubmed step; narrower COLUMN tiles are what helps there.)
    hipLaunchKernelGGL((linear_fwd_mfma<NACC, 4, true>), dim3((unsigned)((N + 127) / 128), (unsigned)(out / (32 * NACC))), dim3(256), 0, st, x, N,
                       d, W, b, out, 0, 0, nullptr, segs);
}

// block shape per wavefront and the number of row-streaming workgroups per output block: the tiny weights of the hot
// path get GT_GRID streams (one output block, all CUs share the rows); large outputs (GCNII layers, 2048 x 4096) already
// have hundreds of output blocks, so each is streamed by ONE workgroup and the slab holds a single partial
inline void gemm_tn_shape(int M1p, int M2p, int &mb, int &nb) {
    mb = M1p / 32 >= 2 ? 2 : 1;
    nb = M2p / 32 >= 3 ? 4 : M2p / 32;
}
inline int gemm_tn_grid(int64_t N, int M1p, int M2p) {
    int mb, nb;
    gemm_tn_shape(M1p, M2p, mb, nb);
    const int64_t blocks = (int64_t)((M1p / 32 + mb - 1) / mb) * ((M2p / 32 + nb - 1) / nb);
    int64_t g = (4 * GT_GRID + blocks - 1) / blocks;             // ~1024 workgroups in total
    if (g > GT_GRID) g = GT_GRID;
    const int64_t need = (N + 4 * 2 * GT_PF - 1) / (4 * 2 * GT_PF);
    if (g > need) g = need;
    // every stream writes (and the reduce re-reads) a full M1p x M2p partial: keep that below ~1/4 of the operand bytes
    int64_t cap = N * (int64_t)(M1p + M2p) / (4 * (int64_t)M1p * M2p);
    if (cap < 8) cap = 8;
    if (g > cap) g = cap;                                        // an upper limit only
    if (g > need) g = need;
    return (int)(g < 1 ? 1 : g);
}
size_t gemm_tn_ws_floats(int64_t N, int M1, int M2) {
    const size_t M1p = (size_t)(M1 + 31) / 32 * 32, M2p = (size_t)(M2 + 31) / 32 * 32;
    const size_t g = (size_t)gemm_tn_grid(N, (int)M1p, (int)M2p);
    return g * M1p * M2p + g * M1p;
}

template <int MB, int NB>
void launch_gemm_tn_persist(const float *A, const float *B, int64_t N, int M1, int M2, int M1p, int M2p, float *slab,
                            float *cs_slab, int g, const float *Yact, int act, hipStream_t st) {
    const unsigned gy = (unsigned)((M1p / 32 + MB - 1) / MB), gz = (unsigned)((M2p / 32 + NB - 1) / NB);
    const size_t lds = (size_t)(MB * NB * 16 * 64 + MB * 64) * sizeof(float);
    if (Yact)
        hipLaunchKernelGGL((gemm_tn_persist<MB, NB, GT_PF / 2, true>), dim3(g, gy, gz), dim3(256), lds, st, A, B, N, M1, M2, M1p, M2p, slab,
                           cs_slab, Yact, act);
    else
        hipLaunchKernelGGL((gemm_tn_persist<MB, NB, GT_PF, false>), dim3(g, gy, gz), dim3(256), lds, st, A, B, N, M1, M2, M1p, M2p, slab,
                           cs_slab, Yact, act);
}

int launch_gemm_tn(const float *A, const float *B, int64_t N, int M1, int M2, float *C, int c_layout, float *colsum,
                   float *ws, hipStream_t st, const float *Yact = nullptr, int act = 0) {
    if (!ws) return dgg_set_error(DGG_ERR_ARG, "gemm_tn: workspace is NULL (dgg_gemm_tn_ws_floats)");
    const int M1p = (M1 + 31) / 32 * 32, M2p = (M2 + 31) / 32 * 32;
    const int g = gemm_tn_grid(N, M1p, M2p);
    float *slab = ws, *cs_slab = ws + (size_t)g * M1p * M2p;
    float *csl = colsum ? cs_slab : nullptr;
    int mb, nb;
    gemm_tn_shape(M1p, M2p, mb, nb);
    if (Yact) mb = 1;                                            // the masked variant carries the y operand too: 32-row blocks
    if (mb == 2 && nb == 4) launch_gemm_tn_persist<2, 4>(A, B, N, M1, M2, M1p, M2p, slab, csl, g, Yact, act, st);
    else if (mb == 2 && nb == 2) launch_gemm_tn_persist<2, 2>(A, B, N, M1, M2, M1p, M2p, slab, csl, g, Yact, act, st);
    else if (mb == 2) launch_gemm_tn_persist<2, 1>(A, B, N, M1, M2, M1p, M2p, slab, csl, g, Yact, act, st);
    else if (nb == 4) launch_gemm_tn_persist<1, 4>(A, B, N, M1, M2, M1p, M2p, slab, csl, g, Yact, act, st);
    else if (nb == 2) launch_gemm_tn_persist<1, 2>(A, B, N, M1, M2, M1p, M2p, slab, csl, g, Yact, act, st);
    else launch_gemm_tn_persist<1, 1>(A, B, N, M1, M2, M1p, M2p, slab, csl, g, Yact, act, st);
    // slabs per reducing workgroup: ~8 (each output then takes g/8 <= 32 float atomics); small outputs keep the 32-way split so
    // that the launch still has a few hundred workgroups
    int split = (M1p * M2p >= 16384) ? (g + 7) / 8 : g;
    split = split < 1 ? 1 : (split > GT_SPLIT ? GT_SPLIT : split);
    hipLaunchKernelGGL(gemm_tn_reduce, dim3((unsigned)((M1p * M2p + 255) / 256), (unsigned)split), dim3(256), 0, st, slab, csl, g, M1,
                       M2, M1p, M2p, C, c_layout, colsum, (int64_t)M1p * M2p, (int64_t)M1p);
    return dgg_check_launch("gemm_tn");
}

}  // namespace

extern "C" {

int dgg_linear_fwd(const float *x, int64_t N, int d, const float *W, const float *b, int out, int w_layout, int act,
                   float *y, void *stream) {
    if (N < 0 || d < 1 || out < 1) return dgg_set_error(DGG_ERR_ARG, "linear_fwd: bad shape");
    if (N == 0) return 0;
    return launch_linear_fwd(x, N, d, W, b, out, w_layout, act, y, (hipStream_t)stream);
}

// Stacks the weights (and biases) of up to 8 layers into the row-stacked nn.Linear layout dgg_linear_fwd_multi reads, in ONE launch
// (the host wrapper used two torch.cat, a fill and the transposes of the [d, out] weights: five launches per forward).
struct PackSegs {
    const float *W[8];
    const float *b[8];
    int out[8], layout[8], first[8];
};
__global__ __launch_bounds__(256) void pack_weights_kernel(PackSegs sg, int nseg, int d, int total, float *__restrict__ Wcat,
                                                           float *__restrict__ bcat) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= total * (d + 1)) return;
    const int c = e / (d + 1), k = e % (d + 1);
    int s = 0;
#pragma unroll
    for (int q = 1; q < 8; q++)
        if (q < nseg && c >= sg.first[q]) s = q;
    const int lc = c - sg.first[s];
    if (k == d) { bcat[c] = sg.b[s] ? sg.b[s][lc] : 0.0f; return; }
    Wcat[(int64_t)c * d + k] = sg.layout[s] == 0 ? sg.W[s][(int64_t)lc * d + k] : sg.W[s][(int64_t)k * sg.out[s] + lc];
}

extern "C" int dgg_linear_pack_weights(int nseg, const float *const *W, const float *const *b, const int *seg_out, const int *seg_layout,
                                       int d, float *Wcat, float *bcat, void *stream) {
    if (nseg < 1 || nseg > 8 || d < 1) return dgg_set_error(DGG_ERR_ARG, "linear_pack_weights: 1..8 layers");
    PackSegs sg{};
    int total = 0;
    for (int s = 0; s < nseg; s++) {
        sg.W[s] = W[s]; sg.b[s] = b ? b[s] : nullptr; sg.out[s] = seg_out[s]; sg.layout[s] = seg_layout[s]; sg.first[s] = total;
        total += seg_out[s];
    }
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)((total * (d + 1) + 255) / 256)), dim3(256), 0, (hipStream_t)stream, sg, nseg, d, total,
                       Wcat, bcat);
    return dgg_check_launch("linear_pack_weights");
}

// Several layers on one input, X read once: Wcat [sum out_s, d] / bcat [sum out_s] (nullable) = the layers' nn.Linear weights
// (layout 0) stacked by rows; layer s has out_s outputs (a multiple of 32; sum <= 256), activation act_s and destination
// y_s [N, out_s].  Bit-identical to nseg calls of dgg_linear_fwd (same k-ordered fmaf chains).
int dgg_linear_fwd_multi(const float *x, int64_t N, int d, const float *Wcat, const float *bcat, int nseg, const int *seg_out,
                         const int *seg_act, float *const *y, void *stream) {
    if (N < 0 || d < 1 || nseg < 1 || nseg > 8) return dgg_set_error(DGG_ERR_ARG, "linear_fwd_multi: bad shape");
    LinSegs segs{};
    int cb = 0;
    for (int sgi = 0; sgi < nseg; sgi++) {
        if (seg_out[sgi] < 32 || seg_out[sgi] % 32 != 0 || cb + seg_out[sgi] / 32 > 8)
            return dgg_set_error(DGG_ERR_UNSUPPORTED, "linear_fwd_multi: layer widths must be multiples of 32 with a total of at most 256");
        for (int q = 0; q < seg_out[sgi] / 32; q++, cb++) {
            segs.y[cb] = y[sgi] + q * 32;
            segs.ld[cb] = seg_out[sgi];
            segs.act[cb] = seg_act[sgi];
        }
    }
    if (cb == 7) return dgg_set_error(DGG_ERR_UNSUPPORTED, "linear_fwd_multi: 7 column blocks (224 outputs) are not tiled");
    if (N == 0) return 0;
    const int out = cb * 32;
    hipStream_t st = (hipStream_t)stream;
    // large inputs: persistent workgroups with the weights resident in LDS (DGG_LINEAR_REG=0/1 forces a path: measurement / tests)
    static const int forced = [] { const char *e = getenv("DGG_LINEAR_REG"); return e ? atoi(e) : -1; }();
    const bool aligned = (reinterpret_cast<uintptr_t>(x) % 16 == 0) && (reinterpret_cast<uintptr_t>(Wcat) % 16 == 0);
    if (d == 128 && aligned && cb <= 6 && (forced >= 0 ? forced != 0 : N >= 32768)) {
        const int64_t nrb = (N + 31) / 32;
        const unsigned grid = (unsigned)std::min<int64_t>(256, (nrb + 3) / 4);   // one workgroup per CU
        switch (cb) {
#define DGG_LIN_REG(NA) case NA: hipLaunchKernelGGL((linear_fwd_reg<128, NA>), dim3(grid), dim3(256), 0, st, x, N, Wcat, bcat, segs, nrb); break
            DGG_LIN_REG(1); DGG_LIN_REG(2); DGG_LIN_REG(3); DGG_LIN_REG(4); DGG_LIN_REG(5); DGG_LIN_REG(6);
#undef DGG_LIN_REG
        }
        return dgg_check_launch("linear_fwd_multi");
    }
    // Inputs with fewer 128-row tiles than CUs (Pubmed: 155): all columns in one workgroup would leave a third of the chip idle and
    // every wavefront alone on its SIMD with nothing to cover its LDS staging -- the columns are split into tiles of 64 (32 for an
    // odd number of blocks) instead, x is then re-read from L2 by the other column tiles (measured at N = 19 717, d = 500, 192
    // outputs: 92 -> 73 us; tiles of 32: 83, of 96: 92).  DGG_LIN_NACC=<blocks per tile> forces a shape (measurement).
    static const int nacc_env = [] { const char *e = getenv("DGG_LIN_NACC"); return e ? atoi(e) : 0; }();
    int nacc_small = 0;
    if ((N + 127) / 128 < 256 && cb > 2) nacc_small = cb % 2 == 0 ? 2 : 1;
    // a few thousand rows (Cora: 22 row tiles): even the tiles of 64 columns leave three quarters of the chip idle -- tiles of 32
    // (Cora shape, d = 1433, 192 outputs: step 0.340 -> 0.311 ms; 64- and 32-row workgroups on top of that: 0.331 / 0.398)
    if (cb > 1 && ((N + 127) / 128) * ((cb + 1) / 2) < 128) nacc_small = 1;
    if (nacc_env > 0 && cb % nacc_env == 0) nacc_small = nacc_env;
    if (nacc_small > 0) {
        switch (nacc_small) {
            case 1: launch_linear_fwd_multi_n<1>(x, N, d, Wcat, bcat, out, segs, st); break;
            case 2: launch_linear_fwd_multi_n<2>(x, N, d, Wcat, bcat, out, segs, st); break;
            case 3: launch_linear_fwd_multi_n<3>(x, N, d, Wcat, bcat, out, segs, st); break;
            default: launch_linear_fwd_multi_n<4>(x, N, d, Wcat, bcat, out, segs, st); break;
        }
        return dgg_check_launch("linear_fwd_multi");
    }
    // one workgroup computes ALL columns of its 128 rows whenever the accumulators fit (<= 6 blocks): X is then read once
    switch (cb) {
        case 1: launch_linear_fwd_multi_n<1>(x, N, d, Wcat, bcat, out, segs, st); break;
        case 2: launch_linear_fwd_multi_n<2>(x, N, d, Wcat, bcat, out, segs, st); break;
        case 3: launch_linear_fwd_multi_n<3>(x, N, d, Wcat, bcat, out, segs, st); break;
        case 4: launch_linear_fwd_multi_n<4>(x, N, d, Wcat, bcat, out, segs, st); break;
        case 5: launch_linear_fwd_multi_n<5>(x, N, d, Wcat, bcat, out, segs, st); break;
        case 6: launch_linear_fwd_multi_n<6>(x, N, d, Wcat, bcat, out, segs, st); break;
        default: launch_linear_fwd_multi_n<4>(x, N, d, Wcat, bcat, out, segs, st); break;   // 8 blocks = two column tiles of 4
    }
    return dgg_check_launch("linear_fwd_multi");
}

// floats of workspace dgg_linear_bwd needs: N*out for the activation backward + the weight-gradient slab
size_t dgg_linear_bwd_ws_floats(int64_t N, int d, int out) { return (size_t)N * out + gemm_tn_ws_floats(N, out, d); }
size_t dgg_gemm_tn_ws_floats(int64_t N, int M1, int M2) { return gemm_tn_ws_floats(N, M1, M2); }

// backward of y = act(x W^T + b).  ws: dgg_linear_bwd_ws_floats(N, d, out) floats.  dx (nullable) is OVERWRITTEN;
// dW (layout of W) and db (nullable) are ACCUMULATED into (caller zeroes them), so that several uses of one
// weight add up.
int dgg_linear_bwd(const float *x, int64_t N, int d, const float *W, int out, int w_layout, int act, const float *y,
                   const float *dy, float *dx, float *dW, float *db, float *ws, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    if (N == 0) return 0;
    if (!ws) return dgg_set_error(DGG_ERR_ARG, "linear_bwd: workspace is NULL (dgg_linear_bwd_ws_floats)");
    const float *dp = dy;
    if (act != 0 && !dx && dW)      // only the weight gradient is wanted: the activation mask rides on the GEMM's operand load
        return launch_gemm_tn(dy, x, N, out, d, dW, w_layout == 0 ? 0 : 1, db, ws + (size_t)N * out, st, y, act);
    if (act != 0) {
        int64_t n = N * out;
        unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
        hipLaunchKernelGGL(act_bwd_kernel, dim3(blocks), dim3(256), 0, st, y, dy, n, act, ws);
        dp = ws;
    }
    if (dx) {
        // dx[N,d] = dp[N,out] * W  ==  linear_fwd with the weight read in the transposed layout
        int rc = launch_linear_fwd(dp, N, out, W, nullptr, d, w_layout == 0 ? 1 : 0, 0, dx, st);
        if (rc != 0) return rc;
    }
    if (dW) return launch_gemm_tn(dp, x, N, out, d, dW, w_layout == 0 ? 0 : 1, db, ws + (size_t)N * out, st);
    return dgg_check_launch("linear_bwd");
}

// dp = dy * act'(y), elementwise (the cotangent of a fused activation epilogue, e.g. the ReLU of dgg_ell_spmm_act_fwd)
int dgg_act_bwd(const float *y, const float *dy, int64_t n, int act, float *dp, void *stream) {
    if (n == 0) return 0;
    const unsigned blocks = (unsigned)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(act_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, y, dy, n, act, dp);
    return dgg_check_launch("act_bwd");
}

// nseg weight gradients over one streamed input (see gemm_tn_multi): C_s (c_layout_s 0: [M1_s][M2], 1: [M2][M1_s]) and colsum_s
// (nullable: bias gradient) are ACCUMULATED into (caller zeroes).  M1_s multiples of 32 with a total of at most 256, M2 <= 128.
// Y_s (nullable) / act_s: the forward output and activation whose derivative masks A_s on the fly.
// ws: dgg_gemm_tn_multi_ws_floats(N, total M1, M2) floats.
size_t dgg_gemm_tn_multi_ws_floats(int64_t N, int M1_total, int M2) {
    const size_t M2p = (size_t)(M2 + 31) / 32 * 32;
    return (size_t)256 * ((size_t)M1_total * M2p + (size_t)M1_total);     // up to 256 row streams
}
int dgg_gemm_tn_multi(int nseg, const float *const *A, const int *M1, const float *const *Y, const int *act, const float *B, int64_t N,
                      int M2, float *const *C, const int *c_layout, float *const *colsum, float *ws, void *stream) {
    if (nseg < 1 || nseg > 8 || M2 < 1) return dgg_set_error(DGG_ERR_UNSUPPORTED, "gemm_tn_multi: 1..8 operands");
    if (!ws) return dgg_set_error(DGG_ERR_ARG, "gemm_tn_multi: workspace is NULL (dgg_gemm_tn_multi_ws_floats)");
    TnSegs segs{};
    int yb = 0;
    for (int sgi = 0; sgi < nseg; sgi++) {
        if (M1[sgi] < 32 || M1[sgi] % 32 != 0 || yb + M1[sgi] / 32 > 8)
            return dgg_set_error(DGG_ERR_UNSUPPORTED, "gemm_tn_multi: operand widths must be multiples of 32 with a total of at most 256");
        for (int q = 0; q < M1[sgi] / 32; q++, yb++) {
            segs.A[yb] = A[sgi];
            segs.Y[yb] = Y ? Y[sgi] : nullptr;
            segs.ld[yb] = M1[sgi];
            segs.o0[yb] = q * 32;
            segs.act[yb] = act ? act[sgi] : 0;
        }
    }
    segs.nyb = yb;
    if (N == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int M2p = (M2 + 31) / 32 * 32, M1tp = yb * 32;
    constexpr int PF = 4;
    int G = 128;                                                 // row streams (a multiple of 8): G * nyb workgroups
    const int64_t need = (N + 4 * 2 * PF - 1) / (4 * 2 * PF);
    while (G > 8 && G / 2 >= need) G /= 2;
    // the hot-path shape (128-wide input, operands in multiples of 64 columns, 16-byte aligned rows): wide tiles
    // (and inputs WIDER than 128 columns -- a multiple of 4 -- as several 128-column tiles of the same kernel: the only form for them)
    bool wide = (M2 == 128 || (M2 > 128 && M2 % 4 == 0)) && (N >= 4096 || M2 > 128) && reinterpret_cast<uintptr_t>(B) % 16 == 0;
    for (int sgi = 0; sgi < nseg; sgi++)
        wide = wide && M1[sgi] % 64 == 0 && reinterpret_cast<uintptr_t>(A[sgi]) % 8 == 0 && (!Y || !Y[sgi] || reinterpret_cast<uintptr_t>(Y[sgi]) % 8 == 0);
    static const int force_wide = [] { const char *e = getenv("DGG_TN_WIDE"); return e ? atoi(e) : -1; }();
    if (force_wide == 0 && M2 <= 128) wide = false;
    if (M2 > 128 && !wide)
        return dgg_set_error(DGG_ERR_UNSUPPORTED, "gemm_tn_multi: an input wider than 128 columns needs M2 % 4 == 0, operand widths in "
                                                  "multiples of 64 and 16-byte aligned rows");
    const int ntile = (M2 + 127) / 128;
    if (wide) {
        constexpr int PFW = 6;
        const int npair = yb / 2 * ntile;
        G = 512 / npair / 8 * 8;
        G = G < 8 ? 8 : G;                                 // ~512 workgroups: two per CU (<= 256 registers); measured: 256 / 384 /
                                                                 //  768 workgroups and prefetch depths 4..10 are equal or slower
        G = G > 256 ? 256 : G;
        const int64_t needw = (N + 4 * 2 * PFW - 1) / (4 * 2 * PFW);
        while (G > 8 && G - 8 >= needw) G -= 8;
    }
    float *slab = ws, *cs_slab = ws + (size_t)G * M1tp * M2p;
    const unsigned grid = (unsigned)(G * yb);
    const int nb = M2p / 32 == 3 ? 4 : M2p / 32;
    const size_t lds = (size_t)(nb * 16 * 64 + 64) * sizeof(float);
    const size_t ldsw = (size_t)(8 * 16 * 64 + 128) * sizeof(float);
    const dim3 gridw((unsigned)(G * (yb / 2) * ntile));
    // weight gradients (2e-4 of max is what they are held to): split-bf16 products on the bf16 matrix cores; DGG_TN_B3=0 keeps the exact
    // fp32 products (v_mfma_f32_32x32x2_f32)
    static const bool tn_b3 = [] { const char *e = getenv("DGG_TN_B3"); return !(e && atoi(e) == 0); }();
    bool hasact = false;
    for (int sgi = 0; sgi < nseg; sgi++) hasact = hasact || (Y && Y[sgi] && act && act[sgi] != 0);
    if (wide && tn_b3 && !hasact) hipLaunchKernelGGL((gemm_tn_wide<6, true, false>), gridw, dim3(256), ldsw, st, segs, B, N, G, M2, M2p, ntile, slab, cs_slab);
    else if (wide && tn_b3) hipLaunchKernelGGL((gemm_tn_wide<6, true, true>), gridw, dim3(256), ldsw, st, segs, B, N, G, M2, M2p, ntile, slab, cs_slab);
    else if (wide) hipLaunchKernelGGL((gemm_tn_wide<6>), gridw, dim3(256), ldsw, st, segs, B, N, G, M2, M2p, ntile, slab, cs_slab);
    else if (nb == 4) hipLaunchKernelGGL((gemm_tn_multi<4, PF>), dim3(grid), dim3(256), lds, st, segs, B, N, M2, M2p, G, slab, cs_slab);
    else if (nb == 2) hipLaunchKernelGGL((gemm_tn_multi<2, PF>), dim3(grid), dim3(256), lds, st, segs, B, N, M2, M2p, G, slab, cs_slab);
    else hipLaunchKernelGGL((gemm_tn_multi<1, PF>), dim3(grid), dim3(256), lds, st, segs, B, N, M2, M2p, G, slab, cs_slab);
    RedSegs rsg{};
    int rb = 0, m1max = 0;
    for (int sgi = 0; sgi < nseg; sgi++) {
        rsg.C[sgi] = C[sgi]; rsg.colsum[sgi] = colsum ? colsum[sgi] : nullptr; rsg.M1[sgi] = M1[sgi]; rsg.M2[sgi] = M2; rsg.M1p[sgi] = M1[sgi];
        rsg.rowbase[sgi] = rb; rsg.c_layout[sgi] = c_layout[sgi];
        rb += M1[sgi];
        m1max = M1[sgi] > m1max ? M1[sgi] : m1max;
    }
    hipLaunchKernelGGL(gemm_tn_reduce_multi, dim3((unsigned)((m1max * M2p + 63) / 64), 1, (unsigned)nseg), dim3(1024), 0, st,
                       slab, cs_slab, G, M2p, (int64_t)M1tp * M2p, (int64_t)M1tp, rsg);
    return dgg_check_launch("gemm_tn_multi");
}

// Several INDEPENDENT small products C_p[M1_p, M2_p] += A_p[N, M1_p]^T B_p[N, M2_p] over the same N rows in one launch (+ one
// reduce per product): the three weight gradients of the k-net (k_embed, k_mu, k_project: autograd of dgm.py:1576-1577, 2051-2063)
// are 32x65, 16x32 and 1x16 -- as three launches each is a latency-bound kernel on an empty chip.  M1_p <= 256 in total
// (padded to 32), M2_p <= 128.  colsum_p (nullable) += column sums of A_p.  ws: dgg_gemm_tn_multi_ws_floats(N, padded total M1, 128).
int dgg_gemm_tn_pairs(int npair, const float *const *A, const int *M1, const float *const *B, const int *M2, int64_t N, float *const *C,
                      float *const *colsum, float *ws, void *stream) {
    if (npair < 1 || npair > 8) return dgg_set_error(DGG_ERR_UNSUPPORTED, "gemm_tn_pairs: 1..8 products");
    if (!ws) return dgg_set_error(DGG_ERR_ARG, "gemm_tn_pairs: workspace is NULL");
    TnSegs segs{};
    int yb = 0, rb[8], m2max = 1;
    for (int p = 0; p < npair; p++) {
        if (M1[p] < 1 || M2[p] < 1 || M2[p] > 128) return dgg_set_error(DGG_ERR_UNSUPPORTED, "gemm_tn_pairs: M2 <= 128");
        rb[p] = yb * 32;
        for (int q = 0; q < (M1[p] + 31) / 32; q++, yb++) {
            if (yb >= 8) return dgg_set_error(DGG_ERR_UNSUPPORTED, "gemm_tn_pairs: at most 256 (padded) rows of output in total");
            segs.A[yb] = A[p]; segs.Y[yb] = nullptr; segs.ld[yb] = M1[p]; segs.o0[yb] = q * 32; segs.act[yb] = 0;
            segs.Bp[yb] = B[p]; segs.m2[yb] = M2[p];
        }
        m2max = M2[p] > m2max ? M2[p] : m2max;
    }
    segs.nyb = yb;
    if (N == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int M2p = 128, M1tp = yb * 32;                         // one slab geometry for all pairs (outputs are tiny)
    constexpr int PF = 4;
    int G = 256;                                                 // (measured: 128 streams 42 us, 256 streams 29 us; prefetch 2 / 8 no better)
    const int64_t need = (N + 4 * 2 * PF - 1) / (4 * 2 * PF);
    while (G > 8 && G / 2 >= need) G /= 2;
    float *slab = ws, *cs_slab = ws + (size_t)G * M1tp * M2p;
    const size_t lds = (size_t)(4 * 16 * 64 + 64) * sizeof(float);
    hipLaunchKernelGGL((gemm_tn_multi<4, PF>), dim3((unsigned)(G * yb)), dim3(256), lds, st, segs, A[0], N, m2max, M2p, G, slab, cs_slab);
    RedSegs rsg{};
    int m1pmax = 0;
    for (int p = 0; p < npair; p++) {
        rsg.C[p] = C[p]; rsg.colsum[p] = colsum ? colsum[p] : nullptr; rsg.M1[p] = M1[p]; rsg.M2[p] = M2[p];
        rsg.M1p[p] = (M1[p] + 31) / 32 * 32; rsg.rowbase[p] = rb[p]; rsg.c_layout[p] = 0;
        m1pmax = rsg.M1p[p] > m1pmax ? rsg.M1p[p] : m1pmax;
    }
    hipLaunchKernelGGL(gemm_tn_reduce_multi, dim3((unsigned)((m1pmax * M2p + 63) / 64), 1, (unsigned)npair), dim3(1024), 0, st,
                       slab, cs_slab, G, M2p, (int64_t)M1tp * M2p, (int64_t)M1tp, rsg);
    return dgg_check_launch("gemm_tn_pairs");
}

// C[M1,M2] += A[N,M1]^T B[N,M2]  (c_layout 1: C stored [M2][M1]); colsum (nullable, [M1]) += column sums of A;
// ws: dgg_gemm_tn_ws_floats(N, M1, M2) floats
int dgg_gemm_tn_acc(const float *A, const float *B, int64_t N, int M1, int M2, float *C, int c_layout, float *colsum,
                    float *ws, void *stream) {
    if (N == 0) return 0;
    return launch_gemm_tn(A, B, N, M1, M2, C, c_layout, colsum, ws, (hipStream_t)stream);
}

}  // extern "C"
