// dgg_topk_fast.hip -- pruned all-pairs scoring + per-row top-64: identical bits to the exhaustive kernel,
// a fraction of its work.
//
// Same contract as allpairs_topk_exhaustive (dgg_topk.hip): for every row i of the N x N score matrix
//   p'_ij = exp(log(exp(-0.05 ||xp_i - xp_j||) + 1e-8) + G_ij)          reference dgm.py:1618-1623, 1213-1229
// keep the 64 largest in (score desc, column asc) order                   reference dgm.py:1404 (torch.sort)
//
// Branch and bound.  Each row keeps the exact top-64 found so far (in the output arrays) and a threshold derived
// from its 64th score.  Every pair is first tested against a CONSERVATIVE upper bound of its score; only pairs
// whose bound reaches the threshold are scored exactly -- with the canonical fp32 arithmetic of dgg_common.h --
// and merged.  A pair rejected by the bound provably cannot enter the top-64, so the result is bit-identical.
//
//   stage A  (every pair, ~6 VALU ops)   perturbed: the raw 32-bit hash of the pair's noise against the row's
//                                        integer threshold (score <= G_ij + log(1+1e-8), distance >= 0);
//                                        unperturbed: bf16-MFMA distance lower bound against the row's radius.
//   stage B  (pairs passing A, batched)  survivors are pushed (ballot-compacted, no atomics) into a wave-private
//                                        LDS ring; every 64 of them are bounded together, one per lane:
//                                        G_ij + log(exp(t*dL_ij) + 1e-8) with dL_ij a rigorous lower bound of the
//                                        distance from the bf16 MFMA Gram tile:
//                                        d2 >= (n_i + n_j)(1 - eps) - 2 <bf16(x_i), bf16(x_j)>,  eps = 2^-8 (1 + slack)
//   stage C  (pairs passing B)           appended to the row's pending list; when 64 are pending the wavefront
//                                        scores them exactly, bitonic-sorts (DPP network) and merges them into
//                                        the row's list and tightens the thresholds (~log2(N/64) times per row).
//
// Decomposition: one wavefront = one workgroup = RBLK blocks of 32 rows; NO workgroup barriers, so a wavefront that
// stops to flush delays nobody.  MFMA orientation D[a][b]: a = column of the tile, b = row, so a LANE holds ONE ROW
// (b = lane & 31) and 16 columns: thresholds, row keys and norms are per-lane registers.  Column tiles (32 columns,
// bf16) are read coalesced from L2 into registers one tile ahead, written to a wave-private padded LDS image
// (conflict-free ds_read_b128) and read back as MFMA fragments.
#include "dgg_common.h"
#include "dgg_api_internal.h"
#include <stdlib.h>

using namespace dgg;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#ifdef DGG_STAMPS   // diagnostic build only: per-section cycle shares (never quote its run time)
__device__ unsigned long long dgg_stamp_acc[8];
#define STAMP(slot)                                                          \
    do {                                                                     \
        unsigned long long now_ = __builtin_readcyclecounter();              \
        stamp_loc[slot] += now_ - stamp_t;                                   \
        stamp_t = now_;                                                      \
    } while (0)
#define COUNT(slot, v) stamp_loc[slot] += (v)
#else
#define STAMP(slot)
#define COUNT(slot, v)
#endif

namespace {

constexpr int CT = 32;             // columns per MFMA tile
constexpr int CAP = 128;           // pending-candidate slots per row (a batch adds at most 64)
constexpr int FLUSH_AT = 64;       // flush a row once this many are pending
constexpr int QN = 128;            // stage-B ring entries
constexpr float EPS_BF16 = 0.0040f;   // 2^-8 (1 + 2^-9) bf16 rounding of both operands + fp32 accumulation slack

// ---- prologue: bf16 copy of the projected features and discounted squared norms ------------------------------
__global__ __launch_bounds__(256) void prep_kernel(const float *__restrict__ xp, int64_t N, int h,
                                                   __bf16 *__restrict__ xb, float *__restrict__ nb) {
    const int lane = threadIdx.x & 63;
    const int64_t j = (int64_t)blockIdx.x * 4 + dgg::wave_id();
    if (j >= N) return;
    float s = 0.0f;
    for (int c = lane; c < h; c += 64) {
        float v = xp[j * h + c];
        xb[j * h + c] = (__bf16)v;
        s += v * v;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) nb[j] = s * (1.0f - EPS_BF16);
}

// exact canonical score of pair (i, j); i is wave-uniform (its features come through the scalar cache)
template <int H>
__device__ __forceinline__ float exact_score(const float *__restrict__ xp, int64_t i, int32_t j, float t, int noise_mode,
                                             uint32_t s0, uint32_t s1) {
    const float *xi = xp + i * H;
    const float4 *xj = reinterpret_cast<const float4 *>(xp + (int64_t)j * H);
    float4 b[H / 4];
#pragma unroll
    for (int c4 = 0; c4 < H / 4; c4++) b[c4] = xj[c4];          // every gather in flight before the chain starts
    float d2 = 0.0f;
#pragma unroll
    for (int c4 = 0; c4 < H / 4; c4++) {
        float df;
        df = __fadd_rn(xi[4 * c4 + 0], -b[c4].x); d2 = __fmaf_rn(df, df, d2);
        df = __fadd_rn(xi[4 * c4 + 1], -b[c4].y); d2 = __fmaf_rn(df, df, d2);
        df = __fadd_rn(xi[4 * c4 + 2], -b[c4].z); d2 = __fmaf_rn(df, df, d2);
        df = __fadd_rn(xi[4 * c4 + 3], -b[c4].w); d2 = __fmaf_rn(df, df, d2);
    }
    float dist = c_sqrt(d2);
    float g = 0.0f;
    if (noise_mode >= 2) g = pair_noise(s0, s1, (uint32_t)i, (uint32_t)j, noise_mode == 3);
    return score_from_dist(dist, t, noise_mode != 0, g);
}

struct RowThr {
    uint32_t a;   // perturbed: stage-A threshold on the raw 32-bit hash; unperturbed: bits of the float radius^2
    float b;      // perturbed: tau_y = log(64th score)
};

__device__ __forceinline__ RowThr thresholds_from_score(float pp63, float t, bool perturb) {
    RowThr r;
    float lg = __logf(pp63);
    if (perturb) {
        r.b = lg;
        float gmin = lg - 1e-3f;                               // margin >> every rounding error in the chain
        float e1 = __expf(gmin * (-1.0f / 0.3f));              // P(G >= gmin) <= e1   (1 - exp(-e1) <= e1)
        float cnt = fminf(e1 * 16777216.0f, 16777216.0f);
        int um = 16777216 - (int)cnt - 2;
        um = um < 0 ? 0 : um;
        r.a = (uint32_t)um << 8;
    } else {
        float d63 = lg / t;                                    // distance of the 64th best
        float D = d63 * (1.0f + 1e-5f) + 1e-4f;
        r.a = __float_as_uint(D * D);
        r.b = 0.0f;
    }
    return r;
}

// upper bound of the perturbed log-score of a pair from the bf16 Gram value and the raw hash
__device__ __forceinline__ float score_upper_bound(float dot, float ni, float nj, uint32_t x, float t, bool zero_noise) {
    float L2 = __fmaf_rn(-2.0f, dot, ni + nj);
    float dL = __fsqrt_rn(fmaxf(L2, 0.0f));
    float lpub = __logf(__expf(t * dL) + 1e-8f);
    // -log(U), U = u24 2^-24: series near 1, where the hardware log is inexact
    uint32_t u24 = x >> 8;
    u24 = u24 == 0 ? 1u : u24;
    float epsu = (float)(16777216u - u24) * 5.9604644775390625e-8f;
    float ser = epsu * (1.0f + epsu * (0.5f + epsu * (0.33333334f + 0.25f * epsu)));
    float nl = epsu < 0.015625f ? ser : -__logf((float)u24 * 5.9604644775390625e-8f);
    float G = zero_noise ? 0.0f : -0.3f * __logf(nl);
    return lpub + G + (3e-5f + 2e-5f * fabsf(lpub));
}

struct FastCtl {
    int nfail;
    int pad[3];
};

template <int H, int NOISE, int RBLK>   // NOISE: 0 none, 2 hash, 3 symmetric hash
__global__ __launch_bounds__(64) void allpairs_topk_fast(
    const float *__restrict__ xp, const __bf16 *__restrict__ xb, const float *__restrict__ nb, int *__restrict__ pend_g,
    int64_t N, int64_t row0, int64_t row1, float t, uint32_t s0, uint32_t s1, int32_t *__restrict__ idx,
    float *__restrict__ val, FastCtl *__restrict__ ctl, int *__restrict__ faillist, float gscale, int2 *__restrict__ cand,
    int *__restrict__ cand_cnt, float *__restrict__ cand_guess) {
    constexpr int KS = H / 16;                 // MFMA k-steps == 16-byte chunks per lane per tile
    constexpr int STRIDE = H * 2 + 16;         // bytes per staged column (padded: conflict-free ds_read_b128)
    constexpr int RW = 32 * RBLK;              // rows per wavefront
    __shared__ __attribute__((aligned(16))) unsigned char colA[CT * STRIDE];
    __shared__ __attribute__((aligned(16))) float nbt[CT];
    __shared__ int cnt[RW];
    __shared__ float row_nb[RW];               // discounted norms / log-thresholds of the wave's rows, readable by any lane
    __shared__ float row_tb[RW];
    __shared__ uint32_t qj[QN], qx[QN], ql[QN];
    __shared__ float qa[QN];

    const int lane = threadIdx.x;
    const int r = lane & 31, hh = lane >> 5;
    const int64_t rbase = row0 + (int64_t)blockIdx.x * RW;
    int *pend = pend_g + (int64_t)blockIdx.x * RW * CAP;

    bf16x8 bfr[RBLK][KS];
    float nbi[RBLK];
    uint32_t k1[RBLK], k2[RBLK], ta[RBLK];
    bool rvalid[RBLK];
    int64_t iv[RBLK];
#pragma unroll
    for (int b = 0; b < RBLK; b++) {
        const int64_t i = rbase + b * 32 + r;
        rvalid[b] = i < row1;
        iv[b] = rvalid[b] ? i : row1 - 1;
#pragma unroll
        for (int s = 0; s < KS; s++) bfr[b][s] = *reinterpret_cast<const bf16x8 *>(xb + iv[b] * H + 16 * s + 8 * hh);
        nbi[b] = nb[iv[b]];
        k1[b] = k2[b] = 0;
        if (NOISE == 2) rowkey(s0, s1, (uint32_t)iv[b], k1[b], k2[b]);
        ta[b] = NOISE == 0 ? __float_as_uint(3.0e38f) : 0u;      // accept everything until 64 candidates are known
        if (!rvalid[b]) ta[b] = NOISE == 0 ? __float_as_uint(-1.0f) : 0xffffffffu;
        if (hh == 0) { row_nb[b * 32 + r] = nbi[b]; row_tb[b * 32 + r] = rvalid[b] ? -3.0e38f : 3.0e38f; }
    }
    for (int e = lane; e < RW; e += 64) cnt[e] = 0;
    for (int e = lane; e < RW * 64; e += 64) {
        int64_t gi = rbase + (e >> 6);
        if (gi < row1) { idx[(gi - row0) * 64 + (e & 63)] = -1; val[(gi - row0) * 64 + (e & 63)] = 0.0f; }
    }
    int qhead = 0, qtail = 0;                  // wave-uniform ring indices (monotone; slot = index & (QN-1))

    uint4 stg[KS];
    float stg_nb = 0.0f;
    auto tile_load = [&](int64_t c0) {
#pragma unroll
        for (int q = 0; q < KS; q++) {
            int ch = q * 64 + lane;
            int64_t gj = c0 + ch / (H / 8);
            stg[q] = make_uint4(0, 0, 0, 0);
            if (gj < N) stg[q] = *reinterpret_cast<const uint4 *>(xb + c0 * H + (int64_t)ch * 8);
        }
        if (NOISE == 0 && lane < CT) stg_nb = (c0 + lane < N) ? nb[c0 + lane] : 3.0e38f;
    };
    auto tile_store = [&]() {
#pragma unroll
        for (int q = 0; q < KS; q++) {
            int ch = q * 64 + lane;
            int jj = ch / (H / 8), part = ch % (H / 8);
            *reinterpret_cast<uint4 *>(&colA[jj * STRIDE + part * 16]) = stg[q];
        }
        if (NOISE == 0 && lane < CT) nbt[lane] = stg_nb;
    };

#ifdef DGG_STAMPS
    unsigned long long stamp_loc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stamp_t = __builtin_readcyclecounter();
#endif

    // exact scoring + merge of the pending candidates of local row flr (whole wavefront cooperates)
    auto do_flush = [&](int flr) {
        const int64_t fi = rbase + flr;                       // wave-uniform
        const int64_t out_row = fi - row0;
        const int n = __builtin_amdgcn_readfirstlane(cnt[flr]);
        int32_t li = idx[out_row * 64 + lane];
        float lv = val[out_row * 64 + lane];
        uint64_t list = li >= 0 ? make_key(lv, li) : DGG_EMPTY_KEY;
        for (int base = 0; base < n; base += 64) {
            int e = base + lane;
            int32_t j = e < n ? pend[flr * CAP + e] : -1;
            uint64_t key = DGG_EMPTY_KEY;
            if (j >= 0) key = make_key(exact_score<H>(xp, fi, j, t, NOISE, s0, s1), j);
            key = wave_sort<false>(key, lane);                // ascending: merges without a reversal
            list = wave_merge_top64_asc(list, key, lane);
        }
        bool empty = list == DGG_EMPTY_KEY;
        idx[out_row * 64 + lane] = empty ? -1 : key_col(list);
        val[out_row * 64 + lane] = empty ? 0.0f : key_val(list);
        uint64_t k63 = shfl_u64(list, 63);
        if (lane == 0) cnt[flr] = 0;
        if (k63 != DGG_EMPTY_KEY) {                           // uniform
            RowThr th = thresholds_from_score(key_val(k63), t, NOISE != 0);
#pragma unroll
            for (int b = 0; b < RBLK; b++)
                if (flr == b * 32 + r) ta[b] = th.a;
            if (lane == 0) row_tb[flr] = th.b;
        }
        COUNT(5, 1);
    };
    auto flush_ready = [&](int threshold) {
#pragma unroll
        for (int b = 0; b < RBLK; b++) {
            int mycnt = cnt[b * 32 + r];
            uint64_t need = __ballot(mycnt >= threshold && rvalid[b]) & 0xffffffffull;
            while (need) {
                int rr = __builtin_ctzll(need);
                need &= need - 1;
                do_flush(b * 32 + rr);
            }
        }
    };
    // stage B on up to 64 queued survivors, one per lane
    auto drain = [&]() {
        const int n = qtail - qhead;                          // uniform, 1..64 used
        const int e = (qhead + lane) & (QN - 1);
        const bool live = lane < n;
        uint32_t j = qj[e], x = qx[e], lr = ql[e];
        float dot = qa[e];
        qhead += n < 64 ? n : 64;
        if (live) {
            float nj = nb[j];
            float yub = score_upper_bound(dot, row_nb[lr], nj, x, t, NOISE == 3 && (int64_t)j == rbase + lr);
            if (yub >= row_tb[lr]) {
                int slot = atomicAdd(&cnt[lr], 1);
                pend[lr * CAP + slot] = (int)j;
            }
        }
        COUNT(4, 1);
        flush_ready(FLUSH_AT);
    };

    const int ntiles = (int)((N + CT - 1) / CT);
    tile_load(0);
    for (int tl = 0; tl < ntiles; tl++) {
        const int64_t cb = (int64_t)tl * CT;
        __syncthreads();                      // single-wave workgroup: orders the LDS image (previous reads done)
        tile_store();
        __syncthreads();
        if (tl + 1 < ntiles) tile_load(cb + CT);
        bf16x8 af[KS];
#pragma unroll
        for (int s = 0; s < KS; s++) af[s] = *reinterpret_cast<const bf16x8 *>(&colA[r * STRIDE + (16 * s + 8 * hh) * 2]);
        const uint32_t jbase = (uint32_t)cb | (uint32_t)(4 * hh);
        const bool tail_tile = cb + CT > N;   // uniform: some columns of this tile do not exist
#pragma unroll
        for (int b = 0; b < RBLK; b++) {
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 16; q++) acc[q] = 0.0f;
#pragma unroll
            for (int s = 0; s < KS; s++) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s], bfr[b][s], acc, 0, 0, 0);
            const int lr = b * 32 + r;
            STAMP(0);
            {
                const uint32_t base = jbase ^ k1[b];
                unsigned long long pm[16];
                unsigned long long anym = 0ull;
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const uint32_t a = (uint32_t)((q & 3) + 8 * (q >> 2));
                    uint32_t x;
                    if (NOISE == 2) {
                        x = base ^ a;
                        x *= 0x7feb352dU; x ^= x >> 15; x += k2[b]; x *= 0x846ca68bU;
                    } else {
                        const uint32_t j = jbase + a;
                        x = pair_u24(s0, s1, (uint32_t)iv[b], j, true) << 8;
                        if (j == (uint32_t)iv[b]) x = 0xffffffffu;          // zero-noise diagonal: decided in stage B
                    }
                    bool pa = x >= ta[b];
                    if (tail_tile) pa = pa && ((int64_t)(jbase + a) < N);
                    pm[q] = __ballot(pa);
                    anym |= pm[q];
                }
                STAMP(1);
                if (anym != 0ull) {
#pragma unroll
                    for (int q = 0; q < 16; q++) {
                        const unsigned long long m = pm[q];
                        if (m == 0ull) continue;                             // wave-uniform
                        const uint32_t a = (uint32_t)((q & 3) + 8 * (q >> 2));
                        const uint32_t j = jbase + a;
                        if ((m >> lane) & 1ull) {
                            uint32_t x;
                            if (NOISE == 2) {
                                x = base ^ a;
                                x *= 0x7feb352dU; x ^= x >> 15; x += k2[b]; x *= 0x846ca68bU;
                            } else {
                                x = pair_u24(s0, s1, (uint32_t)iv[b], j, true) << 8;
                            }
                            const int pos = (qtail + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32),
                                                          __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))) & (QN - 1);
                            qj[pos] = j; qx[pos] = x; ql[pos] = (uint32_t)lr; qa[pos] = acc[q];
                        }
                        qtail += __builtin_popcountll(m);
                        if (qtail - qhead >= 64) { STAMP(2); drain(); STAMP(3); }
                    }
                }
                STAMP(2);
            }
        }
    }
    // drain the ring, then flush everything still pending
    while (qtail != qhead) drain();
    flush_ready(1);
#ifdef DGG_STAMPS
    STAMP(3);
    if (lane == 0)
        for (int q = 0; q < 8; q++) atomicAdd(&dgg_stamp_acc[q], stamp_loc[q]);
#endif
}

constexpr int RBLK_DEFAULT = 2;
template <int H>
int launch_fast(const float *xp, int64_t N, int64_t row0, int64_t row1, float t, int noise_mode, uint32_t s0, uint32_t s1,
                int32_t *idx, float *val, void *ws, hipStream_t st) {
    constexpr int RBLK = RBLK_DEFAULT;
    __bf16 *xb = reinterpret_cast<__bf16 *>(ws);
    size_t off = (((size_t)N * H * 2 + 255) / 256) * 256;
    float *nb = reinterpret_cast<float *>(reinterpret_cast<char *>(ws) + off);
    off += (((size_t)N * 4 + 255) / 256) * 256;
    int *pend = reinterpret_cast<int *>(reinterpret_cast<char *>(ws) + off);
    const size_t rows64 = ((size_t)(row1 - row0) + 127) / 128 * 128;
    off += rows64 * CAP * 4;
    FastCtl *ctl = reinterpret_cast<FastCtl *>(reinterpret_cast<char *>(ws) + off);
    int *faillist = reinterpret_cast<int *>(reinterpret_cast<char *>(ws) + off + 256);
    if (dgg_check_hip(hipMemsetAsync(ctl, 0, sizeof(FastCtl), st), "fast memset") != 0) return DGG_ERR_HIP;
    hipLaunchKernelGGL(prep_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, st, xp, N, H, xb, nb);
    dim3 grid((unsigned)((row1 - row0 + 32 * RBLK - 1) / (32 * RBLK)));
    if (noise_mode == 0) {
        return dgg_set_error(DGG_ERR_UNSUPPORTED, "unperturbed scores: dgg_topk_sweep.hip (N >= 8192) or the exhaustive kernel");
    } else if (noise_mode == 2)
        hipLaunchKernelGGL((allpairs_topk_fast<H, 2, RBLK>), grid, dim3(64), 0, st, xp, xb, nb, pend, N, row0, row1, t, s0, s1, idx, val, ctl,
                           faillist, 1.0f, nullptr, nullptr, nullptr);
    else
        hipLaunchKernelGGL((allpairs_topk_fast<H, 3, RBLK>), grid, dim3(64), 0, st, xp, xb, nb, pend, N, row0, row1, t, s0, s1, idx, val, ctl,
                           faillist, 1.0f, nullptr, nullptr, nullptr);
    return dgg_check_launch("allpairs_topk_fast");
}

}  // namespace

#ifdef DGG_STAMPS
extern "C" int dgg_debug_read_stamps(unsigned long long *out8, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out8, HIP_SYMBOL(dgg_stamp_acc), 64);
    if (e == hipSuccess && reset) {
        unsigned long long z[8] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(dgg_stamp_acc), z, 64);
    }
    return e == hipSuccess ? 0 : 3;
}
#endif

size_t dgg_allpairs_fast_ws_bytes(int64_t N, int h) {
    size_t rows = ((size_t)N + 127) / 128 * 128;
    return (((size_t)N * h * 2 + 255) / 256) * 256 + (((size_t)N * 4 + 255) / 256) * 256 + rows * CAP * 4 + 256 + rows * 4;
}

bool dgg_allpairs_fast_supported(int h, int noise_mode, int K) {
    return K == 64 && (h == 16 || h == 32 || h == 64 || h == 128) && noise_mode >= 2;   // (unperturbed scores: dgg_topk_sweep.hip)
}

int dgg_allpairs_topk_fast_impl(const float *xp, int64_t N, int h, int64_t row0, int64_t row1, float t, int noise_mode,
                                uint32_t s0, uint32_t s1, int K, int32_t *idx, float *val, void *workspace, size_t ws_bytes,
                                hipStream_t st) {
    if (!dgg_allpairs_fast_supported(h, noise_mode, K))
        return dgg_set_error(DGG_ERR_UNSUPPORTED, "pruned all-pairs path needs K=64, latent_dim in {16,32,64,128}, in-kernel noise");
    if (!workspace || ws_bytes < dgg_allpairs_fast_ws_bytes(N, h))
        return dgg_set_error(DGG_ERR_ARG, "pruned all-pairs path: workspace too small (dgg_allpairs_workspace_bytes)");
    if (row1 <= row0) return 0;
    switch (h) {
        case 16: return launch_fast<16>(xp, N, row0, row1, t, noise_mode, s0, s1, idx, val, workspace, st);
        case 32: return launch_fast<32>(xp, N, row0, row1, t, noise_mode, s0, s1, idx, val, workspace, st);
        case 64: return launch_fast<64>(xp, N, row0, row1, t, noise_mode, s0, s1, idx, val, workspace, st);
        default: return launch_fast<128>(xp, N, row0, row1, t, noise_mode, s0, s1, idx, val, workspace, st);
    }
}
