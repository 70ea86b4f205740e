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

// Unperturbed scores: the radius of a row's 64 nearest neighbours is GUESSED from a pilot (PILOT_T sampled column tiles: the
// larger of the two half-rows' PILOT_M-th smallest bf16 distance bounds, i.e. about the 16th smallest of 5120 sampled columns,
// which admits a few hundred of the N columns with a relative spread of ~25 %) instead of starting at infinity, where the branch-and-bound would exact-score ~64 (1 + ln(N/64))
// = 535 candidates per row at N = 100k before its threshold has converged (that, not the MFMA sweep, was 80 % of the kernel).
// A guess that turns out too tight is DETECTED (fewer than 64 entries, or a 64th distance beyond the guess) and the row is
// redone without pruning by topk_fast_fallback, so the result stays exact.
struct FastCtl {
    int nfail;
    int pad[3];
};
constexpr int PILOT_T = 160;         // sampled column tiles (5120 columns)
constexpr int PILOT_M = 8;           // order statistic kept per half-row: the radius covers >= 2 * PILOT_M samples
constexpr int TAU_SAMPLE = 256;     // candidates fast_finalize looks at for its pruning threshold
constexpr int CAPF = 2048;          // candidate slots per row of the unperturbed two-phase path (expected ~400 at the pilot's radius)

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

// ---- unperturbed scores: dedicated candidate sweep ---------------------------------------------------------------------------
// The branch-and-bound kernel above is built around wavefronts that stop to flush; without flushes (the unperturbed path
// collects candidates at the pilot's radius and settles them in fast_finalize) nothing stops, so the sweep is restructured as
// a plain tiled kernel: a workgroup of 4 wavefronts owns 128 rows (one 32-row MFMA block per wavefront) and streams ALL
// columns in tiles of 128, staged ONCE per workgroup through a double-buffered, padded LDS image (register prefetch of the next
// tile during the MFMAs of the current one, one barrier per tile); per 32x32 Gram block: 4 bf16 MFMAs, 16 bounds per lane,
// ONE compare of their minimum against the row's radius.  782 workgroups at N = 100k: three per CU.
constexpr int NPC = 128;           // columns per staged tile
template <int H>
__global__ __launch_bounds__(256, H <= 64 ? 4 : 2) void np_sweep(const __bf16 *__restrict__ xb, const float *__restrict__ nb, int64_t N, int64_t row0,
                                                int64_t row1, float gscale, int2 *__restrict__ cand, int *__restrict__ cand_cnt,
                                                float *__restrict__ cand_guess) {
    constexpr int KS = H / 16, STRIDE = H * 2 + 16, CPT = H / 8;     // 16-byte chunks per column
    constexpr int LQ = NPC * CPT / 256;                              // chunks per thread and tile
    __shared__ __attribute__((aligned(16))) unsigned char colA[2][NPC * STRIDE];
    __shared__ __attribute__((aligned(16))) float nbt[2][NPC];
    const int tid = threadIdx.x, lane = tid & 63, wave = dgg::wave_id(), r = lane & 31, hh = lane >> 5;
    const int lr = wave * 32 + r;                                    // row of this lane inside the workgroup
    const int64_t i = row0 + (int64_t)blockIdx.x * 128 + lr;
    const bool rvalid = i < row1;
    const int64_t ic = rvalid ? i : row1 - 1;
    bf16x8 bfr[KS];
#pragma unroll
    for (int s = 0; s < KS; s++) bfr[s] = *reinterpret_cast<const bf16x8 *>(xb + ic * H + 16 * s + 8 * hh);
    const float nbi = nb[ic];
    uint4 stg[LQ];                                                   // (plain arrays: a struct passed by reference to the lambdas is not
    float stg_nb = 0.0f;                                             //  promoted to registers and round-trips through scratch memory)
    // (zero + conditional load: the unconditional clamped form makes the compiler keep `stg` in scratch memory -- 80 bytes per lane)
    auto tile_load = [&](int64_t c0) {
#pragma unroll
        for (int q = 0; q < LQ; q++) {
            const int ch = q * 256 + tid;
            stg[q] = make_uint4(0, 0, 0, 0);
            if (c0 + ch / CPT < N) stg[q] = *reinterpret_cast<const uint4 *>(xb + c0 * H + (int64_t)ch * 8);
        }
        if (tid < NPC) stg_nb = (c0 + tid < N) ? nb[c0 + tid] : 3.0e38f;
    };
    auto tile_store = [&](int buf) {
#pragma unroll
        for (int q = 0; q < LQ; q++) {
            const int ch = q * 256 + tid;
            *reinterpret_cast<uint4 *>(&colA[buf][(ch / CPT) * STRIDE + (ch % CPT) * 16]) = stg[q];
        }
        if (tid < NPC) nbt[buf][tid] = stg_nb;
    };
    // bounds of one 32-column block of the staged tile against this wavefront's 32 rows
    auto bounds = [&](int buf, int sub, float (&L2)[16]) {
        bf16x8 af[KS];
#pragma unroll
        for (int s = 0; s < KS; s++) af[s] = *reinterpret_cast<const bf16x8 *>(&colA[buf][(sub * 32 + r) * STRIDE + (16 * s + 8 * hh) * 2]);
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; q++) acc[q] = 0.0f;
#pragma unroll
        for (int s = 0; s < KS; s++) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s], bfr[s], acc, 0, 0, 0);
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) {
            const float4 nj4 = *reinterpret_cast<const float4 *>(&nbt[buf][sub * 32 + 8 * g4 + 4 * hh]);
            const float njs[4] = {nj4.x, nj4.y, nj4.z, nj4.w};
#pragma unroll
            for (int u = 0; u < 4; u++) L2[4 * g4 + u] = __fmaf_rn(-2.0f, acc[4 * g4 + u], nbi + njs[u]);
        }
    };
    const int ntiles = (int)((N + NPC - 1) / NPC);
    // ---- pilot: PILOT_T / 4 tiles spread over the column range (per-workgroup offset) -> radius guess (see allpairs_topk_fast)
    float guess = 3.0e38f;
    constexpr int PT = PILOT_T / 4;
    if (ntiles >= 2 * PT) {
        float tm[PILOT_M];
#pragma unroll
        for (int q = 0; q < PILOT_M; q++) tm[q] = 3.0e38f;
        const int stride_t = ntiles / PT;
        const int first_t = (int)(((uint32_t)blockIdx.x * 2654435761u) % (uint32_t)stride_t);
        for (int pt = 0; pt < PT; pt++) {
            tile_load((int64_t)(first_t + pt * stride_t) * NPC);
            __syncthreads();
            tile_store(0);
            __syncthreads();
#pragma unroll
            for (int sub = 0; sub < NPC / 32; sub++) {
                float L2[16];
                bounds(0, sub, L2);
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    float v = L2[q];
                    if (v < tm[PILOT_M - 1]) {
#pragma unroll
                        for (int m = 0; m < PILOT_M; m++) { const float lo = fminf(v, tm[m]); v = fmaxf(v, tm[m]); tm[m] = lo; }
                    }
                }
            }
        }
        const float other = __shfl_xor(tm[PILOT_M - 1], 32, 64);
        guess = fmaxf(fmaxf(tm[PILOT_M - 1], other), 0.0f) * gscale + 1e-6f;
        __syncthreads();
    }
    // Candidate appends: the two lanes of a row (columns 4*hh.. of every 8) each own HALF of the row's list and a private counter
    // in a register.  (A shared LDS counter per row -- atomicAdd with return -- put one LDS round trip on every append site a
    // wavefront entered: ~16 sites per tile, most of the 9000 cycles a tile took; 5.9 -> 5.3 ms.)
    int2 *cl = cand + ((int64_t)blockIdx.x * 128 + lr) * CAPF + hh * (CAPF / 2);
    int mycnt = 0;
    // ---- sweep  (measured and rejected: two register stages, i.e. the loads of tile tl + 3 in flight while tile tl + 1 waits
    // in registers -- 7.0 ms against 5.3)
    // Candidate test.  A lane holds 16 bounds per 32x32 block and a wavefront ~4 hits among its 1024: testing the 16 positions
    // one by one costs a compare + divergent branch each, and the wavefront enters whenever ANY lane hits (5.3 -> 4.4 ms).
    // The loop is VALU-bound: ~570 dynamic VALU instructions per wavefront and 128-column tile (16 fma + 16 tags + 32 min/med3
    // per block, the append paths, the rescans), 4 wavefronts per SIMD; the staging skeleton alone takes 0.9 ms.  Instead the column position q is written into the 4 low mantissa bits of each bound (a slightly SMALLER bound:
    // still a lower bound) and the lane's two smallest tagged bounds come out of a min / med3 chain: one test per block for
    // the first hit, one for the second (12 % of blocks for some lane), and only a lane with two hits rescans its 16 values.
    const float rad2f = rvalid ? __int_as_float(__float_as_int(guess) | 15) : -INFINITY;
    auto append = [&](float tagged, uint32_t colbase) {
        const int bits = __float_as_int(tagged), q = bits & 15;
        if (mycnt < CAPF / 2) cl[mycnt] = make_int2((int)(colbase + (uint32_t)((q & 3) + 8 * (q >> 2))), bits & ~15);
        mycnt++;
    };
    tile_load(0);
    tile_store(0);
    __syncthreads();
    for (int tl = 0; tl < ntiles; tl++) {
        const int buf = tl & 1;
        if (tl + 1 < ntiles) tile_load((int64_t)(tl + 1) * NPC);         // in flight during the MFMAs below
        const uint32_t cbase = (uint32_t)tl * NPC + (uint32_t)(4 * hh);
#pragma unroll
        for (int sub = 0; sub < NPC / 32; sub++) {
            float vb[16];
            bounds(buf, sub, vb);
            float m1 = INFINITY, m2 = INFINITY;                      // the two smallest tagged bounds (m1 <= m2)
#pragma unroll
            for (int q = 0; q < 16; q++) {
                vb[q] = __int_as_float((__float_as_int(vb[q]) & ~15) | q);
                m2 = __builtin_amdgcn_fmed3f(m1, m2, vb[q]);
                m1 = fminf(m1, vb[q]);
            }
            if (__ballot(m1 <= rad2f) != 0ull) {                     // wave-uniform
                const uint32_t colbase = cbase + (uint32_t)(sub * 32);
                if (m1 <= rad2f) append(m1, colbase);
                if (__ballot(m2 <= rad2f) != 0ull) {
                    if (m2 <= rad2f) {
                        append(m2, colbase);
                        const int q1 = __float_as_int(m1) & 15, q2 = __float_as_int(m2) & 15;
#pragma unroll
                        for (int q = 0; q < 16; q++)
                            if (q != q1 && q != q2 && vb[q] <= rad2f) append(vb[q], colbase);
                    }
                }
            }
        }
        if (tl + 1 < ntiles) tile_store(buf ^ 1);
        __syncthreads();
    }
    const int c1 = __shfl_xor(mycnt, 32, 64);                    // the other half's count
    if (rvalid && hh == 0) {                                     // packed (low 16 bits: half 0, high: half 1; 0xffff = overflow)
        const int n0 = mycnt <= CAPF / 2 ? mycnt : 0xffff, n1 = c1 <= CAPF / 2 ? c1 : 0xffff;
        cand_cnt[(int64_t)blockIdx.x * 128 + lr] = n0 | (n1 << 16);
        cand_guess[(int64_t)blockIdx.x * 128 + lr] = guess;
    }
}

// settle one row's candidate list (unperturbed scores): (1) tau = 64th smallest UPPER bound of d^2 over the candidates
// (U = L + 2 eps (n_i + n_j) with L the stored bf16 lower bound): at least 64 candidates have d^2 <= tau, so a candidate with
// L > tau cannot be among the 64 nearest; (2) exact canonical score of the survivors (typically 70-100 of ~400), sorted and
// merged; (3) verification of the pilot's guess: the list is exact iff it is full and its 64th exact distance lies inside
// the guessed radius (every pair the sweep pruned had d^2 >= L > guess).  Rows that fail go to the fail list.
template <int H>
__global__ __launch_bounds__(256) void fast_finalize(const float *__restrict__ xp, const float *__restrict__ nb, int64_t N, int64_t row0,
                                                     int64_t row1, float t, const int2 *__restrict__ cand, const int *__restrict__ cand_cnt,
                                                     const float *__restrict__ cand_guess, FastCtl *__restrict__ ctl,
                                                     int *__restrict__ faillist, int32_t *__restrict__ idx, float *__restrict__ val) {
    const int lane = threadIdx.x & 63;
    const int64_t lrow = (int64_t)blockIdx.x * 4 + dgg::wave_id();
    const int64_t i = row0 + lrow;
    if (i >= row1) return;
    const int packed = cand_cnt[lrow], n0 = packed & 0xffff, n1 = (packed >> 16) & 0xffff;   // two half lists (np_sweep)
    const int n = n0 + n1;
    const float guess = cand_guess[lrow];
    const int2 *cl0 = cand + lrow * CAPF;
    auto cand_at = [&](int e) { return cl0[e < n0 ? e : CAPF / 2 + (e - n0)]; };
    bool ok = n0 <= CAPF / 2 && n1 <= CAPF / 2 && n >= 64;
    uint64_t list = DGG_EMPTY_KEY;
    if (ok) {
        const float ni = nb[i];                                  // discounted norms: n (1 - eps)
        constexpr float SL = 2.0f * EPS_BF16 / (1.0f - EPS_BF16) * 1.0001f;
        // (1) 64 smallest upper bounds: keys ordered by the COMPLEMENT of the bound's bits (bounds clamped at 0: bit-monotone)
        // (any subset gives a valid tau -- 64 of ITS upper bounds lie below it: the first 256 candidates cost four sort + merge
        //  rounds instead of seven and leave ~20 more survivors for stage 2)
        uint64_t ub = DGG_EMPTY_KEY;
        const int n1s = n < TAU_SAMPLE ? n : TAU_SAMPLE;
        for (int base = 0; base < n1s; base += 64) {
            const int e = base + lane;
            uint64_t key = DGG_EMPTY_KEY;
            if (e < n1s) {
                const int2 c = cand_at(e);
                const float L = __int_as_float(c.y);
                const float U = fmaxf(L + SL * (ni + nb[c.x]), 0.0f) * 1.00001f + 1e-7f;
                key = ((uint64_t)(~__float_as_uint(U)) << 32) | (uint32_t)(e + 1);
            }
            key = wave_sort<false>(key, lane);
            ub = wave_merge_top64_asc(ub, key, lane);
        }
        const float tau = __uint_as_float(~(uint32_t)(shfl_u64(ub, 63) >> 32));
        // (2) exact scores of the candidates whose lower bound can still reach tau
        for (int base = 0; base < n; base += 64) {
            const int e = base + lane;
            int32_t j = -1;
            if (e < n) {
                const int2 c = cand_at(e);
                if (__int_as_float(c.y) <= tau) j = c.x;
            }
            if (__ballot(j >= 0) == 0ull) continue;
            uint64_t key = DGG_EMPTY_KEY;
            if (j >= 0) key = make_key(exact_score<H>(xp, i, j, t, 0, 0u, 0u), j);
            key = wave_sort<false>(key, lane);
            list = wave_merge_top64_asc(list, key, lane);
        }
        // (3) verify the guess
        const uint64_t k63 = shfl_u64(list, 63);
        if (k63 == DGG_EMPTY_KEY) ok = false;
        else if (guess < 1.0e38f) {
            const float d63 = __logf(fmaxf(key_val(k63), 1e-37f)) / t;
            ok = d63 * d63 * (1.0f + 1e-4f) + 1e-6f <= guess;
        }
    }
    if (ok) {
        idx[lrow * 64 + lane] = key_col(list);
        val[lrow * 64 + lane] = key_val(list);
    } else if (lane == 0) {
        faillist[atomicAdd(&ctl->nfail, 1)] = (int)lrow;
    }
}

// rows whose guessed radius failed verification: redone from scratch, one workgroup per row, LANE = COLUMN, every column scored
// exactly (25 MB of gathers per row at N = 100k; a handful of rows)
template <int H>
__global__ __launch_bounds__(256) void topk_fast_fallback(const float *__restrict__ xp, int64_t N, int64_t row0, float t,
                                                         const FastCtl *__restrict__ ctl, const int *__restrict__ faillist,
                                                         int32_t *__restrict__ idx, float *__restrict__ val) {
    __shared__ uint64_t lists[4][64];
    const int lane = threadIdx.x & 63, wave = dgg::wave_id();
    const int nfail = ctl->nfail;
    for (int f = blockIdx.x; f < nfail; f += gridDim.x) {
        const int lrow = faillist[f];
        const int64_t i = row0 + lrow;
        uint64_t list = DGG_EMPTY_KEY;
        for (int64_t j0 = (int64_t)wave * 64; j0 < N; j0 += 256) {
            const int64_t j = j0 + lane;
            uint64_t key = DGG_EMPTY_KEY;
            if (j < N) key = make_key(exact_score<H>(xp, i, (int32_t)j, t, 0, 0u, 0u), (int32_t)j);
            key = wave_sort<false>(key, lane);
            list = wave_merge_top64_asc(list, key, lane);
        }
        lists[wave][lane] = list;
        __syncthreads();
        if (wave == 0) {
            for (int w = 1; w < 4; w++) list = wave_merge_top64_asc(list, wave_sort<false>(lists[w][lane], lane), lane);
            const bool empty = list == DGG_EMPTY_KEY;
            idx[(int64_t)lrow * 64 + lane] = empty ? -1 : key_col(list);
            val[(int64_t)lrow * 64 + lane] = empty ? 0.0f : key_val(list);
        }
        __syncthreads();
    }
}

constexpr int RBLK_DEFAULT = 2;
// headroom factor on the pilot's radius guess; tests shrink it (DGG_FAST_GUESS_SCALE) to force the verification / fallback path
static float g_fast_guess_scale = [] {
    const char *e = getenv("DGG_FAST_GUESS_SCALE");
    return e ? (float)atof(e) : 1.02f;
}();

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
        // two-phase: sweep (candidate lists at the pilot's radius) -> finalize (one wavefront per row) -> fallback (failed rows)
        char *w2 = reinterpret_cast<char *>(faillist) + rows64 * 4;
        int *cand_cnt = reinterpret_cast<int *>(w2);
        float *cand_guess = reinterpret_cast<float *>(w2 + rows64 * 4);
        int2 *cand = reinterpret_cast<int2 *>(w2 + 2 * rows64 * 4);
        hipLaunchKernelGGL(np_sweep<H>, dim3((unsigned)((row1 - row0 + 127) / 128)), dim3(256), 0, st, xb, nb, N, row0, row1, g_fast_guess_scale,
                           cand, cand_cnt, cand_guess);
        hipLaunchKernelGGL(fast_finalize<H>, dim3((unsigned)((row1 - row0 + 3) / 4)), dim3(256), 0, st, xp, nb, N, row0, row1, t, cand, cand_cnt,
                           cand_guess, ctl, faillist, idx, val);
        hipLaunchKernelGGL(topk_fast_fallback<H>, dim3(256), dim3(256), 0, st, xp, N, row0, t, ctl, faillist, idx, val);
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
    return (((size_t)N * h * 2 + 255) / 256) * 256 + (((size_t)N * 4 + 255) / 256) * 256 + rows * CAP * 4 + 256 + rows * 4 +
           2 * rows * 4 + rows * (size_t)CAPF * sizeof(int2);    // + counts, guesses and candidate lists of the unperturbed path
}

bool dgg_allpairs_fast_supported(int h, int noise_mode, int K) {
    return K == 64 && (h == 16 || h == 32 || h == 64 || h == 128) && noise_mode != 1;
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
