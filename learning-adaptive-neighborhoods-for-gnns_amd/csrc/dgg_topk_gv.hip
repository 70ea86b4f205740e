// dgg_topk_gv.hip -- "guess and verify": the fastest exact all-pairs top-64 for PERTURBED scores.
//
// Same contract and same bits as allpairs_topk_exhaustive (dgg_topk.hip) / allpairs_topk_np (dgg_topk_np.hip):
//   p'_ij = exp(log(exp(-0.05 ||xp_i - xp_j||) + 1e-8) + G_ij),  64 largest per row     reference dgm.py:1618-1623,
//                                                                                         1213-1229, 1404
// The adaptive noise prefilter (dgg_topk_np.hip) learns each row's threshold while it sweeps, which costs ~1200
// exactly-scored candidates and ~20 merge stops per row.  Here the threshold is GUESSED up front and the result
// VERIFIED afterwards, so the sweep is a pure integer loop and only ~230 candidates per row are scored:
//
//   K0 pilot     262k random pairs -> M = mean p^(1/0.3): for a threshold v, the expected number of pairs of a row
//                with log-score >= v is  N * M * exp(-v/0.3)  (Gumbel tail); solve for a count of 128 -> gmin0.
//   K1 sweep     lane = row, column wave-uniform: keep pairs whose noise alone could reach gmin0 (one unsigned
//                compare of the raw hash), append their column to the row's candidate list.  No features touched.
//   K2 finalize  one wavefront per row: exact canonical score of the candidates (64 per pass), DPP bitonic sort/merge
//                -> top-64.  VERIFY: every rejected pair has log-score < gmin0 + 1e-8, so the list is exact iff its
//                64th log-score >= gmin0 + margin.  Rows that fail (too few / too many candidates, or an unusually
//                distant node) are appended to a fail list ...
//   K3 fallback  ... and redone by the adaptive kernel (row-list form), which needs no guess.
// All four launches are asynchronous on one stream; nothing is read back by the host.
#include "dgg_common.h"
#include "dgg_api_internal.h"

using namespace dgg;

namespace {

constexpr int NSEG = 4;            // column segments swept by separate wavefronts (fixed threshold: no coupling)
constexpr int CAPS = 160;          // candidate slots per (row, segment)
constexpr int CAPF = NSEG * CAPS;  // candidate slots per row in the fixed-threshold sweep
constexpr float TARGET_MAX = 128.0f;   // expected number of pairs per row above the guessed threshold (need 64) ...
constexpr float TARGET_MIN = 80.0f;    // ... lowered when distances matter (small M) so the candidate lists still fit
constexpr float ADMIT_MAX = 0.56f * CAPF;  // pairs the noise filter may admit per row on average (358): a segment's 160 slots see 90 (7 sigma of
                                           // headroom) and the transposed list of the triangular sweep -- 448 slots, up to ALL of a late row's
                                           // candidates -- 358 (4.7 sigma).  (0.8 until round 6: 2.8 sigma per segment, 39 rows of 100 000
                                           // overflowed on N(0, 0.7) features, and the last rows' transposed lists could not hold their share.)
constexpr int PILOT_PAIRS = 262144;
constexpr float DTIGHT = 0.1124f;      // 0.3 ln(128 / 88): gv_finalize's first stage aims at an expected 88 pairs above its threshold

struct GvCtl {                     // device-side control block (workspace head)
    float msum;                    // sum over pilot pairs of p^(1/0.3)
    int nfail;                     // rows that failed verification
    float gmin0;                   // guessed log-score threshold (written by K1's first wave for K2)
    int pad;
};

template <int H>
__device__ __forceinline__ float exact_score_gv(const float *__restrict__ xp, int64_t i, int32_t j, float t, bool sym,
                                                uint32_t s0, uint32_t s1) {
    const float *xi = xp + i * H;                               // wave-uniform: scalar loads
    const float4 *xj = reinterpret_cast<const float4 *>(xp + (int64_t)j * H);
    float d2 = 0.0f;
#pragma unroll
    for (int c8 = 0; c8 < H / 8; c8++) {
        float4 b0 = xj[2 * c8], b1 = xj[2 * c8 + 1];
        float df;
        df = __fadd_rn(xi[8 * c8 + 0], -b0.x); d2 = __fmaf_rn(df, df, d2);
        df = __fadd_rn(xi[8 * c8 + 1], -b0.y); d2 = __fmaf_rn(df, df, d2);
        df = __fadd_rn(xi[8 * c8 + 2], -b0.z); d2 = __fmaf_rn(df, df, d2);
        df = __fadd_rn(xi[8 * c8 + 3], -b0.w); d2 = __fmaf_rn(df, df, d2);
        df = __fadd_rn(xi[8 * c8 + 4], -b1.x); d2 = __fmaf_rn(df, df, d2);
        df = __fadd_rn(xi[8 * c8 + 5], -b1.y); d2 = __fmaf_rn(df, df, d2);
        df = __fadd_rn(xi[8 * c8 + 6], -b1.z); d2 = __fmaf_rn(df, df, d2);
        df = __fadd_rn(xi[8 * c8 + 7], -b1.w); d2 = __fmaf_rn(df, df, d2);
    }
    float dist = c_sqrt(d2);
    float g = pair_noise(s0, s1, (uint32_t)i, (uint32_t)j, sym);
    return score_from_dist(dist, t, true, g);
}
// the same with the pair's 24-bit uniform already at hand (gv_finalize hashes every candidate before it decides to gather it)
template <int H>
__device__ __forceinline__ float exact_score_gv_u(const float *__restrict__ xp, int64_t i, int32_t j, float t, bool diag, uint32_t u24) {
    const float *xi = xp + i * H;
    const float4 *xj = reinterpret_cast<const float4 *>(xp + (int64_t)j * H);
    float d2 = 0.0f;
#pragma unroll
    for (int c8 = 0; c8 < H / 8; c8++) {
        float4 b0 = xj[2 * c8], b1 = xj[2 * c8 + 1];
        float df;
        df = __fadd_rn(xi[8 * c8 + 0], -b0.x); d2 = __fmaf_rn(df, df, d2);
        df = __fadd_rn(xi[8 * c8 + 1], -b0.y); d2 = __fmaf_rn(df, df, d2);
        df = __fadd_rn(xi[8 * c8 + 2], -b0.z); d2 = __fmaf_rn(df, df, d2);
        df = __fadd_rn(xi[8 * c8 + 3], -b0.w); d2 = __fmaf_rn(df, df, d2);
        df = __fadd_rn(xi[8 * c8 + 4], -b1.x); d2 = __fmaf_rn(df, df, d2);
        df = __fadd_rn(xi[8 * c8 + 5], -b1.y); d2 = __fmaf_rn(df, df, d2);
        df = __fadd_rn(xi[8 * c8 + 6], -b1.z); d2 = __fmaf_rn(df, df, d2);
        df = __fadd_rn(xi[8 * c8 + 7], -b1.w); d2 = __fmaf_rn(df, df, d2);
    }
    return score_from_dist(c_sqrt(d2), t, true, diag ? 0.0f : gumbel_u24(u24));
}

// pairs whose raw hash is below the result have noise G < gmin (with margin): they cannot reach log-score gmin + 1e-8
__device__ __forceinline__ uint32_t hash_threshold_from_gmin(float gmin) {
    float e1 = __expf((gmin - 1e-3f) * (-1.0f / 0.3f));        // P(G >= gmin - 1e-3) = 1 - exp(-e1) <= e1
    float c = fminf(e1 * 16777216.0f, 16777216.0f);
    int um = 16777216 - (int)c - 2;
    um = um < 0 ? 0 : um;
    return (uint32_t)um << 8;
}

// K0: moment of the score distribution from random pairs.  PILOT_PER_THREAD pairs per thread, one atomic per WORKGROUP (same-address
// atomics serialise at ~15 ns each: one per wavefront of 1024 workgroups was 60 us of a kernel with 5 us of work)
constexpr int PILOT_PER_THREAD = 4;
template <int H>
__global__ __launch_bounds__(256) void gv_pilot(const float *__restrict__ xp, int64_t N, float t, uint32_t s0, uint32_t s1,
                                                GvCtl *ctl) {
    __shared__ float part[4];
    float m = 0.0f;
    for (int q = 0; q < PILOT_PER_THREAD; q++) {
        const uint32_t gid = (blockIdx.x * PILOT_PER_THREAD + q) * 256 + threadIdx.x;
        uint32_t a = mix32(gid * 2u + 1u + s0), b = mix32(gid * 2u + 2u + s1 * 0x9E3779B9u);
        int64_t i = (int64_t)(((uint64_t)a * (uint64_t)N) >> 32), j = (int64_t)(((uint64_t)b * (uint64_t)N) >> 32);
        float d2 = 0.0f;
        for (int c = 0; c < H; c++) { float df = xp[i * H + c] - xp[j * H + c]; d2 = fmaf(df, df, d2); }
        float lp = __logf(__expf(t * sqrtf(d2)) + 1e-8f);
        m += (i == j) ? 0.0f : __expf(lp * (1.0f / 0.3f));
    }
    const int lane = threadIdx.x & 63;
    m = wave_sum_dpp(m, lane);
    if (lane == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&ctl->msum, part[0] + part[1] + part[2] + part[3]);
}

// row keys of every node (symmetric noise: a pair below the diagonal is keyed by its COLUMN, dgm.py:1216-1223): computed once
// instead of two mix32 per column and per wavefront on the scalar unit
__global__ void gv_rowkeys(int64_t N, uint32_t s0, uint32_t s1, uint2 *__restrict__ keys) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N) return;
    uint32_t k1, k2;
    rowkey(s0, s1, (uint32_t)j, k1, k2);
    keys[j] = make_uint2(k1, k2);
}

// K1: fixed-threshold sweep.  lane = row; pend/cnt are indexed by LOCAL row (i - row0)
template <bool SYM>
__global__ __launch_bounds__(64) void gv_sweep(int64_t N, int64_t row0, int64_t row1, uint32_t s0, uint32_t s1,
                                               GvCtl *ctl, int *__restrict__ pend_g, int *__restrict__ cnt_g,
                                               const uint2 *__restrict__ colkeys) {
    const int lane = threadIdx.x;
    const int64_t i = row0 + (int64_t)blockIdx.x * 64 + lane;
    const bool rvalid = i < row1;
    const uint32_t iu = (uint32_t)(rvalid ? i : row1 - 1);
    const int64_t lrow = (int64_t)blockIdx.x * 64 + lane;
    const int seg = blockIdx.y;                                  // this wavefront sweeps columns [c_lo, c_hi)
    const int64_t per = ((N + NSEG - 1) / NSEG + 15) / 16 * 16;
    const int64_t c_lo = seg * per < N ? seg * per : N, c_hi = c_lo + per < N ? c_lo + per : N;
    int *pend = pend_g + lrow * CAPF + seg * CAPS;
    // guessed threshold: expected TARGET pairs per row with log-score above it
    const float M = fmaxf(ctl->msum * (1.0f / PILOT_PAIRS), 1e-30f);
    // the noise test admits TARGET / M pairs per row: keep that below the list capacity when M is small
    const float target = fminf(TARGET_MAX, fmaxf(TARGET_MIN, ADMIT_MAX * M));
    const float gmin0 = 0.3f * __logf(fmaxf((float)N * M / target, 1e-30f));
    if (blockIdx.x == 0 && seg == 0 && lane == 0) ctl->gmin0 = gmin0;
    const uint32_t ta = rvalid ? hash_threshold_from_gmin(gmin0) : 0xffffffffu;
    uint32_t k1, k2;
    rowkey(s0, s1, iu, k1, k2);
    int cnt = 0;
    auto col_step = [&](uint32_t j) {                            // j: wave-uniform column
        uint32_t x;
        if (!SYM) {
            x = j ^ k1;
            x *= 0x7feb352dU; x ^= x >> 15; x += k2; x *= 0x846ca68bU;
        } else {
            uint32_t kj1, kj2;
            rowkey(s0, s1, j, kj1, kj2);                         // scalar ALU
            uint32_t xa = pair_u24_keyed(k1, k2, j) << 8;
            uint32_t xb = pair_u24_keyed(kj1, kj2, iu) << 8;
            x = j > iu ? xa : xb;
            if (j == iu) x = 0xffffffffu;                        // zero-noise diagonal: always a candidate
        }
        if (x >= ta) {
            if (cnt < CAPS) pend[cnt] = (int)j;
            cnt++;
        }
    };
    constexpr int UB = 16;                                       // columns hashed together: independent chains (ILP)
    const int64_t NB = c_lo + (c_hi - c_lo) / UB * UB;
    if (!SYM) {
        for (int64_t j0 = c_lo; j0 < NB; j0 += UB) {
            uint32_t x[UB];
#pragma unroll
            for (int u = 0; u < UB; u++) {
                uint32_t v = (uint32_t)(j0 + u) ^ k1;
                v *= 0x7feb352dU; v ^= v >> 15; v += k2; v *= 0x846ca68bU;
                x[u] = v;
            }
#pragma unroll
            for (int u = 0; u < UB; u++) asm volatile("" : "+v"(x[u]));   // keep the 16 chains interleaved (no sinking)
#pragma unroll
            for (int u = 0; u < UB; u++) {
                if (x[u] >= ta) {
                    if (cnt < CAPS) pend[cnt] = (int)(j0 + u);
                    cnt++;
                }
            }
        }
        for (int64_t j = NB; j < c_hi; j++) col_step((uint32_t)j);
    } else {
        // symmetric noise is keyed on (min, max) (dgm.py:1216-1223).  For the 64 rows of this wavefront the columns split
        // into j < every row (key = the COLUMN's: wave-uniform, computed on the scalar unit; the row id is hashed),
        // j > every row (key = the row's: the asymmetric loop) and the <= 64 columns in between (generic step).
        const int64_t i_first = row0 + (int64_t)blockIdx.x * 64, i_last = i_first + 63;
        const int64_t below_hi = c_hi < i_first ? c_hi : (i_first > c_lo ? i_first : c_lo);      // [c_lo, below_hi): j < i
        const int64_t above_lo = i_last + 1 > c_lo ? (i_last + 1 < c_hi ? i_last + 1 : c_hi) : c_lo;   // [above_lo, c_hi): j > i
        int64_t j0 = c_lo;
        for (; j0 + UB <= below_hi; j0 += UB) {
            uint32_t x[UB];
#pragma unroll
            for (int u = 0; u < UB; u++) {
                const uint2 kj = colkeys[j0 + u];                // wave-uniform address: scalar loads (s_load_dwordx16)
                uint32_t v = iu ^ kj.x;
                v *= 0x7feb352dU; v ^= v >> 15; v += kj.y; v *= 0x846ca68bU;
                x[u] = v;
            }
#pragma unroll
            for (int u = 0; u < UB; u++) asm volatile("" : "+v"(x[u]));
#pragma unroll
            for (int u = 0; u < UB; u++) {
                if (x[u] >= ta) {
                    if (cnt < CAPS) pend[cnt] = (int)(j0 + u);
                    cnt++;
                }
            }
        }
        int64_t mid_hi = above_lo > j0 ? above_lo : j0;
        // align the start of the "above" loop so that its unrolled body covers whole groups; leftovers go generic
        for (; j0 < mid_hi; j0++) col_step((uint32_t)j0);
        for (; j0 + UB <= c_hi; j0 += UB) {
            uint32_t x[UB];
#pragma unroll
            for (int u = 0; u < UB; u++) {
                uint32_t v = (uint32_t)(j0 + u) ^ k1;
                v *= 0x7feb352dU; v ^= v >> 15; v += k2; v *= 0x846ca68bU;
                x[u] = v;
            }
#pragma unroll
            for (int u = 0; u < UB; u++) asm volatile("" : "+v"(x[u]));
#pragma unroll
            for (int u = 0; u < UB; u++) {
                if (x[u] >= ta) {
                    if (cnt < CAPS) pend[cnt] = (int)(j0 + u);
                    cnt++;
                }
            }
        }
        for (; j0 < c_hi; j0++) col_step((uint32_t)j0);
    }
    if (rvalid) cnt_g[lrow * NSEG + seg] = cnt;
}

// K1', symmetric noise on a WHOLE graph (rows == all N nodes): the noise of pair (i, j) is the noise of (j, i), so only the pairs
// j > i are hashed -- half of the N^2 steps of gv_sweep<true> -- and a hit is recorded twice: column j in row i's own list
// (private counter, as above) and row i in the TRANSPOSED list of node j.  The transposed appends happen AFTER the wavefront's
// sweep, every lane walking its own list: 64 returning atomics in flight per step instead of one per column with a hit (a
// returning atomic inside the column loop stalls the wavefront for a memory round trip: 3.1 ms against 2.6 for the rectangular
// sweep).  Row j's candidates are then its own list (columns > j), its transposed list (rows < j) and the zero-noise diagonal.
// (Row shards keep the rectangular sweep: a remote row's hits would never reach a local transposed list.)
constexpr int CAPT = 448;          // transposed-list slots per node (expected <= ~230)
__global__ __launch_bounds__(64) void gv_sweep_tri(int64_t N, uint32_t s0, uint32_t s1, GvCtl *ctl, int *__restrict__ pend_g,
                                                   int *__restrict__ cnt_g, int *__restrict__ pendT, int *__restrict__ cntT) {
    const int lane = threadIdx.x;
    const int64_t i0 = (int64_t)blockIdx.x * 64, i = i0 + lane;
    const bool rvalid = i < N;
    const uint32_t iu = (uint32_t)(rvalid ? i : N - 1);
    const int seg = blockIdx.y;
    const int64_t per = ((N + NSEG - 1) / NSEG + 15) / 16 * 16;
    const int64_t c_lo0 = seg * per < N ? seg * per : N, c_hi = c_lo0 + per < N ? c_lo0 + per : N;
    const float M = fmaxf(ctl->msum * (1.0f / PILOT_PAIRS), 1e-30f);
    const float target = fminf(TARGET_MAX, fmaxf(TARGET_MIN, ADMIT_MAX * M));
    const float gmin0 = 0.3f * __logf(fmaxf((float)N * M / target, 1e-30f));
    if (blockIdx.x == 0 && seg == 0 && lane == 0) ctl->gmin0 = gmin0;
    int cnt = 0;
    if (c_hi > i0) {                                             // (a segment wholly below the wavefront's rows has nothing to do)
        const int64_t c_lo = c_lo0 > i0 ? c_lo0 : i0;
        int *pend = pend_g + i * CAPF + seg * CAPS;
        const uint32_t ta = rvalid ? hash_threshold_from_gmin(gmin0) : 0xffffffffu;
        uint32_t k1, k2;
        rowkey(s0, s1, iu, k1, k2);
        // a hit that no longer fits the row's own list (the row itself fails verification and is redone alone) still reaches its PARTNER's
        // transposed list, on the spot: the partner's candidates stay complete.  (Until round 6 an overflow was only published and
        // gv_finalize failed EVERY row on seeing it: 24 overflowing lists among 400 000 -- the guess admits up to 0.8 of the slots on
        // average -- sent all 100 000 rows through the exhaustive fallback: 132 ms for a 2.3 ms stage, on N(0, 0.7) features.)
        auto spill = [&](uint32_t j) {
            const int slot = atomicAdd(&cntT[j], 1);
            if (slot < CAPT) pendT[(int64_t)j * CAPT + slot] = (int)iu;
        };
        // the <= 64 columns that straddle the wavefront's own rows: row i takes column j only if j > i (j == i: the diagonal)
        int64_t j0 = c_lo;
        const int64_t diag_hi = (i0 + 64 < c_hi) ? i0 + 64 : c_hi;
        for (; j0 < diag_hi; j0++) {
            const uint32_t j = (uint32_t)j0;
            const uint32_t x = pair_u24_keyed(k1, k2, j) << 8;
            if (rvalid && (j == iu || (j > iu && x >= ta))) {    // zero-noise diagonal: always a candidate of its own row
                if (cnt < CAPS) pend[cnt] = (int)j; else if (j != iu) spill(j);
                cnt++;
            }
        }
        constexpr int UB = 16;
        for (; j0 + UB <= c_hi; j0 += UB) {
            uint32_t x[UB];
#pragma unroll
            for (int u = 0; u < UB; u++) {
                uint32_t v = (uint32_t)(j0 + u) ^ k1;
                v *= 0x7feb352dU; v ^= v >> 15; v += k2; v *= 0x846ca68bU;
                x[u] = v;
            }
#pragma unroll
            for (int u = 0; u < UB; u++) asm volatile("" : "+v"(x[u]));   // keep the chains interleaved (no sinking)
#pragma unroll
            for (int u = 0; u < UB; u++) {
                if (x[u] >= ta) {
                    if (rvalid) { if (cnt < CAPS) pend[cnt] = (int)(j0 + u); else spill((uint32_t)(j0 + u)); }
                    cnt++;
                }
            }
        }
        for (; j0 < c_hi; j0++) {
            const uint32_t j = (uint32_t)j0;
            if ((pair_u24_keyed(k1, k2, j) << 8) >= ta) {
                if (rvalid) { if (cnt < CAPS) pend[cnt] = (int)j; else spill(j); }
                cnt++;
            }
        }
        // transposed appends, lane-parallel: every lane walks its own list, one returning atomic per entry
        const int nmine = rvalid ? (cnt < CAPS ? cnt : CAPS) : 0;
        int nmax = nmine;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) nmax = max(nmax, __shfl_xor(nmax, off, 64));
        for (int e = 0; e < nmax; e++) {
            if (e < nmine) {
                const int j = pend[e];
                if (j != (int)iu) {
                    const int slot = atomicAdd(&cntT[j], 1);
                    if (slot < CAPT) pendT[(int64_t)j * CAPT + slot] = (int)iu;
                }
            }
        }
    }
    if (rvalid) cnt_g[i * NSEG + seg] = cnt;
}

// K2: exact scoring of the candidates, top-64, verification.  One wavefront per row.
template <int H, bool SYM>
__global__ __launch_bounds__(256) void gv_finalize(const float *__restrict__ xp, int64_t N, int64_t row0, int64_t row1,
                                                   float t, uint32_t s0, uint32_t s1, GvCtl *ctl,
                                                   const int *__restrict__ pend_g, const int *__restrict__ cnt_g,
                                                   int *__restrict__ faillist, int32_t *__restrict__ idx,
                                                   float *__restrict__ val, const int *__restrict__ pendT = nullptr,
                                                   const int *__restrict__ cntT = nullptr) {
    const int lane = threadIdx.x & 63;
    const int64_t lrow = (int64_t)blockIdx.x * 4 + dgg::wave_id();
    const int64_t i = row0 + lrow;
    if (i >= row1) return;
    // the row's candidates sit in NSEG segment lists (+ the transposed list of the triangular sweep); view them as one
    // concatenated list of n entries
    int ns[NSEG], off[NSEG + 1];
    off[0] = 0;
    bool ok = true;
#pragma unroll
    for (int s = 0; s < NSEG; s++) {
        ns[s] = cnt_g[lrow * NSEG + s];
        ok = ok && ns[s] <= CAPS;
        off[s + 1] = off[s] + (ns[s] <= CAPS ? ns[s] : CAPS);
    }
    const int nT = cntT ? cntT[i] : 0;
    ok = ok && nT <= CAPT;
    const int n = off[NSEG] + (nT <= CAPT ? nT : CAPT);
    const int *pl = pend_g + lrow * CAPF;
    uint64_t list = DGG_EMPTY_KEY;
    ok = ok && n >= 64;
    if (ok) {
        // Two stages.  The sweep admitted every pair whose noise can reach gmin0 (an expected TARGET_MAX of them clear it with their
        // distance).  Stage A gathers and scores only the candidates whose noise can reach the TIGHTER g_tight = gmin0 + DTIGHT
        // (an expected TARGET_TIGHT clear that) and verifies the 64th score against g_tight: true for most rows, and 1/3 of the
        // 256-byte gathers never happen.  A row that fails stage A scores the rest (stage B) and is verified against gmin0 as before.
        const float gmin0 = ctl->gmin0, g_tight = gmin0 + DTIGHT;
        const uint32_t ta_tight = hash_threshold_from_gmin(g_tight);
        auto cand_at = [&](int e) {
            int s = 0;
#pragma unroll
            for (int q = 1; q < NSEG; q++) s += (e >= off[q]) ? 1 : 0;
            int o = off[0];
#pragma unroll
            for (int q = 1; q < NSEG; q++) o = (s == q) ? off[q] : o;
            return e >= off[NSEG] ? pendT[i * CAPT + (e - off[NSEG])] : pl[s * CAPS + (e - o)];
        };
        auto stage = [&](bool tight_part) {
            for (int base = 0; base < n; base += 64) {
                const int e = base + lane;
                uint64_t key = DGG_EMPTY_KEY;
                if (e < n) {
                    const int32_t j = cand_at(e);
                    const bool diag = SYM && (int64_t)j == i;
                    const uint32_t u24 = pair_u24(s0, s1, (uint32_t)i, (uint32_t)j, SYM);
                    const bool tight = diag || (u24 << 8) >= ta_tight;
                    if (tight == tight_part) key = make_key(exact_score_gv_u<H>(xp, i, j, t, diag, u24), j);
                }
                if (__ballot(key != DGG_EMPTY_KEY) == 0ull) continue;
                key = wave_sort<false>(key, lane);
                list = wave_merge_top64_asc(list, key, lane);
            }
        };
        stage(true);
        uint64_t k63 = shfl_u64(list, 63);
        const bool okA = k63 != DGG_EMPTY_KEY && __logf(key_val(k63)) >= g_tight + 1e-3f;
        if (!okA) {
            stage(false);
            k63 = shfl_u64(list, 63);
            // rejected pairs have log-score < gmin0 + 1e-8: the list is exact iff its 64th entry clears that with margin
            ok = k63 != DGG_EMPTY_KEY && __logf(key_val(k63)) >= gmin0 + 1e-3f;
        }
    }
    if (ok) {
        idx[lrow * 64 + lane] = key_col(list);
        val[lrow * 64 + lane] = key_val(list);
    } else if (lane == 0) {
        int slot = atomicAdd(&ctl->nfail, 1);
        faillist[slot] = (int)lrow;
    }
}

// K3: the rows of the fail list are redone from scratch, one workgroup per row with LANE = COLUMN: every wavefront sweeps
// a quarter of the columns 64 at a time -- noise first, exact score only while the noise can still reach the wavefront's
// 64th log-score -- and the four lists are merged through LDS.  (A failing row is rare -- a handful in 100 000 -- but a
// serial per-row sweep of all N columns would put milliseconds on the critical path for it.)
template <int H, bool SYM>
__global__ __launch_bounds__(256) void gv_fallback(const float *__restrict__ xp, int64_t N, int64_t row0, float t,
                                                  uint32_t s0, uint32_t s1, const GvCtl *ctl,
                                                  const int *__restrict__ faillist, int32_t *__restrict__ idx,
                                                  float *__restrict__ val) {
    __shared__ uint64_t lists[4][64];
    const int lane = threadIdx.x & 63, wave = dgg::wave_id();
    const int nfail = ctl->nfail;
    for (int f = blockIdx.x; f < nfail; f += gridDim.x) {
        const int lrow = faillist[f];
        const int64_t i = row0 + lrow;
        uint64_t list = DGG_EMPTY_KEY;
        float thr = -INFINITY;
        for (int64_t j0 = (int64_t)wave * 64; j0 < N; j0 += 256) {
            const int64_t j = j0 + lane;
            uint64_t key = DGG_EMPTY_KEY;
            if (j < N) {
                const float g = pair_noise(s0, s1, (uint32_t)i, (uint32_t)j, SYM);
                if (!(g + 1e-8f + 1e-3f < thr)) key = make_key(exact_score_gv<H>(xp, i, (int32_t)j, t, SYM, s0, s1), (int32_t)j);
            }
            if (__ballot(key != DGG_EMPTY_KEY) != 0ull) {
                key = wave_sort<false>(key, lane);
                list = wave_merge_top64_asc(list, key, lane);
                const uint64_t k63 = shfl_u64(list, 63);
                if (k63 != DGG_EMPTY_KEY) thr = __logf(key_val(k63));
            }
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

template <int H>
int launch_gv(const float *xp, int64_t N, int64_t row0, int64_t row1, float t, int noise_mode, uint32_t s0, uint32_t s1,
              int32_t *idx, float *val, void *ws, hipStream_t st) {
    const int64_t R = row1 - row0, R64 = (R + 63) / 64 * 64;
    char *w = reinterpret_cast<char *>(ws);
    GvCtl *ctl = reinterpret_cast<GvCtl *>(w);
    int *cnt = reinterpret_cast<int *>(w + 256);
    int *faillist = cnt + R64 * NSEG;
    int *pend = faillist + R64;
    uint2 *colkeys = reinterpret_cast<uint2 *>(pend + R64 * (size_t)CAPF);
    if (dgg_check_hip(hipMemsetAsync(ctl, 0, sizeof(GvCtl), st), "gv memset") != 0) return DGG_ERR_HIP;
    const bool sym = noise_mode == 3;
    hipLaunchKernelGGL(gv_pilot<H>, dim3(PILOT_PAIRS / 256 / PILOT_PER_THREAD), dim3(256), 0, st, xp, N, t, s0, s1, ctl);
    dim3 gsweep((unsigned)(R64 / 64));
    dim3 gsweep2((unsigned)(R64 / 64), NSEG);
    // whole graph + symmetric noise: triangular sweep with transposed lists (after the column keys in the workspace)
    static const int tri_env = [] { const char *e = getenv("DGG_GV_TRI"); return e ? atoi(e) : 1; }();
    const bool tri = sym && row0 == 0 && row1 == N && tri_env != 0;
    int *cntT = reinterpret_cast<int *>(colkeys + N + 64), *pendT = cntT + R64;
    if (tri) {
        if (dgg_check_hip(hipMemsetAsync(cntT, 0, (size_t)R64 * 4, st), "gv memset") != 0) return DGG_ERR_HIP;
        hipLaunchKernelGGL(gv_sweep_tri, gsweep2, dim3(64), 0, st, N, s0, s1, ctl, pend, cnt, pendT, cntT);
    } else if (sym) {
        hipLaunchKernelGGL(gv_rowkeys, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, N, s0, s1, colkeys);
        hipLaunchKernelGGL(gv_sweep<true>, gsweep2, dim3(64), 0, st, N, row0, row1, s0, s1, ctl, pend, cnt, colkeys);
    } else hipLaunchKernelGGL(gv_sweep<false>, gsweep2, dim3(64), 0, st, N, row0, row1, s0, s1, ctl, pend, cnt, colkeys);
    dim3 gfin((unsigned)((R + 3) / 4));
    if (tri) hipLaunchKernelGGL((gv_finalize<H, true>), gfin, dim3(256), 0, st, xp, N, row0, row1, t, s0, s1, ctl, pend, cnt, faillist, idx, val,
                                pendT, cntT);
    else if (sym) hipLaunchKernelGGL((gv_finalize<H, true>), gfin, dim3(256), 0, st, xp, N, row0, row1, t, s0, s1, ctl, pend, cnt, faillist, idx, val);
    else hipLaunchKernelGGL((gv_finalize<H, false>), gfin, dim3(256), 0, st, xp, N, row0, row1, t, s0, s1, ctl, pend, cnt, faillist, idx, val);
    if (sym) hipLaunchKernelGGL((gv_fallback<H, true>), gsweep, dim3(256), 0, st, xp, N, row0, t, s0, s1, ctl, faillist, idx, val);
    else hipLaunchKernelGGL((gv_fallback<H, false>), gsweep, dim3(256), 0, st, xp, N, row0, t, s0, s1, ctl, faillist, idx, val);
    return dgg_check_launch("allpairs_topk_gv");
}

}  // namespace

size_t dgg_allpairs_gv_ws_bytes(int64_t rows, int64_t N) {
    size_t R64 = ((size_t)rows + 63) / 64 * 64;
    return 256 + R64 * 4 * (NSEG + 1) + R64 * (size_t)CAPF * 4 + ((size_t)N + 64) * sizeof(uint2) +   // + the column keys (symmetric noise)
           R64 * 4 + R64 * (size_t)CAPT * 4;                     // + counters and transposed lists of the triangular sweep
}

bool dgg_allpairs_gv_supported(int h, int noise_mode, int K) {
    return K == 64 && (h == 8 || h == 16 || h == 32 || h == 64 || h == 128) && (noise_mode == 2 || noise_mode == 3);
}

int dgg_allpairs_topk_gv_impl(const float *xp, int64_t N, int h, int64_t row0, int64_t row1, float t, int noise_mode,
                              uint32_t s0, uint32_t s1, int K, int32_t *idx, float *val, void *workspace, size_t ws_bytes,
                              hipStream_t st) {
    if (!dgg_allpairs_gv_supported(h, noise_mode, K))
        return dgg_set_error(DGG_ERR_UNSUPPORTED, "guess-and-verify path needs K=64, in-kernel noise, latent_dim in {8,...,128}");
    if (!workspace || ws_bytes < dgg_allpairs_gv_ws_bytes(row1 - row0, N))
        return dgg_set_error(DGG_ERR_ARG, "guess-and-verify path: workspace too small (dgg_allpairs_workspace_bytes)");
    if (row1 <= row0) return 0;
    switch (h) {
        case 8: return launch_gv<8>(xp, N, row0, row1, t, noise_mode, s0, s1, idx, val, workspace, st);
        case 16: return launch_gv<16>(xp, N, row0, row1, t, noise_mode, s0, s1, idx, val, workspace, st);
        case 32: return launch_gv<32>(xp, N, row0, row1, t, noise_mode, s0, s1, idx, val, workspace, st);
        case 64: return launch_gv<64>(xp, N, row0, row1, t, noise_mode, s0, s1, idx, val, workspace, st);
        default: return launch_gv<128>(xp, N, row0, row1, t, noise_mode, s0, s1, idx, val, workspace, st);
    }
}
