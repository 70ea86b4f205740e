// dgg_topk_rsym.hip -- all-pairs top-64 under the RANKED SYMMETRIC noise generator (noise_mode 5): the reference's default
// symmetric perturbation (symmetric_noise=True: G_ij = G_ji, zero diagonal, dgm.py:1216-1223) in O(N * ~300) instead of the N^2
// hash sweep of dgg_topk_gv.hip.
//
// Same contract as the other all-pairs kernels (reference dgm.py:1618-1623, 1213-1229, 1404):
//   p'_ij = exp(log(exp(-0.05 ||xp_i - xp_j||) + 1e-8) + G_ij),  64 largest per row, (score desc, column asc).
//
// Generator (dgg_common.h / oracle ora_ranked_sym_block): every unordered pair is OWNED by one endpoint (circular offsets: node
// o owns partners o+1 .. o+n_o mod N), and an owner produces the noises of its pairs in DECREASING order with the ranked
// generator of noise_mode 4 (Renyi order statistics, keyed bijection for the placement).  Since log p'_ij <= G_ij + 1e-8 for
// every distance, only pairs whose noise reaches a row's 64th log-score can enter its list:
//
//   K0 pilot      random pairs -> M = mean p^(1/0.3); for a threshold g the expected number of pairs of a row with log-score
//                 >= g is N M exp(-g/0.3); solve for TARGET_A (a little above the mean number of ranks L a row has to settle:
//                 64, or ceil(k_i + 8.5) + 1 with the learned degrees) -> gminA.
//   K1 emit       one wavefront per OWNER walks its sequence while the noise is >= gminA, gathers the partner's features and
//                 computes the EXACT score once for both endpoints (the score is symmetric bit for bit): every pair goes to the
//                 owner's own list; pairs whose log-score reaches gminA also go to the partner's inbox (one returning atomic).
//   K2 finalize   one wavefront per row: sort / merge of own list + inbox + the zero-noise diagonal (no gathers).  VERIFY: a pair
//                 that is in neither list has log-score < gminA, so the list is exact iff its L-th log-score clears gminA.
//   tier 2        rows that fail (fewer than L scores above gminA: a node farther from the others than the average, or a full
//                 list) publish how much lower THEIR threshold has to be (from the number of scores they did find); all owners
//                 walk down to the lowest of these but deliver only to the failing rows, each above its own threshold (K3); K4
//                 settles them with exact scores and verifies against the row's threshold.
//   tier 3        rows that fail again (or need more than MAXDEPTH times the pairs) get their whole noise row written out (every
//                 owner walks its complete sequence, K5) and are swept column-parallel (K6).  Graphs of up to SMALL_N nodes
//                 take this tier directly.
// The noise of a pair has no random access (its rank inside the owner's sequence has no closed form), hence the tiers instead
// of a per-row exhaustive fallback: ONE row far from everything makes every owner walk deeper in tier 2, and a tier-3 row costs
// a full walk of all sequences (the per-pair hash generator, dgg_topk_gv.hip, has neither problem and stays selectable).  More
// tier-3 rows than the workspace holds set RsCtl::err (read by the host mirror).
#include "dgg_common.h"
#include "dgg_api_internal.h"

#include <cstdlib>

using namespace dgg;

namespace {

constexpr int CAPO = 384;            // own-list slots per row (pairs the row owns with noise >= gminA, scored)
constexpr int CAPT = 128;            // inbox slots per row (pairs owned by the partner with log-score >= gminA)
constexpr int FB2_ROWS = 8192;       // rows tier 2 can hold
constexpr int FB2CAP = 2048;         // candidate slots per tier-2 row
constexpr int SMALL_N = 1024;        // up to here every row takes tier 3
constexpr float TARGET_MIN = 24.0f;
constexpr float ADMIT_MAX = 0.7f * CAPO;          // own pairs per row the noise test may admit (target / M / 2) before the list fills
constexpr float MAXDEPTH = 32.0f;    // tier 2 walks at most this factor (in pairs) beyond tier 1
constexpr int PILOT_WG = 256, PILOT_PER_THREAD = 4;          // 256 x 256 x 4 = 262144 random pairs
constexpr int PILOT_PAIRS = PILOT_WG * 256 * PILOT_PER_THREAD;

struct RsCtl {
    float msum;                      // pilot: sum of p^(1/0.3)
    int nfail;                       // rows that failed tier 1
    float gminA;                     // guessed log-score threshold of tier 1
    int nfail3;                      // rows that go to tier 3
    int err;                         // tier-3 capacity exceeded: those rows were NOT computed
    int lsum;                        // sum over the rows of the number of ranks to settle
    int need_min;                    // lowest tier-2 threshold over the failing rows (ordered-int image of the float)
    float floor2;                    // lowest threshold tier 2 accepts (gminA - 0.3 ln MAXDEPTH)
    unsigned long long stats[4];     // emitted pairs, delivered pairs, tier-2 deliveries, unused
};

__device__ __forceinline__ int ordered_int(float f) { const int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
__device__ __forceinline__ float from_ordered_int(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

__device__ __forceinline__ uint64_t scan_u64(uint64_t v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t lo = __shfl_up((uint32_t)v, off, 64), hi = __shfl_up((uint32_t)(v >> 32), off, 64);
        uint64_t t = ((uint64_t)hi << 32) | lo;
        if (lane >= off) v += t;
    }
    return v;
}

__device__ __forceinline__ int64_t rsym_owned(int64_t N, int64_t o) {
    return (N - 1) / 2 + ((((N & 1) == 0) && o < N / 2) ? 1 : 0);
}

template <int H>
__global__ __launch_bounds__(256) void rs_pilot(const float *__restrict__ xp, int64_t N, float t, uint32_t s0, uint32_t s1,
                                                const float *__restrict__ klim, int64_t rows, RsCtl *ctl) {
    __shared__ float part[4];
    __shared__ int lpart[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
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
    int ls = 0;                                                  // ranks to settle, summed over the rows of the shard
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < rows; r += (int64_t)PILOT_WG * 256) ls += klim ? klimit_len(klim[r], 64) : 64;
    m = wave_sum_dpp(m, lane);
    for (int off = 32; off >= 1; off >>= 1) ls += __shfl_xor(ls, off, 64);
    if (lane == 0) { part[wave] = m; lpart[wave] = ls; }
    __syncthreads();
    if (threadIdx.x == 0) {                                      // one pair of atomics per workgroup (same-address atomics serialise)
        atomicAdd(&ctl->msum, part[0] + part[1] + part[2] + part[3]);
        atomicAdd(&ctl->lsum, lpart[0] + lpart[1] + lpart[2] + lpart[3]);
    }
}

__global__ void rs_thresh(int64_t N, int64_t rows, float target_a, float maxdepth, RsCtl *ctl) {
    const float M = fmaxf(ctl->msum * (1.0f / PILOT_PAIRS), 1e-30f);
    const float Lbar = (float)ctl->lsum / (float)(rows > 0 ? rows : 1);
    // expected pairs per row with a log-score above gminA: a little more than a row needs, less when distances matter so much
    // (small M) that the noise test would overfill the own lists
    float target = target_a > 0.0f ? target_a : 1.25f * Lbar + 6.0f;
    target = fminf(target, fmaxf(TARGET_MIN, 2.0f * ADMIT_MAX * M));
    const float g = 0.3f * __logf(fmaxf((float)N * M / target, 1e-30f));
    ctl->gminA = g;
    ctl->floor2 = g - 0.3f * __logf(maxdepth);
    ctl->need_min = ordered_int(INFINITY);
}

// small graphs: every row of the shard takes tier 3
__global__ void rs_all3(int64_t row0, int64_t rows, RsCtl *ctl, int *__restrict__ slot3, int *__restrict__ fail3rows) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r == 0) ctl->nfail3 = (int)rows;
    if (r >= rows) return;
    slot3[row0 + r] = (int)r;
    fail3rows[r] = (int)r;
}

template <int H>
__device__ __forceinline__ float exact_score_known(const float *__restrict__ xi, const float *__restrict__ xp, int32_t j, float t, float G) {
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
    return score_from_dist(c_sqrt(d2), t, true, G);
}

// exact score, or 0 when the partial distance over the first 32 features already puts the log-score below `bar`
template <int H>
__device__ __forceinline__ float exact_score_cut(const float *__restrict__ xi, const float *__restrict__ xp, int32_t j, float t, float G, float bar) {
    const float4 *xj = reinterpret_cast<const float4 *>(xp + (int64_t)j * H);
    float d2 = 0.0f;
    auto chain = [&](int c8) {
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
    };
    constexpr int HEAD = H >= 64 ? 4 : H / 8;
#pragma unroll
    for (int c8 = 0; c8 < HEAD; c8++) chain(c8);
    if (HEAD < H / 8) {
        // log p' = G + log(exp(t dist) + 1e-8) decreases with the distance: evaluated at the lower bound (fast math; `bar` carries the margin)
        if (G + __logf(__expf(t * sqrtf(d2)) + 1e-8f) < bar) return 0.0f;
#pragma unroll
        for (int c8 = HEAD; c8 < H / 8; c8++) chain(c8);
    }
    return score_from_dist(c_sqrt(d2), t, true, G);
}

// K1 / K3 / K5: one wavefront per owner walks its sequence in decreasing noise order.
//   TIER 1: while G >= gminA (margin 1e-3): exact score of the pair -> own list of o (if o is a row of the shard); inbox of the
//           partner (if it is one) when the log-score reaches gminA
//   TIER 2: while G >= the lowest threshold a failing row asked for: (partner, noise) to the tier-2 lists of the failing rows
//           among {o, partner} whose own threshold the noise reaches
//   TIER 3: the whole sequence: dense noise rows of the tier-3 rows among {o, partner}
template <int H, int TIER>
__global__ __launch_bounds__(256) void rs_emit(const float *__restrict__ xp, int64_t N, int64_t row0, int64_t row1, float t, uint32_t s0,
                                               uint32_t s1, RsCtl *ctl, int *__restrict__ cntO, int *__restrict__ cntT,
                                               int2 *__restrict__ own, int2 *__restrict__ inbox, const int *__restrict__ slot2,
                                               const float *__restrict__ need2, int *__restrict__ cnt2, int2 *__restrict__ list2,
                                               const int *__restrict__ slot3, float *__restrict__ gdense, int t3cap, int stats_on) {
    if (TIER == 2 && ctl->nfail == 0) return;
    if (TIER == 3 && ctl->nfail3 == 0) return;
    const int lane = threadIdx.x & 63;
    const int64_t o = (int64_t)blockIdx.x * 4 + dgg::wave_id();
    if (o >= N) return;
    const int64_t n = rsym_owned(N, o);
    const bool own_in = o >= row0 && o < row1;
    if (n <= 0) {
        if (TIER == 1 && own_in && lane == 0) cntO[o - row0] = 0;
        return;
    }
    // row shards: an owner outside the shard matters only if one of its partners o + 1 .. o + n (mod N) is a row of the shard, i.e.
    // if the shard's first row lies within n steps ahead of it -- the other owners (up to 3/8 of them on a rank of 8) do not walk
    if (!own_in) {
        const int64_t d0 = row0 > o ? row0 - o : row0 - o + N;
        if (d0 > n) return;
    }
    const float gminA = ctl->gminA;
    const float tau = TIER == 3 ? -INFINITY : (TIER == 2 ? from_ordered_int(ctl->need_min) : gminA) - 1e-3f;
    int so = -1;
    float need_o = INFINITY;
    if (TIER == 2) { so = slot2[o]; if (so >= 0) need_o = need2[so]; }
    if (TIER == 3) { so = slot3[o]; if (so >= t3cap) so = -1; }
    const float *xo = xp + o * H;                                // wave-uniform row: scalar loads
    uint32_t k1, k2;
    rowkey(s0, s1, (uint32_t)o, k1, k2);
    const uint32_t k3 = mix32(k2 ^ 0x68E31DA4u);
    const int b = ranked_bits(n);
    const uint64_t D = (uint64_t)1 << b;
    uint64_t S = 0;
    uint32_t scount = 0;
    int emitted = 0, delivered = 0;
    for (uint64_t rb = 0; rb < D; rb += 64) {
        const uint32_t c = ranked_sigma((uint32_t)(rb + lane), k1, k2, k3, b);
        const bool valid = (int64_t)c < n;
        const uint64_t m = __ballot(valid);
        if (m == 0ull) continue;
        const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        const uint32_t s = scount + pos + 1;                     // 1-based rank
        const uint64_t term = valid ? ranked_term(k1, k3, s, n) : 0ull;
        const uint64_t pre = scan_u64(term, lane) + S;
        const float G = ranked_gumbel(pre);
        const bool emit = valid && G >= tau;
        int64_t p = o + (int64_t)c + 1;
        if (p >= N) p -= N;
        if (emit) {
            if (TIER == 1) {
                const bool p_in = p >= row0 && p < row1;
                if (own_in || p_in) {
                    // the first 128 bytes of the partner's row bound the distance from below (the fmaf chain only grows): a pair whose
                    // log-score cannot reach gminA even at that distance is of no use to either endpoint at this tier -- a row that
                    // needs it has failed tier 1 and gets it again in tier 2 -- so its second half is never fetched (score 0)
                    const float v = exact_score_cut<H>(xo, xp, (int32_t)p, t, G, gminA - 2e-3f);
                    if (own_in && s <= (uint32_t)CAPO) own[(o - row0) * CAPO + (s - 1)] = make_int2((int)p, __float_as_int(v));
                    if (p_in && __logf(v) >= gminA - 1e-3f) {
                        const int slot = atomicAdd(&cntT[p - row0], 1);
                        if (slot < CAPT) inbox[(p - row0) * CAPT + slot] = make_int2((int)o, __float_as_int(v));
                        delivered++;
                    }
                }
            } else if (TIER == 2) {
                if (so >= 0 && G >= need_o - 1e-3f) {
                    const int q = atomicAdd(&cnt2[so], 1);
                    if (q < FB2CAP) list2[(int64_t)so * FB2CAP + q] = make_int2((int)p, __float_as_int(G));
                    delivered++;
                }
                const int sp = slot2[p];
                if (sp >= 0 && G >= need2[sp] - 1e-3f) {
                    const int q = atomicAdd(&cnt2[sp], 1);
                    if (q < FB2CAP) list2[(int64_t)sp * FB2CAP + q] = make_int2((int)o, __float_as_int(G));
                    delivered++;
                }
            } else {
                if (so >= 0) gdense[(int64_t)so * N + p] = G;
                const int sp = slot3[p];
                if (sp >= 0 && sp < t3cap) gdense[(int64_t)sp * N + o] = G;
            }
        }
        const uint64_t em = __ballot(emit);
        emitted += __builtin_popcountll(em);
        S = shfl_u64(pre, 63);
        scount += (uint32_t)__builtin_popcountll(m);
        if (scount >= (uint32_t)n) break;
        if (em != m) break;                                      // noise is decreasing in the rank: the first miss ends the walk
    }
    if (TIER == 1 && own_in && lane == 0) cntO[o - row0] = emitted;
    if (stats_on && TIER != 3) {
        for (int off = 32; off >= 1; off >>= 1) delivered += __shfl_xor(delivered, off, 64);
        if (lane == 0) {
            if (TIER == 1) atomicAdd(&ctl->stats[0], (unsigned long long)emitted);
            atomicAdd(&ctl->stats[TIER], (unsigned long long)delivered);
        }
    }
}

// merge one chunk of keys into the descending 64-entry list (first chunk: sort; few live keys: insert one by one; else sort + merge)
__device__ __forceinline__ uint64_t merge_chunk(uint64_t list, uint64_t key, bool first, int lane) {
    uint64_t live = __ballot(key != DGG_EMPTY_KEY);
    if (live == 0ull) return list;
    if (first) return wave_sort<true>(key, lane);
    if (__builtin_popcountll(live) <= 16) {
        while (live != 0ull) {                                   // wave-uniform
            const int src = __builtin_ctzll(live);
            live &= live - 1;
            const uint64_t kk = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(key >> 32), src) << 32) |
                                (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key, src);
            const int pos = __builtin_popcountll(__ballot(list > kk));
            const uint64_t prev = ((uint64_t)(uint32_t)__shfl_up((int)(list >> 32), 1, 64) << 32) | (uint32_t)__shfl_up((int)(uint32_t)list, 1, 64);
            list = lane < pos ? list : (lane == pos ? kk : prev);
        }
        return list;
    }
    return wave_merge_top64_asc(list, wave_sort<false>(key, lane), lane);
}

__device__ __forceinline__ uint64_t lane_key(uint64_t list, int l) {
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(list >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)list, l);
}

// settle n candidates (entry e -> (column, noise) through `at`) into a descending 64-entry list with exact scores; a candidate
// whose noise cannot reach the L-th log-score found so far is not gathered.  Same list whatever the order of the candidates.
template <int H, typename At>
__device__ __forceinline__ uint64_t settle(const float *__restrict__ xp, int64_t i, float t, int n, int L, int lane, At at) {
    const float *xi = xp + i * H;                                // wave-uniform row: scalar loads
    uint64_t list = DGG_EMPTY_KEY;
    float thr_log = -INFINITY;
    bool first = true;
    for (int base = 0; base < n; base += 64) {
        const int e = base + lane;
        uint64_t key = DGG_EMPTY_KEY;
        if (e < n) {
            const int2 ent = at(e);
            const float G = __int_as_float(ent.y);
            if (!(G + 1e-8f + 1e-3f < thr_log)) key = make_key(exact_score_known<H>(xi, xp, ent.x, t, G), ent.x);
        }
        if (__ballot(key != DGG_EMPTY_KEY) == 0ull) continue;
        list = merge_chunk(list, key, first, lane);
        first = false;
        const uint64_t kL = lane_key(list, L - 1);
        if (kL != DGG_EMPTY_KEY) thr_log = __logf(key_val(kL));
    }
    return list;
}

__device__ __forceinline__ bool list_clears(uint64_t list, int L, float g) {
    const uint64_t kL = lane_key(list, L - 1);
    return kL != DGG_EMPTY_KEY && __logf(key_val(kL)) >= g + 1e-3f;
}

__device__ __forceinline__ void write_row(uint64_t list, int L, int lane, int64_t lrow, int32_t *__restrict__ idx, float *__restrict__ val) {
    const bool empty = list == DGG_EMPTY_KEY || lane >= L;
    idx[lrow * 64 + lane] = empty ? -1 : key_col(list);
    val[lrow * 64 + lane] = empty ? 0.0f : key_val(list);
}

// a row that could not be settled goes to tier 3 (dense noise row), or is reported when that is full
__device__ __forceinline__ void to_tier3(RsCtl *ctl, int64_t i, int64_t lrow, int lane, int t3cap, int *__restrict__ slot3,
                                         int *__restrict__ fail3rows, int32_t *__restrict__ idx, float *__restrict__ val) {
    int s3 = 0;
    if (lane == 0) s3 = atomicAdd(&ctl->nfail3, 1);
    s3 = __builtin_amdgcn_readfirstlane(s3);
    if (s3 < t3cap) {
        if (lane == 0) { slot3[i] = s3; fail3rows[s3] = (int)lrow; }
    } else {
        if (lane == 0) ctl->err = 1;
        idx[lrow * 64 + lane] = -1;
        val[lrow * 64 + lane] = 0.0f;
    }
}

// K2: one wavefront per row of the shard: the scores are already there
__global__ __launch_bounds__(256) void rs_finalize(int64_t N, int64_t row0, int64_t row1, float t, const float *__restrict__ klim, RsCtl *ctl,
                                                   const int *__restrict__ cntO, const int *__restrict__ cntT,
                                                   const int2 *__restrict__ own, const int2 *__restrict__ inbox,
                                                   int *__restrict__ slot2, float *__restrict__ need2, int *__restrict__ fail2rows,
                                                   int *__restrict__ slot3, int *__restrict__ fail3rows, int t3cap,
                                                   int32_t *__restrict__ idx, float *__restrict__ val) {
    const int lane = threadIdx.x & 63;
    const int64_t lrow = (int64_t)blockIdx.x * 4 + dgg::wave_id();
    const int64_t i = row0 + lrow;
    if (i >= row1) return;
    const int L = klim ? __builtin_amdgcn_readfirstlane(klimit_len(klim[lrow], 64)) : 64;
    const int nO = cntO[lrow], nT = cntT[lrow];
    const bool fits = nO <= CAPO && nT <= CAPT;
    const int nOc = nO <= CAPO ? nO : CAPO, nTc = nT <= CAPT ? nT : CAPT;
    const int n = 1 + nOc + nTc;
    const int2 *po = own + lrow * CAPO, *pt = inbox + lrow * CAPT;
    const float gminA = ctl->gminA;
    // entry 0: the zero-noise diagonal (distance 0); then the own list (decreasing noise), then the inbox (arrival order)
    const float vdiag = score_from_dist(0.0f, t, true, 0.0f);
    uint64_t list = DGG_EMPTY_KEY;
    uint32_t thr_bits = 0u;                                      // keys below the L-th score found so far are dropped before the merge
    for (int base = 0; base < n; base += 64) {
        const int e = base + lane;
        uint64_t key = DGG_EMPTY_KEY;
        if (e < n) {
            const int2 ent = e == 0 ? make_int2((int)i, __float_as_int(vdiag)) : (e - 1 < nOc ? po[e - 1] : pt[e - 1 - nOc]);
            if ((uint32_t)ent.y >= thr_bits) key = make_key(__int_as_float(ent.y), ent.x);
        }
        if (__ballot(key != DGG_EMPTY_KEY) == 0ull) continue;
        list = merge_chunk(list, key, base == 0, lane);
        const uint64_t kL = lane_key(list, L - 1);
        if (kL != DGG_EMPTY_KEY) thr_bits = (uint32_t)(kL >> 32);
    }
    const bool ok = fits && list_clears(list, L, gminA);
    if (ok) { write_row(list, L, lane, lrow, idx, val); return; }
    // Tier 2 with the row's own threshold: c of the L ranks cleared gminA, i.e. the row sees about c / TARGET_A of the pairs the
    // average row sees; ask for a threshold low enough for an expected 1.5 L (at least 1.5 x deeper).
    const float lv = list == DGG_EMPTY_KEY ? -INFINITY : __logf(key_val(list));
    const int c = __builtin_popcountll(__ballot(lane < L && lv >= gminA + 1e-3f));
    const float need = gminA - 0.3f * __logf(fmaxf(1.5f * (float)L / (float)(c < 4 ? 4 : c), 1.5f));
    int s2 = 0;
    if (lane == 0) s2 = (need >= ctl->floor2) ? atomicAdd(&ctl->nfail, 1) : FB2_ROWS;
    s2 = __builtin_amdgcn_readfirstlane(s2);
    if (s2 < FB2_ROWS) {
        if (lane == 0) {
            slot2[i] = s2;
            need2[s2] = need;
            fail2rows[s2] = (int)lrow;
            atomicMin(&ctl->need_min, ordered_int(need));
        }
    } else to_tier3(ctl, i, lrow, lane, t3cap, slot3, fail3rows, idx, val);
}

// K4: one wavefront per tier-2 row
template <int H>
__global__ __launch_bounds__(256) void rs_finalize2(const float *__restrict__ xp, int64_t N, int64_t row0, float t,
                                                    const float *__restrict__ klim, RsCtl *ctl, const int *__restrict__ cnt2,
                                                    const int2 *__restrict__ list2, const float *__restrict__ need2,
                                                    const int *__restrict__ fail2rows, int *__restrict__ slot3,
                                                    int *__restrict__ fail3rows, int t3cap, int32_t *__restrict__ idx,
                                                    float *__restrict__ val) {
    const int nf = ctl->nfail < FB2_ROWS ? ctl->nfail : FB2_ROWS;
    const int slot = blockIdx.x * 4 + dgg::wave_id();
    if (slot >= nf) return;
    const int lane = threadIdx.x & 63;
    const int64_t lrow = fail2rows[slot];
    const int64_t i = row0 + lrow;
    const int L = klim ? __builtin_amdgcn_readfirstlane(klimit_len(klim[lrow], 64)) : 64;
    const int n2 = cnt2[slot];
    bool ok = n2 <= FB2CAP;
    const int n = 1 + (n2 <= FB2CAP ? n2 : FB2CAP);
    const int2 *pl = list2 + (int64_t)slot * FB2CAP;
    uint64_t list = settle<H>(xp, i, t, n, L, lane, [&](int e) { return e == 0 ? make_int2((int)i, 0) : pl[e - 1]; });
    ok = ok && (list_clears(list, L, need2[slot]) || (int64_t)(n2 + 1) >= N);
    if (ok) write_row(list, L, lane, lrow, idx, val);
    else to_tier3(ctl, i, lrow, lane, t3cap, slot3, fail3rows, idx, val);
}

// K6: one workgroup per tier-3 row, LANE = COLUMN: every wavefront sweeps a quarter of the columns 64 at a time with the noise
// row written by rs_emit<3> -- exact score only while the noise can still reach the wavefront's L-th log-score
template <int H>
__global__ __launch_bounds__(256) void rs_rows3(const float *__restrict__ xp, int64_t N, int64_t row0, float t,
                                                const float *__restrict__ klim, const RsCtl *ctl, const int *__restrict__ fail3rows,
                                                const float *__restrict__ gdense, int t3cap, int32_t *__restrict__ idx,
                                                float *__restrict__ val) {
    __shared__ uint64_t lists[4][64];
    const int lane = threadIdx.x & 63, wave = dgg::wave_id();
    const int nf = ctl->nfail3 < t3cap ? ctl->nfail3 : t3cap;
    for (int f = blockIdx.x; f < nf; f += gridDim.x) {
        const int64_t lrow = fail3rows[f];
        const int64_t i = row0 + lrow;
        const float *xi = xp + i * H;
        const float *gi = gdense + (int64_t)f * N;
        const int L = klim ? __builtin_amdgcn_readfirstlane(klimit_len(klim[lrow], 64)) : 64;
        uint64_t list = DGG_EMPTY_KEY;
        float thr = -INFINITY;
        for (int64_t j0 = (int64_t)wave * 64; j0 < N; j0 += 256) {
            const int64_t j = j0 + lane;
            uint64_t key = DGG_EMPTY_KEY;
            if (j < N) {
                const float g = j == i ? 0.0f : gi[j];
                if (!(g + 1e-8f + 1e-3f < thr)) key = make_key(exact_score_known<H>(xi, xp, (int32_t)j, t, g), (int32_t)j);
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
            write_row(list, L, lane, lrow, idx, val);
        }
        __syncthreads();
    }
}

struct RsLayout {
    size_t ctl, cntO, cntT, fail2rows, fail3rows, slot2, slot3, cnt2, need2, own, inbox, list2, gdense, total;
    int t3cap;
};

int t3_slots(int64_t rows, int64_t N) {
    if (N <= SMALL_N) return (int)rows;
    int64_t c = ((int64_t)1 << 28) / (N > 0 ? N : 1);
    c = c < 16 ? 16 : (c > 256 ? 256 : c);
    return (int)c;
}

RsLayout make_layout(int64_t rows, int64_t N) {
    RsLayout L;
    const size_t R = ((size_t)rows + 63) / 64 * 64, NN = ((size_t)N + 63) / 64 * 64;
    L.t3cap = t3_slots(rows, N);
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += (bytes + 255) / 256 * 256; return at; };
    L.ctl = take(256);
    L.slot2 = take(NN * 4);                  // (slot2 and slot3 are contiguous: one memset)
    L.slot3 = take(NN * 4);
    L.cntT = take(R * 4);                    // (cntT and cnt2 are contiguous: one memset)
    L.cnt2 = take(FB2_ROWS * 4);
    L.cntO = take(R * 4);
    L.need2 = take(FB2_ROWS * 4);
    L.fail2rows = take(FB2_ROWS * 4);
    L.fail3rows = take((size_t)(L.t3cap > 0 ? L.t3cap : 1) * 4);
    L.own = take(R * (size_t)CAPO * 8);
    L.inbox = take(R * (size_t)CAPT * 8);
    L.list2 = take((size_t)FB2_ROWS * FB2CAP * 8);
    L.gdense = take((size_t)L.t3cap * (size_t)N * 4);
    L.total = o;
    return L;
}

template <int H>
int launch_rsym(const float *xp, int64_t N, int64_t row0, int64_t row1, float t, uint32_t s0, uint32_t s1, const float *klim,
                int32_t *idx, float *val, void *ws, hipStream_t st) {
    const int64_t R = row1 - row0;
    const RsLayout L = make_layout(R, N);
    char *w = reinterpret_cast<char *>(ws);
    RsCtl *ctl = reinterpret_cast<RsCtl *>(w + L.ctl);
    int *cntO = reinterpret_cast<int *>(w + L.cntO), *cntT = reinterpret_cast<int *>(w + L.cntT);
    int *fail2rows = reinterpret_cast<int *>(w + L.fail2rows), *fail3rows = reinterpret_cast<int *>(w + L.fail3rows);
    int *slot2 = reinterpret_cast<int *>(w + L.slot2), *slot3 = reinterpret_cast<int *>(w + L.slot3);
    int *cnt2 = reinterpret_cast<int *>(w + L.cnt2);
    float *need2 = reinterpret_cast<float *>(w + L.need2);
    int2 *own = reinterpret_cast<int2 *>(w + L.own), *inbox = reinterpret_cast<int2 *>(w + L.inbox), *list2 = reinterpret_cast<int2 *>(w + L.list2);
    float *gdense = reinterpret_cast<float *>(w + L.gdense);
    // tuning / test knobs, read per call: expected pairs of a row above the tier-1 threshold (default 1.25 mean(L) + 6), how many
    // times more pairs tier 2 may ask for, statistics, and DGG_RSYM_SMALL=0: small graphs through the tiers as well
    const char *e_t = getenv("DGG_RSYM_TARGET"), *e_d = getenv("DGG_RSYM_DEPTH2"), *e_s = getenv("DGG_RSYM_STATS"), *e_m = getenv("DGG_RSYM_SMALL");
    const float target = e_t ? (float)atof(e_t) : 0.0f;
    const float maxdepth = e_d ? fmaxf((float)atof(e_d), 1.0f) : MAXDEPTH;
    const int stats_on = e_s ? atoi(e_s) : 0;
    const int force_small = e_m ? atoi(e_m) : -1;
    if (dgg_check_hip(hipMemsetAsync(ctl, 0, sizeof(RsCtl), st), "rsym memset") != 0) return DGG_ERR_HIP;
    if (dgg_check_hip(hipMemsetAsync(slot2, 0xFF, L.cntT - L.slot2, st), "rsym memset") != 0) return DGG_ERR_HIP;
    if (dgg_check_hip(hipMemsetAsync(cntT, 0, L.cntO - L.cntT, st), "rsym memset") != 0) return DGG_ERR_HIP;
    const dim3 gown((unsigned)((N + 3) / 4)), grow((unsigned)((R + 3) / 4));
    const bool small = N <= SMALL_N && force_small != 0;
#define RS_EMIT_ARGS xp, N, row0, row1, t, s0, s1, ctl, cntO, cntT, own, inbox, (const int *)slot2, (const float *)need2, cnt2, list2, \
                     (const int *)slot3, gdense, L.t3cap, stats_on
    if (small) {
        hipLaunchKernelGGL(rs_all3, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st, row0, R, ctl, slot3, fail3rows);
    } else {
        hipLaunchKernelGGL(rs_pilot<H>, dim3(PILOT_WG), dim3(256), 0, st, xp, N, t, s0, s1, klim, R, ctl);
        hipLaunchKernelGGL(rs_thresh, dim3(1), dim3(1), 0, st, N, R, target, maxdepth, ctl);
        hipLaunchKernelGGL((rs_emit<H, 1>), gown, dim3(256), 0, st, RS_EMIT_ARGS);
        hipLaunchKernelGGL(rs_finalize, grow, dim3(256), 0, st, N, row0, row1, t, klim, ctl, (const int *)cntO, (const int *)cntT,
                           (const int2 *)own, (const int2 *)inbox, slot2, need2, fail2rows, slot3, fail3rows, L.t3cap, idx, val);
        hipLaunchKernelGGL((rs_emit<H, 2>), gown, dim3(256), 0, st, RS_EMIT_ARGS);
        hipLaunchKernelGGL(rs_finalize2<H>, dim3(FB2_ROWS / 4), dim3(256), 0, st, xp, N, row0, t, klim, ctl, (const int *)cnt2,
                           (const int2 *)list2, (const float *)need2, (const int *)fail2rows, slot3, fail3rows, L.t3cap, idx, val);
    }
    hipLaunchKernelGGL((rs_emit<H, 3>), gown, dim3(256), 0, st, RS_EMIT_ARGS);
#undef RS_EMIT_ARGS
    const int g3 = L.t3cap < 1024 ? (L.t3cap > 0 ? L.t3cap : 1) : 1024;
    hipLaunchKernelGGL(rs_rows3<H>, dim3((unsigned)g3), dim3(256), 0, st, xp, N, row0, t, klim, (const RsCtl *)ctl, (const int *)fail3rows,
                       (const float *)gdense, L.t3cap, idx, val);
    return dgg_check_launch("allpairs_topk_rsym");
}

}  // namespace

size_t dgg_allpairs_rsym_ws_bytes(int64_t rows, int64_t N) { return make_layout(rows, N).total; }

bool dgg_allpairs_rsym_supported(int h, int K) { return K == 64 && (h == 8 || h == 16 || h == 32 || h == 64 || h == 128); }

int dgg_allpairs_topk_rsym_impl(const float *xp, int64_t N, int h, int64_t row0, int64_t row1, float t, uint32_t s0, uint32_t s1, int K,
                                const float *klim, int32_t *idx, float *val, void *workspace, size_t ws_bytes, hipStream_t st) {
    if (!dgg_allpairs_rsym_supported(h, K))
        return dgg_set_error(DGG_ERR_UNSUPPORTED, "ranked symmetric noise needs K = 64 and latent_dim in {8,16,32,64,128}");
    if (N >= ((int64_t)1 << 31)) return dgg_set_error(DGG_ERR_UNSUPPORTED, "ranked symmetric noise needs N < 2^31");
    if (!workspace || ws_bytes < dgg_allpairs_rsym_ws_bytes(row1 - row0, N))
        return dgg_set_error(DGG_ERR_ARG, "ranked symmetric noise: workspace too small (dgg_allpairs_workspace_bytes)");
    if (row1 <= row0) {
        // an empty row shard launches nothing, but the caller reads the status words (error flag, tier-3 count, walk depth) out of
        // the workspace it just allocated: they must read "nothing happened", not uninitialised memory
        const RsLayout L = make_layout(0, N);
        return dgg_check_hip(hipMemsetAsync(reinterpret_cast<char *>(workspace) + L.ctl, 0, sizeof(RsCtl), st), "rsym memset");
    }
    switch (h) {
        case 8: return launch_rsym<8>(xp, N, row0, row1, t, s0, s1, klim, idx, val, workspace, st);
        case 16: return launch_rsym<16>(xp, N, row0, row1, t, s0, s1, klim, idx, val, workspace, st);
        case 32: return launch_rsym<32>(xp, N, row0, row1, t, s0, s1, klim, idx, val, workspace, st);
        case 64: return launch_rsym<64>(xp, N, row0, row1, t, s0, s1, klim, idx, val, workspace, st);
        default: return launch_rsym<128>(xp, N, row0, row1, t, s0, s1, klim, idx, val, workspace, st);
    }
}
