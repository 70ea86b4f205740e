// dgg_topk_ranked.hip -- all-pairs top-64 under the RANKED noise generator (noise_mode 4): O(N * ~150) instead of O(N^2).
//
// Same contract as the other all-pairs kernels (reference dgm.py:1618-1623, 1213-1229, 1404):
//   p'_ij = exp(log(exp(-0.05 ||xp_i - xp_j||) + 1e-8) + G_ij),  64 largest per row, (score desc, column asc).
// With the ranked generator (dgg_common.h) a row's noise is produced in DECREASING order: rank s has
// G_(s) = -0.3 log(sum_{t<=s} E_t/(N-t+1)) and sits at column sigma_i(.).  Since log p'_ij <= G_ij + 1e-8 for every
// distance, a row is finished as soon as the noise of the next rank cannot reach the row's current 64th log-score:
// every pair not yet visited is provably out.  A row therefore visits ~2-3 blocks of 64 ranks (about 113 / 0.76
// columns at N = 100k) instead of N columns, scoring each visited pair exactly (canonical fp32 arithmetic).
//
// One wavefront per row; lane = one position r' of the keyed bijection.  Per block: sigma_i, ballot-compaction of the
// positions that fall inside [0,N) (their order is the rank order), exponential spacing terms, a 64-bit wavefront
// prefix scan carried across blocks, exact scores (gathering the candidate rows of xp), DPP bitonic sort + merge.
#include "dgg_common.h"
#include "dgg_api_internal.h"

using namespace dgg;

namespace {

#ifdef DGG_RK_SHFLSCAN
__device__ __forceinline__ uint64_t wave_inclusive_scan_u64(uint64_t v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t lo = __shfl_up((uint32_t)v, off, 64), hi = __shfl_up((uint32_t)(v >> 32), off, 64);
        uint64_t t = ((uint64_t)hi << 32) | lo;
        if (lane >= off) v += t;
    }
    return v;
}
#else
// 64-lane inclusive prefix sum of 64-bit integers on the DPP network: Kogge-Stone inside the rows of 16 lanes (row_shr 1, 2, 4, 8; a
// lane without a source reads 0), then the row totals by row_bcast:15 (lane 15 of rows 0 / 2 into rows 1 / 3) and row_bcast:31 (lane 31
// into rows 2 and 3).  Two DPP moves + one 64-bit add per step, no LDS crossbar (the shuffle form: twelve ds_bpermute round trips).
template <int CTRL, int ROWMASK>
__device__ __forceinline__ uint64_t dpp_add_u64(uint64_t v) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, ROWMASK, 0xF, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), CTRL, ROWMASK, 0xF, false);
    return v + (((uint64_t)hi << 32) | lo);
}
#ifdef DGG_RK_MOVSCAN
__device__ __forceinline__ uint64_t wave_inclusive_scan_u64(uint64_t v, int) {
    v = dpp_add_u64<0x111, 0xF>(v);                              // row_shr:1
    v = dpp_add_u64<0x112, 0xF>(v);                              // row_shr:2
    v = dpp_add_u64<0x114, 0xF>(v);                              // row_shr:4
    v = dpp_add_u64<0x118, 0xF>(v);                              // row_shr:8
    v = dpp_add_u64<0x142, 0xA>(v);                              // row_bcast:15 -> rows 1, 3
    v = dpp_add_u64<0x143, 0xC>(v);                              // row_bcast:31 -> rows 2, 3
    return v;
}
#else
// ... with the DPP operand ON the add: v_add_co_u32_dpp / v_addc_co_u32_dpp take the shifted lane as source 0 (bound_ctrl: a lane
// without a source adds 0; a row outside row_mask is not written), two instructions per step where the builtin form above compiles
// to two v_mov + two v_mov_dpp + a 64-bit add.  Hand-scheduled hazards (the compiler does not look inside): a DPP read of a register
// needs two wait states after the VALU write -- the other half's add and one s_nop sit in between.
#define DGG_DPP_ADD64(ctrl)                                                                              \
    "v_add_co_u32_dpp %0, vcc, %0, %0 " ctrl " bound_ctrl:0\n"                                           \
    "v_addc_co_u32_dpp %1, vcc, %1, %1, vcc " ctrl " bound_ctrl:0\n"                                     \
    "s_nop 0\n"
__device__ __forceinline__ uint64_t wave_inclusive_scan_u64(uint64_t v, int) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    asm volatile("s_nop 1\n"
                 DGG_DPP_ADD64("row_shr:1 row_mask:0xf bank_mask:0xf")
                 DGG_DPP_ADD64("row_shr:2 row_mask:0xf bank_mask:0xf")
                 DGG_DPP_ADD64("row_shr:4 row_mask:0xf bank_mask:0xf")
                 DGG_DPP_ADD64("row_shr:8 row_mask:0xf bank_mask:0xf")
                 DGG_DPP_ADD64("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 DGG_DPP_ADD64("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 : "+v"(lo), "+v"(hi) : : "vcc");
    return ((uint64_t)hi << 32) | lo;
}
#undef DGG_DPP_ADD64
#endif
#endif

// One block of the ranked walk of row i: positions rb .. rb + 63 of the row's sequence.  Position 0 is the row's OWN column with its
// independent variate (dgg_common.h, DGG_RANKED_DIAG_KEY); position p >= 1 is slot sigma(p - 1) of the n = N - 1 other columns, a
// candidate when the slot is < n.  -> col, cand (the lane holds a candidate), G (its noise), mv (ballot of the lanes that hold a
// RANK, i.e. all candidates but the diagonal); S / scount carry the fixed-point prefix sum and the rank count across blocks.
struct RankedLane { uint32_t col; bool cand; float G; };
typedef float f32x2 __attribute__((ext_vector_type(2)));
// log(exp(t sqrt(d2)) + 1e-8) to ~1e-5 absolute from the raw transcendental instructions (bound tests only, never a score)
__device__ __forceinline__ float fast_logp_bound(float d2, float t) {
    const float e = __builtin_amdgcn_exp2f(t * 1.44269504f * __builtin_amdgcn_sqrtf(d2));
    return 0.693147181f * __builtin_amdgcn_logf(e + 1e-8f);
}
__device__ __forceinline__ RankedLane ranked_block(uint64_t rb, int lane, uint32_t i, int64_t n, int b, uint64_t D, uint32_t k1, uint32_t k2, uint32_t k3,
                                                   uint64_t &S, uint32_t &scount, uint64_t &mv) {
    const uint64_t p = rb + (uint64_t)lane;
    const bool isdiag = p == 0ull;
    const uint32_t c = ranked_sigma((uint32_t)(p - 1ull), k1, k2, k3, b);
    const bool rvalid = !isdiag && p <= D && (int64_t)c < n;
    mv = __ballot(rvalid);
    const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(mv >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mv, 0u));
    const uint32_t s = scount + pos + 1;                         // 1-based rank of this lane's slot (if it holds one)
    // the diagonal's exponential variate comes out of the same instruction stream: rank 1 of 1 under its own key (no division effect)
    const uint64_t term = ranked_term(k1, isdiag ? (k3 ^ DGG_RANKED_DIAG_KEY) : k3, isdiag ? 1u : s, isdiag ? (int64_t)1 : n);
    const uint64_t pre = wave_inclusive_scan_u64(rvalid ? term : 0ull, lane) + S;
    RankedLane o;
    o.G = ranked_gumbel(isdiag ? term : pre);
    o.cand = rvalid || isdiag;
    o.col = isdiag ? i : c + (c >= i ? 1u : 0u);
    S = shfl_u64(pre, 63);                                       // inclusive sum at the last lane = block total + carry
    scount += (uint32_t)__builtin_popcountll(mv);
    return o;
}

// PROBE (dgg_allpairs_ranked_probe): the same walk on every `stride`-th row with a budget of `max_blocks` blocks, NO list written;
// probe[0..5] += rows walked, blocks visited, candidates gathered, candidates scored in full, rows that hit the budget; probe[5] =
// max blocks of a row.  The walk depth depends on the DATA -- a row visits ~ L exp(spread of 0.05 dist / 0.3) ranks -- and this is
// how callers measure it (bench.py) or estimate it before choosing this generator (dgm.py, args.dgg_asym_generator = "auto").
#ifndef DGG_RK_WAVES
#define DGG_RK_WAVES 1
#endif
template <int H, bool PROBE = false>
__global__ __launch_bounds__(256, DGG_RK_WAVES) void allpairs_topk_ranked(const float *__restrict__ xp, int64_t N, int64_t row0,
                                                            int64_t row1, float t, uint32_t s0, uint32_t s1,
                                                            const float *__restrict__ klim, int32_t *__restrict__ idx,
                                                            float *__restrict__ val, int softk_mode, float *__restrict__ w_out,
                                                            float *__restrict__ rs_out, const uint32_t *__restrict__ seed_dev,
                                                            int stride = 1, int max_blocks = 0,
                                                            unsigned long long *__restrict__ probe = nullptr,
                                                            const float *__restrict__ lpub = nullptr) {
    const int lane = threadIdx.x & 63;
    if (seed_dev) { s0 = seed_dev[0]; s1 = seed_dev[1]; }       // seed in device memory: ONE captured hipGraph serves fresh seeds
    // (readfirstlane: the compiler cannot know that dgg::wave_id() is wave-uniform; without it the row's own features are
    //  fetched with VECTOR loads into 64 registers instead of scalar loads)
    const int64_t lrow = ((int64_t)blockIdx.x * 4 + dgg::wave_id()) * (PROBE ? stride : 1);
    const int64_t i = row0 + lrow;
    if (i >= row1) return;
    unsigned long long nblk = 0, ngath = 0, nsc = 0;
    bool budget_hit = false;
    // ranks beyond L cannot receive weight (klimit_len): the search only has to settle the first L of the list
    const int L = klim ? __builtin_amdgcn_readfirstlane(klimit_len(klim[lrow], 64)) : 64;
    uint32_t k1, k2;
    rowkey(s0, s1, (uint32_t)i, k1, k2);
    const uint32_t k3 = mix32(k2 ^ 0x68E31DA4u);
    const int64_t n = N - 1;                                     // ranks = the OTHER columns; the row's own column is position 0
    const int b = ranked_bits(n);
    const uint64_t D = (uint64_t)1 << b;
    const float *xi = xp + i * H;                                // wave-uniform row
    // upper bound of log p_ij over the other nodes j: the distance-free log(1 + 1e-8), or dgg_allpairs_rowmin_bound's (every rank the
    // walk has not reached is another node: the diagonal comes first)
    const float lp = lpub ? lpub[lrow] : 1e-8f;
    uint64_t list = DGG_EMPTY_KEY;
    uint64_t S = 0;                                              // fixed-point prefix sum carried across blocks
    uint32_t scount = 0;                                         // ranks assigned so far
    float thr_log = -INFINITY;                                   // log of the L-th best score so far (-inf: list not full)
    for (uint64_t rb = 0; rb <= D; rb += 64) {
        uint64_t m;
        const RankedLane rl = ranked_block(rb, lane, (uint32_t)i, n, b, D, k1, k2, k3, S, scount, m);
        const uint32_t c = rl.col;
        const float G = rl.G;
        uint64_t key = DGG_EMPTY_KEY;
        // per-candidate version of the stop test: a rank whose noise cannot reach the L-th log-score found so far is
        // not gathered at all (ranks come in decreasing noise order, so these are the tail lanes of the block)
        const bool want = rl.cand && !(G + lp + 1e-3f < thr_log);
        if (PROBE) { nblk++; ngath += __builtin_popcountll(__ballot(want)); }
        // (Measured and rejected: staging the candidate rows through LDS so that 8 lanes read one 128-byte line -- 8x fewer L1 tag
        //  lookups, same bits -- costs two LDS round trips per block: 658 us against 257.)
        if (want) {
            const float4 *xj = reinterpret_cast<const float4 *>(xp + (int64_t)c * H);
            float d2 = 0.0f;
            auto chain = [&](int c8) {
                float4 b0 = xj[2 * c8], b1 = xj[2 * c8 + 1];
#ifdef DGG_RK_SCALAR_SUB
                float df;
                df = __fadd_rn(xi[8 * c8 + 0], -b0.x); d2 = __fmaf_rn(df, df, d2);
                df = __fadd_rn(xi[8 * c8 + 1], -b0.y); d2 = __fmaf_rn(df, df, d2);
                df = __fadd_rn(xi[8 * c8 + 2], -b0.z); d2 = __fmaf_rn(df, df, d2);
                df = __fadd_rn(xi[8 * c8 + 3], -b0.w); d2 = __fmaf_rn(df, df, d2);
                df = __fadd_rn(xi[8 * c8 + 4], -b1.x); d2 = __fmaf_rn(df, df, d2);
                df = __fadd_rn(xi[8 * c8 + 5], -b1.y); d2 = __fmaf_rn(df, df, d2);
                df = __fadd_rn(xi[8 * c8 + 6], -b1.z); d2 = __fmaf_rn(df, df, d2);
                df = __fadd_rn(xi[8 * c8 + 7], -b1.w); d2 = __fmaf_rn(df, df, d2);
#else
                // the differences two at a time (v_pk_add_f32: the same IEEE subtraction, half the issue slots); the squares still
                // enter the ONE ascending fmaf chain of the canonical order
                const f32x2 a0 = {xi[8 * c8 + 0], xi[8 * c8 + 1]}, a1 = {xi[8 * c8 + 2], xi[8 * c8 + 3]};
                const f32x2 a2 = {xi[8 * c8 + 4], xi[8 * c8 + 5]}, a3 = {xi[8 * c8 + 6], xi[8 * c8 + 7]};
                const f32x2 e0 = a0 - f32x2{b0.x, b0.y}, e1 = a1 - f32x2{b0.z, b0.w};
                const f32x2 e2 = a2 - f32x2{b1.x, b1.y}, e3 = a3 - f32x2{b1.z, b1.w};
                d2 = __fmaf_rn(e0.x, e0.x, d2); d2 = __fmaf_rn(e0.y, e0.y, d2);
                d2 = __fmaf_rn(e1.x, e1.x, d2); d2 = __fmaf_rn(e1.y, e1.y, d2);
                d2 = __fmaf_rn(e2.x, e2.x, d2); d2 = __fmaf_rn(e2.y, e2.y, d2);
                d2 = __fmaf_rn(e3.x, e3.x, d2); d2 = __fmaf_rn(e3.y, e3.y, d2);
#endif
            };
            // first cache line of the row (32 features): the running sum of the fmaf chain only grows, so sqrt of the
            // partial sum is a rigorous lower bound of the distance -- if even that cannot reach the L-th log-score, the
            // second line is never fetched
            constexpr int HEAD = H >= 64 ? 4 : H / 8;
#pragma unroll
            for (int c8 = 0; c8 < HEAD; c8++) chain(c8);
            bool alive = true;
            // log p' = G + log(exp(t dist) + 1e-8) is decreasing in dist: evaluate it at the lower bound (fast math + margin)
#ifdef DGG_RK_SLOW_BOUND
            if (HEAD < H / 8) alive = !(G + __logf(__expf(t * sqrtf(d2)) + 1e-8f) + 1e-3f < thr_log);
#else
            // (the hardware's sqrt / exp2 / log2 as they are -- 1 ulp each, no denormal paths: the margin is 1e-3 -- and no test at
            //  all while the list is not full: every candidate of the first block is scored anyway)
            if (HEAD < H / 8 && thr_log > -INFINITY) alive = !(G + fast_logp_bound(d2, t) + 1e-3f < thr_log);
#endif
            if (alive) {
#pragma unroll
                for (int c8 = HEAD; c8 < H / 8; c8++) chain(c8);
                key = make_key(score_from_dist(c_sqrt(d2), t, true, G), (int32_t)c);
            }
        }
        // Settle the block's keys into the list (descending, one entry per lane).  The kernel is bound by vector-instruction issue
        // and the 64-lane bitonic sort + merge is ~270 of its ~650 instructions per block, whatever the number of live keys:
        //  * first block: the list is empty -- sort descending, no merge;
        //  * a later block with few live keys (the usual case: the per-candidate cuts leave a handful): insert them one by one
        //    (position = number of larger entries, the tail shifts down one lane) -- ~10 instructions per key;
        //  * otherwise sort + merge.  All three produce the same list (keys are unique).
        uint64_t live = __ballot(key != DGG_EMPTY_KEY);
        const int nlive = __builtin_popcountll(live);
        if (PROBE) nsc += nlive;
        if (rb == 0) {
            list = wave_sort<true>(key, lane);
        } else if (nlive <= 16) {
            while (live != 0ull) {                               // wave-uniform
                const int src = __builtin_ctzll(live);
                live &= live - 1;
                const uint64_t kk = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(key >> 32), src) << 32) |
                                    (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key, src);
                const int pos = __builtin_popcountll(__ballot(list > kk));
                const uint64_t prev = ((uint64_t)(uint32_t)__shfl_up((int)(list >> 32), 1, 64) << 32) | (uint32_t)__shfl_up((int)(uint32_t)list, 1, 64);
                list = lane < pos ? list : (lane == pos ? kk : prev);
            }
        } else {
            key = wave_sort<false>(key, lane);
            list = wave_merge_top64_asc(list, key, lane);
        }
        const int nvalid = __builtin_popcountll(m);
        if ((int64_t)scount >= n) break;                         // every column visited
        // stop test: the lowest noise of this block bounds every rank still to come
        const uint64_t k63 = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(list >> 32), L - 1) << 32) |
                             (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)list, L - 1);
        if (k63 != DGG_EMPTY_KEY) thr_log = __logf(key_val(k63));
        if (k63 != DGG_EMPTY_KEY && nvalid > 0) {
            const int last = 63 - __builtin_clzll(m);            // last lane with a rank = highest rank in the block
            const float gmin = __shfl(G, last, 64);
            if (gmin + lp + 1e-3f < thr_log) break;
        }
        if (PROBE && max_blocks > 0 && nblk >= (unsigned long long)max_blocks) { budget_hit = true; break; }
    }
    if (PROBE) {
        if (lane == 0) {
            atomicAdd(&probe[0], 1ull); atomicAdd(&probe[1], nblk); atomicAdd(&probe[2], ngath); atomicAdd(&probe[3], nsc);
            if (budget_hit) atomicAdd(&probe[4], 1ull);
            atomicMax(&probe[5], nblk);
        }
        return;
    }
    const bool empty = list == DGG_EMPTY_KEY || lane >= L;
    idx[lrow * 64 + lane] = empty ? -1 : key_col(list);
    const float sv = empty ? 0.0f : key_val(list);
    val[lrow * 64 + lane] = sv;
    if (w_out) {
        // smooth first-k ramp on the settled list (softk_fwd_kernel, dgg_ell.hip; reference dgm.py:1410-1420) while the scores are
        // still in registers: same operations, same bits; saves re-reading idx / val and a launch
        const float f = c_ramp((float)lane, klim[lrow]);
        float v = f;
        if (softk_mode == 0 || softk_mode == 3) {
            const float a = __fmul_rn(sv, f);
            v = softk_mode == 0 ? a : __fadd_rn(__fadd_rn(f, -a), a);
        }
        const float wv = empty ? 0.0f : v;
        w_out[lrow * 64 + lane] = wv;
        const float s_ = wave_sum_butterfly(wv);
        if (lane == 0) rs_out[lrow] = s_;
    }
}

template <int H>
int launch_ranked(const float *xp, int64_t N, int64_t row0, int64_t row1, float t, uint32_t s0, uint32_t s1,
                  const float *klim, int32_t *idx, float *val, hipStream_t st, int softk_mode = 0, float *w = nullptr, float *rs = nullptr,
                  const uint32_t *seed_dev = nullptr, const float *lpub = nullptr) {
    dim3 grid((unsigned)((row1 - row0 + 3) / 4));
    hipLaunchKernelGGL(allpairs_topk_ranked<H>, grid, dim3(256), 0, st, xp, N, row0, row1, t, s0, s1, klim, idx, val, softk_mode, w, rs, seed_dev, 1, 0,
                       (unsigned long long *)nullptr, lpub);
    return dgg_check_launch("allpairs_topk_ranked");
}

// ---- rows WIDER than 64 ranks ("chunked rows") -----------------------------------------------------------------------------
// The learned degree is unbounded (k = relu(kp sd + mu) + 1, reference dgm.py:1580-1584) and the reference ramps over the whole dense
// row (dgm.py:1402-1421): a row needs its first L_i = ceil(k_i + 8.5) + 1 ranks, whatever k_i.  Chunked layout: node i owns the
// M_i = ceil(L_i / 64) consecutive 64-entry CHUNKS [cptr[i], cptr[i+1]) of idx / val / w; rank r of the row is lane r % 64 of its chunk
// r / 64.  With every M_i = 1 this IS the [rows,64] list of the kernels above, and every chunk is a row of that layout to the
// kernels downstream (partition, aggregation, backward), which only need the node of a chunk (cnode).
//
// chunk_partial + chunk_layout: k -> cptr (exclusive scan of M_i), cnode, meta = {total chunks, max M_i, flags}; flags bit 0: some
// row's ramp support exceeds 64 * maxm ranks (it is cut there -- harmless when 64 * maxm covers every column of the graph, otherwise
// callers raise, never truncate silently), bit 2: some learned degree is NaN, bit 1: more chunks than
// `ccap` (cptr is CLAMPED to ccap -- rows beyond it own no chunk -- so no consumer leaves the arrays; re-run with a larger capacity).  Two passes of 128
// workgroups (segment totals, then the scan of each segment from its base): ~10 us at 100 000 rows (one workgroup: 96 us).
constexpr int CL_NB = 128, CL_T = 256;                          // workgroups / threads of the two layout passes
__device__ __forceinline__ int64_t cl_segment(int64_t rows) { return ((rows + CL_NB - 1) / CL_NB + CL_T - 1) / CL_T * CL_T; }

// pass 1: per-segment totals of M_i, the widest row, the "beyond 64 * maxm ranks" flag -> part[3 * CL_NB]
__global__ __launch_bounds__(CL_T) void chunk_partial(const float *__restrict__ k, int64_t rows, int maxm, int32_t *__restrict__ part) {
    __shared__ int ws[CL_T / 64][3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t per = cl_segment(rows), lo = (int64_t)blockIdx.x * per, hi = lo + per < rows ? lo + per : rows;
    const int kcap = 64 * maxm;
    int s = 0, mx = 0, flag = 0;
    for (int64_t i = lo + tid; i < hi; i += CL_T) {
        const float kk = k[i];
        if (!(ceilf(kk + 8.5f) + 1.0f <= (float)kcap)) flag |= 1;       // (also NaN)
        if (kk != kk) flag |= 4;                                         // a learned degree that is not a number
        const int m = (klimit_len(kk, kcap) + 63) >> 6;
        s += m;
        mx = m > mx ? m : mx;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        s += __shfl_xor(s, off, 64);
        const int v = __shfl_xor(mx, off, 64);
        mx = v > mx ? v : mx;
        flag |= __shfl_xor(flag, off, 64);
    }
    if (lane == 0) { ws[wave][0] = s; ws[wave][1] = mx; ws[wave][2] = flag; }
    __syncthreads();
    if (tid == 0) {
        int ts = 0, tm = 0, tf = 0;
        for (int q = 0; q < CL_T / 64; q++) { ts += ws[q][0]; tm = ws[q][1] > tm ? ws[q][1] : tm; tf |= ws[q][2]; }
        part[3 * blockIdx.x] = ts; part[3 * blockIdx.x + 1] = tm; part[3 * blockIdx.x + 2] = tf;
    }
}
// pass 2: every workgroup scans the segment totals for its base, then its own segment in tiles of CL_T rows (coalesced loads, a
// workgroup scan per tile, the running sum carried): cptr, cnode; workgroup 0 writes the totals; all zero cnode beyond the last chunk
__global__ __launch_bounds__(CL_T) void chunk_layout(const float *__restrict__ k, int64_t rows, int maxm, int64_t ccap,
                                                     const int32_t *__restrict__ part, int32_t *__restrict__ cptr,
                                                     int32_t *__restrict__ cnode, int32_t *__restrict__ meta, int32_t *__restrict__ sticky) {
    __shared__ int wsum[CL_T / 64], sbase, stotal, smax, sflag;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t per = cl_segment(rows), lo = (int64_t)blockIdx.x * per, hi = lo + per < rows ? lo + per : rows;
    const int kcap = 64 * maxm;
    if (wave == 0) {                                             // CL_NB = 128 partial sums: two per lane
        int b0 = 0, tot = 0, tm = 0, tf = 0;
#pragma unroll
        for (int q = 0; q < CL_NB / 64; q++) {
            const int g = q * 64 + lane;
            const int v = part[3 * g];
            if (g < (int)blockIdx.x) b0 += v;
            tot += v;
            tm = part[3 * g + 1] > tm ? part[3 * g + 1] : tm;
            tf |= part[3 * g + 2];
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            b0 += __shfl_xor(b0, off, 64);
            tot += __shfl_xor(tot, off, 64);
            const int v = __shfl_xor(tm, off, 64);
            tm = v > tm ? v : tm;
            tf |= __shfl_xor(tf, off, 64);
        }
        if (lane == 0) { sbase = b0; stotal = tot; smax = tm; sflag = tf; }
    }
    __syncthreads();
    int run = sbase;
    const int total = stotal;
    for (int64_t i0 = lo; i0 < hi; i0 += CL_T) {
        const int64_t i = i0 + tid;
        const int m = i < hi ? (klimit_len(k[i], kcap) + 63) >> 6 : 0;
        int incl = m;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(incl, off, 64);
            if (lane >= off) incl += v;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int wbase = 0, tsum = 0;
#pragma unroll
        for (int q = 0; q < CL_T / 64; q++) {
            const int v = wsum[q];
            if (q < wave) wbase += v;
            tsum += v;
        }
        const int first = run + wbase + incl - m;
        if (i < hi) {
            // MEMORY SAFETY under a fixed capacity: a row that starts beyond ccap owns no chunk, one that straddles it is cut there --
            // every consumer walks [cptr[i], cptr[i+1]) and stays inside arrays of ccap chunks; flags bit 1 reports the overflow
            cptr[i] = first < ccap ? first : (int)ccap;
            for (int c = 0; c < m; c++)
                if (first + c < ccap) cnode[first + c] = (int32_t)i;
        }
        run += tsum;
        __syncthreads();
    }
    for (int64_t c = (int64_t)total + (int64_t)blockIdx.x * CL_T + tid; c < ccap; c += (int64_t)CL_NB * CL_T) cnode[c] = 0;   // chunks beyond the last one: node 0, empty
    if (blockIdx.x == 0 && tid == 0) {
        const int flags = sflag | (total > ccap ? 2 : 0);
        cptr[rows] = total < ccap ? total : (int)ccap;
        meta[0] = total;                                         // (the chunks NEEDED: what a caller re-sizes its arrays to)
        meta[1] = smax;
        meta[2] = flags;
        meta[3] = 0;
        // the flags of THIS call are overwritten by the next one (a replayed hipGraph); `sticky` is only ever ORed into and keeps
        // every overflow since the host last cleared it
        if (sticky && flags) atomicOr(sticky, flags);
    }
}

__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int uniform_lane) {
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), uniform_lane) << 32) |
           (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, uniform_lane);
}

// The search of allpairs_topk_ranked with M_i <= MAXM descending 64-lane lists per row in registers (list[0] holds ranks 0..63, list[1]
// ranks 64..127, ...).  A block's live keys are sorted and cascaded down the lists: merged with list[m] by one bitonic half-cleaner
// (upper 64 stay, lower 64 carry on to list[m+1]); a list whose smallest key beats the carry's largest is skipped, and the cascade ends
// when the carry is empty.  Stop test, per-candidate cut and partial-distance bound as above, against the row's L_i-th log-score.
// Output: chunk c = cptr[i] + m takes list[m] (idx -1 / val 0 beyond L_i), w = ramp(64 m + lane - k_i) x score, rs_i = the wavefront's
// butterfly over the per-lane sums of its chunks (the oracle's butterfly_sum for K > 64).
template <int H, int MAXM>
__global__ __launch_bounds__(256) void allpairs_topk_ranked_wide(const float *__restrict__ xp, int64_t N, int64_t row0, int64_t row1, float t,
                                                                 uint32_t s0, uint32_t s1, const float *__restrict__ klim,
                                                                 const int32_t *__restrict__ cptr, int32_t *__restrict__ idx,
                                                                 float *__restrict__ val, int softk_mode, float *__restrict__ w_out,
                                                                 float *__restrict__ rs_out, const uint32_t *__restrict__ seed_dev,
                                                                 unsigned nrow_blocks, int64_t ccap, int skip_heavy,
                                                                 const float *__restrict__ lpub) {
    const int lane = threadIdx.x & 63;
    if (blockIdx.x >= nrow_blocks) {
        // arrays allocated for `ccap` chunks (a capacity fixed ahead of the learned degrees, e.g. inside a captured hipGraph): the
        // chunks beyond the last one are written EMPTY, so that every consumer may walk all ccap chunks
        const int64_t q = (int64_t)cptr[row1 - row0] + (int64_t)(blockIdx.x - nrow_blocks) * 4 + dgg::wave_id();
        if (q < ccap) {
            idx[q * 64 + lane] = -1;
            val[q * 64 + lane] = 0.0f;
            if (w_out) w_out[q * 64 + lane] = 0.0f;
        }
        return;
    }
    if (seed_dev) { s0 = seed_dev[0]; s1 = seed_dev[1]; }
    const int64_t lrow = (int64_t)blockIdx.x * 4 + dgg::wave_id();
    const int64_t i = row0 + lrow;
    if (i >= row1) return;
    const int c0 = __builtin_amdgcn_readfirstlane(cptr[lrow]);
    int Mi = __builtin_amdgcn_readfirstlane(cptr[lrow + 1]) - c0;
    if (Mi <= 0) {                                               // (a fixed capacity ran out before this row: it owns no chunk; flagged by chunk_layout)
        if (w_out && lane == 0) rs_out[lrow] = 0.0f;
        return;
    }
    if (Mi > MAXM) {
        if (skip_heavy) return;                                  // more lists than a wavefront holds in registers: dgg_allpairs_topk_anywide settles the row
        Mi = MAXM;
    }
    const float ki = klim[lrow];
    const int L = __builtin_amdgcn_readfirstlane(klimit_len(ki, 64 * Mi));
    const int mL = (L - 1) >> 6, laneL = (L - 1) & 63;
    uint32_t k1, k2;
    rowkey(s0, s1, (uint32_t)i, k1, k2);
    const uint32_t k3 = mix32(k2 ^ 0x68E31DA4u);
    const int64_t n = N - 1;                                     // (ranked_block: the row's own column first, then the ranks of the others)
    const int b = ranked_bits(n);
    const uint64_t D = (uint64_t)1 << b;
    const float *xi = xp + i * H;
    const float lp = lpub ? lpub[lrow] : 1e-8f;                  // upper bound of log p over the OTHER nodes (allpairs_topk_ranked)
    uint64_t list[MAXM];
#pragma unroll
    for (int m = 0; m < MAXM; m++) list[m] = DGG_EMPTY_KEY;
    uint64_t S = 0;
    uint32_t scount = 0;
    float thr_log = -INFINITY;
    for (uint64_t rb = 0; rb <= D; rb += 64) {
        uint64_t mv;
        const RankedLane rl = ranked_block(rb, lane, (uint32_t)i, n, b, D, k1, k2, k3, S, scount, mv);
        const uint32_t c = rl.col;
        const float G = rl.G;
        uint64_t key = DGG_EMPTY_KEY;
        const bool want = rl.cand && !(G + lp + 1e-3f < thr_log);
        if (want) {
            const float4 *xj = reinterpret_cast<const float4 *>(xp + (int64_t)c * H);
            float d2 = 0.0f;
            auto chain = [&](int c8) {
                float4 b0 = xj[2 * c8], b1 = xj[2 * c8 + 1];
#ifdef DGG_RK_SCALAR_SUB
                float df;
                df = __fadd_rn(xi[8 * c8 + 0], -b0.x); d2 = __fmaf_rn(df, df, d2);
                df = __fadd_rn(xi[8 * c8 + 1], -b0.y); d2 = __fmaf_rn(df, df, d2);
                df = __fadd_rn(xi[8 * c8 + 2], -b0.z); d2 = __fmaf_rn(df, df, d2);
                df = __fadd_rn(xi[8 * c8 + 3], -b0.w); d2 = __fmaf_rn(df, df, d2);
                df = __fadd_rn(xi[8 * c8 + 4], -b1.x); d2 = __fmaf_rn(df, df, d2);
                df = __fadd_rn(xi[8 * c8 + 5], -b1.y); d2 = __fmaf_rn(df, df, d2);
                df = __fadd_rn(xi[8 * c8 + 6], -b1.z); d2 = __fmaf_rn(df, df, d2);
                df = __fadd_rn(xi[8 * c8 + 7], -b1.w); d2 = __fmaf_rn(df, df, d2);
#else
                // the differences two at a time (v_pk_add_f32: the same IEEE subtraction, half the issue slots); the squares still
                // enter the ONE ascending fmaf chain of the canonical order
                const f32x2 a0 = {xi[8 * c8 + 0], xi[8 * c8 + 1]}, a1 = {xi[8 * c8 + 2], xi[8 * c8 + 3]};
                const f32x2 a2 = {xi[8 * c8 + 4], xi[8 * c8 + 5]}, a3 = {xi[8 * c8 + 6], xi[8 * c8 + 7]};
                const f32x2 e0 = a0 - f32x2{b0.x, b0.y}, e1 = a1 - f32x2{b0.z, b0.w};
                const f32x2 e2 = a2 - f32x2{b1.x, b1.y}, e3 = a3 - f32x2{b1.z, b1.w};
                d2 = __fmaf_rn(e0.x, e0.x, d2); d2 = __fmaf_rn(e0.y, e0.y, d2);
                d2 = __fmaf_rn(e1.x, e1.x, d2); d2 = __fmaf_rn(e1.y, e1.y, d2);
                d2 = __fmaf_rn(e2.x, e2.x, d2); d2 = __fmaf_rn(e2.y, e2.y, d2);
                d2 = __fmaf_rn(e3.x, e3.x, d2); d2 = __fmaf_rn(e3.y, e3.y, d2);
#endif
            };
            constexpr int HEAD = H >= 64 ? 4 : H / 8;
#pragma unroll
            for (int c8 = 0; c8 < HEAD; c8++) chain(c8);
            bool alive = true;
#ifdef DGG_RK_SLOW_BOUND
            if (HEAD < H / 8) alive = !(G + __logf(__expf(t * sqrtf(d2)) + 1e-8f) + 1e-3f < thr_log);
#else
            // (the hardware's sqrt / exp2 / log2 as they are -- 1 ulp each, no denormal paths: the margin is 1e-3 -- and no test at
            //  all while the list is not full: every candidate of the first block is scored anyway)
            if (HEAD < H / 8 && thr_log > -INFINITY) alive = !(G + fast_logp_bound(d2, t) + 1e-3f < thr_log);
#endif
            if (alive) {
#pragma unroll
                for (int c8 = HEAD; c8 < H / 8; c8++) chain(c8);
                key = make_key(score_from_dist(c_sqrt(d2), t, true, G), (int32_t)c);
            }
        }
        if (__ballot(key != DGG_EMPTY_KEY) != 0ull) {
            uint64_t carry = wave_sort<true>(key, lane);
            if (rb == 0) {
                list[0] = carry;
            } else {
#pragma unroll
                for (int m = 0; m < MAXM; m++) {
                    if (m < Mi) {                                    // (wave-uniform)
                        const uint64_t cmax = readlane_u64(carry, 0);
                        if (cmax == DGG_EMPTY_KEY) break;            // nothing left to place
                        if (cmax > readlane_u64(list[m], 63)) {      // (else: list[m] keeps all of its entries, the carry goes on whole)
                            const uint64_t rc = shfl_u64(carry, 63 - lane);          // ascending
                            const uint64_t hi = list[m] > rc ? list[m] : rc, lo = list[m] > rc ? rc : list[m];
                            list[m] = bitonic_block<64, 32, true>(hi, lane);
                            carry = bitonic_block<64, 32, true>(lo, lane);
                        }
                    }
                }
            }
        }
        const int nvalid = __builtin_popcountll(mv);
        if ((int64_t)scount >= n) break;
        uint64_t kL = DGG_EMPTY_KEY;
#pragma unroll
        for (int m = 0; m < MAXM; m++)
            if (m == mL) kL = readlane_u64(list[m], laneL);
        if (kL != DGG_EMPTY_KEY) thr_log = __logf(key_val(kL));
        if (kL != DGG_EMPTY_KEY && nvalid > 0) {
            const int last = 63 - __builtin_clzll(mv);
            const float gmin = __shfl(G, last, 64);
            if (gmin + lp + 1e-3f < thr_log) break;
        }
    }
    float rsum = 0.0f;
#pragma unroll
    for (int m = 0; m < MAXM; m++) {
        if (m < Mi) {
            const int r = 64 * m + lane;
            const bool empty = list[m] == DGG_EMPTY_KEY || r >= L;
            const int64_t e = ((int64_t)c0 + m) * 64 + lane;
            idx[e] = empty ? -1 : key_col(list[m]);
            const float sv = empty ? 0.0f : key_val(list[m]);
            val[e] = sv;
            if (w_out) {
                const float f = c_ramp((float)r, ki);
                float v = f;
                if (softk_mode == 0 || softk_mode == 3) {
                    const float a = __fmul_rn(sv, f);
                    v = softk_mode == 0 ? a : __fadd_rn(__fadd_rn(f, -a), a);
                }
                const float wv = empty ? 0.0f : v;
                w_out[e] = wv;
                rsum = m == 0 ? wv : __fadd_rn(rsum, wv);
            }
        }
    }
    if (w_out) {
        const float s_ = wave_sum_butterfly(rsum);
        if (lane == 0) rs_out[lrow] = s_;
    }
}

template <int H>
int launch_ranked_wide(int maxm, const float *xp, int64_t N, int64_t row0, int64_t row1, float t, uint32_t s0, uint32_t s1, const float *klim,
                       const int32_t *cptr, int32_t *idx, float *val, hipStream_t st, int softk_mode, float *w, float *rs, const uint32_t *seed_dev,
                       int64_t ccap, const float *lpub) {
    const int skip_heavy = maxm > DGG_CHUNK_MAXM ? 1 : 0;
    const unsigned nrow_blocks = (unsigned)((row1 - row0 + 3) / 4);
    const int64_t tail = ccap > row1 - row0 ? ccap - (row1 - row0) : 0;         // (every row has at least one chunk)
    dim3 grid(nrow_blocks + (unsigned)((tail + 3) / 4));
#define DGG_RW(MM) hipLaunchKernelGGL((allpairs_topk_ranked_wide<H, MM>), grid, dim3(256), 0, st, xp, N, row0, row1, t, s0, s1, klim, cptr, idx, val, \
                                      softk_mode, w, rs, seed_dev, nrow_blocks, ccap, skip_heavy, lpub)
    if (maxm <= 2) DGG_RW(2);
    else if (maxm <= 4) DGG_RW(4);
    else if (maxm <= 8) DGG_RW(8);
    else if (maxm <= 16) DGG_RW(16);
    else DGG_RW(32);
#undef DGG_RW
    return dgg_check_launch("allpairs_topk_ranked_wide");
}

// same cut for the evaluators that always settle all K ranks
__global__ void klimit_truncate(const float *__restrict__ klim, int64_t rows, int K, int32_t *__restrict__ idx,
                                float *__restrict__ val) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * K) return;
    if ((int)(e % K) >= klimit_len(klim[e / K], K)) { idx[e] = -1; val[e] = 0.0f; }
}

}  // namespace

int dgg_klimit_truncate_impl(const float *klim, int64_t rows, int K, int32_t *idx, float *val, hipStream_t st) {
    if (rows == 0) return 0;
    hipLaunchKernelGGL(klimit_truncate, dim3((unsigned)((rows * K + 255) / 256)), dim3(256), 0, st, klim, rows, K, idx, val);
    return dgg_check_launch("klimit_truncate");
}

int dgg_allpairs_topk_ranked_impl(const float *xp, int64_t N, int h, int64_t row0, int64_t row1, float t, uint32_t s0,
                                  uint32_t s1, int K, const float *klim, int32_t *idx, float *val, hipStream_t st, int softk_mode,
                                  float *w, float *rs, const uint32_t *seed_dev, const float *lpub) {
    if (K != 64) return dgg_set_error(DGG_ERR_UNSUPPORTED, "ranked-noise path needs K = 64");
    if (N >= ((int64_t)1 << 31)) return dgg_set_error(DGG_ERR_UNSUPPORTED, "ranked-noise path needs N < 2^31");
    if (w && (!klim || !rs)) return dgg_set_error(DGG_ERR_ARG, "ranked-noise path: the fused ramp needs the learned k and the row-sum output");
    if (row1 <= row0) return 0;
    switch (h) {
        case 8: return launch_ranked<8>(xp, N, row0, row1, t, s0, s1, klim, idx, val, st, softk_mode, w, rs, seed_dev, lpub);
        case 16: return launch_ranked<16>(xp, N, row0, row1, t, s0, s1, klim, idx, val, st, softk_mode, w, rs, seed_dev, lpub);
        case 32: return launch_ranked<32>(xp, N, row0, row1, t, s0, s1, klim, idx, val, st, softk_mode, w, rs, seed_dev, lpub);
        case 64: return launch_ranked<64>(xp, N, row0, row1, t, s0, s1, klim, idx, val, st, softk_mode, w, rs, seed_dev, lpub);
        case 128: return launch_ranked<128>(xp, N, row0, row1, t, s0, s1, klim, idx, val, st, softk_mode, w, rs, seed_dev, lpub);
        default: return dgg_set_error(DGG_ERR_UNSUPPORTED, "ranked-noise path supports latent_dim in {8,16,32,64,128}");
    }
}

extern "C" {
// Walk statistics of the ranked search (noise_mode 4) on every `stride`-th row of [row0, row1), each walk cut after `max_blocks`
// blocks of 64 ranks (0: no cut); nothing but the six counters is written: probe[0..4] += rows walked, blocks visited, candidates
// gathered, candidates scored in full, rows cut by the budget; probe[5] = max(probe[5], blocks of a row).  Caller zeroes probe.
int dgg_allpairs_ranked_probe(const float *xp, int64_t N, int h, int64_t row0, int64_t row1, float t, uint32_t s0, uint32_t s1,
                              const float *k_limit, int stride, int max_blocks, unsigned long long *probe, const float *lpub, void *stream) {
    if (row0 < 0 || row1 > N || row0 > row1 || stride < 1 || max_blocks < 0 || !probe)
        return dgg_set_error(DGG_ERR_ARG, "allpairs_ranked_probe: bad row range, stride, budget or NULL counters");
    if (N >= ((int64_t)1 << 31)) return dgg_set_error(DGG_ERR_UNSUPPORTED, "ranked-noise path needs N < 2^31");
    if (row1 == row0) return 0;
    const int64_t nrows = (row1 - row0 + stride - 1) / stride;
    const dim3 grid((unsigned)((nrows + 3) / 4));
    hipStream_t st = (hipStream_t)stream;
#define DGG_RANKED_PROBE(HH)                                                                                               \
    hipLaunchKernelGGL((allpairs_topk_ranked<HH, true>), grid, dim3(256), 0, st, xp, N, row0, row1, t, s0, s1, k_limit, nullptr, nullptr, 0, \
                       nullptr, nullptr, nullptr, stride, max_blocks, probe, lpub)
    switch (h) {
        case 8: DGG_RANKED_PROBE(8); break;
        case 16: DGG_RANKED_PROBE(16); break;
        case 32: DGG_RANKED_PROBE(32); break;
        case 64: DGG_RANKED_PROBE(64); break;
        case 128: DGG_RANKED_PROBE(128); break;
        default: return dgg_set_error(DGG_ERR_UNSUPPORTED, "ranked-noise path supports latent_dim in {8,16,32,64,128}");
    }
#undef DGG_RANKED_PROBE
    return dgg_check_launch("allpairs_ranked_probe");
}

// dgg_allpairs_topk (noise_mode DGG_NOISE_RANKED, K = 64, k_limit required) FUSED with dgg_softk_fwd: also w [rows,64] and
// rs [rows] (mode as dgg_softk_fwd: 0 k_times_edge_prob, 1 k_only, 3 straight-through hard).  Same bits as the two calls.
int dgg_allpairs_topk_ranked_softk(const float *xp, int64_t N, int h, int64_t row0, int64_t row1, float t, uint32_t s0, uint32_t s1,
                                   const float *k, int mode, int32_t *idx, float *val, float *w, float *rs, void *stream) {
    if (row0 < 0 || row1 > N || row0 > row1) return dgg_set_error(DGG_ERR_ARG, "allpairs_topk_ranked_softk: bad row range");
    if (mode != 0 && mode != 1 && mode != 3) return dgg_set_error(DGG_ERR_ARG, "allpairs_topk_ranked_softk: mode must be 0, 1 or 3");
    if (!k || !w || !rs) return dgg_set_error(DGG_ERR_ARG, "allpairs_topk_ranked_softk: k, w and rs are required");
    return dgg_allpairs_topk_ranked_impl(xp, N, h, row0, row1, t, s0, s1, 64, k, idx, val, (hipStream_t)stream, mode, w, rs);
}
// the same with the noise seed (s0, s1) read from DEVICE memory at launch time: a captured hipGraph of the step then draws fresh
// noise on every replay (the caller advances seed_dev between replays, e.g. by a captured increment), as training does per forward
int dgg_allpairs_topk_ranked_softk_dseed(const float *xp, int64_t N, int h, int64_t row0, int64_t row1, float t, const uint32_t *seed_dev,
                                         const float *k, int mode, int32_t *idx, float *val, float *w, float *rs, void *stream) {
    if (row0 < 0 || row1 > N || row0 > row1) return dgg_set_error(DGG_ERR_ARG, "allpairs_topk_ranked_softk: bad row range");
    if (mode != 0 && mode != 1 && mode != 3) return dgg_set_error(DGG_ERR_ARG, "allpairs_topk_ranked_softk: mode must be 0, 1 or 3");
    if (!k || !w || !rs || !seed_dev) return dgg_set_error(DGG_ERR_ARG, "allpairs_topk_ranked_softk_dseed: k, w, rs and seed_dev are required");
    return dgg_allpairs_topk_ranked_impl(xp, N, h, row0, row1, t, 0u, 0u, 64, k, idx, val, (hipStream_t)stream, mode, w, rs, seed_dev);
}

// dgg_allpairs_topk_ranked_softk[_dseed] with the rows' upper bounds of log p over their OTHER nodes (lpub [row1-row0], nullable:
// dgg_allpairs_rowmin_bound): the walk stops -- and skips candidates -- on  G + lpub[i]  instead of the distance-free  G + 1e-8.  Same
// result, fewer ranks walked when the latent distances spread over several noise scales.  seed_dev != NULL: the seed is read from device
// memory (s0, s1 ignored).
int dgg_allpairs_topk_ranked_softk_lp(const float *xp, int64_t N, int h, int64_t row0, int64_t row1, float t, uint32_t s0, uint32_t s1,
                                      const uint32_t *seed_dev, const float *lpub, const float *k, int mode, int32_t *idx, float *val, float *w,
                                      float *rs, void *stream) {
    if (row0 < 0 || row1 > N || row0 > row1) return dgg_set_error(DGG_ERR_ARG, "allpairs_topk_ranked_softk_lp: bad row range");
    if (mode != 0 && mode != 1 && mode != 3) return dgg_set_error(DGG_ERR_ARG, "allpairs_topk_ranked_softk_lp: mode must be 0, 1 or 3");
    if (!k || !w || !rs) return dgg_set_error(DGG_ERR_ARG, "allpairs_topk_ranked_softk_lp: k, w and rs are required");
    return dgg_allpairs_topk_ranked_impl(xp, N, h, row0, row1, t, s0, s1, 64, k, idx, val, (hipStream_t)stream, mode, w, rs, seed_dev, lpub);
}

// ---- chunked rows (rows wider than 64 ranks) ----
// Layout of the chunked rows from the learned degrees: cptr [rows+1] (first chunk of every node), cnode [ccap] (node of every chunk),
// meta int32[4 + 384] = {total chunks, max chunks of a row, flags (1: a row needs more than 64*maxm ranks, 2: more than ccap chunks), 0,
// scratch of the two-pass scan}.
int dgg_chunk_layout(const float *k, int64_t rows, int maxm, int64_t ccap, int32_t *cptr, int32_t *cnode, int32_t *meta, int32_t *sticky,
                     void *stream) {
    if (rows < 0 || maxm < 1 || maxm > DGG_CHUNK_MAXM_ANY || ccap < 0 || ccap >= ((int64_t)1 << 31) || !k || !cptr || !cnode || !meta)
        return dgg_set_error(DGG_ERR_ARG, "chunk_layout: bad sizes or NULL arrays (maxm in 1..2^20, ccap < 2^31)");
    if (rows >= ((int64_t)1 << 25)) return dgg_set_error(DGG_ERR_UNSUPPORTED, "chunk_layout: rows < 2^25");
    // (the segment totals of the first pass live in meta[4 .. 4 + 3 * 128): meta is int32 [4 + 384])
    hipLaunchKernelGGL(chunk_partial, dim3(CL_NB), dim3(CL_T), 0, (hipStream_t)stream, k, rows, maxm, meta + 4);
    hipLaunchKernelGGL(chunk_layout, dim3(CL_NB), dim3(CL_T), 0, (hipStream_t)stream, k, rows, maxm, ccap, meta + 4, cptr, cnode, meta, sticky);
    return dgg_check_launch("chunk_layout");
}
// dgg_allpairs_topk_ranked_softk[_dseed] for chunked rows: idx / val / w are [chunks, 64] (chunk c of node i = ranks 64 (c - cptr[i]) ...),
// rs [rows]; maxm = the largest chunk count of a row (meta[1] of dgg_chunk_layout, or the capacity it was called with); ccap = chunks the
// arrays were allocated for (>= cptr[rows]; the chunks beyond the last one are written empty).  seed_dev != NULL: the seed is read from
// device memory (s0, s1 ignored).
int dgg_allpairs_topk_ranked_wide(const float *xp, int64_t N, int h, int64_t row0, int64_t row1, float t, uint32_t s0, uint32_t s1,
                                  const uint32_t *seed_dev, const float *k, int mode, int maxm, const int32_t *cptr, int64_t ccap, int32_t *idx,
                                  float *val, float *w, float *rs, const float *lpub, void *stream) {
    if (row0 < 0 || row1 > N || row0 > row1) return dgg_set_error(DGG_ERR_ARG, "allpairs_topk_ranked_wide: bad row range");
    if (mode != 0 && mode != 1 && mode != 3) return dgg_set_error(DGG_ERR_ARG, "allpairs_topk_ranked_wide: mode must be 0, 1 or 3");
    if (!k || !cptr || !idx || !val || (w && !rs)) return dgg_set_error(DGG_ERR_ARG, "allpairs_topk_ranked_wide: k, cptr, idx, val (and rs with w) are required");
    // maxm > 32: the rows of up to 32 chunks are settled here, the wider ones are LEFT to dgg_allpairs_topk_anywide (noise_mode 4)
    if (maxm < 1 || maxm > DGG_CHUNK_MAXM_ANY) return dgg_set_error(DGG_ERR_ARG, "allpairs_topk_ranked_wide: maxm in 1..2^20");
    if (N >= ((int64_t)1 << 31)) return dgg_set_error(DGG_ERR_UNSUPPORTED, "ranked-noise path needs N < 2^31");
    if (row1 == row0) return 0;
    hipStream_t st = (hipStream_t)stream;
    switch (h) {
        case 16: return launch_ranked_wide<16>(maxm, xp, N, row0, row1, t, s0, s1, k, cptr, idx, val, st, mode, w, rs, seed_dev, ccap, lpub);
        case 32: return launch_ranked_wide<32>(maxm, xp, N, row0, row1, t, s0, s1, k, cptr, idx, val, st, mode, w, rs, seed_dev, ccap, lpub);
        case 64: return launch_ranked_wide<64>(maxm, xp, N, row0, row1, t, s0, s1, k, cptr, idx, val, st, mode, w, rs, seed_dev, ccap, lpub);
        case 128: return launch_ranked_wide<128>(maxm, xp, N, row0, row1, t, s0, s1, k, cptr, idx, val, st, mode, w, rs, seed_dev, ccap, lpub);
        default: return dgg_set_error(DGG_ERR_UNSUPPORTED, "chunked ranked-noise path supports latent_dim in {16,32,64,128}");
    }
}
}
