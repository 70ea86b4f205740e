// dgg_topk_anywide.hip -- all-pairs top-L_i for CHUNKED rows of ANY width, under every noise generator with random access to a pair's
// noise (none / per-pair hash / symmetric per-pair hash) and, for the rows the register lists of dgg_topk_ranked.hip cannot hold
// (more than 32 chunks = 2048 ranks), under the ranked generator.
//
// Contract (reference dgm.py:1402-1421 select_top_k on the dense row, 1580-1584 the unbounded learned degree, 1211-1231 the
// perturbation, 1618-1623 the scores): row i keeps the L_i = ceil(k_i + 8.5) + 1 best columns by
//   p'_ij = exp(log(exp(-0.05 ||xp_i - xp_j||) + 1e-8) + G_ij)      (G = 0 and no log round trip when unperturbed)
// in (score desc, column asc) order, in the M_i = ceil(L_i / 64) chunks [cptr[i], cptr[i+1]) of idx / val / w (dgg_chunk_layout), with
// the first-k ramp and the row sums of dgg_allpairs_topk_ranked_wide: same bits as the oracle's ora_allpairs_topk(K = 64 M) + ora_softk.
//
// How.  Every row owns a THRESHOLD BUFFER of 128 keys per chunk (cap_i = 128 M_i >= 2 L_i) in the workspace.  A producer streams the
// row's candidates, appends every key above the row's current threshold, and when the buffer is full SELECTS the L_i-th largest key
// (bisection on the 64-bit keys: a count pass per step over an L2-resident buffer), keeps the L_i keys above it and raises the
// threshold: a row of N candidates is compacted ~log2(N / L_i) times and admits ~L_i log2(N / L_i) keys in all.  Producers:
//   aw_scan_plain    unperturbed scores: every pair scored canonically, 16 rows per wavefront over LDS-staged column tiles (the
//                    exhaustive kernel's loop, dgg_topk.hip, with the running top-64 replaced by the buffer)
//   aw_scan_hash     per-pair hash noise (symmetric or not): log p' <= G + 1e-8 whatever the distance, so a pair whose RAW HASH lies
//                    below the integer image of the row's threshold is dropped by one unsigned compare (dgg_topk_gv.hip's filter, here
//                    against the row's own moving threshold); the survivors (a fraction of a per cent) are queued in LDS as (row,
//                    column) pairs and scored 64 at a time with full lanes, their features gathered from global memory -- the N^2
//                    part of the kernel is ~12 integer instructions per 64 pairs and touches no feature
//   aw_ranked_walk   ranked generator, rows of more than 32 chunks: the walk of allpairs_topk_ranked_wide (ranks in decreasing noise
//                    order, stop when the next rank's noise cannot reach the threshold) with the buffer in place of the register lists
//   aw_pilot_rows / aw_hash_sweep / aw_hash_score: the hash generators' guess-and-verify FRONT END (below), which leaves only the rows it cannot
//                    settle to aw_scan_hash
// aw_emit (one workgroup per row) sorts the surviving <= L_i keys in LDS (bitonic; rows beyond 4096 keys in buckets of 4096 split off
// by further selections), writes idx / val / w per chunk and the row sum in the chunk-ordered, lane-wise + butterfly order.
// Exactness: a key is dropped only when L_i better keys of the same row are known; the hash filter and the walk's stop test drop a
// pair only when its noise alone, with margins, cannot reach a threshold that is itself the score of a known L_i-th best key.
#include "dgg_common.h"
#include "dgg_api_internal.h"
#include <stdlib.h>

using namespace dgg;

namespace {

constexpr int KSLOT = 128;              // keys of threshold buffer per chunk of 64 ranks
struct HashCtl { int nfail; int pad; unsigned long long capsum; };   // rows a front end handed on; sum of their buffer capacities (keys)
constexpr int PLAIN_FEW_MAX = 2048;
// segments per listed row of the segmented scan (aw_plain_rows_part): the most the scratch admits for the capacities listed
constexpr int PSEG_MAX = 32;
__device__ __forceinline__ int plain_segments(unsigned long long capsum, long long scratch_keys) {
    const long long s = scratch_keys / (long long)(capsum > 0ull ? capsum : 1ull);
    return s < 1 ? 0 : (s > PSEG_MAX ? PSEG_MAX : (int)s);      // 0: the listed rows do not fit at all (aw_scan_plain takes them)
}     // unperturbed front end: failing rows up to which the one-wavefront-per-row fallback is used

__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { const uint64_t o = shfl_xor_u64(v, off); v = o < v ? o : v; }
    return v;
}
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { const uint64_t o = shfl_xor_u64(v, off); v = o > v ? o : v; }
    return v;
}

// A threshold tau (one of the keys) with L <= #{keys >= tau} <= Lmax over buf[0..n), n >= L >= 1, keys distinct and non-zero; Lmax = L:
// the L-th largest key exactly.  One wavefront, buf in global memory (L2-resident).  Bisection on the key values with a count pass per
// step, each a round trip to L2: mid-scan compactions take a WINDOW (any threshold that keeps at least L and leaves room in the buffer is
// as good: the step count falls from ~log2(n) + 1 to ~log2(n / (Lmax - L))), the final one of a row is exact.
__device__ __noinline__ uint64_t wave_select_lth(const uint64_t *__restrict__ buf, int n, int L, int Lmax, int lane) {
    uint64_t mn = ~0ull, mx = 0ull;
    for (int e = lane; e < n; e += 64) { const uint64_t k = buf[e]; mn = k < mn ? k : mn; mx = k > mx ? k : mx; }
    mn = wave_min_u64(mn);
    mx = wave_max_u64(mx);
    if (L >= n || Lmax >= n) return mn;
    uint64_t lo = mn, hi = mx + 1ull;                            // count(>= lo) > Lmax, count(>= hi) < L
    while (hi - lo > 1ull) {
        const uint64_t mid = lo + ((hi - lo) >> 1);
        int c = 0;
        uint64_t m2 = ~0ull;
        int e = lane;
        for (; e + 192 < n; e += 256) {                          // four independent loads in flight
            const uint64_t k0 = buf[e], k1 = buf[e + 64], k2 = buf[e + 128], k3 = buf[e + 192];
            c += (k0 >= mid) + (k1 >= mid) + (k2 >= mid) + (k3 >= mid);
            if (k0 >= mid && k0 < m2) m2 = k0;
            if (k1 >= mid && k1 < m2) m2 = k1;
            if (k2 >= mid && k2 < m2) m2 = k2;
            if (k3 >= mid && k3 < m2) m2 = k3;
        }
        for (; e < n; e += 64) { const uint64_t k = buf[e]; c += k >= mid; if (k >= mid && k < m2) m2 = k; }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off, 64);
        if (c >= L && c <= Lmax) return wave_min_u64(m2);        // the smallest key at or above mid keeps exactly c keys
        if (c > Lmax) lo = mid; else hi = mid;
    }
    return lo;
}
// keeps the keys >= tau at the front of buf (order preserved) -> their count.  Four batches of 64 keys per step, the next step's loads
// issued before this step's stores (they land on positions already read: a store never reaches a key that is still to be loaded)
__device__ __noinline__ int wave_compact_ge(uint64_t *__restrict__ buf, int n, uint64_t tau, int lane) {
    int at = 0;
    uint64_t cur[4], nxt[4];
#pragma unroll
    for (int q = 0; q < 4; q++) { const int e = 64 * q + lane; cur[q] = e < n ? buf[e] : 0ull; }
    for (int base = 0; base < n; base += 256) {
#pragma unroll
        for (int q = 0; q < 4; q++) { const int e = base + 256 + 64 * q + lane; nxt[q] = e < n ? buf[e] : 0ull; }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const bool keep = cur[q] >= tau && cur[q] != 0ull;   // (positions beyond n were loaded as 0, no key is 0)
            const uint64_t m = __ballot(keep);
            const int pos = at + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (keep) buf[pos] = cur[q];
            at += __builtin_popcountll(m);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) cur[q] = nxt[q];
    }
    return at;
}
// window of a mid-scan compaction: keep between L and L + L / 4 + 16 keys, never more than leaves 64 free slots
__device__ __forceinline__ int keep_window(int L, int cap) {
    int hi = L + (L >> 2) + 16;
    hi = hi > cap - 64 ? cap - 64 : hi;
    return hi < L ? L : hi;
}

// pairs whose raw 32-bit hash is below the result have noise G < gmin - 1e-3 (dgg_topk_gv.hip, hash_threshold_from_gmin): P(G >= g) =
// 1 - exp(-e^(-g / 0.3)) <= e^(-g / 0.3) =: e1, i.e. G >= g needs U >= 1 - e1; two units of 2^-24 and 1e-3 of margin cover the rounding
__device__ __forceinline__ uint32_t hash_threshold_from_gmin(float gmin) {
    const float e1 = __expf((gmin - 1e-3f) * (-1.0f / 0.3f));
    const float c = fminf(e1 * 16777216.0f, 16777216.0f);
    int um = 16777216 - (int)c - 2;
    um = um < 0 ? 0 : um;
    return (uint32_t)um << 8;
}
// integer image of a row's key threshold tau (the score of its L-th best key so far): a pair can only beat tau if its log-score
// reaches log(score(tau)); log p' <= G + 1e-8, so its noise has to reach that minus the margins (fast-math log: 1e-3 covers it)
// (lp: upper bound of log p_ij of the row over j != i -- 1e-8 without a distance bound, dgg_allpairs_rowmin_bound's otherwise; the
//  diagonal pair, at distance 0, is passed by the callers whatever its hash)
__device__ __forceinline__ uint32_t uthr_from_key(uint64_t tau, float lp) {
    if (tau == DGG_EMPTY_KEY) return 0u;
    return hash_threshold_from_gmin(__logf(fmaxf(key_val(tau), 1e-37f)) - lp - 1e-3f);
}

struct RowGeom { int64_t base; int cap, L; };                    // buffer of a row: keys[base .. base + cap), L ranks to settle
__device__ __forceinline__ RowGeom row_geom(const int32_t *__restrict__ cptr, const float *__restrict__ klim, int64_t lrow) {
    RowGeom g;
    const int c0 = cptr[lrow], M = cptr[lrow + 1] - c0;
    g.base = (int64_t)c0 * KSLOT;
    g.cap = M * KSLOT;
    g.L = M > 0 ? klimit_len(klim[lrow], 64 * M) : 0;
    return g;
}

// ---- unperturbed scores: every pair scored, 16 rows per wavefront ---------------------------------------------------------------
constexpr int RW = 16, WAVES = 4, RB = RW * WAVES, TN = 64;
template <int H>
__global__ __launch_bounds__(WAVES * 64) void aw_scan_plain(const float *__restrict__ xp, int64_t N, int64_t row0, int64_t row1, float t,
                                                            const float *__restrict__ klim, const int32_t *__restrict__ cptr,
                                                            uint64_t *__restrict__ keys, int32_t *__restrict__ cnt_out,
                                                            const int32_t *__restrict__ rowlist, const HashCtl *__restrict__ ctl, int few_max,
                                                            long long scratch_keys) {
    // rowlist != NULL: the rows of this launch are rowlist[0 .. *nlist) (local row ids: the rows aw_plain_score could not settle), else
    // every row of [row0, row1)
    __shared__ float colT[H * TN];
    __shared__ float rowsL[RB * H];
    __shared__ int s_row[RB];                                    // local row id of every row slot of the workgroup, -1: none
    const int tid = threadIdx.x, lane = tid & 63, wave = dgg::wave_id();
    const int64_t nsel = rowlist ? (int64_t)ctl->nfail : row1 - row0;
    // (few listed rows whose segments fit the scratch: aw_plain_rows_part / _merge settled them)
    if ((int64_t)blockIdx.x * RB >= nsel || (rowlist && nsel <= few_max && plain_segments(ctl->capsum, scratch_keys) > 0)) return;
    if (tid < RB) {
        const int64_t q = (int64_t)blockIdx.x * RB + tid;
        s_row[tid] = q < nsel ? (rowlist ? rowlist[q] : (int)q) : -1;
    }
    __syncthreads();
    for (int e = tid; e < RB * H; e += WAVES * 64) {
        const int r = e / H, c = e % H;
        const int lr_ = s_row[r];
        rowsL[e] = lr_ >= 0 ? xp[(row0 + lr_) * H + c] : 0.0f;
    }
    uint64_t thr[RW];
    int cnt[RW];
#pragma unroll
    for (int r = 0; r < RW; r++) { thr[r] = DGG_EMPTY_KEY; cnt[r] = 0; }
    for (int64_t j0 = 0; j0 < N; j0 += TN) {
        __syncthreads();
        for (int e = tid; e < TN * H; e += WAVES * 64) {
            const int jj = e / H, c = e % H;
            const int64_t gj = j0 + jj;
            colT[c * TN + jj] = gj < N ? xp[gj * H + c] : 0.0f;
        }
        __syncthreads();
        float xj[H];
#pragma unroll
        for (int c = 0; c < H; c++) xj[c] = colT[c * TN + lane];
        const int64_t j = j0 + lane;
        const bool jvalid = j < N;
#pragma unroll
        for (int r = 0; r < RW; r++) {
            const int lr = wave * RW + r;
            const int li = __builtin_amdgcn_readfirstlane(s_row[lr]);
            if (li < 0) continue;                                // wave-uniform
            const int64_t i = row0 + li;
            const float *xi = rowsL + lr * H;
            float d2 = 0.0f;
#pragma unroll
            for (int c = 0; c < H; c++) {
                const float df = __fadd_rn(xi[c], -xj[c]);
                d2 = __fmaf_rn(df, df, d2);
            }
            const float v = score_from_dist(c_sqrt(d2), t, false, 0.0f);
            const uint64_t key = jvalid ? make_key(v, (int32_t)j) : DGG_EMPTY_KEY;
            bool pass = key > thr[r];
            uint64_t m = __ballot(pass);
            if (m != 0ull) {                                     // wave-uniform
                const RowGeom g = row_geom(cptr, klim, i - row0);
                if (g.cap == 0) continue;                        // (a fixed capacity ran out before this row: no chunk)
                uint64_t *buf = keys + g.base;
                if (cnt[r] + __builtin_popcountll(m) > g.cap) {  // full: keep the L best, raise the threshold
                    thr[r] = wave_select_lth(buf, cnt[r], g.L, keep_window(g.L, g.cap), lane);
                    cnt[r] = wave_compact_ge(buf, cnt[r], thr[r], lane);
                    pass = key > thr[r];
                    m = __ballot(pass);
                }
                const int pos = cnt[r] + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (pass) buf[pos] = key;
                cnt[r] += __builtin_popcountll(m);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RW; r++) {
        const int li = __builtin_amdgcn_readfirstlane(s_row[wave * RW + r]);
        if (li < 0) continue;
        const int64_t i = row0 + li;
        const RowGeom g = row_geom(cptr, klim, i - row0);
        int n = cnt[r];
        if (n > g.L && g.cap > 0) {
            uint64_t *buf = keys + g.base;
            const uint64_t tau = wave_select_lth(buf, n, g.L, g.L, lane);
            n = wave_compact_ge(buf, n, tau, lane);
        }
        if (lane == 0) cnt_out[i - row0] = n;
    }
}

// ---- unperturbed scores, FRONT END's second half: the candidates of dgg_topk_sweep.hip's radius sweep scored, verified -------------------
// (pw_pilot / pw_sweep: every pair that is not in the row's candidate sub-lists has d^2 > rad[i]).  One wavefront per row: exact canonical
// squared distances and scores of the candidates streamed through the row's threshold buffer, its L best kept; VERIFIED when the L-th
// best lies inside the radius with margins (sw_finalize's test): rows that fail -- a radius guessed too small, a sub-list that
// overflowed -- are listed for aw_scan_plain (every pair scored: exact whatever the guess was).
template <int H>
__global__ __launch_bounds__(256) void aw_plain_score(const float *__restrict__ xp, int64_t N, int64_t row0, int64_t row1, float t,
                                                      const float *__restrict__ klim, const int32_t *__restrict__ cptr,
                                                      const uint32_t *__restrict__ cand, const int32_t *__restrict__ ncand, int nsub, int cslot,
                                                      const float *__restrict__ rad, uint64_t *__restrict__ keys, int32_t *__restrict__ cnt_out,
                                                      int32_t *__restrict__ nfail, int32_t *__restrict__ faillist,
                                                      unsigned long long *__restrict__ capsum, unsigned long long *__restrict__ failoff) {
    const int lane = threadIdx.x & 63;
    const int64_t lrow = (int64_t)blockIdx.x * 4 + dgg::wave_id();
    const int64_t i = row0 + lrow;
    if (i >= row1) return;
    const RowGeom g = row_geom(cptr, klim, lrow);
    if (g.cap == 0) { if (lane == 0) cnt_out[lrow] = 0; return; }
    const int M = g.cap / KSLOT, capsub = M * cslot / nsub;
    const float R = rad[lrow];
    bool ok = R < 3.0e38f && t < 0.0f;
    int C = 0;
    for (int q = 0; q < nsub; q++) {
        const int c = ncand[lrow * nsub + q];
        ok = ok && c <= capsub;
        C += c;
    }
    ok = ok && (C >= g.L || (int64_t)C >= N);
    uint64_t *buf = keys + g.base;
    const uint32_t *cl = cand + (int64_t)(g.base / KSLOT) * cslot;
    int n = 0;
    if (ok) {
        const float *xi = xp + i * H;                            // wave-uniform: scalar loads
        uint64_t thr = DGG_EMPTY_KEY;
        for (int q = 0; q < nsub; q++) {
            const int nq = ncand[lrow * nsub + q];
            for (int base = 0; base < nq; base += 64) {
                const int e = base + lane;
                uint64_t key = DGG_EMPTY_KEY;
                if (e < nq) {
                    const uint32_t j = cl[(int64_t)q * capsub + e];
                    const float4 *xj = reinterpret_cast<const float4 *>(xp + (int64_t)j * H);
                    float4 bq[H / 4];
#pragma unroll
                    for (int c4 = 0; c4 < H / 4; c4++) bq[c4] = xj[c4];
                    float d2 = 0.0f;
#pragma unroll
                    for (int c4 = 0; c4 < H / 4; c4++) {
                        float df;
                        df = __fadd_rn(xi[4 * c4 + 0], -bq[c4].x); d2 = __fmaf_rn(df, df, d2);
                        df = __fadd_rn(xi[4 * c4 + 1], -bq[c4].y); d2 = __fmaf_rn(df, df, d2);
                        df = __fadd_rn(xi[4 * c4 + 2], -bq[c4].z); d2 = __fmaf_rn(df, df, d2);
                        df = __fadd_rn(xi[4 * c4 + 3], -bq[c4].w); d2 = __fmaf_rn(df, df, d2);
                    }
                    key = (int64_t)j < N ? make_key(score_from_dist(c_sqrt(d2), t, false, 0.0f), (int32_t)j) : DGG_EMPTY_KEY;
                }
                bool pass = key != DGG_EMPTY_KEY && key > thr;
                uint64_t m = __ballot(pass);
                if (m != 0ull) {
                    if (n + __builtin_popcountll(m) > g.cap) {
                        thr = wave_select_lth(buf, n, g.L, keep_window(g.L, g.cap), lane);
                        n = wave_compact_ge(buf, n, thr, lane);
                        pass = pass && key > thr;
                        m = __ballot(pass);
                    }
                    const int p = n + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    if (pass) buf[p] = key;
                    n += __builtin_popcountll(m);
                }
            }
        }
        uint64_t tau;
        if (n > g.L) tau = wave_select_lth(buf, n, g.L, g.L, lane);
        else {
            uint64_t mn = ~0ull;
            for (int e = lane; e < n; e += 64) { const uint64_t k = buf[e]; mn = k < mn ? k : mn; }
            tau = wave_min_u64(mn);
        }
        if ((int64_t)C < N || n < g.L) {
            // the distance of the L-th best (+ margins for the log and the rounding of the canonical exp) inside the radius the sweep tested
            // against: every pair it did not list has d^2 > R (dgg_topk_sweep.hip, sw_finalize's verification)
            const float dL = c_log(fmaxf(key_val(tau), 1e-37f)) / t + 1e-5f;
            // (and a score in the normal range: zero / denormal scores tie over whole shells of distances, the oracle's column order then
            //  decides also among the pairs the sweep did not list -- the row goes to the scan that scores every pair)
            ok = n >= g.L && key_val(tau) >= 4.8e-38f && dL * dL * (1.0f + 1e-5f) <= R * (1.0f - 1e-5f) - 1e-6f * fabsf(R);
        }
        if (ok && n > g.L) n = wave_compact_ge(buf, n, tau, lane);
    }
    if (lane == 0) {
        cnt_out[lrow] = ok ? n : 0;
        if (!ok) {
            // listed for the scans that score every pair; its slice of the segmented scan's scratch starts at (sum of the capacities listed
            // so far) x (segments per row, fixed once every row is listed)
            const int pos = atomicAdd(nfail, 1);
            faillist[pos] = (int32_t)lrow;
            const unsigned long long off = atomicAdd(capsum, (unsigned long long)g.cap);
            if (pos < PLAIN_FEW_MAX) failoff[pos] = off;
        }
    }
}

// the rows the front end could not settle when they are FEW (54 of 100 000 at k ~ 130, ~900 at k ~ 32 on benchmark data): every column scored,
// lane = column, through threshold buffers.  (aw_scan_plain shares column tiles among 16 rows per wavefront: right for every row of a graph,
// but a launch of it costs a workgroup's whole walk -- 23 ms at N = 100 000 -- however few rows it is given.)  One wavefront per row took
// 2.7 ms for its 1 563 dependent blocks of 64 columns, whatever the number of rows: the column range of a row is cut into S SEGMENTS, one
// wavefront each, S = the most the scratch admits (<= PSEG_MAX) for the capacities listed (ctl->capsum); aw_plain_rows_part leaves a
// segment's best <= cap keys in the scratch, aw_plain_rows_merge streams a row's segments through its own buffer.  nlist[0] > few_max:
// left to aw_scan_plain.
// one block of keys into a row's threshold buffer (the moving-threshold append of every scan of this file)
__device__ __forceinline__ void buffer_append(uint64_t *__restrict__ buf, int &n, uint64_t &thr, uint64_t key, int L, int cap, int lane) {
    bool pass = key != DGG_EMPTY_KEY && key > thr;
    uint64_t m = __ballot(pass);
    if (m == 0ull) return;
    if (n + __builtin_popcountll(m) > cap) {
        thr = wave_select_lth(buf, n, L, keep_window(L, cap), lane);
        n = wave_compact_ge(buf, n, thr, lane);
        pass = pass && key > thr;
        m = __ballot(pass);
    }
    const int p = n + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    if (pass) buf[p] = key;
    n += __builtin_popcountll(m);
}
template <int H>
__global__ __launch_bounds__(256) void aw_plain_rows_part(const float *__restrict__ xp, int64_t N, int64_t row0, float t, const float *__restrict__ klim,
                                                          const int32_t *__restrict__ cptr, const int32_t *__restrict__ rowlist,
                                                          const HashCtl *__restrict__ ctl, int few_max, const unsigned long long *__restrict__ failoff,
                                                          uint64_t *__restrict__ scratch, long long scratch_keys, int32_t *__restrict__ pcnt) {
    const int lane = threadIdx.x & 63;
    const int nsel = ctl->nfail;
    if (nsel > few_max) return;
    const int S = plain_segments(ctl->capsum, scratch_keys);
    if (S == 0) return;
    const int64_t per = ((N + S - 1) / S + 63) / 64 * 64;        // columns per segment
    const int nwork = nsel * S, stride = (int)gridDim.x * 4;
    for (int wk = blockIdx.x * 4 + dgg::wave_id(); wk < nwork; wk += stride) {
        const int q = wk / S, sg = wk - q * S;
        const int lrow = __builtin_amdgcn_readfirstlane(rowlist[q]);
        const int64_t i = row0 + lrow;
        const RowGeom g = row_geom(cptr, klim, lrow);
        uint64_t *buf = scratch + (failoff[q] * (unsigned long long)S + (unsigned long long)sg * (unsigned long long)g.cap);
        const float *xi = xp + i * H;                            // wave-uniform: scalar loads
        uint64_t thr = DGG_EMPTY_KEY;
        int n = 0;
        const int64_t jend = (sg + 1) * per < N ? (sg + 1) * per : N;
        for (int64_t j0 = sg * per; j0 < jend; j0 += 64) {
            const int64_t j = j0 + lane;
            uint64_t key = DGG_EMPTY_KEY;
            if (j < jend) {
                const float4 *xj = reinterpret_cast<const float4 *>(xp + j * H);
                float4 bq[H / 4];
#pragma unroll
                for (int c4 = 0; c4 < H / 4; c4++) bq[c4] = xj[c4];
                float d2 = 0.0f;
#pragma unroll
                for (int c4 = 0; c4 < H / 4; c4++) {
                    float df;
                    df = __fadd_rn(xi[4 * c4 + 0], -bq[c4].x); d2 = __fmaf_rn(df, df, d2);
                    df = __fadd_rn(xi[4 * c4 + 1], -bq[c4].y); d2 = __fmaf_rn(df, df, d2);
                    df = __fadd_rn(xi[4 * c4 + 2], -bq[c4].z); d2 = __fmaf_rn(df, df, d2);
                    df = __fadd_rn(xi[4 * c4 + 3], -bq[c4].w); d2 = __fmaf_rn(df, df, d2);
                }
                key = make_key(score_from_dist(c_sqrt(d2), t, false, 0.0f), (int32_t)j);
            }
            buffer_append(buf, n, thr, key, g.L, g.cap, lane);
        }
        if (n > g.L) {
            const uint64_t tau = wave_select_lth(buf, n, g.L, g.L, lane);
            n = wave_compact_ge(buf, n, tau, lane);
        }
        if (lane == 0) pcnt[q * PSEG_MAX + sg] = n;
    }
}
__global__ __launch_bounds__(256) void aw_plain_rows_merge(const float *__restrict__ klim, const int32_t *__restrict__ cptr, uint64_t *__restrict__ keys,
                                                           int32_t *__restrict__ cnt_out, const int32_t *__restrict__ rowlist,
                                                           const HashCtl *__restrict__ ctl, int few_max, const unsigned long long *__restrict__ failoff,
                                                           const uint64_t *__restrict__ scratch, long long scratch_keys, const int32_t *__restrict__ pcnt) {
    const int lane = threadIdx.x & 63;
    const int nsel = ctl->nfail;
    if (nsel > few_max) return;
    const int S = plain_segments(ctl->capsum, scratch_keys);
    if (S == 0) return;
    for (int q = blockIdx.x * 4 + dgg::wave_id(); q < nsel; q += (int)gridDim.x * 4) {
        const int lrow = __builtin_amdgcn_readfirstlane(rowlist[q]);
        const RowGeom g = row_geom(cptr, klim, lrow);
        uint64_t *buf = keys + g.base;
        uint64_t thr = DGG_EMPTY_KEY;
        int n = 0;
        for (int sg = 0; sg < S; sg++) {
            const int ns = pcnt[q * PSEG_MAX + sg];
            const uint64_t *src = scratch + (failoff[q] * (unsigned long long)S + (unsigned long long)sg * (unsigned long long)g.cap);
            for (int base = 0; base < ns; base += 64) {
                const int e = base + lane;
                buffer_append(buf, n, thr, e < ns ? src[e] : DGG_EMPTY_KEY, g.L, g.cap, lane);
            }
        }
        if (n > g.L) {
            const uint64_t tau = wave_select_lth(buf, n, g.L, g.L, lane);
            n = wave_compact_ge(buf, n, tau, lane);
        }
        if (lane == 0) cnt_out[lrow] = n;
    }
}

// ---- per-pair hash noise: integer filter on every pair, exact scores for the survivors -----------------------------------------------
// A wavefront owns 16 rows and walks ALL columns 64 at a time (lane = column) with integer work only: the raw hash of (row, column)
// against the integer image of the row's threshold.  No feature is touched in that loop and there is no barrier in it -- the first
// form of this kernel staged every column tile through LDS for the handful of survivors it holds and spent 9/10 of its time there
// (41 ms at N = 100 000, k ~ 130).  Survivors are queued as (row, column) in LDS across column blocks and scored 64 at a time with
// full lanes: the row's features from LDS, the column's as sixteen independent 16-byte gathers (the ranked search's access pattern).
#define WAVE_FENCE() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
constexpr int QCAP = RW * 64 + 64;                               // queue slots per wavefront: < 64 left over + at most 16 x 64 new
template <int H, bool SYM>
__global__ __launch_bounds__(WAVES * 64) void aw_scan_hash(const float *__restrict__ xp, int64_t N, int64_t row0, int64_t row1, float t,
                                                           uint32_t s0, uint32_t s1, const uint32_t *__restrict__ seed_dev,
                                                           const float *__restrict__ klim, const int32_t *__restrict__ cptr,
                                                           uint64_t *__restrict__ keys, int32_t *__restrict__ cnt_out,
                                                           const float *__restrict__ lpub, const int32_t *__restrict__ rowlist,
                                                           const int32_t *__restrict__ nlist) {
    // rowlist != NULL: the rows of this launch are rowlist[0 .. *nlist) (local row ids: the rows aw_hash_score could not settle), else
    // every row of [row0, row1)
    constexpr int RS = H + 4;                                    // padded row stride (floats): 16-byte aligned, rows 4 banks apart
    __shared__ __attribute__((aligned(16))) float rowsL[RB * RS];
    __shared__ int s_row[RB];                                    // local row id of every row slot of the workgroup, -1: none
    // (NOT volatile: hipcc turns volatile LDS accesses into FLAT instructions with system-scope cache bits and a full memory wait
    //  behind each -- 43 of them in the first form of this kernel.  The lanes of a wavefront exchange data through these arrays;
    //  wavefront-scope fences keep the compiler from caching across the exchanges, the hardware executes a wavefront's LDS
    //  instructions in order)
    __shared__ uint64_t s_thr[WAVES][RW];
    __shared__ int64_t s_base[WAVES][RW];
    __shared__ int s_cnt[WAVES][RW], s_cap[WAVES][RW], s_L[WAVES][RW];
    __shared__ float s_lp[WAVES][RW];                            // the rows' upper bounds of log p over the OTHER nodes
    __shared__ uint32_t s_q[WAVES][QCAP];                        // (row << 28 | column) of the pairs that passed the integer filter
    const int tid = threadIdx.x, lane = tid & 63, wave = dgg::wave_id();
    if (seed_dev) { s0 = seed_dev[0]; s1 = seed_dev[1]; }
    const int64_t nsel = rowlist ? (int64_t)nlist[0] : row1 - row0;
    if ((int64_t)blockIdx.x * RB >= nsel) return;
    if (tid < RB) {
        const int64_t q = (int64_t)blockIdx.x * RB + tid;
        s_row[tid] = q < nsel ? (rowlist ? rowlist[q] : (int)q) : -1;
    }
    __syncthreads();
    for (int e = tid; e < RB * H; e += WAVES * 64) {
        const int r = e / H, c = e % H;
        const int lr_ = s_row[r];
        rowsL[r * RS + c] = lr_ >= 0 ? xp[(row0 + lr_) * H + c] : 0.0f;
    }
    if (lane < RW) {
        s_thr[wave][lane] = DGG_EMPTY_KEY;
        s_cnt[wave][lane] = 0;
        const int li = s_row[wave * RW + lane];
        const RowGeom g0 = li >= 0 ? row_geom(cptr, klim, li) : RowGeom{0, 0, 0};
        s_base[wave][lane] = g0.base; s_cap[wave][lane] = g0.cap; s_L[wave][lane] = g0.L;
        s_lp[wave][lane] = (lpub && li >= 0) ? lpub[li] : 1e-8f;
    }
    __syncthreads();                                             // (the last barrier: the wavefronts are independent from here on)
    uint32_t uthr[RW];                                           // integer image of the rows' thresholds (wave-uniform)
    uint32_t rk1[RW], rk2[RW], rid[RW];                          // the rows' hash keys and global ids (wave-uniform; id 0xffffffff: no row)
#pragma unroll
    for (int r = 0; r < RW; r++) {
        uthr[r] = 0u;
        const int li = __builtin_amdgcn_readfirstlane(s_row[wave * RW + r]);
        rid[r] = li >= 0 ? (uint32_t)(row0 + li) : 0xffffffffu;
        rowkey(s0, s1, rid[r], rk1[r], rk2[r]);
    }
    const bool anyrow = rid[0] != 0xffffffffu;                   // (row slots fill from the front)
    int qn = 0;                                                  // queued pairs of this wavefront (wave-uniform)

    // score up to 64 queued pairs (lane e < nd takes queue entry off + e) and append the keys above their rows' thresholds
    auto drain = [&](int nd, int off) {
        WAVE_FENCE();
        const bool have = lane < nd;
        const uint32_t ent = have ? s_q[wave][off + lane] : 0u;
        const int r = (int)(ent >> 28);
        const int64_t i = row0 + s_row[wave * RW + r], j = (int64_t)(ent & 0x0fffffffu);
        uint64_t key = DGG_EMPTY_KEY;
        if (have) {
            const float4 *xi = reinterpret_cast<const float4 *>(rowsL + (wave * RW + r) * RS);
            const float4 *xj = reinterpret_cast<const float4 *>(xp + j * H);
            float4 bq[H / 4];
#pragma unroll
            for (int c4 = 0; c4 < H / 4; c4++) bq[c4] = xj[c4];  // all gathers of the candidate in flight before the first use
            float d2 = 0.0f;
#pragma unroll
            for (int c4 = 0; c4 < H / 4; c4++) {
                const float4 a = xi[c4], b = bq[c4];
                float df;
                df = __fadd_rn(a.x, -b.x); d2 = __fmaf_rn(df, df, d2);
                df = __fadd_rn(a.y, -b.y); d2 = __fmaf_rn(df, df, d2);
                df = __fadd_rn(a.z, -b.z); d2 = __fmaf_rn(df, df, d2);
                df = __fadd_rn(a.w, -b.w); d2 = __fmaf_rn(df, df, d2);
            }
            const float g = pair_noise(s0, s1, (uint32_t)i, (uint32_t)j, SYM);
            key = make_key(score_from_dist(c_sqrt(d2), t, true, g), (int32_t)j);
        }
        const bool valid = have && key > s_thr[wave][r];
        const int pos = valid ? atomicAdd(&s_cnt[wave][r], 1) : 0;
        const RowGeom g = RowGeom{s_base[wave][r], s_cap[wave][r], s_L[wave][r]};
        if (valid && pos < g.cap) keys[g.base + pos] = key;
        bool over = valid && pos >= g.cap;
        uint64_t pend = __ballot(over);
        while (pend != 0ull) {                                   // a row's buffer is full: keep its L best, raise its threshold, retry its keys
            const int src = __builtin_ctzll(pend);
            const int R = __builtin_amdgcn_readlane(r, src);
            const RowGeom gR = RowGeom{s_base[wave][R], s_cap[wave][R], s_L[wave][R]};
            const bool mine = over && r == R;
            uint64_t tau = ~0ull;                                // (a row without a chunk keeps nothing)
            int kept = 0;
            if (gR.cap > 0) {
                uint64_t *buf = keys + gR.base;
                tau = wave_select_lth(buf, gR.cap, gR.L, keep_window(gR.L, gR.cap), lane);
                kept = wave_compact_ge(buf, gR.cap, tau, lane);
            }
            if (lane == 0) { s_thr[wave][R] = tau; s_cnt[wave][R] = kept; }
            WAVE_FENCE();
            const uint32_t ut = uthr_from_key(tau, s_lp[wave][R]);
#pragma unroll
            for (int q = 0; q < RW; q++) uthr[q] = q == R ? ut : uthr[q];
            if (mine && key > tau) {
                const int p2 = atomicAdd(&s_cnt[wave][R], 1);
                if (p2 < gR.cap) keys[gR.base + p2] = key;       // (kept <= cap / 2 and at most 64 retries: always true)
            }
            over = over && r != R;
            pend = __ballot(over);
        }
    };

    if (anyrow) {
        for (int64_t j0 = 0; j0 < N; j0 += 64) {
            const uint32_t j = (uint32_t)j0 + (uint32_t)lane;
            const bool jvalid = (int64_t)j < N;
            uint32_t ck1 = 0u, ck2 = 0u;                         // symmetric noise: a pair below the diagonal is keyed by its COLUMN
            if (SYM) rowkey(s0, s1, j, ck1, ck2);
            // the 16 rows' filters back to back (no control flow between the independent hash chains), ONE test for "any survivor"
            uint64_t m[RW];
            uint64_t any = 0ull;
#pragma unroll
            for (int r = 0; r < RW; r++) {
                const uint32_t i = rid[r];
                uint32_t k1 = rk1[r], k2 = rk2[r], b = j;
                if (SYM && j < i) { k1 = ck1; k2 = ck2; b = i; }
                uint32_t x = b ^ k1;                             // pair_u24_keyed before the shift
                x *= 0x7feb352dU; x ^= x >> 15; x += k2; x *= 0x846ca68bU;
                const bool pass = jvalid && i != 0xffffffffu && (x >= uthr[r] || j == i);     // (the diagonal pair sits at distance 0: outside the bound)
                m[r] = __ballot(pass);
                any |= m[r];
            }
            if (any != 0ull) {
#pragma unroll
                for (int r = 0; r < RW; r++) {
                    if (m[r] != 0ull) {                          // wave-uniform
                        const bool pass = (m[r] >> lane) & 1ull;
                        const int pos = qn + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m[r] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m[r], 0u));
                        if (pass) s_q[wave][pos] = ((uint32_t)r << 28) | j;
                        qn += __builtin_popcountll(m[r]);
                    }
                }
            }
            if (qn >= 64) {
                WAVE_FENCE();
                int off = 0;
                for (; off + 64 <= qn; off += 64) drain(64, off);
                const int rest = qn - off;                       // < 64 pairs stay queued: moved to the front
                const uint32_t mv = lane < rest ? s_q[wave][off + lane] : 0u;
                WAVE_FENCE();
                if (lane < rest) s_q[wave][lane] = mv;
                qn = rest;
                WAVE_FENCE();
            }
        }
        WAVE_FENCE();
        if (qn > 0) drain(qn, 0);
    }
    WAVE_FENCE();
#pragma unroll
    for (int r = 0; r < RW; r++) {
        const int li = __builtin_amdgcn_readfirstlane(s_row[wave * RW + r]);
        if (li < 0) continue;
        const RowGeom g = RowGeom{s_base[wave][r], s_cap[wave][r], s_L[wave][r]};
        int n = s_cnt[wave][r];
        n = n > g.cap ? g.cap : n;
        if (n > g.L && g.cap > 0) {
            uint64_t *buf = keys + g.base;
            const uint64_t tau = wave_select_lth(buf, n, g.L, g.L, lane);
            n = wave_compact_ge(buf, n, tau, lane);
        }
        if (lane == 0) cnt_out[li] = n;
    }
}

// ---- ranked generator, rows of more than `min_m` chunks: the walk of allpairs_topk_ranked_wide on the threshold buffer -----------------
// ---- per-pair hash noise, FRONT END: a threshold per row guessed from a pilot, one integer sweep, verification ------------------------
// The moving threshold of aw_scan_hash settles slowly: a row admits ~ c L (1 + ln(N / c L)) candidates, c = 1 / E[(p / p_max)^(1/0.3)]
// the factor by which the integer filter (which sees the noise and the row's distance BOUND, not the distance) over-admits -- 1 500
// gathered and scored rows of xp per row at L = 140 with the nearest-neighbour bound, 3 800 without.  dgg_topk_gv.hip's answer, per ROW
// here: guess the log-score v_i above which ~1.25 L_i + 32 of the row's pairs lie from the Gumbel tail, N M_i exp(-v / 0.3), with
// M_i = mean_j p_ij^(1/0.3) from 64 sampled columns (aw_pilot_rows); sweep all pairs against that FIXED integer threshold
// (aw_hash_sweep: lane = row, the column wave-uniform, ~8 integer instructions per 64 pairs, survivors appended to the row's candidate
// list); score the candidates once (aw_hash_score) and VERIFY: every pair the sweep dropped has log-score < v_i - 1e-3 + lp - lp..., i.e.
// below v_i, so the row is exact iff its L_i-th best log-score clears v_i.  Rows that fail -- too few candidates, a list that overflows,
// a pilot that missed -- are listed and redone by aw_scan_hash (exact whatever the guess was).  Needs the rows' distance bound (lpub):
// without it the filter over-admits 4.5 x at the benchmark's features and every list overflows (measured: 31 ms against 24).
constexpr int CSLOT = 512;              // candidate slots (32-bit columns) per chunk
constexpr int HSEG = 4;                 // column segments swept by separate wavefronts (a row's candidate slots are split among them)

template <int H>
__global__ __launch_bounds__(256) void aw_pilot_rows(const float *__restrict__ xp, int64_t N, int64_t row0, int64_t row1, float t, uint32_t s0,
                                                     uint32_t s1, const uint32_t *__restrict__ seed_dev, const float *__restrict__ klim,
                                                     const int32_t *__restrict__ cptr, const float *__restrict__ lpub,
                                                     uint32_t *__restrict__ uthr_row, float *__restrict__ gmin_row, HashCtl *__restrict__ ctl,
                                                     float tscale) {
    const int lane = threadIdx.x & 63;
    if (seed_dev) { s0 = seed_dev[0]; s1 = seed_dev[1]; }
    if (blockIdx.x == 0 && threadIdx.x == 0) ctl->nfail = 0;
    const int64_t lrow = (int64_t)blockIdx.x * 4 + dgg::wave_id();
    const int64_t i = row0 + lrow;
    if (i >= row1) return;
    const RowGeom g = row_geom(cptr, klim, lrow);
    const uint32_t a = mix32(mix32((uint32_t)i * 64u + (uint32_t)lane + s0) ^ (s1 * 0x9E3779B9u + 0x632BE5ABu));
    const int64_t j = (int64_t)(((uint64_t)a * (uint64_t)N) >> 32);
    const float4 *xi = reinterpret_cast<const float4 *>(xp + i * H), *xj = reinterpret_cast<const float4 *>(xp + j * H);
    float d2 = 0.0f;
#pragma unroll
    for (int c4 = 0; c4 < H / 4; c4++) {
        const float4 p = xi[c4], q = xj[c4];
        d2 += (p.x - q.x) * (p.x - q.x) + (p.y - q.y) * (p.y - q.y) + (p.z - q.z) * (p.z - q.z) + (p.w - q.w) * (p.w - q.w);
    }
    float m = j == i ? 0.0f : __expf(__logf(__expf(t * sqrtf(d2)) + 1e-8f) * (1.0f / 0.3f));
    m = wave_sum_dpp(m, lane) * (1.0f / 64.0f);
    if (lane == 0) {
        // expected pairs of the row with log-score >= v:  N M exp(-v / 0.3)  ->  v for a count of `target`; the filter passes the pairs
        // with G >= v - lp, i.e. target / (M exp(-lp / 0.3)) candidates: they have to fit the row's slots
        const float lp = lpub ? lpub[lrow] : 1e-8f;
        const float target = tscale * (float)g.L + 32.0f;      // (1.25 L + 32; tests shrink the factor to force the verification to fail)
        const float v = 0.3f * __logf(fmaxf((float)N * m, 1e-30f) / target);
        const float ncand = target / fmaxf(m * __expf(-lp * (1.0f / 0.3f)), 1e-30f);
        const bool usable = g.cap > 0 && m > 0.0f && v == v && fabsf(v) < 60.0f && ncand < 0.7f * (float)((g.cap / KSLOT) * CSLOT) &&
                            target < 0.8f * (float)g.cap;
        gmin_row[lrow] = usable ? v : INFINITY;                  // (+inf: the row is left to the moving-threshold scan)
        uthr_row[lrow] = usable ? hash_threshold_from_gmin(v - lp) : 0xffffffffu;
    }
}

// the integer sweep: lane = row, column wave-uniform (its key words on the scalar unit); a surviving pair costs one 4-byte store into the
// lane's own list (row, segment): no ballots, no atomics
template <bool SYM>
__global__ __launch_bounds__(64) void aw_hash_sweep(int64_t N, int64_t row0, int64_t row1, uint32_t s0, uint32_t s1,
                                                    const uint32_t *__restrict__ seed_dev, const int32_t *__restrict__ cptr,
                                                    const uint32_t *__restrict__ uthr_row, uint32_t *__restrict__ cand, int32_t *__restrict__ ncand) {
    const int lane = threadIdx.x;
    if (seed_dev) { s0 = seed_dev[0]; s1 = seed_dev[1]; }
    const int seg = blockIdx.x % HSEG;
    const int64_t lrow = (int64_t)(blockIdx.x / HSEG) * 64 + lane;
    const bool rvalid = lrow < row1 - row0;
    const int64_t lr = rvalid ? lrow : row1 - row0 - 1;
    const uint32_t i = (uint32_t)(row0 + lr);
    uint32_t rk1, rk2;
    rowkey(s0, s1, i, rk1, rk2);
    const uint32_t uthr = rvalid ? uthr_row[lr] : 0xffffffffu;
    const int c0 = cptr[lr], M = cptr[lr + 1] - c0;
    const int capseg = M * (CSLOT / HSEG);
    uint32_t *cb = cand + (int64_t)c0 * CSLOT + (int64_t)seg * capseg;
    int cnt = 0;
    const int64_t per = (N + HSEG - 1) / HSEG;
    const int64_t j0 = (int64_t)seg * per, j1 = j0 + per < N ? j0 + per : N;
    const bool skip = uthr == 0xffffffffu;                       // (the row goes to the moving-threshold scan: not even its diagonal is listed)
    // symmetric noise: a pair below the diagonal is keyed by its COLUMN (the key words of 64 columns at once, one per lane: two mix32 per
    // column on the scalar unit -- one unit for four SIMDs -- made this sweep 2.6x slower than the asymmetric one).  Which side of the
    // diagonal a column lies on is the same for all 64 rows of the wavefront except in the <= 2 blocks that straddle them: blocks wholly
    // ABOVE the diagonal run the asymmetric loop (row keys, no column keys at all), blocks wholly BELOW it the same loop with the column's
    // key words as scalars and the row as the counter -- the per-lane choice (three selects per pair) only in the straddling blocks
    // (6.4 -> see DESIGN: the symmetric sweep was 1.8x the asymmetric one).
    const uint32_t imin = (uint32_t)(row0 + (int64_t)(blockIdx.x / HSEG) * 64);
    const int64_t imax_l = (int64_t)imin + 63 < row1 - 1 ? (int64_t)imin + 63 : row1 - 1;
    const uint32_t imax = (uint32_t)imax_l;
    for (int64_t jb = j0; jb < j1; jb += 64) {
        const int nc = j1 - jb < 64 ? (int)(j1 - jb) : 64;
        const bool below = SYM && (uint64_t)jb + (uint64_t)nc <= (uint64_t)imin;     // every column < every row of the wavefront
        const bool above = !SYM || (uint64_t)jb > (uint64_t)imax;                      // every column > every row
        if (above) {
            // sixteen hash chains at a time, interleaved (gv_sweep's form: a chain is two quarter-rate multiplies deep), then the tests
            constexpr int UB = 16;
            int c = 0;
            for (; c + UB <= nc; c += UB) {
                uint32_t x[UB];
#pragma unroll
                for (int u = 0; u < UB; u++) {
                    uint32_t v = ((uint32_t)jb + (uint32_t)(c + u)) ^ rk1;
                    v *= 0x7feb352dU; v ^= v >> 15; v += rk2; v *= 0x846ca68bU;
                    x[u] = v;
                }
#pragma unroll
                for (int u = 0; u < UB; u++) asm volatile("" : "+v"(x[u]));   // (keeps the chains interleaved: no sinking into the tests)
#pragma unroll
                for (int u = 0; u < UB; u++) {
                    const uint32_t j = (uint32_t)jb + (uint32_t)(c + u);
                    if (!skip && (x[u] >= uthr || (!SYM && j == i))) {
                        if (cnt < capseg) cb[cnt] = j;
                        cnt++;
                    }
                }
            }
            for (; c < nc; c++) {
                const uint32_t j = (uint32_t)jb + (uint32_t)c;   // wave-uniform
                uint32_t x = j ^ rk1;
                x *= 0x7feb352dU; x ^= x >> 15; x += rk2; x *= 0x846ca68bU;
                if (!skip && (x >= uthr || (!SYM && j == i))) {
                    if (cnt < capseg) cb[cnt] = j;
                    cnt++;
                }
            }
            continue;
        }
        uint32_t ck1, ck2;
        rowkey(s0, s1, (uint32_t)jb + (uint32_t)lane, ck1, ck2);
        if (below) {
            constexpr int UB = 16;
            int c = 0;
            for (; c + UB <= nc; c += UB) {
                uint32_t x[UB];
#pragma unroll
                for (int u = 0; u < UB; u++) {
                    const uint32_t c1 = (uint32_t)__builtin_amdgcn_readlane((int)ck1, c + u), c2 = (uint32_t)__builtin_amdgcn_readlane((int)ck2, c + u);
                    uint32_t v = i ^ c1;
                    v *= 0x7feb352dU; v ^= v >> 15; v += c2; v *= 0x846ca68bU;
                    x[u] = v;
                }
#pragma unroll
                for (int u = 0; u < UB; u++) asm volatile("" : "+v"(x[u]));
#pragma unroll
                for (int u = 0; u < UB; u++) {
                    if (!skip && x[u] >= uthr) {
                        if (cnt < capseg) cb[cnt] = (uint32_t)jb + (uint32_t)(c + u);
                        cnt++;
                    }
                }
            }
            for (; c < nc; c++) {
                const uint32_t j = (uint32_t)jb + (uint32_t)c;
                const uint32_t c1 = (uint32_t)__builtin_amdgcn_readlane((int)ck1, c), c2 = (uint32_t)__builtin_amdgcn_readlane((int)ck2, c);
                uint32_t x = i ^ c1;
                x *= 0x7feb352dU; x ^= x >> 15; x += c2; x *= 0x846ca68bU;
                if (!skip && x >= uthr) {
                    if (cnt < capseg) cb[cnt] = j;
                    cnt++;
                }
            }
            continue;
        }
        for (int c = 0; c < nc; c++) {
            const uint32_t j = (uint32_t)jb + (uint32_t)c;
            const uint32_t c1 = (uint32_t)__builtin_amdgcn_readlane((int)ck1, c), c2 = (uint32_t)__builtin_amdgcn_readlane((int)ck2, c);
            uint32_t k1 = rk1, k2 = rk2, b = j;
            if (j < i) { k1 = c1; k2 = c2; b = i; }
            uint32_t x = b ^ k1;
            x *= 0x7feb352dU; x ^= x >> 15; x += k2; x *= 0x846ca68bU;
            if (!skip && (x >= uthr || j == i)) {
                if (cnt < capseg) cb[cnt] = j;
                cnt++;
            }
        }
    }
    if (rvalid) ncand[lrow * HSEG + seg] = cnt;
}

// score the candidates of a row, keep its L best, verify the guess; one wavefront per row
template <int H, bool SYM>
__global__ __launch_bounds__(256) void aw_hash_score(const float *__restrict__ xp, int64_t N, int64_t row0, int64_t row1, float t, uint32_t s0,
                                                     uint32_t s1, const uint32_t *__restrict__ seed_dev, const float *__restrict__ klim,
                                                     const int32_t *__restrict__ cptr, const uint32_t *__restrict__ cand,
                                                     const int32_t *__restrict__ ncand, const float *__restrict__ gmin_row,
                                                     uint64_t *__restrict__ keys, int32_t *__restrict__ cnt_out, HashCtl *__restrict__ ctl,
                                                     int32_t *__restrict__ faillist) {
    const int lane = threadIdx.x & 63;
    if (seed_dev) { s0 = seed_dev[0]; s1 = seed_dev[1]; }
    const int64_t lrow = (int64_t)blockIdx.x * 4 + dgg::wave_id();
    const int64_t i = row0 + lrow;
    if (i >= row1) return;
    const RowGeom g = row_geom(cptr, klim, lrow);
    if (g.cap == 0) { if (lane == 0) cnt_out[lrow] = 0; return; }
    const float gmin = gmin_row[lrow];
    const int M = g.cap / KSLOT, capseg = M * (CSLOT / HSEG);
    int nseg[HSEG], C = 0;
    bool ok = gmin < INFINITY;
#pragma unroll
    for (int q = 0; q < HSEG; q++) {
        nseg[q] = ncand[lrow * HSEG + q];
        ok = ok && nseg[q] <= capseg;
        C += nseg[q];
    }
    ok = ok && C >= g.L;
    uint64_t *buf = keys + g.base;
    const uint32_t *cl = cand + (int64_t)(g.base / KSLOT) * CSLOT;
    int n = 0;
    if (ok) {
        const float *xi = xp + i * H;                            // wave-uniform: scalar loads
        uint64_t thr = DGG_EMPTY_KEY;                            // the candidates stream through the row's threshold buffer (more of them than it holds: compacted on the way)
#pragma unroll
        for (int q = 0; q < HSEG; q++) {
            for (int base = 0; base < nseg[q]; base += 64) {
                const int e = base + lane;
                uint64_t key = DGG_EMPTY_KEY;
                if (e < nseg[q]) {
                    const uint32_t j = cl[(int64_t)q * capseg + e];
                    const float4 *xj = reinterpret_cast<const float4 *>(xp + (int64_t)j * H);
                    float4 bq[H / 4];
#pragma unroll
                    for (int c4 = 0; c4 < H / 4; c4++) bq[c4] = xj[c4];
                    float d2 = 0.0f;
#pragma unroll
                    for (int c4 = 0; c4 < H / 4; c4++) {
                        float df;
                        df = __fadd_rn(xi[4 * c4 + 0], -bq[c4].x); d2 = __fmaf_rn(df, df, d2);
                        df = __fadd_rn(xi[4 * c4 + 1], -bq[c4].y); d2 = __fmaf_rn(df, df, d2);
                        df = __fadd_rn(xi[4 * c4 + 2], -bq[c4].z); d2 = __fmaf_rn(df, df, d2);
                        df = __fadd_rn(xi[4 * c4 + 3], -bq[c4].w); d2 = __fmaf_rn(df, df, d2);
                    }
                    const float gn = pair_noise(s0, s1, (uint32_t)i, j, SYM);
                    key = make_key(score_from_dist(c_sqrt(d2), t, true, gn), (int32_t)j);
                }
                bool pass = key != DGG_EMPTY_KEY && key > thr;
                uint64_t m = __ballot(pass);
                if (m != 0ull) {
                    if (n + __builtin_popcountll(m) > g.cap) {
                        thr = wave_select_lth(buf, n, g.L, keep_window(g.L, g.cap), lane);
                        n = wave_compact_ge(buf, n, thr, lane);
                        pass = pass && key > thr;
                        m = __ballot(pass);
                    }
                    const int p = n + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    if (pass) buf[p] = key;
                    n += __builtin_popcountll(m);
                }
            }
        }
        // every pair the sweep dropped has G < v - lp - 1e-3, i.e. log-score < v - 1e-3 + 1e-8: the list is exact iff its L-th best
        // clears that (fast-math log of the score: 1e-4 of margin)
        uint64_t tau;
        if (n > g.L) tau = wave_select_lth(buf, n, g.L, g.L, lane);
        else {
            uint64_t mn = ~0ull;
            for (int e = lane; e < n; e += 64) { const uint64_t k = buf[e]; mn = k < mn ? k : mn; }
            tau = wave_min_u64(mn);
        }
        ok = n >= g.L && __logf(fmaxf(key_val(tau), 1e-37f)) - 1e-4f >= gmin - 1e-3f + 1e-6f;
        if (ok && n > g.L) n = wave_compact_ge(buf, n, tau, lane);
    }
    if (lane == 0) {
        cnt_out[lrow] = ok ? n : 0;
        if (!ok) faillist[atomicAdd(&ctl->nfail, 1)] = (int32_t)lrow;
    }
}

// (the walk's block generator of dgg_topk_ranked.hip, restated: position 0 = the row's own column with its independent variate, position
//  p >= 1 = slot sigma(p - 1) of the n = N - 1 other columns; dgg_common.h, DGG_RANKED_DIAG_KEY)
__device__ __forceinline__ uint64_t scan_u64(uint64_t v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t lo = __shfl_up((uint32_t)v, off, 64), hi = __shfl_up((uint32_t)(v >> 32), off, 64);
        const uint64_t t = ((uint64_t)hi << 32) | lo;
        if (lane >= off) v += t;
    }
    return v;
}
template <int H>
__global__ __launch_bounds__(256) void aw_ranked_walk(const float *__restrict__ xp, int64_t N, int64_t row0, int64_t row1, float t, uint32_t s0,
                                                      uint32_t s1, const uint32_t *__restrict__ seed_dev, const float *__restrict__ klim,
                                                      const int32_t *__restrict__ cptr, int min_m, uint64_t *__restrict__ keys,
                                                      int32_t *__restrict__ cnt_out, const float *__restrict__ lpub) {
    const int lane = threadIdx.x & 63;
    if (seed_dev) { s0 = seed_dev[0]; s1 = seed_dev[1]; }
    const int64_t lrow = (int64_t)blockIdx.x * 4 + dgg::wave_id();
    const int64_t i = row0 + lrow;
    if (i >= row1) return;
    const int M = __builtin_amdgcn_readfirstlane(cptr[lrow + 1] - cptr[lrow]);
    if (M <= min_m) return;
    const RowGeom g = row_geom(cptr, klim, lrow);
    uint64_t *buf = keys + g.base;
    uint32_t k1, k2;
    rowkey(s0, s1, (uint32_t)i, k1, k2);
    const uint32_t k3 = mix32(k2 ^ 0x68E31DA4u);
    const int64_t n = N - 1;
    const int b = ranked_bits(n);
    const uint64_t D = (uint64_t)1 << b;
    const float *xi = xp + i * H;
    const float lp = lpub ? lpub[lrow] : 1e-8f;                  // upper bound of log p over the OTHER nodes
    uint64_t S = 0, thr = DGG_EMPTY_KEY;
    uint32_t scount = 0;
    int cnt = 0;
    float thr_log = -INFINITY;
    for (uint64_t rb = 0; rb <= D; rb += 64) {
        const uint64_t p = rb + (uint64_t)lane;
        const bool isdiag = p == 0ull;
        const uint32_t cs = ranked_sigma((uint32_t)(p - 1ull), k1, k2, k3, b);
        const bool rvalid = !isdiag && p <= D && (int64_t)cs < n;
        const uint64_t mv = __ballot(rvalid);
        const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(mv >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mv, 0u));
        const uint32_t s = scount + pos + 1;
        const uint64_t term = ranked_term(k1, isdiag ? (k3 ^ DGG_RANKED_DIAG_KEY) : k3, isdiag ? 1u : s, isdiag ? (int64_t)1 : n);
        const uint64_t pre = scan_u64(rvalid ? term : 0ull, lane) + S;
        const float G = ranked_gumbel(isdiag ? term : pre);
        const uint32_t c = isdiag ? (uint32_t)i : cs + (cs >= (uint32_t)i ? 1u : 0u);
        uint64_t key = DGG_EMPTY_KEY;
        const bool want = (rvalid || isdiag) && !(G + lp + 1e-3f < thr_log);
        if (want) {
            const float4 *xj = reinterpret_cast<const float4 *>(xp + (int64_t)c * H);
            float d2 = 0.0f;
#pragma unroll
            for (int c4 = 0; c4 < H / 4; c4++) {
                const float4 bq = xj[c4];
                float df;
                df = __fadd_rn(xi[4 * c4 + 0], -bq.x); d2 = __fmaf_rn(df, df, d2);
                df = __fadd_rn(xi[4 * c4 + 1], -bq.y); d2 = __fmaf_rn(df, df, d2);
                df = __fadd_rn(xi[4 * c4 + 2], -bq.z); d2 = __fmaf_rn(df, df, d2);
                df = __fadd_rn(xi[4 * c4 + 3], -bq.w); d2 = __fmaf_rn(df, df, d2);
            }
            key = make_key(score_from_dist(c_sqrt(d2), t, true, G), (int32_t)c);
        }
        bool pass = key != DGG_EMPTY_KEY && key > thr;
        uint64_t m = __ballot(pass);
        if (m != 0ull) {
            if (cnt + __builtin_popcountll(m) > g.cap) {
                thr = wave_select_lth(buf, cnt, g.L, keep_window(g.L, g.cap), lane);
                cnt = wave_compact_ge(buf, cnt, thr, lane);
                thr_log = __logf(fmaxf(key_val(thr), 1e-37f));
                pass = pass && key > thr;
                m = __ballot(pass);
            }
            const int q = cnt + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (pass) buf[q] = key;
            cnt += __builtin_popcountll(m);
        }
        const int nvalid = __builtin_popcountll(mv);
        S = shfl_u64(pre, 63);
        scount += (uint32_t)nvalid;
        if ((int64_t)scount >= n) break;
        if (thr != DGG_EMPTY_KEY && nvalid > 0) {                // the lowest noise of this block bounds every rank still to come
            const int last = 63 - __builtin_clzll(mv);
            const float gmin = __shfl(G, last, 64);
            if (gmin + lp + 1e-3f < thr_log) break;
        }
    }
    if (cnt > g.L) {
        const uint64_t tau = wave_select_lth(buf, cnt, g.L, g.L, lane);
        cnt = wave_compact_ge(buf, cnt, tau, lane);
    }
    if (lane == 0) cnt_out[lrow] = cnt;
}

// ---- emit: sort a row's surviving keys, write chunks, ramp, row sum ---------------------------------------------------------------------
constexpr int ET = 256;                                          // threads of an emit workgroup
// descending bitonic sort of sk[0..P2) (P2 a power of two >= 64) by the workgroup
__device__ __forceinline__ void wg_sort_desc(uint64_t *sk, int P2, int tid) {
    for (int kb = 2; kb <= P2; kb <<= 1) {
        for (int j = kb >> 1; j > 0; j >>= 1) {
            for (int q = tid; q < (P2 >> 1); q += ET) {
                const int a = ((q & ~(j - 1)) << 1) | (q & (j - 1)), bq = a + j;
                const bool down = (a & kb) == 0 || kb == P2;     // this block ends descending
                const uint64_t ka = sk[a], kbv = sk[bq];
                if ((ka < kbv) == down) { sk[a] = kbv; sk[bq] = ka; }
            }
            __syncthreads();
        }
    }
}
// count of keys in [lo_incl, hi_excl) over buf[0..n) by the workgroup (result in every thread); *mn = the smallest such key
__device__ __forceinline__ int wg_count_range(const uint64_t *__restrict__ buf, int n, uint64_t lo_incl, uint64_t hi_excl, int tid,
                                              int *s_red, unsigned long long *s_min, uint64_t *mn) {
    int c = 0;
    uint64_t m2 = ~0ull;
    for (int e = tid; e < n; e += ET) {
        const uint64_t k = buf[e];
        const bool in = k >= lo_incl && k < hi_excl;
        c += in;
        if (in && k < m2) m2 = k;
    }
    __syncthreads();
    if (tid == 0) { *s_red = 0; *s_min = ~0ull; }
    __syncthreads();
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off, 64);
    m2 = wave_min_u64(m2);
    if ((tid & 63) == 0) { atomicAdd(s_red, c); atomicMin(s_min, (unsigned long long)m2); }
    __syncthreads();
    *mn = (uint64_t)*s_min;
    return *s_red;
}
template <int P>
__global__ __launch_bounds__(ET) void aw_emit(const uint64_t *__restrict__ keys, const int32_t *__restrict__ cnt, const float *__restrict__ klim,
                                              const int32_t *__restrict__ cptr, int64_t rows, int min_m, int64_t ccap, int32_t *__restrict__ idx,
                                              float *__restrict__ val, int softk_mode, float *__restrict__ w_out, float *__restrict__ rs_out) {
    __shared__ uint64_t sk[P];
    __shared__ int s_red;
    __shared__ unsigned long long s_min;
    const int tid = threadIdx.x, lane = tid & 63;
    if ((int64_t)blockIdx.x >= rows) {
        // arrays allocated for `ccap` chunks (a fixed capacity): the chunks beyond the last one are written EMPTY (min_m = 0 only: with
        // the ranked generator dgg_allpairs_topk_ranked_wide does it)
        const int64_t q = (int64_t)cptr[rows] + ((int64_t)blockIdx.x - rows) * 4 + (tid >> 6);
        if (q < ccap) {
            idx[q * 64 + lane] = -1;
            val[q * 64 + lane] = 0.0f;
            if (w_out) w_out[q * 64 + lane] = 0.0f;
        }
        return;
    }
    const int64_t lrow = blockIdx.x;
    const int c0 = cptr[lrow], M = cptr[lrow + 1] - c0;
    if (M <= min_m) {
        if (M <= 0 && min_m == 0 && w_out && tid == 0) rs_out[lrow] = 0.0f;     // (a fixed capacity ran out before this row)
        return;
    }
    const float ki = klim[lrow];
    const int L = klimit_len(ki, 64 * M);
    int n = cnt[lrow];
    n = n > L ? L : n;
    const uint64_t *buf = keys + (int64_t)c0 * KSLOT;
    auto emit = [&](int r, uint64_t key) {                       // rank r of the row <- key (DGG_EMPTY_KEY: no entry)
        const bool empty = key == DGG_EMPTY_KEY || r >= L;
        const int64_t e = ((int64_t)c0 + (r >> 6)) * 64 + (r & 63);
        idx[e] = empty ? -1 : key_col(key);
        const float sv = empty ? 0.0f : key_val(key);
        val[e] = sv;
        if (w_out) {
            const float f = c_ramp((float)r, ki);
            float v = f;
            if (softk_mode == 0 || softk_mode == 3) {
                const float a = __fmul_rn(sv, f);
                v = softk_mode == 0 ? a : __fadd_rn(__fadd_rn(f, -a), a);
            }
            w_out[e] = empty ? 0.0f : v;
        }
    };
    int done = 0;
    uint64_t hi = ~0ull;                                         // keys at or above `hi` are emitted already
    while (done < n) {
        const int rest = n - done;
        int take = rest;
        uint64_t tau = 0ull;
        if (rest > P) {                                          // the next P ranks: tau = the P-th largest key below hi
            take = P;
            uint64_t lo = 0ull, up = hi, mn;                     // count[lo, hi) >= take, count[up, hi) < take
            while (up - lo > 1ull) {
                const uint64_t mid = lo + ((up - lo) >> 1);
                const int c = wg_count_range(buf, n, mid, hi, tid, &s_red, &s_min, &mn);
                if (c == take) { lo = mn; break; }
                if (c > take) lo = mid; else up = mid;
            }
            tau = lo;
        }
        // the keys of [tau, hi) into LDS (their count is `take`), padded to a power of two with empty keys
        int P2 = 64;
        while (P2 < take) P2 <<= 1;
        __syncthreads();
        if (tid == 0) s_red = 0;
        for (int e = tid; e < P2; e += ET) sk[e] = DGG_EMPTY_KEY;
        __syncthreads();
        for (int e = tid; e < n; e += ET) {
            const uint64_t k = buf[e];
            if (k >= tau && k < hi) {
                const int p = atomicAdd(&s_red, 1);
                if (p < P2) sk[p] = k;
            }
        }
        __syncthreads();
        wg_sort_desc(sk, P2, tid);
        for (int q = tid; q < take; q += ET) emit(done + q, sk[q]);
        done += take;
        hi = tau;
    }
    for (int r = n + tid; r < 64 * M; r += ET) emit(r, DGG_EMPTY_KEY);          // ranks without a candidate (fewer than L columns exist)
    if (w_out) {
        // row sum in the order of allpairs_topk_ranked_wide / the oracle's butterfly_sum for K > 64: every lane adds its entry of each
        // chunk in chunk order, then the 64-lane butterfly
        __threadfence_block();
        __syncthreads();
        if (tid < 64) {
            float rsum = 0.0f;
            for (int m = 0; m < M; m++) {
                const float wv = w_out[((int64_t)c0 + m) * 64 + lane];
                rsum = m == 0 ? wv : __fadd_rn(rsum, wv);
            }
            const float s_ = wave_sum_butterfly(rsum);
            if (lane == 0) rs_out[lrow] = s_;
        }
    }
}

struct HashWs { uint32_t *cand; int32_t *ncand; uint32_t *uthr; float *gmin; int32_t *fail; HashCtl *ctl; unsigned long long *failoff; int32_t *pcnt; uint64_t *scratch; long long scratch_keys; void *plain_ws; };
template <int H>
int launch_anywide(const float *xp, int64_t N, int64_t row0, int64_t row1, float t, int noise_mode, uint32_t s0, uint32_t s1, const uint32_t *seed_dev,
                   const float *k, int mode, int maxm, int min_m, const int32_t *cptr, int64_t ccap, int32_t *idx, float *val, float *w, float *rs,
                   uint64_t *keys, int32_t *cnt, const float *lpub, const HashWs &hx, hipStream_t st) {
    const int64_t rows = row1 - row0;
    const dim3 gscan((unsigned)((rows + RB - 1) / RB));
    if (noise_mode == 0) {
        const char *ef = getenv("DGG_ANYWIDE_PLAIN_FRONT"), *ew = getenv("DGG_ANYWIDE_PLAIN_FEW");      // (read per call: tests switch between calls)
        const int few_max = ew ? (atoi(ew) < PLAIN_FEW_MAX ? atoi(ew) : PLAIN_FEW_MAX) : PLAIN_FEW_MAX;
        const int32_t *rowlist = nullptr;
        if ((!ef || atoi(ef) != 0) && hx.plain_ws && N >= 1024) {   // radius sweep on the matrix cores; the rows it cannot settle: every pair scored
            if (dgg_check_hip(hipMemsetAsync(hx.ctl, 0, sizeof(HashCtl), st), "anywide memset") != 0) return DGG_ERR_HIP;
            int rc0 = dgg_plain_wide_front_impl(xp, N, H, row0, row1, k, cptr, CSLOT, hx.cand, hx.ncand, hx.gmin, hx.plain_ws, st);
            if (rc0) return rc0;
            hipLaunchKernelGGL(aw_plain_score<H>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, xp, N, row0, row1, t, k, cptr, hx.cand, hx.ncand,
                               dgg_plain_wide_sublists(), CSLOT, hx.gmin, keys, cnt, &hx.ctl->nfail, hx.fail, &hx.ctl->capsum, hx.failoff);
            rowlist = hx.fail;
            hipLaunchKernelGGL(aw_plain_rows_part<H>, dim3(2048), dim3(256), 0, st, xp, N, row0, t, k, cptr, rowlist, hx.ctl, few_max, hx.failoff, hx.scratch,
                               hx.scratch_keys, hx.pcnt);
            hipLaunchKernelGGL(aw_plain_rows_merge, dim3(PLAIN_FEW_MAX / 4), dim3(256), 0, st, k, cptr, keys, cnt, rowlist, hx.ctl, few_max, hx.failoff,
                               hx.scratch, hx.scratch_keys, hx.pcnt);
        }
        hipLaunchKernelGGL(aw_scan_plain<H>, gscan, dim3(WAVES * 64), 0, st, xp, N, row0, row1, t, k, cptr, keys, cnt, rowlist, hx.ctl, few_max, hx.scratch_keys);
        if (rowlist && getenv("DGG_ANYWIDE_DEBUG")) {            // diagnostics only (synchronises): rows the radius front end handed on
            int nf = -1;
            if (hipStreamSynchronize(st) == hipSuccess && hipMemcpy(&nf, &hx.ctl->nfail, 4, hipMemcpyDeviceToHost) == hipSuccess)
                fprintf(stderr, "dgg anywide (unperturbed): %d of %lld rows not settled by the radius sweep\n", nf, (long long)rows);
        }
    }
    else if (noise_mode == 2 || noise_mode == 3) {
        // (read per call: tests switch the front end off / shrink its target between calls)
        const char *ef = getenv("DGG_ANYWIDE_HASH_FRONT"), *et = getenv("DGG_ANYWIDE_HASH_TARGET");
        const bool front = !ef || atoi(ef) != 0;
        const float tscale = et ? (float)atof(et) : 1.25f;
        const bool sym = noise_mode == 3;
        const int32_t *rowlist = nullptr, *nlist = nullptr;
        if (front && lpub) {                                     // guess-and-verify; the rows it cannot settle go through the moving threshold
            const dim3 g4((unsigned)((rows + 3) / 4)), gsw((unsigned)(((rows + 63) / 64) * HSEG));
            hipLaunchKernelGGL(aw_pilot_rows<H>, g4, dim3(256), 0, st, xp, N, row0, row1, t, s0, s1, seed_dev, k, cptr, lpub, hx.uthr, hx.gmin, hx.ctl, tscale);
            if (sym) hipLaunchKernelGGL(aw_hash_sweep<true>, gsw, dim3(64), 0, st, N, row0, row1, s0, s1, seed_dev, cptr, hx.uthr, hx.cand, hx.ncand);
            else hipLaunchKernelGGL(aw_hash_sweep<false>, gsw, dim3(64), 0, st, N, row0, row1, s0, s1, seed_dev, cptr, hx.uthr, hx.cand, hx.ncand);
            if (sym) hipLaunchKernelGGL((aw_hash_score<H, true>), g4, dim3(256), 0, st, xp, N, row0, row1, t, s0, s1, seed_dev, k, cptr, hx.cand, hx.ncand, hx.gmin, keys, cnt, hx.ctl, hx.fail);
            else hipLaunchKernelGGL((aw_hash_score<H, false>), g4, dim3(256), 0, st, xp, N, row0, row1, t, s0, s1, seed_dev, k, cptr, hx.cand, hx.ncand, hx.gmin, keys, cnt, hx.ctl, hx.fail);
            rowlist = hx.fail;
            nlist = &hx.ctl->nfail;
        }
        if (sym) hipLaunchKernelGGL((aw_scan_hash<H, true>), gscan, dim3(WAVES * 64), 0, st, xp, N, row0, row1, t, s0, s1, seed_dev, k, cptr, keys, cnt, lpub, rowlist, nlist);
        else hipLaunchKernelGGL((aw_scan_hash<H, false>), gscan, dim3(WAVES * 64), 0, st, xp, N, row0, row1, t, s0, s1, seed_dev, k, cptr, keys, cnt, lpub, rowlist, nlist);
    }
    else
        hipLaunchKernelGGL(aw_ranked_walk<H>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, xp, N, row0, row1, t, s0, s1, seed_dev, k, cptr,
                           min_m, keys, cnt, lpub);
    int rc = dgg_check_launch("allpairs_topk_anywide: candidate pass");
    if (rc) return rc;
    const int64_t tail = (min_m == 0 && ccap > rows) ? ccap - rows : 0;        // (every row has at least one chunk)
    const dim3 gemit((unsigned)(rows + (tail + 3) / 4));
    if ((int64_t)maxm * 64 <= 1024)
        hipLaunchKernelGGL(aw_emit<1024>, gemit, dim3(ET), 0, st, keys, cnt, k, cptr, rows, min_m, ccap, idx, val, mode, w, rs);
    else
        hipLaunchKernelGGL(aw_emit<4096>, gemit, dim3(ET), 0, st, keys, cnt, k, cptr, rows, min_m, ccap, idx, val, mode, w, rs);
    return dgg_check_launch("allpairs_topk_anywide: emit");
}

}  // namespace

extern "C" {

// bytes of workspace dgg_allpairs_topk_anywide needs for arrays of `ccap` chunks and `rows` rows: 128 keys of 8 bytes per chunk + a count per row
static inline size_t aw_al(size_t b) { return (b + 255) & ~(size_t)255; }
constexpr int NSUBMAX = 8;              // candidate sub-lists per row at most (hash front end: 4 column segments; unperturbed: 2 halves x 4)
// bytes of workspace dgg_allpairs_topk_anywide needs for arrays of `ccap` chunks, `rows` rows of a graph of N nodes, latent width h:
// keys (128 x 8 B per chunk) | counts | candidate columns of the front ends (512 x 4 B per chunk) | per row: candidate counts per
// sub-list, integer thresholds, guessed log-scores / radii, fail list | control block | segmented-scan tables + scratch | the unperturbed
// front end's fp16 copy of xp
// scratch of the unperturbed front end's segmented scan (keys): 64 per node, between 2^16 and 2^22 (32 MB)
static long long aw_scratch_keys(int64_t N) { const long long v = 64ll * (long long)N; return v < (1ll << 16) ? (1ll << 16) : (v > (1ll << 22) ? (1ll << 22) : v); }
size_t dgg_allpairs_anywide_ws_bytes(int64_t ccap, int64_t rows, int64_t N, int h) {
    if (ccap < 0 || rows < 0 || N < 0) return 0;
    return aw_al((size_t)ccap * KSLOT * sizeof(uint64_t)) + aw_al((size_t)rows * 4) + aw_al((size_t)ccap * CSLOT * sizeof(uint32_t)) +
           aw_al((size_t)rows * NSUBMAX * 4) + 3 * aw_al((size_t)rows * 4) + 256 + aw_al((size_t)PLAIN_FEW_MAX * 8) +
           aw_al((size_t)PLAIN_FEW_MAX * PSEG_MAX * 4) + aw_al((size_t)aw_scratch_keys(N) * 8) + aw_al(dgg_plain_wide_front_ws_bytes(rows, N, h));
}

// All-pairs top-L_i on CHUNKED rows of any width (include/dgg_hip.h).  noise_mode 0 (unperturbed), 2 (per-pair hash), 3 (symmetric per-pair
// hash): every row of [row0, row1), min_m must be 0.  noise_mode 4 (ranked generator): only the rows of MORE than min_m chunks (the others
// belong to dgg_allpairs_topk_ranked_wide, which also writes the spare chunks of a fixed capacity).
int dgg_allpairs_topk_anywide(const float *xp, int64_t N, int h, int64_t row0, int64_t row1, float t, int noise_mode, uint32_t s0, uint32_t s1,
                              const uint32_t *seed_dev, const float *k, int mode, int maxm, int min_m, const int32_t *cptr, int64_t ccap,
                              int32_t *idx, float *val, float *w, float *rs, const float *lpub, void *workspace, size_t ws_bytes, void *stream) {
    // lpub (nullable, [row1-row0]): dgg_allpairs_rowmin_bound's upper bounds of log p over a row's OTHER nodes; tightens the integer filter
    // of the hash generators and the stop tests of the ranked walk (the result does not depend on it)
    if (row0 < 0 || row1 > N || row0 > row1) return dgg_set_error(DGG_ERR_ARG, "allpairs_topk_anywide: bad row range");
    if (mode != 0 && mode != 1 && mode != 3) return dgg_set_error(DGG_ERR_ARG, "allpairs_topk_anywide: mode must be 0, 1 or 3");
    if (!xp || !k || !cptr || !idx || !val || (w && !rs)) return dgg_set_error(DGG_ERR_ARG, "allpairs_topk_anywide: xp, k, cptr, idx, val (and rs with w) are required");
    if (noise_mode != 0 && noise_mode != 2 && noise_mode != 3 && noise_mode != 4)
        return dgg_set_error(DGG_ERR_UNSUPPORTED, "allpairs_topk_anywide: noise_mode none (0), hash (2), symmetric hash (3) or ranked (4)");
    if (maxm < 1 || maxm > DGG_CHUNK_MAXM_ANY || min_m < 0 || (noise_mode != 4 && min_m != 0))
        return dgg_set_error(DGG_ERR_ARG, "allpairs_topk_anywide: maxm in 1..2^20; min_m = 0 unless the generator is the ranked one");
    if (N >= ((int64_t)1 << 28) || ccap >= ((int64_t)1 << 24)) return dgg_set_error(DGG_ERR_UNSUPPORTED, "allpairs_topk_anywide: N < 2^28, ccap < 2^24");
    if (!workspace || ws_bytes < dgg_allpairs_anywide_ws_bytes(ccap, row1 - row0, N, h))
        return dgg_set_error(DGG_ERR_ARG, "allpairs_topk_anywide: workspace missing or smaller than dgg_allpairs_anywide_ws_bytes");
    if (row1 == row0) return 0;
    char *wsp = reinterpret_cast<char *>(workspace);
    uint64_t *keys = reinterpret_cast<uint64_t *>(wsp); wsp += aw_al((size_t)ccap * KSLOT * sizeof(uint64_t));
    int32_t *cnt = reinterpret_cast<int32_t *>(wsp); wsp += aw_al((size_t)(row1 - row0) * 4);
    HashWs hx;
    hx.cand = reinterpret_cast<uint32_t *>(wsp); wsp += aw_al((size_t)ccap * CSLOT * sizeof(uint32_t));
    hx.ncand = reinterpret_cast<int32_t *>(wsp); wsp += aw_al((size_t)(row1 - row0) * NSUBMAX * 4);
    hx.uthr = reinterpret_cast<uint32_t *>(wsp); wsp += aw_al((size_t)(row1 - row0) * 4);
    hx.gmin = reinterpret_cast<float *>(wsp); wsp += aw_al((size_t)(row1 - row0) * 4);
    hx.fail = reinterpret_cast<int32_t *>(wsp); wsp += aw_al((size_t)(row1 - row0) * 4);
    hx.ctl = reinterpret_cast<HashCtl *>(wsp); wsp += 256;
    hx.failoff = reinterpret_cast<unsigned long long *>(wsp); wsp += aw_al((size_t)PLAIN_FEW_MAX * 8);
    hx.pcnt = reinterpret_cast<int32_t *>(wsp); wsp += aw_al((size_t)PLAIN_FEW_MAX * PSEG_MAX * 4);
    hx.scratch_keys = aw_scratch_keys(N);
    hx.scratch = reinterpret_cast<uint64_t *>(wsp); wsp += aw_al((size_t)hx.scratch_keys * 8);
    hx.plain_ws = wsp;
    hipStream_t st = (hipStream_t)stream;
    switch (h) {
        case 16: return launch_anywide<16>(xp, N, row0, row1, t, noise_mode, s0, s1, seed_dev, k, mode, maxm, min_m, cptr, ccap, idx, val, w, rs, keys, cnt, lpub, hx, st);
        case 32: return launch_anywide<32>(xp, N, row0, row1, t, noise_mode, s0, s1, seed_dev, k, mode, maxm, min_m, cptr, ccap, idx, val, w, rs, keys, cnt, lpub, hx, st);
        case 64: return launch_anywide<64>(xp, N, row0, row1, t, noise_mode, s0, s1, seed_dev, k, mode, maxm, min_m, cptr, ccap, idx, val, w, rs, keys, cnt, lpub, hx, st);
        case 128: return launch_anywide<128>(xp, N, row0, row1, t, noise_mode, s0, s1, seed_dev, k, mode, maxm, min_m, cptr, ccap, idx, val, w, rs, keys, cnt, lpub, hx, st);
        default: return dgg_set_error(DGG_ERR_UNSUPPORTED, "allpairs_topk_anywide supports latent_dim in {16,32,64,128}");
    }
}

}  // extern "C"
