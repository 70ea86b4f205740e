// dgg_topk.hip -- pair scoring + per-row top-K selection (exhaustive variants).
//
// Replaces, for the live class DGG_LearnableK_debug (reference dgm.py:1178-1292):
//   edge_prob_net "u-v-dist"            dgm.py:1607-1627   exp(-0.05 * ||xp_u - xp_v||)
//   Gumbel perturbation                 dgm.py:1211-1229   exp(log(p + 1e-8) + G)
//   torch.sort(descending) + first-K'   dgm.py:1404        -> ELL (idx, score), K' <= 64 per row
// The dense [N,N] tensors of the reference never exist: a workgroup owns a block of rows, streams
// column tiles of the projected features through LDS, and every wavefront keeps the running top-64 of its
// rows in registers (one list entry per lane), merged with a 64-lane bitonic network.
//
// Kernels here evaluate EVERY candidate pair with the canonical arithmetic of dgg_common.h
// (the pruned fast path lives in dgg_topk_fast.hip and must return identical bits).
#include "dgg_common.h"
#include "dgg_api_internal.h"

#include <cstdlib>

using namespace dgg;

namespace {

constexpr int RW = 16;            // rows per wavefront
constexpr int WAVES = 4;          // wavefronts per workgroup
constexpr int RB = RW * WAVES;    // rows per workgroup
constexpr int TN = 64;            // columns per tile (one per lane)

// ---- all-pairs, exhaustive --------------------------------------------------------------------------
// LDS: colT[H][TN] (transposed column tile, conflict-free lane-strided reads) + rows[RB][H]
template <int H>
__global__ __launch_bounds__(WAVES * 64) void allpairs_topk_exhaustive(
    const float *__restrict__ xp, int64_t N, int64_t row0, int64_t row1, float t, int noise_mode,
    const float *__restrict__ G, int64_t ldG, uint32_t s0, uint32_t s1, int K,
    int32_t *__restrict__ idx, float *__restrict__ val) {
    __shared__ float colT[H * TN];
    __shared__ float rowsL[RB * H];
    const int tid = threadIdx.x, lane = tid & 63, wave = dgg::wave_id();
    const int64_t rbase = row0 + (int64_t)blockIdx.x * RB;

    for (int e = tid; e < RB * H; e += WAVES * 64) {
        int r = e / H, c = e % H;
        int64_t gi = rbase + r;
        rowsL[e] = gi < row1 ? xp[gi * H + c] : 0.0f;
    }
    uint64_t list[RW];
    uint64_t thr[RW];
#pragma unroll
    for (int r = 0; r < RW; r++) { list[r] = DGG_EMPTY_KEY; thr[r] = DGG_EMPTY_KEY; }
    const bool perturb = noise_mode != 0;
    const bool sym = noise_mode == 3;

    for (int64_t j0 = 0; j0 < N; j0 += TN) {
        __syncthreads();
        // stage the column tile transposed: coalesced read of TN rows of H floats
        for (int e = tid; e < TN * H; e += WAVES * 64) {
            int jj = e / H, c = e % H;
            int64_t gj = j0 + jj;
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
            const int64_t i = rbase + lr;
            if (i >= row1) continue;                      // wave-uniform
            const float *xi = rowsL + lr * H;
            float d2 = 0.0f;
#pragma unroll
            for (int c = 0; c < H; c++) {
                float df = __fadd_rn(xi[c], -xj[c]);
                d2 = __fmaf_rn(df, df, d2);
            }
            float dist = c_sqrt(d2);
            float g = 0.0f;
            if (noise_mode == 1) g = jvalid ? G[i * ldG + j] : 0.0f;
            else if (noise_mode >= 2) g = pair_noise(s0, s1, (uint32_t)i, (uint32_t)j, sym);
            float v = score_from_dist(dist, t, perturb, g);
            uint64_t key = jvalid ? make_key(v, (int32_t)j) : DGG_EMPTY_KEY;
            bool pass = key > thr[r];
            if (__ballot(pass) != 0ull) {                 // wave-uniform
                uint64_t cand = pass ? key : DGG_EMPTY_KEY;
                cand = wave_sort_desc(cand, lane);
                list[r] = wave_merge_top64(list[r], cand, lane);
                thr[r] = shfl_u64(list[r], 63);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RW; r++) {
        const int64_t i = rbase + wave * RW + r;
        if (i >= row1) continue;
        if (lane < K) {
            int32_t c = key_col(list[r]);
            bool empty = list[r] == DGG_EMPTY_KEY;
            idx[(i - row0) * K + lane] = empty ? -1 : c;
            val[(i - row0) * K + lane] = empty ? 0.0f : key_val(list[r]);
        }
    }
}

// ---- candidates from a CSR graph (edge-list mode): one wavefront per row -----------------------------
__global__ __launch_bounds__(256) void edgelist_topk_kernel(
    const float *__restrict__ xp, int64_t N, int h, const int64_t *__restrict__ rowptr,
    const int32_t *__restrict__ col, float t, int noise_mode, const float *__restrict__ G, int64_t ldG,
    uint32_t s0, uint32_t s1, int K, int32_t *__restrict__ idx, float *__restrict__ val) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * (blockDim.x >> 6) + dgg::wave_id();
    if (i >= N) return;
    const bool perturb = noise_mode != 0, sym = noise_mode == 3;
    const float *xi = xp + i * h;
    uint64_t list = DGG_EMPTY_KEY;
    const int64_t e0 = rowptr[i], e1 = rowptr[i + 1];
    for (int64_t eb = e0; eb < e1; eb += 64) {
        int64_t e = eb + lane;
        uint64_t key = DGG_EMPTY_KEY;
        if (e < e1) {
            int32_t j = col[e];
            const float *xj = xp + (int64_t)j * h;
            float d2 = 0.0f;
            for (int c = 0; c < h; c++) {
                float df = __fadd_rn(xi[c], -xj[c]);
                d2 = __fmaf_rn(df, df, d2);
            }
            float dist = c_sqrt(d2);
            float g = 0.0f;
            if (noise_mode == 1) g = G[i * ldG + j];
            else if (noise_mode >= 2) g = pair_noise(s0, s1, (uint32_t)i, (uint32_t)j, sym);
            key = make_key(score_from_dist(dist, t, perturb, g), j);
        }
        key = wave_sort_desc(key, lane);
        list = wave_merge_top64(list, key, lane);
    }
    if (lane < K) {
        bool empty = list == DGG_EMPTY_KEY;
        idx[i * K + lane] = empty ? -1 : key_col(list);
        val[i * K + lane] = empty ? 0.0f : key_val(list);
    }
}

// the same rows for the latent widths the generator is built with (16 / 32 / 64 / 128): a lane fetches ITS candidate's row as H/4
// sixteen-byte loads that are all in flight before the first subtraction (the loop above issues h dependent four-byte gathers: on
// a citation graph -- Pubmed: 5.5 candidates a row -- its time is h load latencies, not bandwidth); the chain over the features is
// the same ascending fmaf chain, so the bits do not change.  With `kk` the ramp of select_top_k (dgm.py:1410-1420, softk_fwd_kernel's
// arithmetic) is applied while the sorted list is still in registers: w and the row sums come out of the same launch.
template <int H>
__device__ __forceinline__ void edgelist_row_full(
    const int64_t i, const int lane, const float *__restrict__ xp, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, float t,
    int noise_mode, const float *__restrict__ G, int64_t ldG, uint32_t s0, uint32_t s1, int K, int32_t *__restrict__ idx,
    float *__restrict__ val, const float *__restrict__ kk, int mode, float *__restrict__ w, float *__restrict__ rs,
    int32_t *__restrict__ overflow) {
    const bool perturb = noise_mode != 0, sym = noise_mode == 3;
    const float4 *xi4 = reinterpret_cast<const float4 *>(xp + i * H);
    uint64_t list = DGG_EMPTY_KEY;
    const int64_t e0 = rowptr[i], e1 = rowptr[i + 1];
    for (int64_t eb = e0; eb < e1; eb += 64) {
        const int64_t e = eb + lane;
        uint64_t key = DGG_EMPTY_KEY;
        if (e < e1) {
            const int32_t j = col[e];
            const float4 *xj4 = reinterpret_cast<const float4 *>(xp + (int64_t)j * H);
            float d2 = 0.0f;
            constexpr int CH = H / 4 < 16 ? H / 4 : 16;      // float4s in flight per lane
#pragma unroll
            for (int c0 = 0; c0 < H / 4; c0 += CH) {
                float4 v[CH];
#pragma unroll
                for (int u = 0; u < CH; u++) v[u] = xj4[c0 + u];
#pragma unroll
                for (int u = 0; u < CH; u++) {
                    const float4 a = xi4[c0 + u];
                    float df = __fadd_rn(a.x, -v[u].x);
                    d2 = __fmaf_rn(df, df, d2);
                    df = __fadd_rn(a.y, -v[u].y);
                    d2 = __fmaf_rn(df, df, d2);
                    df = __fadd_rn(a.z, -v[u].z);
                    d2 = __fmaf_rn(df, df, d2);
                    df = __fadd_rn(a.w, -v[u].w);
                    d2 = __fmaf_rn(df, df, d2);
                }
            }
            const float dist = c_sqrt(d2);
            float g = 0.0f;
            if (noise_mode == 1) g = G[i * ldG + j];
            else if (noise_mode >= 2) g = pair_noise(s0, s1, (uint32_t)i, (uint32_t)j, sym);
            key = make_key(score_from_dist(dist, t, perturb, g), j);
        }
        key = wave_sort_desc(key, lane);
        list = wave_merge_top64(list, key, lane);
    }
    const bool empty = list == DGG_EMPTY_KEY;
    const float sc = empty ? 0.0f : key_val(list);
    if (lane < K) {
        idx[i * K + lane] = empty ? -1 : key_col(list);
        val[i * K + lane] = sc;
    }
    if (kk) {
        float wv = 0.0f;
        if (lane < K) {
            const float f = c_ramp((float)lane, kk[i]);
            float v = f;
            if (mode == 0 || mode == 3) {
                const float a = __fmul_rn(sc, f);
                v = mode == 0 ? a : __fadd_rn(__fadd_rn(f, -a), a);
            }
            wv = empty ? 0.0f : v;
            w[i * K + lane] = wv;
        }
        const float s = wave_sum_butterfly(wv);
        if (lane == 0) rs[i] = s;
        // the ELL keeps K candidates of a row: exact while the ramp's support k + 8.5 fits or the row has no more candidates than that
        if (overflow && lane == 0 && e1 - e0 > K && kk[i] + 8.5f > (float)K) atomicOr(overflow, 1);
    }
}

// A citation graph has ~5 candidates a row: a wavefront per row keeps 5 of 64 lanes busy and the launch is 20 000 wavefronts of
// dependent loads (rowptr -> col -> rows) -- 28 us on the Pubmed shape.  Here the first `npack` workgroups take FOUR rows per
// wavefront, sixteen lanes each, and settle the rows with at most 16 candidates: the sort is the 16-lane part of the bitonic
// network (DPP only), the ramp and the row sum stay inside the group -- adding the zeros of lanes 16..63 in the full butterfly
// is exact, so the row sums keep their bits.  The remaining workgroups are one wavefront per row with the full-width code above and
// leave at once unless their row is wider than 16 (5 % of Pubmed's rows; walking those inside the packed wavefronts, one after
// the other, made the slowest wavefront -- four rows, one of them a hub of 300 -- the whole launch: 45 us).
template <int H>
__global__ __launch_bounds__(256) void edgelist_topk_pack4(
    const float *__restrict__ xp, int64_t N, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, float t,
    int noise_mode, const float *__restrict__ G, int64_t ldG, uint32_t s0, uint32_t s1, int K, int32_t *__restrict__ idx,
    float *__restrict__ val, const float *__restrict__ kk, int mode, float *__restrict__ w, float *__restrict__ rs,
    int32_t *__restrict__ overflow, unsigned npack) {
    const int lane = threadIdx.x & 63, g = lane >> 4, l = lane & 15;
    if (blockIdx.x >= npack) {                                    // one wavefront per row: the rows wider than a lane group
        const int64_t iw = (int64_t)(blockIdx.x - npack) * (blockDim.x >> 6) + dgg::wave_id();
        if (iw >= N || rowptr[iw + 1] - rowptr[iw] <= 16) return;
        edgelist_row_full<H>(iw, lane, xp, rowptr, col, t, noise_mode, G, ldG, s0, s1, K, idx, val, kk, mode, w, rs, overflow);
        return;
    }
    const int64_t i0 = ((int64_t)blockIdx.x * (blockDim.x >> 6) + dgg::wave_id()) * 4;
    if (i0 >= N) return;
    const int64_t i = i0 + g;
    const int64_t e0 = i < N ? rowptr[i] : 0;
    const int n = i < N ? (int)(rowptr[i + 1] - e0) : 17;
    const bool rowok = n <= 16;                                   // this group's row is settled here (else: by its own wavefront, or beyond N)
    const bool perturb = noise_mode != 0, sym = noise_mode == 3;
    const bool have = rowok && l < n;
    const int32_t j = have ? col[e0 + l] : 0;
    const float4 *xi4 = reinterpret_cast<const float4 *>(xp + (rowok ? i : 0) * H);
    const float4 *xj4 = reinterpret_cast<const float4 *>(xp + (int64_t)j * H);
    float d2 = 0.0f;
    constexpr int CH = H / 4 < 8 ? H / 4 : 8;                     // float4s of each operand in flight per lane
#pragma unroll
    for (int c0 = 0; c0 < H / 4; c0 += CH) {
        float4 v[CH], a[CH];
#pragma unroll
        for (int u = 0; u < CH; u++) { v[u] = xj4[c0 + u]; a[u] = xi4[c0 + u]; }
#pragma unroll
        for (int u = 0; u < CH; u++) {
            float df = __fadd_rn(a[u].x, -v[u].x);
            d2 = __fmaf_rn(df, df, d2);
            df = __fadd_rn(a[u].y, -v[u].y);
            d2 = __fmaf_rn(df, df, d2);
            df = __fadd_rn(a[u].z, -v[u].z);
            d2 = __fmaf_rn(df, df, d2);
            df = __fadd_rn(a[u].w, -v[u].w);
            d2 = __fmaf_rn(df, df, d2);
        }
    }
    uint64_t key = DGG_EMPTY_KEY;
    if (have) {
        const float dist = c_sqrt(d2);
        float gn = 0.0f;
        if (noise_mode == 1) gn = G[i * ldG + j];
        else if (noise_mode >= 2) gn = pair_noise(s0, s1, (uint32_t)i, (uint32_t)j, sym);
        key = make_key(score_from_dist(dist, t, perturb, gn), j);
    }
    // descending sort inside every group of 16 lanes: blocks of 2, 4, 8 as in wave_sort, the last merge with the final direction
    key = bitonic_block<2, 1, true>(key, lane);
    key = bitonic_block<4, 2, true>(key, lane);
    key = bitonic_block<8, 4, true>(key, lane);
    key = bitonic_block<64, 8, true>(key, lane);                  // (KB = 64: "up" everywhere; distances 8, 4, 2, 1 stay inside the group)
    const bool empty = key == DGG_EMPTY_KEY;
    const float sc = empty ? 0.0f : key_val(key);
    if (rowok) {
#pragma unroll
        for (int q = 0; q < 4; q++) {                             // the group's 16 ranks, then the empty tail of the row
            const int r = l + 16 * q;
            if (r < K) {
                idx[i * K + r] = (q == 0 && !empty) ? key_col(key) : -1;
                val[i * K + r] = q == 0 ? sc : 0.0f;
            }
        }
    }
    if (kk) {
        float wv = 0.0f;
        if (rowok && l < K) {
            const float f = c_ramp((float)l, kk[i]);
            float v = f;
            if (mode == 0 || mode == 3) {
                const float a = __fmul_rn(sc, f);
                v = mode == 0 ? a : __fadd_rn(__fadd_rn(f, -a), a);
            }
            wv = empty ? 0.0f : v;
        }
        if (rowok) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int r = l + 16 * q;
                if (r < K) w[i * K + r] = q == 0 ? wv : 0.0f;
            }
        }
        float sm = wv;                                            // wave_sum_butterfly's last four steps (the first two add exact zeros)
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) sm = __fadd_rn(sm, __shfl_xor(sm, off, 64));
        if (rowok && l == 0) rs[i] = sm;
        if (overflow && rowok && l == 0 && n > K && kk[i] + 8.5f > (float)K) atomicOr(overflow, 1);      // (lists narrower than 16)
    }
}

template <int H>
__global__ __launch_bounds__(256) void edgelist_topk_vec(
    const float *__restrict__ xp, int64_t N, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, float t,
    int noise_mode, const float *__restrict__ G, int64_t ldG, uint32_t s0, uint32_t s1, int K, int32_t *__restrict__ idx,
    float *__restrict__ val, const float *__restrict__ kk, int mode, float *__restrict__ w, float *__restrict__ rs,
    int32_t *__restrict__ overflow) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * (blockDim.x >> 6) + dgg::wave_id();
    if (i >= N) return;
    edgelist_row_full<H>(i, lane, xp, rowptr, col, t, noise_mode, G, ldG, s0, s1, K, idx, val, kk, mode, w, rs, overflow);
}

// same for latent widths beyond 128 (PPI: 2048): the candidates of a row are scored ONE AT A TIME by the whole wavefront --
// lanes over the features, coalesced 256-byte segments, 64 interleaved fmaf chains + xor butterfly (the canonical order for
// wide latents, oracle pair_dist) -- and the score of candidate q of a batch is kept by lane q for the sort / merge
__global__ __launch_bounds__(256) void edgelist_topk_wide_kernel(
    const float *__restrict__ xp, int64_t N, int h, const int64_t *__restrict__ rowptr,
    const int32_t *__restrict__ col, float t, int noise_mode, const float *__restrict__ G, int64_t ldG,
    uint32_t s0, uint32_t s1, int K, int32_t *__restrict__ idx, float *__restrict__ val) {
    // ONE WORKGROUP PER ROW: graphs of this regime are small (PPI: ~2 000 nodes, latent 2048), so a wavefront per row leaves
    // most of the chip idle.  The candidates of a 64-candidate batch are dealt to the four wavefronts in groups of CQ; every
    // wavefront streams its candidates' 8 KB rows (lanes over the features, UF x CQ coalesced 256-byte loads in flight) and
    // leaves the squared distances in LDS; wavefront 0 turns them into scores and merges.  The arithmetic per pair is the
    // canonical one (64 interleaved fmaf chains + butterfly): same bits as before.
    __shared__ float d2s[64];
    const int lane = threadIdx.x & 63, wave = dgg::wave_id();
    const int64_t i = blockIdx.x;
    const bool perturb = noise_mode != 0, sym = noise_mode == 3;
    const float *xi = xp + i * h;
    uint64_t list = DGG_EMPTY_KEY;
    const int64_t e0 = rowptr[i], e1 = rowptr[i + 1];
    for (int64_t eb = e0; eb < e1; eb += 64) {
        const int n = e1 - eb < 64 ? (int)(e1 - eb) : 64;
        const int32_t jl = eb + lane < e1 ? col[eb + lane] : 0;
        constexpr int CQ = 4, UF = 4;                            // candidates per group, feature blocks of 64 per iteration
        for (int q0 = wave * CQ; q0 < n; q0 += 4 * CQ) {
            const float *xj[CQ];
            float d2[CQ];
#pragma unroll
            for (int u = 0; u < CQ; u++) {
                xj[u] = xp + (int64_t)bcast(jl, q0 + u < n ? q0 + u : n - 1) * h;
                d2[u] = 0.0f;
            }
            int c = lane;
            for (; c + 64 * (UF - 1) < h; c += 64 * UF) {        // UF * (CQ + 1) loads issued before the first use
                float xv[UF], xw[UF][CQ];
#pragma unroll
                for (int f = 0; f < UF; f++) {
                    xv[f] = xi[c + 64 * f];
#pragma unroll
                    for (int u = 0; u < CQ; u++) xw[f][u] = xj[u][c + 64 * f];
                }
#pragma unroll
                for (int f = 0; f < UF; f++)                     // ascending feature order inside every lane's chain
#pragma unroll
                    for (int u = 0; u < CQ; u++) {
                        const float df = __fadd_rn(xv[f], -xw[f][u]);
                        d2[u] = __fmaf_rn(df, df, d2[u]);
                    }
            }
            for (; c < h; c += 64) {
                const float xv = xi[c];
#pragma unroll
                for (int u = 0; u < CQ; u++) {
                    const float df = __fadd_rn(xv, -xj[u][c]);
                    d2[u] = __fmaf_rn(df, df, d2[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < CQ; u++) {
                const float tot = wave_sum_butterfly(d2[u]);
                if (lane == 0 && q0 + u < n) d2s[q0 + u] = tot;
            }
        }
        __syncthreads();
        if (wave == 0) {
            uint64_t key = DGG_EMPTY_KEY;
            if (lane < n) {
                float g = 0.0f;
                if (noise_mode == 1) g = G[i * ldG + jl];
                else if (noise_mode >= 2) g = pair_noise(s0, s1, (uint32_t)i, (uint32_t)jl, sym);
                key = make_key(score_from_dist(c_sqrt(d2s[lane]), t, perturb, g), jl);
            }
            key = wave_sort_desc(key, lane);
            list = wave_merge_top64(list, key, lane);
        }
        __syncthreads();
    }
    if (wave == 0 && lane < K) {
        bool empty = list == DGG_EMPTY_KEY;
        idx[i * K + lane] = empty ? -1 : key_col(list);
        val[i * K + lane] = empty ? 0.0f : key_val(list);
    }
}

// ---- selection only: dense score rows -> top-K (test entry; bit-exact target) ------------------------
__global__ __launch_bounds__(256) void select_scores_kernel(const float *__restrict__ scores, int64_t R, int64_t N,
                                                            int K, int32_t *__restrict__ idx, float *__restrict__ val) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * (blockDim.x >> 6) + dgg::wave_id();
    if (i >= R) return;
    uint64_t list = DGG_EMPTY_KEY, thr = DGG_EMPTY_KEY;
    for (int64_t j0 = 0; j0 < N; j0 += 64) {
        int64_t j = j0 + lane;
        uint64_t key = j < N ? make_key(scores[i * N + j], (int32_t)j) : DGG_EMPTY_KEY;
        bool pass = key > thr;
        if (__ballot(pass) != 0ull) {
            uint64_t cand = wave_sort_desc(pass ? key : DGG_EMPTY_KEY, lane);
            list = wave_merge_top64(list, cand, lane);
            thr = shfl_u64(list, 63);
        }
    }
    if (lane < K) {
        bool empty = list == DGG_EMPTY_KEY;
        idx[i * K + lane] = empty ? -1 : key_col(list);
        val[i * K + lane] = empty ? 0.0f : key_val(list);
    }
}

template <int H>
int launch_exhaustive(const float *xp, int64_t N, int64_t row0, int64_t row1, float t, int noise_mode, const float *G,
                      int64_t ldG, uint32_t s0, uint32_t s1, int K, int32_t *idx, float *val, hipStream_t st) {
    int64_t rows = row1 - row0;
    if (rows <= 0) return 0;
    dim3 grid((unsigned)((rows + RB - 1) / RB));
    hipLaunchKernelGGL(allpairs_topk_exhaustive<H>, grid, dim3(WAVES * 64), 0, st, xp, N, row0, row1, t, noise_mode, G,
                       ldG, s0, s1, K, idx, val);
    return dgg_check_launch("allpairs_topk_exhaustive");
}

}  // namespace

int dgg_allpairs_topk_exhaustive_impl(const float *xp, int64_t N, int h, int64_t row0, int64_t row1, float t,
                                      int noise_mode, const float *G, int64_t ldG, uint32_t s0, uint32_t s1, int K,
                                      int32_t *idx, float *val, hipStream_t st) {
    switch (h) {
        case 8: return launch_exhaustive<8>(xp, N, row0, row1, t, noise_mode, G, ldG, s0, s1, K, idx, val, st);
        case 16: return launch_exhaustive<16>(xp, N, row0, row1, t, noise_mode, G, ldG, s0, s1, K, idx, val, st);
        case 32: return launch_exhaustive<32>(xp, N, row0, row1, t, noise_mode, G, ldG, s0, s1, K, idx, val, st);
        case 64: return launch_exhaustive<64>(xp, N, row0, row1, t, noise_mode, G, ldG, s0, s1, K, idx, val, st);
        case 128: return launch_exhaustive<128>(xp, N, row0, row1, t, noise_mode, G, ldG, s0, s1, K, idx, val, st);
        default: return dgg_set_error(DGG_ERR_UNSUPPORTED, "all-pairs scoring supports latent_dim in {8,16,32,64,128}");
    }
}

namespace {
inline bool edgelist_vec_ok(const float *xp, int h) {
    return (h == 16 || h == 32 || h == 64 || h == 128) && (uintptr_t)xp % 16 == 0;
}
void launch_edgelist_vec(const float *xp, int64_t N, int h, const int64_t *rowptr, const int32_t *col, float t, int noise_mode,
                         const float *G, int64_t ldG, uint32_t s0, uint32_t s1, int K, int32_t *idx, float *val, const float *k,
                         int mode, float *w, float *rs, int32_t *overflow, hipStream_t st) {
    // (four rows per wavefront -- see edgelist_topk_pack4; DGG_EL_PACK=0 keeps a wavefront per row)
    static const bool pack = [] { const char *e = getenv("DGG_EL_PACK"); return !e || atoi(e) != 0; }();
    const unsigned npack = (unsigned)((N + 15) / 16);
    const dim3 grid((unsigned)(pack ? npack + (N + 3) / 4 : (N + 3) / 4)), block(256);
#define DGG_EL_VEC(HH)                                                                                                       \
    if (pack) hipLaunchKernelGGL(edgelist_topk_pack4<HH>, grid, block, 0, st, xp, N, rowptr, col, t, noise_mode, G, ldG, s0, s1, K, idx, val, \
                                 k, mode, w, rs, overflow, npack);                                                           \
    else hipLaunchKernelGGL(edgelist_topk_vec<HH>, grid, block, 0, st, xp, N, rowptr, col, t, noise_mode, G, ldG, s0, s1, K, idx, val, k, \
                            mode, w, rs, overflow)
    if (h == 16) { DGG_EL_VEC(16); }
    else if (h == 32) { DGG_EL_VEC(32); }
    else if (h == 64) { DGG_EL_VEC(64); }
    else { DGG_EL_VEC(128); }
#undef DGG_EL_VEC
}
}  // namespace

extern "C" {

int dgg_edgelist_topk(const float *xp, int64_t N, int h, const int64_t *rowptr, const int32_t *col, float t,
                      int noise_mode, const float *G, int64_t ldG, uint32_t s0, uint32_t s1, int K, int32_t *idx,
                      float *val, void *stream) {
    if (K < 1 || K > 64) return dgg_set_error(DGG_ERR_UNSUPPORTED, "ELL width K must be in [1,64]");
    if (noise_mode == 1 && !G) return dgg_set_error(DGG_ERR_ARG, "explicit noise requested but G is NULL");
    if (noise_mode < 0 || noise_mode > 3)
        return dgg_set_error(DGG_ERR_UNSUPPORTED, "edgelist_topk: noise_mode must be none / explicit / hash / symmetric hash");
    if (N == 0) return 0;
    if (h > 128)
        hipLaunchKernelGGL(edgelist_topk_wide_kernel, dim3((unsigned)N), dim3(256), 0, (hipStream_t)stream, xp, N, h,
                           rowptr, col, t, noise_mode, G, ldG, s0, s1, K, idx, val);
    else if (edgelist_vec_ok(xp, h))
        launch_edgelist_vec(xp, N, h, rowptr, col, t, noise_mode, G, ldG, s0, s1, K, idx, val, nullptr, 0, nullptr, nullptr, nullptr,
                            (hipStream_t)stream);
    else
        hipLaunchKernelGGL(edgelist_topk_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, xp, N, h,
                           rowptr, col, t, noise_mode, G, ldG, s0, s1, K, idx, val);
    return dgg_check_launch("edgelist_topk");
}

// dgg_edgelist_topk followed by dgg_softk_fwd in one launch (same bits as the two calls): latent widths 16 / 32 / 64 / 128 only.
// overflow (nullable, int32[1], ORed into): set when a row has more than K candidates AND a learned degree with k + 8.5 > K
int dgg_edgelist_topk_softk(const float *xp, int64_t N, int h, const int64_t *rowptr, const int32_t *col, float t,
                            int noise_mode, const float *G, int64_t ldG, uint32_t s0, uint32_t s1, int K, const float *k,
                            int mode, int32_t *idx, float *val, float *w, float *rs, int32_t *overflow, void *stream) {
    if (K < 1 || K > 64) return dgg_set_error(DGG_ERR_UNSUPPORTED, "ELL width K must be in [1,64]");
    if (noise_mode == 1 && !G) return dgg_set_error(DGG_ERR_ARG, "explicit noise requested but G is NULL");
    if (noise_mode < 0 || noise_mode > 3)
        return dgg_set_error(DGG_ERR_UNSUPPORTED, "edgelist_topk: noise_mode must be none / explicit / hash / symmetric hash");
    if (mode != 0 && mode != 1 && mode != 3) return dgg_set_error(DGG_ERR_ARG, "edgelist_topk_softk: mode must be 0 (k_times), 1 (k_only) or 3 (hard)");
    if (!k || !w || !rs) return dgg_set_error(DGG_ERR_ARG, "edgelist_topk_softk: k, w and rs are required");
    if (!edgelist_vec_ok(xp, h))
        return dgg_set_error(DGG_ERR_UNSUPPORTED, "edgelist_topk_softk: latent_dim must be 16, 32, 64 or 128 (16-byte aligned rows)");
    if (N == 0) return 0;
    launch_edgelist_vec(xp, N, h, rowptr, col, t, noise_mode, G, ldG, s0, s1, K, idx, val, k, mode, w, rs, overflow, (hipStream_t)stream);
    return dgg_check_launch("edgelist_topk_softk");
}

int dgg_select_scores(const float *scores, int64_t R, int64_t N, int K, int32_t *idx, float *val, void *stream) {
    if (K < 1 || K > 64) return dgg_set_error(DGG_ERR_UNSUPPORTED, "ELL width K must be in [1,64]");
    if (R == 0) return 0;
    hipLaunchKernelGGL(select_scores_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, scores, R,
                       N, K, idx, val);
    return dgg_check_launch("select_scores");
}

}  // extern "C"
