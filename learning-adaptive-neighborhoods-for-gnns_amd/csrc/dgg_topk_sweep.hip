// dgg_topk_sweep.hip -- UNPERTURBED all-pairs scoring + per-row top-64 (noise_mode 0): the reference's own default
// (`perturb_edge_prob=False`, train_small_graphs.py:158-163; scores dgm.py:1618-1623, sort dgm.py:1404).  The only path whose
// result depends on all N^2 distances; identical bits to the exhaustive kernel (dgg_topk.hip).
//
// Guess, sweep, verify -- in two phases, so that the radius each row is swept with is TIGHT:
//
//   sw_prep      xw[j] = [ fp16(xp_j) | c_hi c_mid c_lo 1 1 1 0.. ]  (H + 16 16-bit values per node), c_j = -nb_j / 2, nb_j = the
//                discounted squared norm: the "augmented" K-step (bf16) folds the norms and the row's radius INTO the MFMA chain:
//                D_ij = <x^_i, x^_j> + c_j + t_i  with the row side [ 1 1 1 t_hi t_mid t_lo 0.. ];   t_i = (R_i - nb_i) / 2
//                => D_ij >= 0  <=>  L_ij := nb_i + nb_j - 2 <x^_i, x^_j>  <=  R_i    (L_ij: rigorous lower bound of d^2_ij)
//                so "is (i, j) inside row i's radius" is the SIGN BIT of an accumulator register: no VALU arithmetic per pair.
//   sw_pilot     loose radius per row from ~N/22 sampled columns (8th smallest bound per half row).
//   sw_sweep<A>  phase A: every 4th column tile, loose radius; hits recorded WITH a bound value (~140 per row).
//   sw_select    per row: the cut at which m = 32 of the phase-A hits have their UPPER bound inside (a 1/4 sample: >= 64 true
//                neighbours in N with probability ~0.9999) becomes the TIGHT radius R_i (never above the loose one); the phase-A
//                hits inside it are compacted for sw_finalize.
//   sw_sweep<B>  phase B: the other 3/4 of the tiles, tight radius; hits recorded as bare columns (~100 per row).
//   sw_finalize  per row: kept phase-A hits + phase-B hits (~135 columns): exact fp32 squared distances, the 64th smallest by
//                bisection, the ~65 columns inside it scored canonically and sorted once; VERIFIED: the list is exact iff it is
//                full and its 64th distance lies inside R_i (every pair the sweeps rejected has d^2 >= L > R_i).  Rows that fail
//                are redone by sw_fallback_* (every column scored): the result is exact whatever the guesses were.
//
// Sweep kernel: a workgroup of 4 wavefronts owns 128 * RBLK rows (RBLK 32-row MFMA blocks per wavefront, B operands in
// registers for the whole kernel) and streams its share of the column tiles (128 columns, interleaved over CS workgroups per
// row block) through a double-buffered, padded LDS image; per 32 x 32 block: KS fp16 MFMAs + the augmented bf16 one
// (v_mfma_f32_32x32x16_f16 / _bf16; the chain of block n + 1 is issued before the sign tests of block n),
// 16 v_alignbit (sign bits -> one 16-bit hit mask per lane); the masks of the RBLK blocks that share a column block are packed
// and tested with ONE wave-uniform branch (a branch per block was entered for 72 % of the blocks and cost 0.3 of the 1.2 ms).
// A lane appends its hits to its PRIVATE list (lane, segment; the record carries the row block), so there are no atomics and
// no LDS traffic on the hit path.
#include "dgg_common.h"
#include "dgg_api_internal.h"
#include <stdlib.h>
#include <math.h>
#include <stdint.h>

using namespace dgg;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int TC = 128;                 // columns per staged tile
// The Gram products run on FP16 operands (v_mfma_f32_32x32x16_f16: the bf16 rate, 11 significant bits instead of 8): the bound
//   d^2 >= (n_i + n_j)(1 - eps) - 2 delta sqrt(H) (sqrt n_i + sqrt n_j) - 2 <x^_i, x^_j>
// is 8 x tighter than with bf16 operands (a row's candidate shell shrinks from 0.8 % to 0.2 % of n_i + n_j).
//   eps   = 2^-10 (1 + 2^-12) = 0.00097681 for the rounding of both operands, rest (7e-5) for fp32 accumulation order, the
//           norm sums and the splits of c and t;
//   delta = 2^-14: an operand below the smallest normal fp16 may be flushed to zero by the matrix pipe.
// A node with a feature beyond the fp16 range ("wild", |x| > 60000) is handled conservatively: as a column it is a hit for
// every row (c = +3e38), as a row it accepts every column (t = +3e38: its lists overflow and the fallback settles it).
constexpr float EPS_F16 = 0.00105f;
constexpr float DELTA_F16 = 6.103515625e-5f;
constexpr float WILD = 60000.0f;
constexpr int CAPA_ROW = 512;           // phase-A record slots per row (all its sub-lists together; expected ~100)
constexpr int CAPB_ROW = 384;           // phase-B record slots per row (expected ~110)
constexpr int LOOSE_TARGET = 350;       // columns the loose radius should admit per row (of N)
constexpr int PILOT_M = 8;              // order statistic kept per half row by the pilot (sample = N * 2 * PILOT_M / LOOSE_TARGET columns;
                                        // measured: 5 of 220 saves 0.03 ms in sw_select and fails 20 x more rows at N = 500k)

struct SweepCtl {
    int nfail;
    int stats_on;
    unsigned long long nA, nAkept, nB;  // diagnostics (stats_on): recorded phase-A hits, those inside the tight radius, phase-B hits
    int pad[8];
};

struct Plan {            // host-side launch geometry, also what the workspace layout depends on
    int64_t rows, N, npad;
    int ntiles, nA, nB, rblk, rw, nrb, rbx, csa, csb, capa, capb, pt;
};

__host__ __device__ inline int tileA(int w) { return 4 * w; }
__host__ __device__ inline int tileB(int w) { return w + w / 3 + 1; }

// exact split of an fp32 value into three bf16 pieces (hi + mid + lo == c)
__device__ __forceinline__ void split3(float c, __bf16 &hi, __bf16 &mid, __bf16 &lo) {
    if (!(fabsf(c) < 2.9e38f)) { hi = (__bf16)c; mid = lo = (__bf16)0.0f; return; }      // the +-3e38 / -inf markers
    const float fh = __uint_as_float(__float_as_uint(c) & 0xffff0000u);
    const float r1 = c - fh;
    const float fm = __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
    const float fl = r1 - fm;
    hi = (__bf16)fh; mid = (__bf16)fm; lo = (__bf16)fl;
}

// ---- prologue: fp16 copy + augmented K-step (bf16) of every node, discounted norms --------------------------------------------
template <int H>
__global__ __launch_bounds__(256) void sw_prep(const float *__restrict__ xp, int64_t N, int64_t npad, uint16_t *__restrict__ xw,
                                               float *__restrict__ nb) {
    constexpr int HW = H + 16;
    const int lane = threadIdx.x & 63;
    const int64_t j = (int64_t)blockIdx.x * 4 + dgg::wave_id();
    if (j >= npad) return;
    float s = 0.0f;
    bool wild = false;
    for (int c = lane; c < H; c += 64) {
        float v = j < N ? xp[j * H + c] : 0.0f;
        s += v * v;
        if (!(fabsf(v) <= WILD)) { wild = true; v = v > 0.0f ? WILD : (v < 0.0f ? -WILD : 0.0f); }
        const _Float16 hv = (_Float16)v;
        xw[j * HW + c] = __builtin_bit_cast(uint16_t, hv);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    wild = __ballot(wild) != 0ull;
    if (lane < 16) {
        float nbj = s * (1.0f - EPS_F16) - 2.0f * DELTA_F16 * sqrtf((float)H) * sqrtf(s) * 1.0001f;
        float c = -0.5f * nbj;
        if (wild) { nbj = INFINITY; c = 3.0e38f; }
        if (j >= N) c = -INFINITY;
        __bf16 p[3];
        split3(c, p[0], p[1], p[2]);
        __bf16 v = (__bf16)0.0f;
        if (lane < 3) v = p[lane];
        else if (lane < 6) v = (__bf16)1.0f;
        xw[j * HW + H + lane] = __builtin_bit_cast(uint16_t, v);
        if (lane == 0 && j < N) nb[j] = nbj;
    }
}

// B operand (row side) of the augmented K-step for the lane's row: [1 1 1 t_hi t_mid t_lo 0 0] in the lower half of the K range
__device__ __forceinline__ bf16x8 aug_row(float t, int hh) {
    bf16x8 v;
#pragma unroll
    for (int q = 0; q < 8; q++) v[q] = (__bf16)0.0f;
    if (hh == 0) {
        __bf16 a, b, c;
        split3(t, a, b, c);
        v[0] = v[1] = v[2] = (__bf16)1.0f;
        v[3] = a; v[4] = b; v[5] = c;
    }
    return v;
}

// one K-step of the chain: the data steps on fp16 operands, the last (augmented) step on bf16 (its entries need the fp32 exponent
// range); the registers are 128-bit containers either way
template <bool AUG>
__device__ __forceinline__ f32x16 mfma_step(const bf16x8 &a, const bf16x8 &b, const f32x16 &acc) {
    if constexpr (AUG) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
}

// staged column tiles: TC columns x (H + 16) 16-bit values, padded to a stride of (H + 16) * 2 + 16 bytes (odd multiple of 16:
// conflict-free ds_read_b128 by the 16-lane groups of the LDS)
template <int H>
struct Tile {
    static constexpr int HW = H + 16, KS1 = H / 16 + 1, STRIDE = HW * 2 + 16, CPC = HW / 8, LQ = TC * CPC / 256;
    static constexpr int BYTES = TC * STRIDE;
};

// ---- pilot: loose radius -----------------------------------------------------------------------------------------------------
// one wavefront per 32-row block (4 per workgroup); PT sampled tiles (stride over the whole column range, per-workgroup offset).
// Per lane the PILOT_M largest block maxima of E = <x^_i, x^_j> + c_j are kept (a block's second-largest value is ignored: the
// pilot is a heuristic, exactness comes from the verification) -> L = nb_i - 2 E, radius = max over the two half rows * gscale.
template <int H>
__global__ __launch_bounds__(256) void sw_pilot(const uint16_t *__restrict__ xw, const float *__restrict__ nb, int64_t row0, int64_t row1,
                                                int ntiles, int pt_tiles, float gscale, float *__restrict__ tloose,
                                                const float *__restrict__ klim, int kpad) {
    using TL = Tile<H>;
    constexpr int HW = TL::HW, KS1 = TL::KS1, STRIDE = TL::STRIDE, CPC = TL::CPC, LQ = TL::LQ;
    __shared__ __attribute__((aligned(16))) unsigned char colA[TL::BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = dgg::wave_id(), r = lane & 31, hh = lane >> 5;
    const int64_t i = row0 + (int64_t)blockIdx.x * 128 + wave * 32 + r;
    const bool rvalid = i < row1;
    const int64_t ic = rvalid ? i : row1 - 1;
    bf16x8 bfr[KS1];
#pragma unroll
    for (int s = 0; s < KS1 - 1; s++) bfr[s] = *reinterpret_cast<const bf16x8 *>(xw + ic * HW + 16 * s + 8 * hh);
    bfr[KS1 - 1] = aug_row(0.0f, hh);
    float tm[PILOT_M];
#pragma unroll
    for (int q = 0; q < PILOT_M; q++) tm[q] = -3.0e38f;
    const int stride_t = ntiles / pt_tiles;
    const int first_t = (int)(((uint32_t)blockIdx.x * 2654435761u) % (uint32_t)stride_t);
    // (the next sampled tile is fetched into registers while the current one is worked on: the tiles are scattered over the
    //  column range, and a load -> store -> compute loop paid a full memory latency per tile, 0.10 ms in all)
    uint4 stg[LQ];
    auto tile_load = [&](int pt) {
        const int64_t c0 = (int64_t)(first_t + pt * stride_t) * TC;
#pragma unroll
        for (int q = 0; q < LQ; q++) {
            stg[q] = make_uint4(0, 0, 0, 0);
            if (pt < pt_tiles) stg[q] = *reinterpret_cast<const uint4 *>(xw + c0 * HW + (int64_t)(q * 256 + tid) * 8);
        }
    };
    tile_load(0);
    for (int pt = 0; pt < pt_tiles; pt++) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < LQ; q++) {
            const int ch = q * 256 + tid;
            *reinterpret_cast<uint4 *>(&colA[(ch / CPC) * STRIDE + (ch % CPC) * 16]) = stg[q];
        }
        __syncthreads();
        tile_load(pt + 1);
#pragma unroll
        for (int sub = 0; sub < TC / 32; sub++) {
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 16; q++) acc[q] = 0.0f;
#pragma unroll
            for (int s = 0; s < KS1; s++) {
                const bf16x8 af = *reinterpret_cast<const bf16x8 *>(&colA[(sub * 32 + r) * STRIDE + (16 * s + 8 * hh) * 2]);
                acc = (s == KS1 - 1) ? mfma_step<true>(af, bfr[s], acc) : mfma_step<false>(af, bfr[s], acc);
            }
            float m = acc[0];
#pragma unroll
            for (int q = 1; q < 15; q += 2) m = fmaxf(fmaxf(m, acc[q]), acc[q + 1]);
            m = fmaxf(m, acc[15]);
            if (m > tm[PILOT_M - 1]) {
#pragma unroll
                for (int q = 0; q < PILOT_M; q++) { const float hi = fmaxf(m, tm[q]); m = fminf(m, tm[q]); tm[q] = hi; }
            }
        }
    }
    // L of the PILOT_M-th best of this half row; the radius covers both halves.  With the learned degrees only the first
    // Lr = ceil(k_i + 8.5) + 1 ranks of a row carry weight (k_limit, include/dgg_hip.h): the radius is taken from the
    // ceil(PILOT_M Lr / 64)-th best instead -- proportionally fewer candidates in every later stage.
    const float nbi = nb[ic];
    float tsel = tm[PILOT_M - 1];
    if (klim) {
        const int Lr = klimit_len(klim[ic - row0], 64);
        int mi = (PILOT_M * Lr + 63) / 64 + kpad;
        mi = mi < 3 ? 3 : (mi > PILOT_M ? PILOT_M : mi);
#pragma unroll
        for (int q = 0; q < PILOT_M - 1; q++) tsel = (q == mi - 1) ? tm[q] : tsel;
    }
    float L = fmaf(-2.0f, tsel, nbi);
    L = fmaxf(L, __shfl_xor(L, 32, 64));
    const float R = fmaxf(L, 0.0f) * gscale + 1e-6f;
    if (rvalid && hh == 0) tloose[i - row0] = nbi < 3.0e38f ? 0.5f * (R - nbi) + 1e-6f * (fabsf(R) + fabsf(nbi)) + 1e-7f : 3.0e38f;   // (wild row)
}

// ---- sweep ---------------------------------------------------------------------------------------------------------------------
// VALUES = true (phase A): tiles 4w, records (column, D bits); false (phase B): the other tiles, records bare columns.
// Workgroup -> (row block, segment): same-XCD workgroups (blockIdx % 8, observed round-robin placement; speed only) walk the same
// segment's tiles at about the same time, so a staged tile is an L2 hit for all but the first of them.
template <int H, int RBLK, bool VALUES>
__global__ __launch_bounds__(256, 2) void sw_sweep(const uint16_t *__restrict__ xw, const float *__restrict__ trow, int64_t npad, int64_t row0, int64_t row1,
                                                   int nset, int nrb, int rbx, int CS, int cap, void *__restrict__ lists,
                                                   unsigned short *__restrict__ cnts) {
    using TL = Tile<H>;
    constexpr int HW = TL::HW, KS1 = TL::KS1, STRIDE = TL::STRIDE, CPC = TL::CPC, LQ = TL::LQ;
    static_assert(RBLK == 2 || RBLK == 4, "row blocks per wavefront");
    __shared__ __attribute__((aligned(16))) unsigned char colA[2][TL::BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = dgg::wave_id(), r = lane & 31, hh = lane >> 5;
    const int xcd = blockIdx.x & 7, kk = blockIdx.x >> 3;
    const int seg = kk / rbx, rowblk = (kk % rbx) * 8 + xcd;
    if (rowblk >= nrb) return;                                          // (uniform per workgroup)
    const int64_t rbase = row0 + (int64_t)rowblk * (128 * RBLK) + (int64_t)wave * (32 * RBLK);

    bf16x8 bfr[RBLK][KS1];
#pragma unroll
    for (int b = 0; b < RBLK; b++) {
        const int64_t i = rbase + b * 32 + r;
        const bool rvalid = i < row1;
        const int64_t ic = rvalid ? i : row1 - 1;
#pragma unroll
        for (int s = 0; s < KS1 - 1; s++) bfr[b][s] = *reinterpret_cast<const bf16x8 *>(xw + ic * HW + 16 * s + 8 * hh);
        bfr[b][KS1 - 1] = aug_row(rvalid ? trow[ic - row0] : -INFINITY, hh);
    }
    // this lane's private list: (row group = the wavefront's RBLK x 32 rows, lane, segment); byte offsets relative to the workgroup's
    // first list (always < 2^32)
    constexpr uint32_t REC = VALUES ? 8u : 4u;
    const int64_t grp = (int64_t)rowblk * 4 + wave;
    char *const wg_lists = reinterpret_cast<char *>(lists) + (int64_t)rowblk * 4 * 64 * CS * cap * REC;
    const uint32_t pos0 = (uint32_t)(((wave * 64 + lane) * CS + seg) * cap) * REC;
    uint32_t pos = pos0;
    const uint32_t lim = pos0 + (uint32_t)cap * REC;

    uint4 stg[LQ];
    auto tile_load = [&](int w) {
        const int64_t c0 = (int64_t)(VALUES ? tileA(w) : tileB(w)) * TC;
#pragma unroll
        for (int q = 0; q < LQ; q++) {      // (zero + conditional load: the unconditional form makes the compiler keep `stg` in scratch memory)
            stg[q] = make_uint4(0, 0, 0, 0);
            if (c0 < npad) stg[q] = *reinterpret_cast<const uint4 *>(xw + c0 * HW + (int64_t)(q * 256 + tid) * 8);
        }
    };
    auto tile_store = [&](int buf) {
#pragma unroll
        for (int q = 0; q < LQ; q++) {
            const int ch = q * 256 + tid;
            *reinterpret_cast<uint4 *>(&colA[buf][(ch / CPC) * STRIDE + (ch % CPC) * 16]) = stg[q];
        }
    };
    auto chain = [&](const bf16x8 (&af)[KS1], int b) {
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; q++) acc[q] = 0.0f;
#pragma unroll
        for (int s = 0; s < KS1 - 1; s++) acc = mfma_step<false>(af[s], bfr[b][s], acc);
        return mfma_step<true>(af[KS1 - 1], bfr[b][KS1 - 1], acc);
    };
    // sign bits of the 16 accumulators of a block -> bit q of a 16-bit mask (1 = outside the radius); two independent chains
    auto signs = [&](const f32x16 &acc) {
        uint32_t m0 = 0u, m1 = 0u;
#pragma unroll
        for (int q = 7; q >= 0; q--) {
            m0 = __builtin_amdgcn_alignbit(m0, __float_as_uint(acc[q]), 31u);
            m1 = __builtin_amdgcn_alignbit(m1, __float_as_uint(acc[q + 8]), 31u);
        }
        return m0 | (m1 << 8);
    };
    // the lane's LARGEST D of a block stands for each of its hits in that block (a lane with two hits in one block -- 3 % of the
    // hitting lanes -- gets the better value twice; sw_select allows for it)
    auto block_best = [&](const f32x16 &acc) {
        float best = acc[0];
#pragma unroll
        for (int q = 1; q < 15; q += 2) best = fmaxf(fmaxf(best, acc[q]), acc[q + 1]);
        return fmaxf(best, acc[15]);
    };
    // hits of two row blocks (b0, b0 + 1) against one column block: pm = mask(b0) | mask(b0 + 1) << 16
    auto record = [&](uint32_t pm, int b0, uint32_t colb, float best0, float best1) {
        uint32_t hm = ~pm;
        while (hm != 0u) {
            const int p = __builtin_ctz(hm);
            hm &= hm - 1u;
            const uint32_t q = (uint32_t)p & 15u, up = (uint32_t)p >> 4;
            const uint32_t rec = (colb + ((q & 3u) | ((q & 12u) << 1))) | (((uint32_t)b0 + up) << 28);
            if (pos < lim) {
                if (VALUES) *reinterpret_cast<int2 *>(wg_lists + pos) = make_int2((int)rec, __float_as_int(up ? best1 : best0));
                else *reinterpret_cast<uint32_t *>(wg_lists + pos) = rec;
            }
            pos += REC;
        }
    };
    auto load_af = [&](bf16x8 (&af)[KS1], int buf, int sub) {
#pragma unroll
        for (int s = 0; s < KS1; s++) af[s] = *reinterpret_cast<const bf16x8 *>(&colA[buf][(sub * 32 + r) * STRIDE + (16 * s + 8 * hh) * 2]);
    };

    int w = seg;
    if (w < nset) { tile_load(w); tile_store(0); }
    __syncthreads();
    for (int it = 0; w < nset; w += CS, it++) {
        const int buf = it & 1;
        const bool more = w + CS < nset;
        if (more) tile_load(w + CS);                                    // in flight during the MFMAs below
        const uint32_t cbase = (uint32_t)(VALUES ? tileA(w) : tileB(w)) * TC + (uint32_t)(4 * hh);
        // software pipeline over the tile's 4 x RBLK blocks: the MFMA chain of block n + 1 is issued before the sign tests of
        // block n, so the matrix pipe runs while the wavefront does its vector work
        bf16x8 afA[KS1], afB[KS1];
        load_af(afA, buf, 0);
        f32x16 cur = chain(afA, 0);
#pragma unroll
        for (int sub = 0; sub < TC / 32; sub++) {
            bf16x8 (&af)[KS1] = (sub & 1) ? afB : afA;
            bf16x8 (&afn)[KS1] = (sub & 1) ? afA : afB;
            if (sub + 1 < TC / 32) load_af(afn, buf, sub + 1);
            uint32_t pm[RBLK / 2];
            float best[VALUES ? RBLK : 1];
#pragma unroll
            for (int b = 0; b < RBLK; b++) {
                f32x16 nxt;
                const bool last = (sub + 1 == TC / 32) && (b + 1 == RBLK);
                if (!last) nxt = (b + 1 < RBLK) ? chain(af, b + 1) : chain(afn, 0);
                const uint32_t m = signs(cur);
                if (b & 1) pm[b / 2] |= m << 16; else pm[b / 2] = m;
                if (VALUES) best[VALUES ? b : 0] = block_best(cur);
                if (!last) cur = nxt;
            }
            uint32_t all = pm[0];
#pragma unroll
            for (int g = 1; g < RBLK / 2; g++) all &= pm[g];
            if (__ballot(all != 0xffffffffu) != 0ull) {                 // wave-uniform: some lane has a hit in these RBLK blocks
                const uint32_t colb = cbase + (uint32_t)(sub * 32);
#pragma unroll
                for (int g = 0; g < RBLK / 2; g++)
                    if (RBLK == 2 || __ballot(pm[g] != 0xffffffffu) != 0ull)
                        record(pm[g], 2 * g, colb, VALUES ? best[VALUES ? 2 * g : 0] : 0.0f, VALUES ? best[VALUES ? 2 * g + 1 : 0] : 0.0f);
            }
        }
        if (more) tile_store(buf ^ 1);
        __syncthreads();
    }
    const uint32_t n = (pos - pos0) / REC;
    cnts[((grp * 64 + lane) * CS) + seg] = (unsigned short)(n > 0xffffu ? 0xffffu : n);
}

// ---- row minimum: a rigorous LOWER BOUND of the distance from every row to its nearest OTHER node --------------------------------
// The noise generators' early-out tests bound a pair's log-score by its noise alone, log p'_ij <= G_ij + 1e-8, i.e. they assume the
// candidate could sit at distance 0.  With d_lb(i) <= min_{j != i} ||xp_i - xp_j|| the bound tightens to
//   log p'_ij <= G_ij + log(exp(t d_lb(i)) + 1e-8)     for every j != i
// which is what decides how deep the ranked search walks on spread latents (a row visits ~ L exp(D / 0.3) ranks, D = the spread of
// t d it has to allow for: DESIGN.md section 6) and how many candidates the per-pair hash filter of the chunked rows admits.
// Same matrix-core sweep as sw_sweep -- the sign test replaced by a running maximum of E_ij = <x^_i, x^_j> + c_j per lane (the augmented
// K-step with t = 0), L_ij = nb_i - 2 E_ij the rigorous lower bound of d^2 (header of this file) -- over ALL column tiles, the diagonal
// masked in the tiles that overlap the wavefront's own rows; column segments combine by an atomic max on an order-preserving integer
// image of E.  ~1 ms at N = 100 000, h = 64: only worth it when the walk is deep (the callers decide from a pilot).
__device__ __forceinline__ uint32_t fkey(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float fkey_inv(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }
template <int H, int RBLK>
__global__ __launch_bounds__(256, 2) void sw_rowmin(const uint16_t *__restrict__ xw, int64_t npad, int64_t row0, int64_t row1, int ntiles, int nrb,
                                                    int rbx, int CS, uint32_t *__restrict__ ekey) {
    using TL = Tile<H>;
    constexpr int HW = TL::HW, KS1 = TL::KS1, STRIDE = TL::STRIDE, CPC = TL::CPC, LQ = TL::LQ;
    __shared__ __attribute__((aligned(16))) unsigned char colA[2][TL::BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = dgg::wave_id(), r = lane & 31, hh = lane >> 5;
    const int xcd = blockIdx.x & 7, kk = blockIdx.x >> 3;
    const int seg = kk / rbx, rowblk = (kk % rbx) * 8 + xcd;
    if (rowblk >= nrb) return;
    const int64_t rbase = row0 + (int64_t)rowblk * (128 * RBLK) + (int64_t)wave * (32 * RBLK);
    bf16x8 bfr[RBLK][KS1];
    float em[RBLK];
#pragma unroll
    for (int b = 0; b < RBLK; b++) {
        const int64_t i = rbase + b * 32 + r;
        const int64_t ic = i < row1 ? i : row1 - 1;
#pragma unroll
        for (int s = 0; s < KS1 - 1; s++) bfr[b][s] = *reinterpret_cast<const bf16x8 *>(xw + ic * HW + 16 * s + 8 * hh);
        bfr[b][KS1 - 1] = aug_row(0.0f, hh);
        em[b] = -3.0e38f;
    }
    uint4 stg[LQ];
    auto tile_load = [&](int w) {
        const int64_t c0 = (int64_t)w * TC;
#pragma unroll
        for (int q = 0; q < LQ; q++) {
            stg[q] = make_uint4(0, 0, 0, 0);
            if (c0 < npad) stg[q] = *reinterpret_cast<const uint4 *>(xw + c0 * HW + (int64_t)(q * 256 + tid) * 8);
        }
    };
    auto tile_store = [&](int buf) {
#pragma unroll
        for (int q = 0; q < LQ; q++) {
            const int ch = q * 256 + tid;
            *reinterpret_cast<uint4 *>(&colA[buf][(ch / CPC) * STRIDE + (ch % CPC) * 16]) = stg[q];
        }
    };
    int w = seg;
    if (w < ntiles) { tile_load(w); tile_store(0); }
    __syncthreads();
    for (int it = 0; w < ntiles; w += CS, it++) {
        const int buf = it & 1;
        const bool more = w + CS < ntiles;
        if (more) tile_load(w + CS);
        const int64_t c0 = (int64_t)w * TC;
        const bool diag = c0 < rbase + 32 * RBLK && c0 + TC > rbase;        // (wave-uniform) the tile holds columns of this wavefront's rows
#pragma unroll 1
        for (int sub = 0; sub < TC / 32; sub++) {                       // (not unrolled: 16 chains in flight spilled 700 bytes per lane)
            bf16x8 af[KS1];
#pragma unroll
            for (int s = 0; s < KS1; s++) af[s] = *reinterpret_cast<const bf16x8 *>(&colA[buf][(sub * 32 + r) * STRIDE + (16 * s + 8 * hh) * 2]);
#pragma unroll
            for (int b = 0; b < RBLK; b++) {
                f32x16 acc;
#pragma unroll
                for (int q = 0; q < 16; q++) acc[q] = 0.0f;
#pragma unroll
                for (int s = 0; s < KS1 - 1; s++) acc = mfma_step<false>(af[s], bfr[b][s], acc);
                acc = mfma_step<true>(af[KS1 - 1], bfr[b][KS1 - 1], acc);
                float m = acc[0];
#pragma unroll
                for (int q = 1; q < 15; q += 2) m = fmaxf(fmaxf(m, acc[q]), acc[q + 1]);
                m = fmaxf(m, acc[15]);
                if (diag) {                                             // the pair (i, i) is not a neighbour: the maximum without it
                    const int64_t o = rbase + b * 32 + r - (c0 + sub * 32);   // the row's own column inside this 32-column block?
                    if (o >= 0 && o < 32 && ((o >> 2) & 1) == hh) {
                        const int qs = (int)((o & 3) | ((o >> 3) << 2));
                        m = -3.0e38f;
#pragma unroll
                        for (int q = 0; q < 16; q++) m = q == qs ? m : fmaxf(m, acc[q]);
                    }
                }
                em[b] = fmaxf(em[b], m);
            }
        }
        if (more) tile_store(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int b = 0; b < RBLK; b++) {
        const int64_t i = rbase + b * 32 + r;
        const float e = fmaxf(em[b], __shfl_xor(em[b], 32, 64));
        if (hh == 0 && i < row1) atomicMax(&ekey[i - row0], fkey(e));
    }
}
// E_max -> upper bound of log p_ij over j != i:  d^2 >= nb_i - 2 E_max  (- the rounding of this line),  log p = log(exp(t d) + 1e-8)
__global__ void sw_rowmin_finish(const uint32_t *__restrict__ ekey, const float *__restrict__ nb, int64_t row0, int64_t rows, float t,
                                 float *__restrict__ lpub) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= rows) return;
    const float nbi = nb[row0 + q], e = fkey_inv(ekey[q]);
    float l2 = fmaf(-2.0f, e, nbi);
    l2 -= 4e-7f * (fabsf(nbi) + 2.0f * fabsf(e)) + 1e-12f;                 // fp32 rounding of the line above
    float ub = 1e-8f;                                                     // no usable bound: the distance-free one, log(1 + 1e-8)
    if (nbi < 3.0e38f && e > -1.0e38f && e < 1.0e38f && l2 > 0.0f && t < 0.0f) {
        const float dlb = sqrtf(l2) * (1.0f - 1e-6f);
        ub = __logf(__expf(t * dlb) + 1e-8f) + 2e-6f * (1.0f + fabsf(t * dlb));      // fast-math exp / log: 2e-6 relative covers them
        ub = fminf(ub, 1e-8f);
    }
    lpub[q] = ub;
}

// ---- unperturbed scores on CHUNKED rows (rows of any width): radius per row guessed from a sampled sweep, one full sweep, candidates ------
// Front end of dgg_allpairs_topk_anywide for noise_mode 0 (dgg_topk_anywide.hip scores the candidates, verifies, and redoes the rows
// that fail with its exhaustive scan).  Row i needs its L_i = ceil(k_i + 8.5) + 1 nearest columns, L_i anything up to N:
//   pw_pilot   one wavefront per 32 rows over SAMPLED column tiles (stride over the column range): every half-lane keeps the 8 largest
//              E_ij = <x^_i, x^_j> + c_j of the first P_i tiles, P_i chosen per row so that about 8 of the 2.5 L_i + 64 columns the
//              radius should admit fall into the half-lane's sample; the 8th largest (the smaller of the two halves) is the row's
//              threshold e_i:  E_ij >= e_i  <=>  L_ij = nb_i - 2 E_ij <= R_i = nb_i - 2 e_i   (L_ij: rigorous lower bound of d^2)
//   pw_sweep   sw_rowmin's loop over ALL tiles with the comparison E >= e_i per accumulator; a hit appends its column to the lane's own
//              sub-list (row, half, segment): no atomics.  Every pair that is NOT listed has d^2 >= L_ij > R_i.
constexpr int PW_CS = 4;                // column segments of pw_sweep (sub-lists per row: 2 halves x PW_CS)
constexpr int PW_PTMAX = 64;            // sampled tiles of pw_pilot at most
template <int H>
__global__ __launch_bounds__(256) void pw_pilot(const uint16_t *__restrict__ xw, const float *__restrict__ nb, int64_t N, int64_t row0, int64_t row1,
                                                int ntiles, const float *__restrict__ klim, const int32_t *__restrict__ cptr,
                                                float *__restrict__ ethr, float *__restrict__ rad, float tfac, float tadd) {
    using TL = Tile<H>;
    constexpr int HW = TL::HW, KS1 = TL::KS1, STRIDE = TL::STRIDE, CPC = TL::CPC, LQ = TL::LQ;
    __shared__ __attribute__((aligned(16))) unsigned char colA[TL::BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = dgg::wave_id(), r = lane & 31, hh = lane >> 5;
    const int64_t i = row0 + (int64_t)blockIdx.x * 128 + wave * 32 + r;
    const bool rvalid = i < row1;
    const int64_t ic = rvalid ? i : row1 - 1, lr = ic - row0;
    bf16x8 bfr[KS1];
#pragma unroll
    for (int s = 0; s < KS1 - 1; s++) bfr[s] = *reinterpret_cast<const bf16x8 *>(xw + ic * HW + 16 * s + 8 * hh);
    bfr[KS1 - 1] = aug_row(0.0f, hh);
    const int M = cptr[lr + 1] - cptr[lr];
    const float target = tfac * (float)klimit_len(klim[lr], 64 * (M > 0 ? M : 1)) + tadd;
    const int pt_all = ntiles < PW_PTMAX ? ntiles : PW_PTMAX;
    int P = (int)ceilf(8.0f * (float)N / (64.0f * target));
    P = P < 1 ? 1 : (P > pt_all ? pt_all : P);
    float tm[8];
#pragma unroll
    for (int q = 0; q < 8; q++) tm[q] = -3.0e38f;
    const int stride_t = ntiles / pt_all;
    const int first_t = (int)(((uint32_t)blockIdx.x * 2654435761u) % (uint32_t)stride_t);
    uint4 stg[LQ];
    auto tile_load = [&](int pt) {
        const int64_t c0 = (int64_t)(first_t + pt * stride_t) * TC;
#pragma unroll
        for (int q = 0; q < LQ; q++) {
            stg[q] = make_uint4(0, 0, 0, 0);
            if (pt < pt_all) stg[q] = *reinterpret_cast<const uint4 *>(xw + c0 * HW + (int64_t)(q * 256 + tid) * 8);
        }
    };
    tile_load(0);
    for (int pt = 0; pt < pt_all; pt++) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < LQ; q++) {
            const int ch = q * 256 + tid;
            *reinterpret_cast<uint4 *>(&colA[(ch / CPC) * STRIDE + (ch % CPC) * 16]) = stg[q];
        }
        __syncthreads();
        tile_load(pt + 1);
        const bool mine = pt < P;                                       // (per lane: the row's sample ends after its P tiles)
#pragma unroll 1
        for (int sub = 0; sub < TC / 32; sub++) {
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 16; q++) acc[q] = 0.0f;
#pragma unroll
            for (int s = 0; s < KS1; s++) {
                const bf16x8 af = *reinterpret_cast<const bf16x8 *>(&colA[(sub * 32 + r) * STRIDE + (16 * s + 8 * hh) * 2]);
                acc = (s == KS1 - 1) ? mfma_step<true>(af, bfr[s], acc) : mfma_step<false>(af, bfr[s], acc);
            }
            if (mine) {
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    float m = acc[q];
                    if (m > tm[7]) {
#pragma unroll
                        for (int u = 0; u < 8; u++) { const float hi = fmaxf(m, tm[u]); m = fminf(m, tm[u]); tm[u] = hi; }
                    }
                }
            }
        }
    }
    // the 8th largest of the half-lane's sample; fewer than 8 columns sampled (tiny graphs): every column is admitted
    float e = 64 * P >= 16 && (int64_t)P * TC < N ? tm[7] : -3.0e38f;
    e = fminf(e, __shfl_xor(e, 32, 64));                                // the more permissive of the two halves
    if (rvalid && hh == 0) {
        const float nbi = nb[ic];
        const bool usable = nbi < 3.0e38f && e > -1.0e38f && M > 0;
        ethr[lr] = usable ? e : -3.0e38f;                               // (-3e38: every column is a hit: the lists overflow, the row falls back)
        rad[lr] = usable ? fmaf(-2.0f, e, nbi) - 4e-7f * (fabsf(nbi) + 2.0f * fabsf(e)) : 3.0e38f;
    }
}
template <int H, int RBLK>
__global__ __launch_bounds__(256, 2) void pw_sweep(const uint16_t *__restrict__ xw, int64_t npad, int64_t row0, int64_t row1, int ntiles, int nrb, int rbx,
                                                   const float *__restrict__ ethr, const int32_t *__restrict__ cptr, int cslot,
                                                   uint32_t *__restrict__ cand, int32_t *__restrict__ ncand) {
    using TL = Tile<H>;
    constexpr int HW = TL::HW, KS1 = TL::KS1, STRIDE = TL::STRIDE, CPC = TL::CPC, LQ = TL::LQ, CS = PW_CS;
    __shared__ __attribute__((aligned(16))) unsigned char colA[2][TL::BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = dgg::wave_id(), r = lane & 31, hh = lane >> 5;
    const int xcd = blockIdx.x & 7, kk = blockIdx.x >> 3;
    const int seg = kk / rbx, rowblk = (kk % rbx) * 8 + xcd;
    if (rowblk >= nrb) return;
    const int64_t rbase = row0 + (int64_t)rowblk * (128 * RBLK) + (int64_t)wave * (32 * RBLK);
    bf16x8 bfr[RBLK][KS1];
    float et[RBLK];
    uint32_t *cb[RBLK];
    int cap[RBLK], cnt[RBLK];
#pragma unroll
    for (int b = 0; b < RBLK; b++) {
        const int64_t i = rbase + b * 32 + r;
        const bool rv = i < row1;
        const int64_t ic = rv ? i : row1 - 1, lr = ic - row0;
#pragma unroll
        for (int s = 0; s < KS1 - 1; s++) bfr[b][s] = *reinterpret_cast<const bf16x8 *>(xw + ic * HW + 16 * s + 8 * hh);
        bfr[b][KS1 - 1] = aug_row(0.0f, hh);
        et[b] = rv ? ethr[lr] : 3.0e38f;                                // (rows beyond the range: no hit ever)
        const int c0 = cptr[lr], M = cptr[lr + 1] - c0;
        cap[b] = M * cslot / (2 * CS);
        cb[b] = cand + (int64_t)c0 * cslot + (int64_t)(hh * CS + seg) * cap[b];
        cnt[b] = 0;
    }
    uint4 stg[LQ];
    auto tile_load = [&](int w) {
        const int64_t c0 = (int64_t)w * TC;
#pragma unroll
        for (int q = 0; q < LQ; q++) {
            stg[q] = make_uint4(0, 0, 0, 0);
            if (c0 < npad) stg[q] = *reinterpret_cast<const uint4 *>(xw + c0 * HW + (int64_t)(q * 256 + tid) * 8);
        }
    };
    auto tile_store = [&](int buf) {
#pragma unroll
        for (int q = 0; q < LQ; q++) {
            const int ch = q * 256 + tid;
            *reinterpret_cast<uint4 *>(&colA[buf][(ch / CPC) * STRIDE + (ch % CPC) * 16]) = stg[q];
        }
    };
    int w = seg;
    if (w < ntiles) { tile_load(w); tile_store(0); }
    __syncthreads();
    for (int it = 0; w < ntiles; w += CS, it++) {
        const int buf = it & 1;
        const bool more = w + CS < ntiles;
        if (more) tile_load(w + CS);
        const uint32_t cbase = (uint32_t)w * TC + (uint32_t)(4 * hh);
#pragma unroll 1
        for (int sub = 0; sub < TC / 32; sub++) {
            bf16x8 af[KS1];
#pragma unroll
            for (int s = 0; s < KS1; s++) af[s] = *reinterpret_cast<const bf16x8 *>(&colA[buf][(sub * 32 + r) * STRIDE + (16 * s + 8 * hh) * 2]);
#pragma unroll
            for (int b = 0; b < RBLK; b++) {
                f32x16 acc;
#pragma unroll
                for (int q = 0; q < 16; q++) acc[q] = 0.0f;
#pragma unroll
                for (int s = 0; s < KS1 - 1; s++) acc = mfma_step<false>(af[s], bfr[b][s], acc);
                acc = mfma_step<true>(af[KS1 - 1], bfr[b][KS1 - 1], acc);
                uint32_t hm = 0u;                                        // bit q: accumulator q is inside the row's radius
#pragma unroll
                for (int q = 0; q < 16; q++) hm |= acc[q] >= et[b] ? (1u << q) : 0u;
                while (hm != 0u) {
                    const uint32_t q = (uint32_t)__builtin_ctz(hm);
                    hm &= hm - 1u;
                    if (cnt[b] < cap[b]) cb[b][cnt[b]] = cbase + (uint32_t)(sub * 32) + ((q & 3u) | ((q & 12u) << 1));
                    cnt[b]++;
                }
            }
        }
        if (more) tile_store(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int b = 0; b < RBLK; b++) {
        const int64_t i = rbase + b * 32 + r;
        if (i < row1) ncand[(i - row0) * (2 * CS) + hh * CS + seg] = cnt[b];
    }
}

// ---- select: tight radius from the phase-A hits -------------------------------------------------------------------------------
// one wavefront per row.  D (loose) = <.,.> + c_j + t_loose is what phase A recorded; L = R_loose - 2 D is the pair's LOWER bound
// and U = L + SL (nb_i + nb_j) an UPPER bound of d^2 (both roundings of the bf16 products the other way).  The verification
// needs 64 columns with TRUE distance inside the radius, so the tight radius is the m-th smallest UPPER bound of the sample
// (m = 26 of a 1/4 sample: >= 64 in N with probability 0.995, whatever the density of the shell the radius falls into):
// D' = D - SL (nb_i + nb_j) / 2, cut = m-th largest D' (>= 0: never looser than the loose radius), t_tight = t_loose - cut.
// A phase-A hit lies inside the tight radius iff D >= cut.
// U = L + SL_UPPER (nb_i + nb_j) is an upper bound of d^2 (both roundings the other way)
constexpr float SL_UPPER = (EPS_F16 + 0.00097682f) / (1.0f - EPS_F16) * 1.0001f;
// (only the choice of the tight radius uses U -- it decides speed, not exactness -- so the flush-to-zero allowance is left out)
constexpr int KCAP = 128;               // phase-A hits inside the tight radius kept per row (expected ~40)
constexpr uint32_t COLMASK = 0x0fffffffu;   // record = column | row block << 28

// the 2 * CS lane lists that hold a row's records (see sw_sweep): list id of (half h, segment g) and the row's block tag
struct RowLists {
    int64_t base;      // (grp * 64 + r) : list id = (base + 32 * h) * CS + g
    uint32_t tag;
};
__device__ __forceinline__ RowLists row_lists(int64_t lrow, int rblk) {
    RowLists rl;
    const int64_t grp = lrow >> (rblk == 4 ? 7 : 6);                   // rblk is 2 or 4: no division
    rl.base = grp * 64 + (lrow & 31);
    rl.tag = (uint32_t)((lrow >> 5) & (rblk - 1));
    return rl;
}

constexpr int SELCAP = CAPA_ROW;        // phase-A hits of one row that sw_select can hold (expected ~140; the loose radius has a heavy tail)
__global__ __launch_bounds__(256) void sw_select(const int2 *__restrict__ listA, const unsigned short *__restrict__ cntA, int64_t rows, int64_t row0,
                                                 int CSA, int capA, int rblk, int m, const float *__restrict__ nb,
                                                 const float *__restrict__ tloose, float *__restrict__ ttight, int32_t *__restrict__ kept,
                                                 int *__restrict__ keptn, SweepCtl *__restrict__ ctl, const float *__restrict__ klim, int mpad) {
    // one wavefront per row.  Pass over the row's 2 * CSA lane lists: its own records (tag) are compacted into LDS as (column, D, D');
    // the cut is found by BISECTION on D' (count of D' >= cut by ballots; any cut is valid -- the verification decides -- so 11
    // halvings replace three 64-lane sorts); the records with D >= cut go to the row's kept list.
    __shared__ int32_t scol[4][SELCAP];
    __shared__ float sd[4][SELCAP], sdp[4][SELCAP];
    const int lane = threadIdx.x & 63, wave = dgg::wave_id();
    const int64_t lrow = (int64_t)blockIdx.x * 4 + wave;
    if (lrow >= rows) return;
    const RowLists rl = row_lists(lrow, rblk);
    const float nbi = nb[row0 + lrow];
    if (klim && m <= 64) {                                              // k_limit: the order statistic scales with the ranks the row has to settle
        const int Lr = __builtin_amdgcn_readfirstlane(klimit_len(klim[lrow], 64));
        const int mfull = m;
        m = (m * Lr + 63) / 64 + mpad;
        m = m < 8 ? 8 : (m > mfull ? mfull : m);
    }
    int total = 0;
    bool over = false;
    // A sweep lane records its LARGEST D of a 16-column block for each of its hits in that block (sw_sweep, block_best): of records that
    // share (row, D) -- consecutive entries of one lane list -- only the first is known to have that D; the others take no part in the
    // order statistic (D' = -inf; they stay candidates: their true D is not larger).  At N = 100 000 3 % of the hitting lanes hold two
    // hits of a block; at N = 8192 the loose radius admits 4 % of the columns, most hits shared a block, the statistic saw them all at
    // the best one's distance and the tight radius came out too small for 20 % of the rows.
    auto take = [&](bool mine, const int2 &c, float nbj) {            // compaction of the row's own records into LDS
        const int py = __shfl_up(c.y, 1, 64), px = __shfl_up(c.x, 1, 64);
        const bool dup = mine && lane > 0 && py == c.y && (((uint32_t)px ^ (uint32_t)c.x) >> 28) == 0u && (((uint32_t)px ^ (uint32_t)c.x) & COLMASK) < 32u;
        const unsigned long long mk = __ballot(mine);
        const int at = total + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
        if (mine && at < SELCAP) {
            const float d = __int_as_float(c.y);
            scol[wave][at] = (int32_t)((uint32_t)c.x & COLMASK);
            sd[wave][at] = d;
            sdp[wave][at] = dup ? -3.0e38f : d - 0.5f * SL_UPPER * (nbi + nbj);
        }
        total += __builtin_popcountll(mk);
    };
    // The lane lists are read LG at a time: counts, first chunks and norm gathers of a group are each ONE round trip (a loop
    // over the lists paid three dependent memory latencies per list: 0.29 ms, all of it waiting).
    constexpr int LG = 8;
    const int nl = 2 * CSA;
    auto list_id = [&](int s) { const int h2 = s >= CSA ? 1 : 0; return (rl.base + 32 * h2) * CSA + (s - h2 * CSA); };   // (no division)
    const int myc = lane < nl ? (int)cntA[list_id(lane)] : 0;
    for (int g0 = 0; g0 < nl; g0 += LG) {
        int n[LG];
        int2 c[LG];
        float nbj[LG];
        bool mine[LG];
#pragma unroll
        for (int k = 0; k < LG; k++) {
            const int s = g0 + k;
            n[k] = s < nl ? __builtin_amdgcn_readlane(myc, s < nl ? s : 0) : 0;
            if (n[k] > capA) { over = true; n[k] = capA; }
            c[k] = make_int2(0, 0);
            if (lane < n[k]) c[k] = listA[list_id(s) * capA + lane];
        }
#pragma unroll
        for (int k = 0; k < LG; k++) {
            mine[k] = lane < n[k] && ((uint32_t)c[k].x >> 28) == rl.tag;
            nbj[k] = mine[k] ? nb[(uint32_t)c[k].x & COLMASK] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < LG; k++)
            if (n[k] > 0) take(mine[k], c[k], nbj[k]);                  // (wave-uniform)
#pragma unroll
        for (int k = 0; k < LG; k++) {                                  // lists longer than one chunk
            for (int base = 64; base < n[k]; base += 64) {
                const int e = base + lane;
                int2 cc = make_int2(0, 0);
                if (e < n[k]) cc = listA[list_id(g0 + k) * capA + e];
                const bool m2 = e < n[k] && ((uint32_t)cc.x >> 28) == rl.tag;
                take(m2, cc, m2 ? nb[(uint32_t)cc.x & COLMASK] : 0.0f);
            }
        }
    }
    const float tl = tloose[lrow];
    float tt = tl, sel = 0.0f;
    int nk = 0;
    if (!over && total <= SELCAP) {
        float v[SELCAP / 64];
        float hi = 0.0f;
#pragma unroll
        for (int k = 0; k < SELCAP / 64; k++) {
            v[k] = (k * 64 + lane < total) ? sdp[wave][k * 64 + lane] : -3.0e38f;
            hi = fmaxf(hi, v[k]);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) hi = fmaxf(hi, __shfl_xor(hi, off, 64));
        const int nch = (total + 63) >> 6;                              // chunks that hold data (wave-uniform)
        auto count_ge = [&](float x) {
            int c = 0;
#pragma unroll
            for (int k = 0; k < SELCAP / 64; k++)
                if (k < nch) c += __builtin_popcountll(__ballot(v[k] >= x));
            return c;
        };
        if (total >= m && count_ge(0.0f) >= m) {
            float lo = 0.0f;                                            // invariant: count(D' >= lo) >= m
            hi = hi * 1.0001f + 1e-30f;
            for (int it = 0; it < 11; it++) {
                const float mid = 0.5f * (lo + hi);
                if (count_ge(mid) >= m) lo = mid; else hi = mid;
            }
            const float cut = lo;
            const float slack = 1e-6f * (fabsf(tl) + cut) + 1e-7f;
            tt = fminf(tl, tl - cut + slack);
            sel = fmaxf(cut - 2.0f * slack, 0.0f);
        }
        // the row's phase-A hits inside the tight radius (D >= sel), compacted for sw_finalize
        for (int base = 0; base < total; base += 64) {
            const int e = base + lane;
            const bool keep = e < total && sd[wave][e] >= sel;
            const unsigned long long mk = __ballot(keep);
            const int at = nk + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
            if (keep && at < KCAP) kept[lrow * KCAP + at] = scol[wave][e];
            nk += __builtin_popcountll(mk);
        }
    } else over = true;
    if (lane == 0) {
        ttight[lrow] = tt;
        keptn[lrow] = over ? (KCAP + 1) : nk;                          // > KCAP: the row goes to the fallback
        if (ctl->stats_on) { atomicAdd(&ctl->nA, (unsigned long long)total); atomicAdd(&ctl->nAkept, (unsigned long long)nk); }
    }
}

// exact canonical squared distance of pair (i, j) (the fmaf chain of the oracle's pair_dist); i is wave-uniform: its features
// come through the scalar cache
template <int H>
__device__ __forceinline__ float exact_d2(const float *__restrict__ xp, int64_t i, int32_t j) {
    const float *xi = xp + i * H;
    const float4 *xj = reinterpret_cast<const float4 *>(xp + (int64_t)j * H);
    float4 b[H / 4];
#pragma unroll
    for (int c4 = 0; c4 < H / 4; c4++) b[c4] = xj[c4];
    float d2 = 0.0f;
#pragma unroll
    for (int c4 = 0; c4 < H / 4; c4++) {
        float df;
        df = __fadd_rn(xi[4 * c4 + 0], -b[c4].x); d2 = __fmaf_rn(df, df, d2);
        df = __fadd_rn(xi[4 * c4 + 1], -b[c4].y); d2 = __fmaf_rn(df, df, d2);
        df = __fadd_rn(xi[4 * c4 + 2], -b[c4].z); d2 = __fmaf_rn(df, df, d2);
        df = __fadd_rn(xi[4 * c4 + 3], -b[c4].w); d2 = __fmaf_rn(df, df, d2);
    }
    return d2;
}
template <int H>
__device__ __forceinline__ float exact_score0(const float *__restrict__ xp, int64_t i, int32_t j, float t) {
    return score_from_dist(c_sqrt(exact_d2<H>(xp, i, j)), t, false, 0.0f);
}

// ---- finalize -----------------------------------------------------------------------------------------------------------------
// one wavefront per row: the row's candidates (phase-A hits inside the tight radius + its phase-B records, ~160 columns) get
// their EXACT squared distance (one candidate per lane and batch, FCAP / 64 batches held in registers); the 64th smallest is
// found by bisection on the bit patterns (ballot counts; ~10 halvings: it stops as soon as exactly 64 lie below), and only the
// candidates within dist_64 + 4e-6 (the margin covers ties and the 1-ulp non-monotonicity of the canonical exp) -- 64 to ~66
// columns -- go through sqrt / exp and the ONE 64-lane sort.  (Scoring, sorting and merging every batch of 64 candidates
// was 2.5 sorts + merges per row: 0.63 ms.)
constexpr float SCORE_MIN_NORMAL = 4.8e-38f;   // 4 x the smallest normal float: below it exp(t d) has lost the bits that order distances
constexpr int FCAP = 320;               // candidates of one row that sw_finalize can hold; more: the row goes to the fallback
template <int H>
__global__ __launch_bounds__(256) void sw_finalize(const float *__restrict__ xp, const float *__restrict__ nb, int64_t row0, int64_t row1, float t,
                                                   const int32_t *__restrict__ kept, const int *__restrict__ keptn,
                                                   const uint32_t *__restrict__ listB, const unsigned short *__restrict__ cntB, int CSB, int capB,
                                                   int rblk, const float *__restrict__ ttight, SweepCtl *__restrict__ ctl,
                                                   int *__restrict__ faillist, int32_t *__restrict__ idx, float *__restrict__ val,
                                                   const float *__restrict__ klim) {
    __shared__ int32_t ccol[4][FCAP];
    __shared__ float cd2[4][FCAP];
    const int lane = threadIdx.x & 63, wave = dgg::wave_id();
    const int64_t lrow = (int64_t)blockIdx.x * 4 + wave;
    const int64_t i = row0 + lrow;
    if (i >= row1) return;
    const RowLists rl = row_lists(lrow, rblk);
    int32_t *cc = ccol[wave];
    // ranks the row has to settle: 64, or ceil(k_i + 8.5) + 1 with the learned degrees (the others come back as idx = -1)
    const int Lr = klim ? __builtin_amdgcn_readfirstlane(klimit_len(klim[lrow], 64)) : 64;
    int total = 0, nB = 0;                                             // wave-uniform
    bool ok = true;
    int why = 7;          // (diagnostics, DGG_SWEEP_STATS=1: ctl->pad[why]++ for a failed row -- 0 phase-A overflow, 1 phase-B overflow, 2 more than
                          //  FCAP candidates, 3 fewer than Lr, 4 list not full, 5 score below the normal range, 6 last distance outside the radius)
    auto push = [&](bool keep, int32_t col) {
        const unsigned long long m = __ballot(keep);
        const int at = total + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (keep && at < FCAP) cc[at] = col;
        total += __builtin_popcountll(m);
    };
    {   // phase-A hits inside the tight radius (compacted by sw_select)
        int n = keptn[lrow];
        if (n > KCAP) { ok = false; n = 0; why = 0; }                   // overflow (of the kept list or of a phase-A lane list)
        for (int base = 0; base < n; base += 64) {
            const int e = base + lane;
            push(e < n, e < n ? kept[lrow * KCAP + e] : 0);
        }
    }
    const int nkept = total;
    {   // phase-B records: the lane lists are read LG at a time (counts and first chunks of a group: one round trip each)
        constexpr int LG = 8;
        const int nl = 2 * CSB;
        auto list_id = [&](int s) { const int h2 = s >= CSB ? 1 : 0; return (rl.base + 32 * h2) * CSB + (s - h2 * CSB); };   // (no division)
        const int myc = lane < nl ? (int)cntB[list_id(lane)] : 0;
        if (__ballot(myc > capB) != 0ull) { if (ok) why = 1; ok = false; }
        for (int g0 = 0; ok && g0 < nl; g0 += LG) {
            int n[LG];
            uint32_t rec[LG];
#pragma unroll
            for (int k = 0; k < LG; k++) {
                const int s = g0 + k;
                n[k] = s < nl ? __builtin_amdgcn_readlane(myc, s < nl ? s : 0) : 0;
                rec[k] = 0xffffffffu;
                if (lane < n[k]) rec[k] = listB[list_id(s) * capB + lane];
            }
#pragma unroll
            for (int k = 0; k < LG; k++)
                if (n[k] > 0) push(lane < n[k] && (rec[k] >> 28) == rl.tag, (int32_t)(rec[k] & COLMASK));
#pragma unroll
            for (int k = 0; k < LG; k++) {
                for (int base = 64; base < n[k]; base += 64) {
                    const int e = base + lane;
                    const uint32_t r2 = e < n[k] ? listB[list_id(g0 + k) * capB + e] : 0xffffffffu;
                    push(e < n[k] && (r2 >> 28) == rl.tag, (int32_t)(r2 & COLMASK));
                }
            }
        }
    }
    nB = total - nkept;
    if (total > FCAP || total < Lr) { if (ok) why = total > FCAP ? 2 : 3; ok = false; }
    uint64_t list = DGG_EMPTY_KEY;
    if (ok) {
        // exact squared distances, FCAP / 64 candidates per lane
        uint32_t u[FCAP / 64];
#pragma unroll
        for (int k = 0; k < FCAP / 64; k++) {
            u[k] = 0xffffffffu;
            if (k * 64 < total) {                                       // wave-uniform
                const int e = k * 64 + lane;
                if (e < total) {
                    const float d2 = exact_d2<H>(xp, i, cc[e]);
                    cd2[wave][e] = d2;
                    u[k] = __float_as_uint(d2);                         // d2 >= 0: the bit patterns are ordered like the values
                }
            }
        }
        const int nch = (total + 63) >> 6;                              // chunks that hold data (wave-uniform)
        auto count_le = [&](uint32_t x) {
            int c = 0;
#pragma unroll
            for (int k = 0; k < FCAP / 64; k++)
                if (k < nch) c += __builtin_popcountll(__ballot(u[k] <= x));
            return c;
        };
        // bisection: smallest tau with count(d2 <= tau) >= Lr -- stopped early once a value with exactly Lr below is met
        uint32_t lo = 0u, hi = 0u;
#pragma unroll
        for (int k = 0; k < FCAP / 64; k++) hi = max(hi, u[k] == 0xffffffffu ? 0u : u[k]);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) hi = max(hi, (uint32_t)__shfl_xor((int)hi, off, 64));
        uint32_t tau = hi;                                              // count(<= hi) = total >= Lr
        while (lo < hi) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            const int c = count_le(mid);
            if (c >= Lr) { hi = mid; tau = mid; if (c == Lr) break; } else lo = mid + 1u;
        }
        const float dcut = c_sqrt(__uint_as_float(tau)) + 4e-6f;
        const float d2cut = dcut * dcut * (1.0f + 1e-6f);
        // the candidates inside the cut: compacted, scored, sorted (one batch unless there are ties at the cut)
        int nin = 0;
        int32_t *sel = cc;                                              // compacted in place (positions only move down)
        for (int base = 0; base < total; base += 64) {
            const int e = base + lane;
            const float d2 = e < total ? cd2[wave][e] : 0.0f;
            const int32_t col = e < total ? cc[e] : 0;
            const bool in = e < total && d2 <= d2cut;
            const unsigned long long m = __ballot(in);
            const int at = nin + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (in) { sel[at] = col; cd2[wave][at] = d2; }
            nin += __builtin_popcountll(m);
        }
        for (int base = 0; base < nin; base += 64) {
            const int e = base + lane;
            uint64_t key = DGG_EMPTY_KEY;
            if (e < nin) key = make_key(score_from_dist(c_sqrt(cd2[wave][e]), t, false, 0.0f), sel[e]);
            key = wave_sort<false>(key, lane);
            list = wave_merge_top64_asc(list, key, lane);
        }
        // verification: full list, and its 64th distance (+ margins for the log and the rounding of the canonical exp) inside the
        // radius the sweeps tested against: R = nb_i + 2 t_tight
        const uint64_t k63 = shfl_u64(list, Lr - 1);                    // the last rank that carries weight
        // (a score below the normal range -- t d < -87: features hundreds of units apart -- no longer identifies a distance: zero and
        //  denormal scores tie over whole shells and the oracle breaks ties by column, also among the columns the sweeps rejected; such
        //  a row is left to the fallback, which scores every column)
        if (k63 == DGG_EMPTY_KEY || key_val(k63) < SCORE_MIN_NORMAL) { ok = false; why = k63 == DGG_EMPTY_KEY ? 4 : 5; }
        else {
            const float d63 = c_log(fmaxf(key_val(k63), 1e-37f)) / t + 1e-5f;
            const float R = fmaf(2.0f, ttight[lrow], nb[i]);
            ok = d63 * d63 * (1.0f + 1e-5f) <= R * (1.0f - 1e-5f) - 1e-7f * nb[i];
            if (!ok) why = 6;
        }
    }
    if (ok) {
        const bool live = lane < Lr;                                    // (ranks beyond Lr are not settled: idx = -1, as dgg_klimit_truncate leaves them)
        idx[lrow * 64 + lane] = live ? key_col(list) : -1;
        val[lrow * 64 + lane] = live ? key_val(list) : 0.0f;
    } else if (lane == 0) {
        faillist[atomicAdd(&ctl->nfail, 1)] = (int)lrow;
    }
    if (ctl->stats_on && lane == 0) { atomicAdd(&ctl->nB, (unsigned long long)nB); if (!ok) atomicAdd(&ctl->pad[why], 1); }
}

// ---- fallback: rows whose radius failed verification, every column scored exactly ------------------------------------------------
// The first FB_MAX failed rows are split over FB_NCH column chunks each (a handful of failed rows must not run as a handful of
// workgroups: 3 rows took 0.69 ms that way): sw_fallback_part keeps a chunk's best 64, sw_fallback_merge merges a row's chunks.
// Rows beyond FB_MAX (every guess wrong: adversarial input) are redone by sw_fallback_rows, one workgroup per row at a time.
constexpr int FB_MAX = 4096, FB_NCH = 32, FB_CHUNK_CAP = 32;     // FB_MAX * FB_NCH partial lists in the workspace
// chunks per failed row: few failed rows (the usual handful) are cut finer, so that the scan of a row is spread over more workgroups
__device__ __forceinline__ int fb_chunks(int nf) { const int c = (FB_MAX * FB_NCH) / (nf > 0 ? nf : 1); return c > FB_CHUNK_CAP ? FB_CHUNK_CAP : (c < FB_NCH ? FB_NCH : c); }
// merge a block of keys into a descending 64-entry list: keys that cannot enter (below the current 64th) are dropped first, a
// handful of survivors is inserted one by one, otherwise sort + merge.  (The fallback's blocks after the first few hold 0-2 keys that
// still enter: the full 64-lane sort + merge per block was most of its time.)
__device__ __noinline__ uint64_t fb_merge(uint64_t list, uint64_t key, int lane) {      // (noinline: ONE copy of the unrolled sort + merge network -- a kernel that runs for microseconds pays ~0.1 us per 64 bytes of code it touches for the first time)
    const uint64_t k63 = shfl_u64(list, 63);
    if (key <= k63) key = DGG_EMPTY_KEY;                              // (k63 == DGG_EMPTY_KEY while the list is not full: nothing dropped)
    uint64_t live = __ballot(key != DGG_EMPTY_KEY);
    if (live == 0ull) return list;
    if (__builtin_popcountll(live) <= 12 && k63 != DGG_EMPTY_KEY) {
        while (live != 0ull) {                                          // wave-uniform
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
template <int H>
__device__ __forceinline__ uint64_t fallback_scan(const float *__restrict__ xp, int64_t i, int64_t c0, int64_t c1, float t, uint64_t (&lists)[4][64],
                                                  int lane, int wave) {
    uint64_t list = DGG_EMPTY_KEY;
    // four 64-column blocks per step: their gathers are independent and in flight together (a block per step paid one memory latency
    // per block)
    for (int64_t j0 = c0 + (int64_t)wave * 64; j0 < c1; j0 += 4 * 256) {
        uint64_t key[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int64_t j = j0 + (int64_t)q * 256 + lane;
            key[q] = j < c1 ? make_key(exact_score0<H>(xp, i, (int32_t)j, t), (int32_t)j) : DGG_EMPTY_KEY;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) list = fb_merge(list, key[q], lane);
    }
    lists[wave][lane] = list;
    __syncthreads();
    if (wave == 0)
        for (int w = 1; w < 4; w++) list = fb_merge(list, lists[w][lane], lane);
    __syncthreads();
    return list;                                                       // valid in wavefront 0
}
template <int H>
__global__ __launch_bounds__(256) void sw_fallback_part(const float *__restrict__ xp, int64_t N, int64_t row0, float t, const SweepCtl *__restrict__ ctl,
                                                        const int *__restrict__ faillist, uint64_t *__restrict__ partial) {
    __shared__ uint64_t lists[4][64];
    const int lane = threadIdx.x & 63, wave = dgg::wave_id();
    const int nf = ctl->nfail < FB_MAX ? ctl->nfail : FB_MAX;
    const int nch = fb_chunks(nf);
    const int64_t chunk = (N + nch - 1) / nch;
    for (int item = blockIdx.x; item < nf * nch; item += gridDim.x) {
        const int f = item / nch, c = item % nch;
        const int64_t c0 = c * chunk, c1 = (c0 + chunk < N) ? c0 + chunk : N;
        const uint64_t list = fallback_scan<H>(xp, row0 + faillist[f], c0, c1 > c0 ? c1 : c0, t, lists, lane, wave);
        if (wave == 0) partial[(int64_t)item * 64 + lane] = list;
    }
}
__global__ __launch_bounds__(256) void sw_fallback_merge(const SweepCtl *__restrict__ ctl, const int *__restrict__ faillist,
                                                         const uint64_t *__restrict__ partial, int32_t *__restrict__ idx, float *__restrict__ val) {
    // one WORKGROUP per failed row: each wavefront merges a quarter of the row's partial lists (loaded eight at a time: independent
    // loads, one round trip), wavefront 0 merges the four results
    __shared__ uint64_t lists[4][64];
    const int lane = threadIdx.x & 63, wave = dgg::wave_id();
    const int nf = ctl->nfail < FB_MAX ? ctl->nfail : FB_MAX;
    const int nch = fb_chunks(nf);
    for (int f = blockIdx.x; f < nf; f += gridDim.x) {
        uint64_t list = DGG_EMPTY_KEY;
        const uint64_t *pf = partial + (int64_t)f * nch * 64 + lane;
        for (int c0 = wave * 8; c0 < nch; c0 += 32) {
            uint64_t key[8];
#pragma unroll
            for (int q = 0; q < 8; q++) key[q] = c0 + q < nch ? pf[(int64_t)(c0 + q) * 64] : DGG_EMPTY_KEY;
#pragma unroll
            for (int q = 0; q < 8; q++) list = fb_merge(list, key[q], lane);
        }
        lists[wave][lane] = list;
        __syncthreads();
        if (wave == 0) {
            for (int w = 1; w < 4; w++) list = fb_merge(list, lists[w][lane], lane);
            const int lrow = faillist[f];
            const bool empty = list == DGG_EMPTY_KEY;
            idx[(int64_t)lrow * 64 + lane] = empty ? -1 : key_col(list);
            val[(int64_t)lrow * 64 + lane] = empty ? 0.0f : key_val(list);
        }
        __syncthreads();
    }
}
template <int H>
__global__ __launch_bounds__(256) void sw_fallback_rows(const float *__restrict__ xp, int64_t N, int64_t row0, float t, const SweepCtl *__restrict__ ctl,
                                                        const int *__restrict__ faillist, int32_t *__restrict__ idx, float *__restrict__ val) {
    __shared__ uint64_t lists[4][64];
    const int lane = threadIdx.x & 63, wave = dgg::wave_id();
    const int nfail = ctl->nfail;
    for (int f = FB_MAX + blockIdx.x; f < nfail; f += gridDim.x) {
        const int lrow = faillist[f];
        const uint64_t list = fallback_scan<H>(xp, row0 + lrow, 0, N, t, lists, lane, wave);
        if (wave == 0) {
            const bool empty = list == DGG_EMPTY_KEY;
            idx[(int64_t)lrow * 64 + lane] = empty ? -1 : key_col(list);
            val[(int64_t)lrow * 64 + lane] = empty ? 0.0f : key_val(list);
        }
    }
}

// ---- host side -------------------------------------------------------------------------------------------------------------------
static float env_float(const char *name, float dflt) {
    const char *e = getenv(name);
    return e ? (float)atof(e) : dflt;
}
// headroom on the pilot's radius; tests shrink it (DGG_FAST_GUESS_SCALE) to force the verification / fallback path
static const float g_guess_scale = env_float("DGG_FAST_GUESS_SCALE", 1.02f);
// order statistic (of the UPPER bounds) of the phase-A sample that becomes the tight radius (<= 64); 0 disables the tightening (B runs with the loose radius)
static const int g_select_m = (int)env_float("DGG_SWEEP_M", 32.0f);
static const int g_stats = (int)env_float("DGG_SWEEP_STATS", 0.0f);

Plan make_plan(int64_t rows, int64_t N, int h) {
    Plan p;
    p.rows = rows; p.N = N;
    p.ntiles = (int)((N + TC - 1) / TC);
    p.npad = (int64_t)p.ntiles * TC;
    p.nA = (p.ntiles + 3) / 4;
    p.nB = p.ntiles - p.nA;
    p.rblk = h <= 64 ? 4 : 2;
    p.rw = 128 * p.rblk;
    p.nrb = (int)((rows + p.rw - 1) / p.rw);
    p.rbx = (p.nrb + 7) / 8;
    // column segments per row block: the count that minimises (rounds of 512 resident workgroups) x (tiles per workgroup) --
    // a phase of 588 workgroups runs a second round at 15 % occupancy (phase A took 0.57 ms with 3 segments, 0.36 with 5)
    auto segs = [&](int nset) {
        int best = 1;
        int64_t best_cost = INT64_MAX;
        for (int cs = 1; cs <= 16; cs++) {
            if (cs > 1 && nset / cs < 4) break;                         // at least a few tiles per workgroup
            const int64_t rounds = ((int64_t)p.nrb * cs + 511) / 512;
            const int64_t cost = rounds * ((nset + cs - 1) / cs + 2);   // + 2: per-workgroup prologue, in tiles
            if (cost < best_cost) { best_cost = cost; best = cs; }
        }
        return best;
    };
    p.csb = segs(p.nB);
    p.csa = segs(p.nA);
    p.capa = p.rblk * CAPA_ROW / (2 * p.csa);                          // per lane list: the records of the lane's RBLK rows
    p.capb = p.rblk * CAPB_ROW / (2 * p.csb);
    // pilot sample: the radius is the PILOT_M-th largest of 4 pt block maxima per half row (a block = 16 columns), and it should admit
    // the fraction q = LOOSE_TARGET / N of the columns: PILOT_M = 4 pt (1 - (1 - q)^16).  (Linearised -- pt = ntiles 2 PILOT_M /
    // LOOSE_TARGET -- it under-samples small graphs: 2 tiles at N = 8192, where the PILOT_M-th of 8 maxima is their minimum and the
    // radius admitted 21 % of the columns; 98 % of the rows overflowed their lists into the exhaustive fallback.)
    const double q = (double)LOOSE_TARGET / (double)(N > LOOSE_TARGET ? N : LOOSE_TARGET + 1);
    int pt = (int)ceil((double)PILOT_M / (4.0 * (1.0 - pow(1.0 - q, 16.0))));
    pt = pt > p.ntiles ? p.ntiles : pt;
    p.pt = pt < 1 ? 1 : pt;
    return p;
}

struct Layout {
    size_t xw, nb, ctl, fail, tl, tt, kept, keptn, cnta, cntb, la, lb, part, total;
};
Layout make_layout(const Plan &p, int h) {
    Layout L;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    L.xw = take((size_t)p.npad * (h + 16) * 2);
    L.nb = take((size_t)p.N * 4);
    L.ctl = take(sizeof(SweepCtl));
    L.fail = take((size_t)p.rows * 4);
    L.tl = take((size_t)p.rows * 4);
    L.tt = take((size_t)p.rows * 4);
    L.keptn = take((size_t)p.rows * 4);
    L.kept = take((size_t)p.rows * KCAP * 4);
    // (per-row arrays at their worst case over the segment counts, so that the size depends on rows and N only)
    const size_t rpad = (size_t)p.nrb * p.rw;                          // the lane lists cover whole workgroups of rows
    L.cnta = take(rpad * 2 * 16 * 2);
    L.cntb = take(rpad * 2 * 16 * 2);
    L.la = take(rpad * CAPA_ROW * 8);
    L.lb = take(rpad * CAPB_ROW * 4);
    L.part = take((size_t)(p.rows < FB_MAX ? p.rows : FB_MAX) * FB_NCH * 64 * 8);
    L.total = off;
    return L;
}

template <int H, int RBLK>
int launch_sweep(const float *xp, int64_t N, int64_t row0, int64_t row1, float t, const float *klim, int32_t *idx, float *val, void *ws, hipStream_t st) {
    const Plan p = make_plan(row1 - row0, N, H);
    const Layout L = make_layout(p, H);
    char *w = reinterpret_cast<char *>(ws);
    uint16_t *xw = reinterpret_cast<uint16_t *>(w + L.xw);
    float *nb = reinterpret_cast<float *>(w + L.nb);
    SweepCtl *ctl = reinterpret_cast<SweepCtl *>(w + L.ctl);
    int *faillist = reinterpret_cast<int *>(w + L.fail);
    float *tl = reinterpret_cast<float *>(w + L.tl), *tt = reinterpret_cast<float *>(w + L.tt);
    int32_t *kept = reinterpret_cast<int32_t *>(w + L.kept);
    int *keptn = reinterpret_cast<int *>(w + L.keptn);
    unsigned short *cnta = reinterpret_cast<unsigned short *>(w + L.cnta), *cntb = reinterpret_cast<unsigned short *>(w + L.cntb);
    int2 *la = reinterpret_cast<int2 *>(w + L.la);
    uint32_t *lb = reinterpret_cast<uint32_t *>(w + L.lb);
    uint64_t *part = reinterpret_cast<uint64_t *>(w + L.part);
    // (the control block is set by a memset + a 4-byte memset pattern: no host memory is read asynchronously)
    if (dgg_check_hip(hipMemsetAsync(ctl, 0, sizeof(SweepCtl), st), "sweep memset") != 0) return DGG_ERR_HIP;
    if (g_stats && dgg_check_hip(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(&ctl->stats_on), 1, 1, st), "sweep memset") != 0) return DGG_ERR_HIP;
    const int64_t rows = row1 - row0;
    hipLaunchKernelGGL(sw_prep<H>, dim3((unsigned)((p.npad + 3) / 4)), dim3(256), 0, st, xp, N, p.npad, xw, nb);
    hipLaunchKernelGGL(sw_pilot<H>, dim3((unsigned)((rows + 127) / 128)), dim3(256), 0, st, xw, nb, row0, row1, p.ntiles, p.pt, g_guess_scale, tl, klim, (int)env_float("DGG_SWEEP_KPAD", 0.0f));
    hipLaunchKernelGGL((sw_sweep<H, RBLK, true>), dim3((unsigned)(8 * p.rbx * p.csa)), dim3(256), 0, st, xw, tl, p.npad, row0, row1, p.nA, p.nrb, p.rbx, p.csa,
                       p.capa, (void *)la, cnta);
    const int m = g_select_m > 64 ? 64 : g_select_m;
    hipLaunchKernelGGL(sw_select, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, la, cnta, rows, row0, p.csa, p.capa, p.rblk, m > 0 ? m : (1 << 30), nb, tl, tt, kept, keptn, ctl, klim, (int)env_float("DGG_SWEEP_MPAD", 8.0f));
    hipLaunchKernelGGL((sw_sweep<H, RBLK, false>), dim3((unsigned)(8 * p.rbx * p.csb)), dim3(256), 0, st, xw, tt, p.npad, row0, row1, p.nB, p.nrb, p.rbx, p.csb,
                       p.capb, (void *)lb, cntb);
    hipLaunchKernelGGL(sw_finalize<H>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, xp, nb, row0, row1, t, kept, keptn, lb, cntb, p.csb, p.capb,
                       p.rblk, tt, ctl, faillist, idx, val, klim);
    hipLaunchKernelGGL(sw_fallback_part<H>, dim3(1024), dim3(256), 0, st, xp, N, row0, t, ctl, faillist, part);
    hipLaunchKernelGGL(sw_fallback_merge, dim3(1024), dim3(256), 0, st, ctl, faillist, part, idx, val);
    hipLaunchKernelGGL(sw_fallback_rows<H>, dim3(512), dim3(256), 0, st, xp, N, row0, t, ctl, faillist, idx, val);
    return dgg_check_launch("allpairs_topk_sweep");
}

}  // namespace

size_t dgg_allpairs_sweep_ws_bytes(int64_t rows, int64_t N, int h) { return make_layout(make_plan(rows, N, h), h).total; }
// byte offset of the control block {int nfail; int stats_on; u64 nA, nAkept, nB} inside the workspace (diagnostics)
size_t dgg_allpairs_sweep_ctl_offset(int64_t rows, int64_t N, int h) { return make_layout(make_plan(rows, N, h), h).ctl; }

bool dgg_allpairs_sweep_supported(int h, int noise_mode, int K) {
    return K == 64 && (h == 16 || h == 32 || h == 64 || h == 128) && noise_mode == 0;
}

int dgg_allpairs_topk_sweep_impl(const float *xp, int64_t N, int h, int64_t row0, int64_t row1, float t, int K, const float *klim, int32_t *idx, float *val,
                                 void *workspace, size_t ws_bytes, hipStream_t st) {
    if (!dgg_allpairs_sweep_supported(h, 0, K))
        return dgg_set_error(DGG_ERR_UNSUPPORTED, "unperturbed sweep needs K=64 and latent_dim in {16,32,64,128}");
    if (row1 <= row0) return 0;
    if (!workspace || ws_bytes < dgg_allpairs_sweep_ws_bytes(row1 - row0, N, h))
        return dgg_set_error(DGG_ERR_ARG, "unperturbed sweep: workspace too small (dgg_allpairs_workspace_bytes)");
    switch (h) {
        case 16: return launch_sweep<16, 4>(xp, N, row0, row1, t, klim, idx, val, workspace, st);
        case 32: return launch_sweep<32, 4>(xp, N, row0, row1, t, klim, idx, val, workspace, st);
        case 64: return launch_sweep<64, 4>(xp, N, row0, row1, t, klim, idx, val, workspace, st);
        default: return launch_sweep<128, 2>(xp, N, row0, row1, t, klim, idx, val, workspace, st);
    }
}

// row-minimum sweep: workspace = [xw npad (h+16) fp16][nb N f32][ekey rows u32]
namespace {
struct RowminLayout { size_t xw, nb, ekey, total; };
RowminLayout rowmin_layout(int64_t rows, int64_t N, int h) {
    RowminLayout L;
    const int64_t npad = (N + TC - 1) / TC * TC;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    L.xw = take((size_t)npad * (h + 16) * 2);
    L.nb = take((size_t)N * 4);
    L.ekey = take((size_t)rows * 4);
    L.total = off;
    return L;
}
template <int H, int RBLK>
int launch_rowmin(const float *xp, int64_t N, int64_t row0, int64_t row1, float t, float *lpub, void *ws, hipStream_t st) {
    const int64_t rows = row1 - row0;
    const RowminLayout L = rowmin_layout(rows, N, H);
    char *w = reinterpret_cast<char *>(ws);
    uint16_t *xw = reinterpret_cast<uint16_t *>(w + L.xw);
    float *nb = reinterpret_cast<float *>(w + L.nb);
    uint32_t *ekey = reinterpret_cast<uint32_t *>(w + L.ekey);
    const int ntiles = (int)((N + TC - 1) / TC);
    const int64_t npad = (int64_t)ntiles * TC;
    const int rw = 128 * RBLK, nrb = (int)((rows + rw - 1) / rw), rbx = (nrb + 7) / 8;
    int cs = 1;                                                         // column segments: fill the chip (512 resident workgroups) with few rounds
    {
        int64_t best = INT64_MAX;
        for (int c = 1; c <= 32; c++) {
            if (c > 1 && ntiles / c < 4) break;
            const int64_t rounds = ((int64_t)nrb * c + 511) / 512, cost = rounds * ((ntiles + c - 1) / c + 2);
            if (cost < best) { best = cost; cs = c; }
        }
    }
    if (dgg_check_hip(hipMemsetAsync(ekey, 0, (size_t)rows * 4, st), "rowmin memset") != 0) return DGG_ERR_HIP;     // key 0 = below every float
    hipLaunchKernelGGL(sw_prep<H>, dim3((unsigned)((npad + 3) / 4)), dim3(256), 0, st, xp, N, npad, xw, nb);
    hipLaunchKernelGGL((sw_rowmin<H, RBLK>), dim3((unsigned)(8 * rbx * cs)), dim3(256), 0, st, xw, npad, row0, row1, ntiles, nrb, rbx, cs, ekey);
    hipLaunchKernelGGL(sw_rowmin_finish, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, ekey, nb, row0, rows, t, lpub);
    return dgg_check_launch("allpairs_rowmin_bound");
}
}  // namespace

extern "C" {
// bytes of workspace dgg_allpairs_rowmin_bound needs
size_t dgg_allpairs_rowmin_ws_bytes(int64_t rows, int64_t N, int h) { return rowmin_layout(rows, N, h).total; }
// lpub [row1-row0]: for every row i a rigorous UPPER bound of log p_ij = log(exp(t ||xp_i - xp_j||) + 1e-8) over all j != i (reference
// dgm.py:1618-1623), from an fp16-MFMA lower bound of the squared distance to the row's nearest other node (dgg_topk_sweep.hip).  The
// early-out tests of the noise generators take it in place of the distance-free bound 1e-8.  latent_dim in {16, 32, 64, 128}, t < 0.
int dgg_allpairs_rowmin_bound(const float *xp, int64_t N, int h, int64_t row0, int64_t row1, float t, float *lpub, void *workspace, size_t ws_bytes,
                              void *stream) {
    if (!xp || !lpub || row0 < 0 || row1 > N || row0 > row1) return dgg_set_error(DGG_ERR_ARG, "allpairs_rowmin_bound: bad row range or NULL arrays");
    if (h != 16 && h != 32 && h != 64 && h != 128) return dgg_set_error(DGG_ERR_UNSUPPORTED, "allpairs_rowmin_bound: latent_dim in {16,32,64,128}");
    if (row1 == row0) return 0;
    if (!workspace || ws_bytes < dgg_allpairs_rowmin_ws_bytes(row1 - row0, N, h)) return dgg_set_error(DGG_ERR_ARG, "allpairs_rowmin_bound: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    switch (h) {
        case 16: return launch_rowmin<16, 4>(xp, N, row0, row1, t, lpub, workspace, st);
        case 32: return launch_rowmin<32, 4>(xp, N, row0, row1, t, lpub, workspace, st);
        case 64: return launch_rowmin<64, 4>(xp, N, row0, row1, t, lpub, workspace, st);
        default: return launch_rowmin<128, 2>(xp, N, row0, row1, t, lpub, workspace, st);
    }
}
}

// front end of the unperturbed chunked rows (dgg_topk_anywide.hip): xw / nb / ethr live in the caller's workspace
namespace {
template <int H, int RBLK>
int launch_plain_front(const float *xp, int64_t N, int64_t row0, int64_t row1, const float *klim, const int32_t *cptr, int cslot, uint32_t *cand,
                       int32_t *ncand, float *rad, void *ws, hipStream_t st) {
    const int64_t rows = row1 - row0;
    const RowminLayout L = rowmin_layout(rows, N, H);
    char *w = reinterpret_cast<char *>(ws);
    uint16_t *xw = reinterpret_cast<uint16_t *>(w + L.xw);
    float *nb = reinterpret_cast<float *>(w + L.nb);
    float *ethr = reinterpret_cast<float *>(w + L.ekey);
    const int ntiles = (int)((N + TC - 1) / TC);
    const int64_t npad = (int64_t)ntiles * TC;
    const int rw = 128 * RBLK, nrb = (int)((rows + rw - 1) / rw), rbx = (nrb + 7) / 8;
    hipLaunchKernelGGL(sw_prep<H>, dim3((unsigned)((npad + 3) / 4)), dim3(256), 0, st, xp, N, npad, xw, nb);
    hipLaunchKernelGGL(pw_pilot<H>, dim3((unsigned)((rows + 127) / 128)), dim3(256), 0, st, xw, nb, N, row0, row1, ntiles, klim, cptr, ethr, rad,
                       env_float("DGG_PLAIN_TARGET_FACTOR", 2.5f), env_float("DGG_PLAIN_TARGET_ADD", 64.0f));
    hipLaunchKernelGGL((pw_sweep<H, RBLK>), dim3((unsigned)(8 * rbx * PW_CS)), dim3(256), 0, st, xw, npad, row0, row1, ntiles, nrb, rbx, ethr, cptr, cslot,
                       cand, ncand);
    return dgg_check_launch("allpairs_topk_anywide: unperturbed front end");
}
}  // namespace
size_t dgg_plain_wide_front_ws_bytes(int64_t rows, int64_t N, int h) { return rowmin_layout(rows, N, h).total; }
int dgg_plain_wide_sublists(void) { return 2 * PW_CS; }
int dgg_plain_wide_front_impl(const float *xp, int64_t N, int h, int64_t row0, int64_t row1, const float *klim, const int32_t *cptr, int cslot,
                              uint32_t *cand, int32_t *ncand, float *rad, void *ws, hipStream_t st) {
    switch (h) {
        case 16: return launch_plain_front<16, 4>(xp, N, row0, row1, klim, cptr, cslot, cand, ncand, rad, ws, st);
        case 32: return launch_plain_front<32, 4>(xp, N, row0, row1, klim, cptr, cslot, cand, ncand, rad, ws, st);
        case 64: return launch_plain_front<64, 4>(xp, N, row0, row1, klim, cptr, cslot, cand, ncand, rad, ws, st);
        default: return launch_plain_front<128, 2>(xp, N, row0, row1, klim, cptr, cslot, cand, ncand, rad, ws, st);
    }
}
