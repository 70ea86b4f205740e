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
//   stage A  (every pair, ~8 VALU ops)   perturbed: the 24-bit uniform of the pair's noise against the row's
//                                        integer threshold (score <= G_ij + log(1+1e-8), distance >= 0);
//                                        unperturbed: bf16-MFMA distance lower bound against the row's radius.
//   stage B  (pairs passing A)           upper bound  G_ij + log(exp(t*dL_ij) + 1e-8),  dL_ij a rigorous lower
//                                        bound of the distance from a bf16 MFMA Gram tile:
//                                        d2 >= (n_i + n_j)(1 - eps) - 2 <bf16(x_i), bf16(x_j)>,  eps = 2^-8 (1 + slack)
//   stage C  (pairs passing B)           appended to the row's pending buffer in LDS; when 64 are pending the
//                                        wavefront scores them exactly, bitonic-sorts and merges them into the
//                                        row's list, and tightens the thresholds (about log2(N/64) times per row).
//
// Layout: a workgroup (4 wavefronts) owns 128 rows, one wavefront 32 rows.  MFMA orientation D[a][b]: a = column
// of the tile, b = row, so that a LANE holds ONE ROW (b = lane & 31) and 16 columns: thresholds, row keys and norms
// are per-lane registers.  Column tiles (bf16, padded rows: conflict-free ds_read_b128) are staged through LDS,
// double-buffered, one barrier per 64 columns.
#include "dgg_common.h"
#include "dgg_api_internal.h"

using namespace dgg;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int WAVES = 4;
constexpr int RB = 32 * WAVES;     // rows per workgroup
constexpr int CT = 32;             // columns per MFMA tile
constexpr int SC = 64;             // columns staged per barrier
constexpr int CAP = 96;            // pending-candidate slots per row
constexpr int FLUSH_AT = 64;       // flush a row once this many are pending (a tile adds at most 32)
constexpr float EPS_BF16 = 0.0040f;   // 2^-8 (1 + 2^-9) bf16 rounding of both operands + fp32 accumulation slack

// ---- prologue: bf16 copy of the projected features and discounted squared norms ------------------------------
__global__ __launch_bounds__(256) void prep_kernel(const float *__restrict__ xp, int64_t N, int h,
                                                   __bf16 *__restrict__ xb, float *__restrict__ nb) {
    const int lane = threadIdx.x & 63;
    const int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
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

__device__ __forceinline__ int col_of_reg(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

// exact canonical score of pair (i, j); xi is wave-uniform
template <int H>
__device__ __forceinline__ float exact_score(const float *__restrict__ xp, int64_t i, int32_t j, float t, int noise_mode,
                                             uint32_t s0, uint32_t s1) {
    const float *xi = xp + i * H;
    const float4 *xj = reinterpret_cast<const float4 *>(xp + (int64_t)j * H);
    float d2 = 0.0f;
#pragma unroll 4
    for (int c4 = 0; c4 < H / 4; c4++) {
        float4 b = xj[c4];
        float df;
        df = __fadd_rn(xi[4 * c4 + 0], -b.x); d2 = __fmaf_rn(df, df, d2);
        df = __fadd_rn(xi[4 * c4 + 1], -b.y); d2 = __fmaf_rn(df, df, d2);
        df = __fadd_rn(xi[4 * c4 + 2], -b.z); d2 = __fmaf_rn(df, df, d2);
        df = __fadd_rn(xi[4 * c4 + 3], -b.w); d2 = __fmaf_rn(df, df, d2);
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

// score exactly and merge the pending candidates of local row lr (whole wavefront cooperates)
template <int H>
__device__ __forceinline__ void flush_row(const float *__restrict__ xp, int64_t i, int64_t out_row, int lr, int lane,
                                          int *pend, int *cnt, uint32_t *thrA, float *thrB, float t, int noise_mode,
                                          uint32_t s0, uint32_t s1, int32_t *__restrict__ idx, float *__restrict__ val) {
    const int n = __builtin_amdgcn_readfirstlane(cnt[lr]);
    int32_t li = idx[out_row * 64 + lane];
    float lv = val[out_row * 64 + lane];
    uint64_t list = li >= 0 ? make_key(lv, li) : DGG_EMPTY_KEY;
    for (int base = 0; base < n; base += 64) {
        int e = base + lane;
        int32_t j = e < n ? pend[lr * CAP + e] : -1;
        uint64_t key = DGG_EMPTY_KEY;
        if (j >= 0) key = make_key(exact_score<H>(xp, i, j, t, noise_mode, s0, s1), j);
        key = wave_sort_desc(key, lane);
        list = wave_merge_top64(list, key, lane);
    }
    bool empty = list == DGG_EMPTY_KEY;
    idx[out_row * 64 + lane] = empty ? -1 : key_col(list);
    val[out_row * 64 + lane] = empty ? 0.0f : key_val(list);
    uint64_t k63 = shfl_u64(list, 63);
    if (lane == 0) {
        cnt[lr] = 0;
        if (k63 != DGG_EMPTY_KEY) {
            RowThr th = thresholds_from_score(key_val(k63), t, noise_mode != 0);
            thrA[lr] = th.a;
            thrB[lr] = th.b;
        }
    }
}

template <int H, int NOISE>   // NOISE: 0 none, 2 hash, 3 symmetric hash
__global__ __launch_bounds__(WAVES * 64, 2) void allpairs_topk_fast(
    const float *__restrict__ xp, const __bf16 *__restrict__ xb, const float *__restrict__ nb, int64_t N, int64_t row0,
    int64_t row1, float t, uint32_t s0, uint32_t s1, int32_t *__restrict__ idx, float *__restrict__ val) {
    constexpr int KS = H / 16;                 // MFMA k-steps
    constexpr int STRIDE = H * 2 + 16;         // bytes per staged column (padded)
    constexpr int CHUNKS = SC * H * 2 / 16;    // 16-byte pieces per stage
    constexpr int CPT = (CHUNKS + WAVES * 64 - 1) / (WAVES * 64);
    __shared__ __attribute__((aligned(16))) unsigned char colA[2][SC * STRIDE];
    __shared__ float nbt[2][SC];
    __shared__ int pend[RB * CAP];
    __shared__ int cnt[RB];
    __shared__ uint32_t thrA[RB];
    __shared__ float thrB[RB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int lr = wave * 32 + r;                               // local row of this lane
    const int64_t rbase = row0 + (int64_t)blockIdx.x * RB;
    const int64_t i = rbase + lr;
    const bool rvalid = i < row1;
    const int64_t iv = rvalid ? i : row1 - 1;                   // clamp loads of invalid rows

    // per-row state
    if (tid < RB) {
        cnt[tid] = 0;
        thrA[tid] = NOISE == 0 ? __float_as_uint(3.0e38f) : 0u;   // accept everything until 64 candidates are known
        thrB[tid] = -3.0e38f;
    }
    for (int e = tid; e < RB * 64; e += WAVES * 64) {
        int64_t gi = rbase + (e >> 6);
        if (gi < row1) { idx[(gi - row0) * 64 + (e & 63)] = -1; val[(gi - row0) * 64 + (e & 63)] = 0.0f; }
    }
    // B operand (rows): lane holds xb[row][16s + 8hh .. +8) for every k-step
    bf16x8 bfr[KS];
#pragma unroll
    for (int s = 0; s < KS; s++) bfr[s] = *reinterpret_cast<const bf16x8 *>(xb + iv * H + 16 * s + 8 * hh);
    const float nbi = nb[iv];
    uint32_t k1 = 0, k2 = 0;
    if (NOISE == 2) rowkey(s0, s1, (uint32_t)iv, k1, k2);

    // staging registers
    uint4 stg[CPT];
    float stg_nb = 0.0f;
    auto stage_load = [&](int64_t c0) {
#pragma unroll
        for (int q = 0; q < CPT; q++) {
            int ch = tid + q * WAVES * 64;
            int jj = ch / (H / 8), part = ch % (H / 8);
            int64_t gj = c0 + jj;
            stg[q] = make_uint4(0, 0, 0, 0);
            if (ch < CHUNKS && gj < N) stg[q] = *reinterpret_cast<const uint4 *>(xb + gj * H + part * 8);
        }
        if (tid < SC) stg_nb = (c0 + tid < N) ? nb[c0 + tid] : 3.0e38f;
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int q = 0; q < CPT; q++) {
            int ch = tid + q * WAVES * 64;
            int jj = ch / (H / 8), part = ch % (H / 8);
            if (ch < CHUNKS) *reinterpret_cast<uint4 *>(&colA[buf][jj * STRIDE + part * 16]) = stg[q];
        }
        if (tid < SC) nbt[buf][tid] = stg_nb;
    };

    stage_load(0);
    stage_store(0);
    __syncthreads();

    const int nstages = (int)((N + SC - 1) / SC);
    for (int st = 0; st < nstages; st++) {
        const int buf = st & 1;
        const int64_t c0 = (int64_t)st * SC;
        if (st + 1 < nstages) stage_load(c0 + SC);
#pragma unroll
        for (int tile = 0; tile < SC / CT; tile++) {
            const int64_t cb = c0 + tile * CT;
            if (cb >= N) break;                                  // uniform
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 16; q++) acc[q] = 0.0f;
#pragma unroll
            for (int s = 0; s < KS; s++) {
                bf16x8 af = *reinterpret_cast<const bf16x8 *>(&colA[buf][(tile * CT + r) * STRIDE + (16 * s + 8 * hh) * 2]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr[s], acc, 0, 0, 0);
            }
            const uint32_t ta = rvalid ? thrA[lr] : (NOISE == 0 ? 0u : 0xffffffffu);
            const float tb = thrB[lr];
            const int jb = (int)cb + 4 * hh;
            if (NOISE == 0) {
                // stage A == stage B: distance lower bound against the row's radius
                const float rad2 = __uint_as_float(ta);
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const int a = (q & 3) + 8 * (q >> 2);
                    float nj = nbt[buf][tile * CT + a + 4 * hh];
                    float L2 = __fmaf_rn(-2.0f, acc[q], nbi + nj);
                    bool pass = rvalid && (L2 <= rad2);
                    if (pass) {
                        int slot = atomicAdd(&cnt[lr], 1);
                        pend[lr * CAP + slot] = jb + a;
                    }
                }
            } else {
                uint32_t xs[16];
                uint32_t mask = 0;
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const int a = (q & 3) + 8 * (q >> 2);
                    const uint32_t j = (uint32_t)(jb + a);
                    uint32_t x;
                    if (NOISE == 2) {
                        x = j ^ k1;
                        x *= 0x7feb352dU; x ^= x >> 15; x += k2; x *= 0x846ca68bU;
                    } else {
                        x = pair_u24(s0, s1, (uint32_t)iv, j, true) << 8;
                        if (j == (uint32_t)iv) x = 0xffffffffu;             // zero-noise diagonal: decide in stage B
                    }
                    xs[q] = x;
                    mask |= (x >= ta ? 1u : 0u) << q;
                }
                if (__ballot(mask != 0) != 0ull) {
#pragma unroll
                    for (int q = 0; q < 16; q++) {
                        const bool pa = (mask >> q) & 1u;
                        if (__ballot(pa) == 0ull) continue;                  // wave-uniform
                        const int a = (q & 3) + 8 * (q >> 2);
                        const int j = jb + a;
                        float nj = nbt[buf][tile * CT + a + 4 * hh];
                        float L2 = __fmaf_rn(-2.0f, acc[q], nbi + nj);
                        float dL = __fsqrt_rn(fmaxf(L2, 0.0f));
                        float lpub = __logf(__expf(t * dL) + 1e-8f);
                        // noise upper estimate: -log(U), U = u24 2^-24; series near 1 (hardware log is inexact there)
                        uint32_t u24 = xs[q] >> 8;
                        u24 = u24 == 0 ? 1u : u24;
                        float epsu = (float)(16777216u - u24) * 5.9604644775390625e-8f;
                        float ser = epsu * (1.0f + epsu * (0.5f + epsu * (0.33333334f + 0.25f * epsu)));
                        float nl = epsu < 0.015625f ? ser : -__logf((float)u24 * 5.9604644775390625e-8f);
                        float G = -0.3f * __logf(nl);
                        if (NOISE == 3 && j == (int)iv) G = 0.0f;
                        float yub = lpub + G + (3e-5f + 2e-5f * fabsf(lpub));
                        bool pass = pa && rvalid && (yub >= tb) && (j < N);
                        if (pass) {
                            int slot = atomicAdd(&cnt[lr], 1);
                            pend[lr * CAP + slot] = j;
                        }
                    }
                }
            }
            // flush rows of this wavefront whose buffer could overflow on the next tile
            int mycnt = cnt[lr];
            uint64_t need = __ballot(mycnt >= FLUSH_AT) & 0xffffffffull;
            while (need) {
                int rr = __builtin_ctzll(need);
                need &= need - 1;
                int flr = wave * 32 + rr;
                int64_t fi = rbase + flr;
                flush_row<H>(xp, fi, fi - row0, flr, lane, pend, cnt, thrA, thrB, t, NOISE, s0, s1, idx, val);
            }
        }
        if (st + 1 < nstages) stage_store(buf ^ 1);
        __syncthreads();
    }
    // final flush of everything still pending
    {
        int mycnt = cnt[lr];
        uint64_t need = __ballot(mycnt > 0 && rvalid) & 0xffffffffull;
        while (need) {
            int rr = __builtin_ctzll(need);
            need &= need - 1;
            int flr = wave * 32 + rr;
            int64_t fi = rbase + flr;
            flush_row<H>(xp, fi, fi - row0, flr, lane, pend, cnt, thrA, thrB, t, NOISE, s0, s1, idx, val);
        }
    }
}

template <int H>
int launch_fast(const float *xp, int64_t N, int64_t row0, int64_t row1, float t, int noise_mode, uint32_t s0, uint32_t s1,
                int32_t *idx, float *val, void *ws, hipStream_t st) {
    __bf16 *xb = reinterpret_cast<__bf16 *>(ws);
    float *nb = reinterpret_cast<float *>(reinterpret_cast<char *>(ws) + (((size_t)N * H * 2 + 255) / 256) * 256);
    hipLaunchKernelGGL(prep_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, st, xp, N, H, xb, nb);
    dim3 grid((unsigned)((row1 - row0 + RB - 1) / RB));
    if (noise_mode == 0)
        hipLaunchKernelGGL((allpairs_topk_fast<H, 0>), grid, dim3(WAVES * 64), 0, st, xp, xb, nb, N, row0, row1, t, s0, s1, idx, val);
    else if (noise_mode == 2)
        hipLaunchKernelGGL((allpairs_topk_fast<H, 2>), grid, dim3(WAVES * 64), 0, st, xp, xb, nb, N, row0, row1, t, s0, s1, idx, val);
    else
        hipLaunchKernelGGL((allpairs_topk_fast<H, 3>), grid, dim3(WAVES * 64), 0, st, xp, xb, nb, N, row0, row1, t, s0, s1, idx, val);
    return dgg_check_launch("allpairs_topk_fast");
}

}  // namespace

size_t dgg_allpairs_fast_ws_bytes(int64_t N, int h) {
    return (((size_t)N * h * 2 + 255) / 256) * 256 + (size_t)N * 4;
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
