// dgg_topk_np.hip -- all-pairs top-64 for the PERTURBED score, pruned by the noise alone ("noise prefilter").
//
// Same contract and same bits as allpairs_topk_exhaustive (dgg_topk.hip):
//   p'_ij = exp(log(exp(-0.05 ||xp_i - xp_j||) + 1e-8) + G_ij),  64 largest per row     reference dgm.py:1618-1623,
//                                                                                         1213-1229, 1404
// Observation: log(exp(t d) + 1e-8) <= log(1 + 1e-8) for every distance d >= 0, so  log p'_ij <= G_ij + 1e-8.
// A pair whose NOISE is below the row's current 64th log-score (minus a safety margin) can never enter the row's
// top-64, whatever its distance.  G is a monotone function of the pair's 24-bit uniform, so the test is ONE integer
// compare on the raw 32-bit hash of the pair: no features are touched for the ~99 % of pairs that fail it.
// Survivors go to the row's pending list; when ~60 are pending, the wavefront scores them exactly (canonical fp32
// arithmetic of dgg_common.h: gathers the 64 candidate rows of xp), sorts them with the DPP bitonic network, merges
// them into the row's list (kept in the output arrays) and tightens the row's integer threshold.
//
// One wavefront = one workgroup = 64 rows, LANE = ROW: row key, threshold and pending count are per-lane registers;
// the column index is wave-uniform (scalar), so the sweep costs ~10 VALU instructions per 64 pairs and needs no LDS,
// no barriers and no atomics.  The distance only enters through the exact scoring of survivors; when distances
// dominate the noise (large |t| * spread of d) the MFMA-bounded kernel of dgg_topk_fast.hip prunes better and the
// dispatcher picks it instead.
#include "dgg_common.h"
#include "dgg_api_internal.h"

using namespace dgg;

namespace {

constexpr int STEP = 8;            // columns between flush checks
constexpr int FLUSH_AT = 56;       // flush a row once this many are pending (STEP more may arrive: <= 64 per flush)
constexpr int CAPN = 64;           // pending slots per row

template <int H>
__device__ __forceinline__ float exact_score_np(const float *__restrict__ xp, int64_t i, int32_t j, float t, bool sym,
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

// integer stage threshold from the row's 64th score: pairs with hash x < result cannot enter the list
__device__ __forceinline__ uint32_t noise_threshold(float pp63) {
    float gmin = __logf(pp63) - 1e-3f;                         // margin >> every rounding error in the chain
    float e1 = __expf(gmin * (-1.0f / 0.3f));                  // P(G >= gmin) = 1 - exp(-e1) <= e1
    float c = fminf(e1 * 16777216.0f, 16777216.0f);
    int um = 16777216 - (int)c - 2;
    um = um < 0 ? 0 : um;
    return (uint32_t)um << 8;
}

template <int H, bool SYM>
__global__ __launch_bounds__(64) void allpairs_topk_np(const float *__restrict__ xp, int *__restrict__ pend_g, int64_t N,
                                                       int64_t row0, int64_t row1, float t, uint32_t s0, uint32_t s1,
                                                       int32_t *__restrict__ idx, float *__restrict__ val) {
    const int lane = threadIdx.x;
    const int64_t rbase = row0 + (int64_t)blockIdx.x * 64;
    const int64_t i = rbase + lane;
    const bool rvalid = i < row1;
    const uint32_t iu = (uint32_t)(rvalid ? i : row1 - 1);
    int *pend = pend_g + ((int64_t)blockIdx.x * 64 + lane) * CAPN;   // this lane's pending list

    uint32_t k1, k2;
    rowkey(s0, s1, iu, k1, k2);
    uint32_t ta = rvalid ? 0u : 0xffffffffu;                    // accept everything until 64 candidates are known
    int cnt = 0;
    for (int e = lane; e < 64 * 64; e += 64) {
        int64_t gi = rbase + (e >> 6);
        if (gi < row1) { idx[(gi - row0) * 64 + (e & 63)] = -1; val[(gi - row0) * 64 + (e & 63)] = 0.0f; }
    }

    auto do_flush = [&](int fl) {
        const int64_t fi = rbase + fl;                          // wave-uniform
        const int64_t out_row = fi - row0;
        const int n = __shfl(cnt, fl, 64);
        const int *pl = pend_g + ((int64_t)blockIdx.x * 64 + fl) * CAPN;
        int32_t li = idx[out_row * 64 + lane];
        float lv = val[out_row * 64 + lane];
        uint64_t list = li >= 0 ? make_key(lv, li) : DGG_EMPTY_KEY;
        int32_t j = lane < n ? pl[lane] : -1;
        uint64_t key = DGG_EMPTY_KEY;
        if (j >= 0) key = make_key(exact_score_np<H>(xp, fi, j, t, SYM, s0, s1), j);
        key = wave_sort<false>(key, lane);                      // ascending: merges without a reversal
        list = wave_merge_top64_asc(list, key, lane);
        bool empty = list == DGG_EMPTY_KEY;
        idx[out_row * 64 + lane] = empty ? -1 : key_col(list);
        val[out_row * 64 + lane] = empty ? 0.0f : key_val(list);
        uint64_t k63 = shfl_u64(list, 63);
        if (lane == fl) {
            cnt = 0;
            if (k63 != DGG_EMPTY_KEY) ta = noise_threshold(key_val(k63));
        }
    };

    auto col_step = [&](uint32_t j) {                            // j: wave-uniform column
        uint32_t x;
        if (!SYM) {
            x = j ^ k1;
            x *= 0x7feb352dU; x ^= x >> 15; x += k2; x *= 0x846ca68bU;
        } else {
            uint32_t kj1, kj2;                                   // key of the column node: scalar ALU
            rowkey(s0, s1, j, kj1, kj2);
            uint32_t xa = pair_u24_keyed(k1, k2, j) << 8;       // i < j : keyed on the row
            uint32_t xb = pair_u24_keyed(kj1, kj2, iu) << 8;    // j < i : keyed on the column
            x = j > iu ? xa : xb;
            if (j == iu) x = 0xffffffffu;                        // zero-noise diagonal: always a candidate
        }
        if (x >= ta) {
            pend[cnt] = (int)j;
            cnt++;
        }
    };
    auto flush_full = [&]() {
        uint64_t need = __ballot(cnt >= FLUSH_AT);
        while (need) {
            int fl = __builtin_ctzll(need);
            need &= need - 1;
            do_flush(fl);
        }
    };
    const int64_t Nfull = N / STEP * STEP;
    for (int64_t j0 = 0; j0 < Nfull; j0 += STEP) {
#pragma unroll
        for (int u = 0; u < STEP; u++) col_step((uint32_t)(j0 + u));
        flush_full();
    }
    for (int64_t j = Nfull; j < N; j++) col_step((uint32_t)j);   // ragged tail (< STEP columns: cannot overflow)
    uint64_t need = __ballot(cnt > 0 && rvalid);
    while (need) {
        int fl = __builtin_ctzll(need);
        need &= need - 1;
        do_flush(fl);
    }
}

template <int H>
int launch_np(const float *xp, int64_t N, int64_t row0, int64_t row1, float t, int noise_mode, uint32_t s0, uint32_t s1,
              int32_t *idx, float *val, void *ws, hipStream_t st) {
    int *pend = reinterpret_cast<int *>(ws);
    dim3 grid((unsigned)((row1 - row0 + 63) / 64));
    if (noise_mode == 2)
        hipLaunchKernelGGL((allpairs_topk_np<H, false>), grid, dim3(64), 0, st, xp, pend, N, row0, row1, t, s0, s1, idx, val);
    else
        hipLaunchKernelGGL((allpairs_topk_np<H, true>), grid, dim3(64), 0, st, xp, pend, N, row0, row1, t, s0, s1, idx, val);
    return dgg_check_launch("allpairs_topk_np");
}

}  // namespace

size_t dgg_allpairs_np_ws_bytes(int64_t N) { return (((size_t)N + 63) / 64 * 64) * CAPN * 4; }

bool dgg_allpairs_np_supported(int h, int noise_mode, int K) {
    return K == 64 && (h == 8 || h == 16 || h == 32 || h == 64 || h == 128) && (noise_mode == 2 || noise_mode == 3);
}

int dgg_allpairs_topk_np_impl(const float *xp, int64_t N, int h, int64_t row0, int64_t row1, float t, int noise_mode,
                              uint32_t s0, uint32_t s1, int K, int32_t *idx, float *val, void *workspace, size_t ws_bytes,
                              hipStream_t st) {
    if (!dgg_allpairs_np_supported(h, noise_mode, K))
        return dgg_set_error(DGG_ERR_UNSUPPORTED, "noise-prefilter path needs K=64, in-kernel noise, latent_dim in {8,...,128}");
    if (!workspace || ws_bytes < dgg_allpairs_np_ws_bytes(N))
        return dgg_set_error(DGG_ERR_ARG, "noise-prefilter path: workspace too small (dgg_allpairs_workspace_bytes)");
    if (row1 <= row0) return 0;
    switch (h) {
        case 8: return launch_np<8>(xp, N, row0, row1, t, noise_mode, s0, s1, idx, val, workspace, st);
        case 16: return launch_np<16>(xp, N, row0, row1, t, noise_mode, s0, s1, idx, val, workspace, st);
        case 32: return launch_np<32>(xp, N, row0, row1, t, noise_mode, s0, s1, idx, val, workspace, st);
        case 64: return launch_np<64>(xp, N, row0, row1, t, noise_mode, s0, s1, idx, val, workspace, st);
        default: return launch_np<128>(xp, N, row0, row1, t, noise_mode, s0, s1, idx, val, workspace, st);
    }
}
