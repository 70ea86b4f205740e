// dgg_hardlit.hip -- the debug class's LITERAL `return_hard_or_soft` with dgg_hard=True (reference dgm.py:1294-1311), as an
// opt-in small-N compatibility path (args.dgg_hard_literal; N <= 8192).
//
//   adj_hard = ones_like(edge_p); adj_hard.scatter_(-1, idxs, (edge_p > 0.5).float()); out = (adj_hard - edge_p).detach() + edge_p
//
// edge_p is the ALREADY UNSORTED soft adjacency (select_top_k, dgm.py:1402-1421) and idxs the sort permutation of the perturbed
// scores (rank -> column, all N ranks): hard[i, idxs[i, r]] = [soft[i, r] > 0.5], i.e. the indicator read at COLUMN r lands on the
// column of RANK r.  The one of a strong neighbour at column c therefore sits at "the column whose score has rank c" -- which
// needs the FULL ranking of the row's N scores (non-candidates included: their probability is 0, perturbed exp(log(1e-8) + G)).
// One workgroup per row: the N scores with the canonical arithmetic of dgg_common.h, a bitonic sort of the (score, column) keys
// in LDS, then the lookups.  O(N^2 log^2 N): this is the reference's dense semantics, kept for drop-in parity only (the value is
// not a function of the graph); the default dgg_hard is the straight-through adjacency (dgg_softk_fwd mode 3).
#include "dgg_common.h"
#include "dgg_api_internal.h"

using namespace dgg;

namespace {

__global__ __launch_bounds__(256) void literal_hard_fwd(const float *__restrict__ xp, int64_t N, int h, const int64_t *__restrict__ rowptr,
                                                        const int32_t *__restrict__ col, float t, int noise_mode, const float *__restrict__ G,
                                                        int64_t ldG, uint32_t s0, uint32_t s1, const int32_t *__restrict__ idx,
                                                        const float *__restrict__ w, int K, float threshold, int npad,
                                                        int32_t *__restrict__ hidx, float *__restrict__ hval, int32_t *__restrict__ hsrc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t *keys = reinterpret_cast<uint64_t *>(smem);
    const int tid = threadIdx.x;
    const int64_t i = blockIdx.x;
    const bool perturb = noise_mode != 0, sym = noise_mode == 3;
    const float *xi = xp + i * h;
    auto noise = [&](int64_t j) {
        if (noise_mode == 1) return G[i * ldG + j];
        if (noise_mode >= 2) return pair_noise(s0, s1, (uint32_t)i, (uint32_t)j, sym);
        return 0.0f;
    };
    auto cand_score = [&](int64_t j) {                                // u-v-dist (dgm.py:1618-1623) + perturbation (dgm.py:1213-1229)
        const float d2 = pair_d2_thread(xi, xp + j * h, h);
        return score_from_dist(c_sqrt(d2), t, perturb, noise(j));
    };
    // scores of all N columns: candidates by distance, non-candidates from probability 0
    for (int64_t j = tid; j < npad; j += 256) {
        uint64_t key = DGG_EMPTY_KEY;
        if (j < N) {
            float v;
            if (rowptr == nullptr) v = cand_score(j);
            else v = perturb ? c_exp(__fadd_rn(c_log(__fadd_rn(0.0f, 1e-8f)), noise(j))) : 0.0f;
            key = make_key(v, (int32_t)j);
        }
        keys[j] = key;
    }
    if (rowptr != nullptr) {
        __syncthreads();
        for (int64_t e = rowptr[i] + tid; e < rowptr[i + 1]; e += 256) {
            const int32_t j = col[e];
            keys[j] = make_key(cand_score(j), j);
        }
    }
    __syncthreads();
    // bitonic sort, descending: (score desc, column asc) -- the reference's torch.sort(descending=True), ties broken by column
    for (int kb = 2; kb <= npad; kb <<= 1) {
        for (int d = kb >> 1; d > 0; d >>= 1) {
            for (int e = tid; e < npad; e += 256) {
                const int p = e ^ d;
                if (p > e) {
                    const uint64_t a = keys[e], b = keys[p];
                    const bool desc = (e & kb) == 0;
                    if ((a < b) == desc) { keys[e] = b; keys[p] = a; }
                }
            }
            __syncthreads();
        }
    }
    // the ones: for every strong neighbour (soft weight > threshold) at column c, the column of rank c
    if (tid < 64) {
        const int lane = tid;
        int32_t c = -1;
        float wv = 0.0f;
        if (lane < K) { c = idx[i * K + lane]; wv = w[i * K + lane]; }
        const bool strong = c >= 0 && wv > threshold;
        const int32_t hc = strong ? key_col(keys[c]) : -1;
        const unsigned long long m = __ballot(strong);
        const int at = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        const int n = __builtin_popcountll(m);
        // soft value AT the column that receives the one (0 unless that column is itself in the row's list): the forward value is
        // (1 - s) + s, the gradient goes to s
        int src = -1;
        float sat = 0.0f;
        for (int q = 0; q < K; q++) {
            const int32_t cq = __shfl(c, q, 64);
            const float wq = __shfl(wv, q, 64);
            if (strong && cq == hc) { src = q; sat = wq; }
        }
        if (lane < K) { hidx[i * K + lane] = -1; hval[i * K + lane] = 0.0f; hsrc[i * K + lane] = -1; }
        if (strong) {
            hidx[i * K + at] = hc;
            hval[i * K + at] = __fadd_rn(__fadd_rn(1.0f, -sat), sat);
            hsrc[i * K + at] = src;
        }
        (void)n;
    }
}

// d out / d soft = 1 at the positions of the ones (the output is `.to_sparse()`: only its stored entries carry gradient)
__global__ __launch_bounds__(256) void literal_hard_bwd(const int32_t *__restrict__ hsrc, const float *__restrict__ g, int64_t N, int K,
                                                        float *__restrict__ dw) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= N * K) return;
    const int src = hsrc[e];
    if (src >= 0) dw[(e / K) * K + src] = g[e];                        // (ranks are distinct: every soft slot is hit at most once)
}

}  // namespace

extern "C" {

int dgg_literal_hard_fwd(const float *xp, int64_t N, int h, const int64_t *rowptr, const int32_t *col, float t, int noise_mode,
                         const float *G, int64_t ldG, uint32_t s0, uint32_t s1, const int32_t *idx, const float *w, int K, float threshold,
                         int32_t *hidx, float *hval, int32_t *hsrc, void *stream) {
    if (N < 1 || N > 8192) return dgg_set_error(DGG_ERR_UNSUPPORTED, "literal dgg_hard path: 1 <= N <= 8192 (full per-row ranking in LDS)");
    if (K < 1 || K > 64) return dgg_set_error(DGG_ERR_UNSUPPORTED, "ELL width K must be in [1,64]");
    if (noise_mode < 0 || noise_mode > 3) return dgg_set_error(DGG_ERR_ARG, "literal dgg_hard path: noise_mode none / explicit / hash / symmetric hash");
    if (noise_mode == 1 && !G) return dgg_set_error(DGG_ERR_ARG, "explicit noise requested but G is NULL");
    int npad = 64;
    while (npad < N) npad <<= 1;
    hipLaunchKernelGGL(literal_hard_fwd, dim3((unsigned)N), dim3(256), (size_t)npad * 8, (hipStream_t)stream, xp, N, h, rowptr, col, t, noise_mode,
                       G, ldG, s0, s1, idx, w, K, threshold, npad, hidx, hval, hsrc);
    return dgg_check_launch("literal_hard_fwd");
}

int dgg_literal_hard_bwd(const int32_t *hsrc, const float *g, int64_t N, int K, float *dw, void *stream) {
    if (N <= 0) return 0;
    if (dgg_check_hip(hipMemsetAsync(dw, 0, (size_t)N * K * 4, (hipStream_t)stream), "literal_hard_bwd memset") != 0) return DGG_ERR_HIP;
    hipLaunchKernelGGL(literal_hard_bwd, dim3((unsigned)((N * K + 255) / 256)), dim3(256), 0, (hipStream_t)stream, hsrc, g, N, K, dw);
    return dgg_check_launch("literal_hard_bwd");
}

}  // extern "C"
