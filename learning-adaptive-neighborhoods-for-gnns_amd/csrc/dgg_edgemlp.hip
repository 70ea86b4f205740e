// dgg_edgemlp.hip -- the edge-MLP scorers of the live class on a candidate edge list:
//   u-v-A_uv dgm.py:1628-1644, u-v-deg 1645-1670, u-v-deg-dist 1671-1702, edge_conv 1703-1719, A_uv 1720-1725.
// The reference gathers [x_u, x_v, extras] per edge and runs sigmoid(W2 act(W1 . + b1) + b2) as two GEMMs over E rows.
// By linearity the first layer splits into per-NODE products AB = xp [Wa | Wb]^T ([N, 2*hw], one MFMA GEMM,
// dgg_linear_fwd) and per-edge terms, so an edge costs 2*hw gathered floats instead of a (2h+extra) x hw product:
//   z_o = A[u][o] + B[v][o] (+ deg_u wdu_o + deg_v wdv_o) (+ ex_e wex_o) + b1_o;  p = sigmoid(sum_o act(z_o) w2_o + b2)
// in exactly the operation order of oracle/dgg_oracle.c (mlp_edge_p).  The scorer writes one probability per candidate
// edge; perturbation + top-K (edgelist_topk_p) and the soft top-k are shared with the u-v-dist path.
#include "dgg_common.h"
#include <stdlib.h>
#include "dgg_api_internal.h"

using namespace dgg;

namespace {

__device__ __forceinline__ float act_apply(float z, int act) { return act == 1 ? (z > 0.0f ? z : __fmul_rn(0.01f, z)) : z; }

// ---- forward: one thread per candidate edge --------------------------------------------------------------------------
__global__ __launch_bounds__(256) void edge_mlp_fwd_kernel(
    const float *__restrict__ AB, const float *__restrict__ xp, int h, int hw, const int32_t *__restrict__ erow,
    const int32_t *__restrict__ col, int64_t E, const float *__restrict__ deg, const float *__restrict__ ex_in, int ex_mode,
    float t_ex, const float *__restrict__ wdu, const float *__restrict__ wdv, const float *__restrict__ wex,
    const float *__restrict__ b1, const float *__restrict__ w2, const float *__restrict__ b2, int act,
    float *__restrict__ p_edge, float *__restrict__ ex_out) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    const int64_t u = erow[e], v = col[e];
    float ex = 0.0f;
    if (ex_mode == 1) {
        ex = ex_in[e];
    } else if (ex_mode == 2) {                                    // exp(t ||xp_u - xp_v||), canonical chain (dgm.py:1684-1686)
        ex = c_exp(__fmul_rn(t_ex, c_sqrt(pair_d2_thread(xp + u * h, xp + v * h, h))));
    }
    if (ex_out) ex_out[e] = ex;
    const float *A = AB + u * 2 * hw, *B = AB + v * 2 * hw + hw;
    const float du = deg ? deg[u] : 0.0f, dv = deg ? deg[v] : 0.0f;
    float s = 0.0f;
    auto term = [&](float a, float b, int o) {                     // (one hidden unit; the sum over o stays in the oracle's order)
        float z = __fadd_rn(a, b);
        if (deg) { z = __fmaf_rn(du, wdu[o], z); z = __fmaf_rn(dv, wdv[o], z); }
        if (ex_mode != 0) z = __fmaf_rn(ex, wex[o], z);
        z = __fadd_rn(z, b1[o]);
        s = __fmaf_rn(act_apply(z, act), w2[o], s);
    };
    if ((hw & 15) == 0 && (reinterpret_cast<uintptr_t>(AB) & 15) == 0) {
        // 16-byte loads, sixteen hidden units (8 loads) in flight: a thread's two rows are its own cache lines, and one 4-byte load per
        // unit and operand left the 1.6 wavefronts per SIMD of a citation graph waiting on 128 dependent round trips (41 -> 12 us)
        for (int o = 0; o < hw; o += 16) {
            float4 a4[4], b4[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                a4[c] = *reinterpret_cast<const float4 *>(A + o + 4 * c);
                b4[c] = *reinterpret_cast<const float4 *>(B + o + 4 * c);
            }
#pragma unroll
            for (int c = 0; c < 4; c++) {
                term(a4[c].x, b4[c].x, o + 4 * c); term(a4[c].y, b4[c].y, o + 4 * c + 1);
                term(a4[c].z, b4[c].z, o + 4 * c + 2); term(a4[c].w, b4[c].w, o + 4 * c + 3);
            }
        }
    } else {
        for (int o = 0; o < hw; o++) term(A[o], B[o], o);
    }
    s = __fadd_rn(s, b2[0]);
    p_edge[e] = __fdiv_rn(1.0f, __fadd_rn(1.0f, c_exp(-s)));
}

// ---- perturbation + per-row top-K on given edge probabilities: one wavefront per row ----------------------------------
// key payload = position of the candidate in its row (columns of a row ascend -> same tie order as the column key)
__global__ __launch_bounds__(256) void edgelist_topk_p_kernel(
    const float *__restrict__ p_edge, int64_t N, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    int noise_mode, const float *__restrict__ G, int64_t ldG, uint32_t s0, uint32_t s1, int K, int32_t *__restrict__ idx,
    float *__restrict__ val, int32_t *__restrict__ eid) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + dgg::wave_id();
    if (i >= N) return;
    const bool sym = noise_mode == 3;
    uint64_t list = DGG_EMPTY_KEY;
    const int64_t e0 = rowptr[i], e1 = rowptr[i + 1];
    for (int64_t eb = e0; eb < e1; eb += 64) {
        const int64_t e = eb + lane;
        uint64_t key = DGG_EMPTY_KEY;
        if (e < e1) {
            float v = p_edge[e];
            if (noise_mode != 0) {
                const int32_t j = col[e];
                const float g = noise_mode == 1 ? G[i * ldG + j] : pair_noise(s0, s1, (uint32_t)i, (uint32_t)j, sym);
                v = c_exp(__fadd_rn(c_log(__fadd_rn(v, 1e-8f)), g));
            }
            key = make_key(v, (int32_t)(e - e0));
        }
        key = wave_sort_desc(key, lane);
        list = wave_merge_top64(list, key, lane);
    }
    if (lane < K) {
        const bool empty = list == DGG_EMPTY_KEY;
        const int64_t e = e0 + key_col(list);
        idx[i * K + lane] = empty ? -1 : col[e];
        val[i * K + lane] = empty ? 0.0f : key_val(list);
        eid[i * K + lane] = empty ? -1 : (int32_t)e;
    }
}

__device__ __forceinline__ bool ellrow_ok(const int64_t *rowptr, int K) { return rowptr == nullptr && K <= 64; }

// d B_j += sum of the rows of dz_rec that belong to node j's records (consecutive: nodeptr); hw / 4 lanes per node, 16-byte loads
__global__ __launch_bounds__(256) void edge_mlp_colsum(const int *__restrict__ nodeptr, int64_t ncols, const float *__restrict__ dz_rec,
                                                       int64_t dz_rows, int hw, float *__restrict__ dAB) {
    const int LPE = hw / 4;
    const int64_t g = ((int64_t)blockIdx.x * 256 + threadIdx.x) / LPE;
    const int c = threadIdx.x % LPE;
    if (g >= ncols) return;
    const int p0 = nodeptr[g];
    int p1 = nodeptr[g + 1];
    if (p1 > dz_rows) p1 = (int)dz_rows;                         // (records beyond the buffer went the atomic way)
    if (p0 >= p1) return;
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    constexpr int UQ = 8;                                        // rows in flight (a hub's few hundred rows are walked by one lane group)
    for (int pb = p0; pb < p1; pb += UQ) {
        float4 v[UQ];
#pragma unroll
        for (int u = 0; u < UQ; u++) v[u] = reinterpret_cast<const float4 *>(dz_rec + (int64_t)(pb + u < p1 ? pb + u : p1 - 1) * hw)[c];
#pragma unroll
        for (int u = 0; u < UQ; u++)
            if (pb + u < p1) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    float4 *o = reinterpret_cast<float4 *>(dAB + g * 2 * hw + hw) + c;
    float4 t = *o;
    t.x += acc.x; t.y += acc.y; t.z += acc.z; t.w += acc.w;
    *o = t;
}

// ---- backward: one wavefront per row, LPE = hw/VEC lanes per selected entry -------------------------------------------
// dA_i: registers -> plain store; dB_j: float atomics (256-B rows at hw = 64); parameter gradients: registers across the
// rows of a persistent workgroup -> LDS -> one atomic per workgroup and element.
template <int VEC>
__global__ __launch_bounds__(256) void edge_mlp_bwd_kernel(
    const float *__restrict__ AB, int64_t N, int hw, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ idx,
    const int32_t *__restrict__ eid, const float *__restrict__ val, const float *__restrict__ dval, int K,
    const float *__restrict__ deg,
    const float *__restrict__ ex, const float *__restrict__ wdu, const float *__restrict__ wdv,
    const float *__restrict__ wex, const float *__restrict__ b1, const float *__restrict__ w2,
    const float *__restrict__ b2, int act, int perturb, float *__restrict__ dAB, float *__restrict__ dpar,
    float *__restrict__ dex, const float *__restrict__ wrow = nullptr, const int *__restrict__ recpos = nullptr,
    float *__restrict__ dz_rec = nullptr, int64_t dz_rows = 0) {
    // dz_rec (VEC 4, ELL rows; dgg_edge_mlp_bwd_partp): the neighbour-side term d z of every selected entry goes, as ONE 16-byte store
    // per lane, to row recpos[entry] of dz_rec -- the entry's place among the destination-ordered records of the payload partition,
    // where the records of a node are consecutive -- and edge_mlp_colsum adds each node's rows; without it: hw float atomics per entry
    // onto dAB (6.9 M at the Pubmed shape: most of the kernel's time).  wrow: the weights the partition was built from (an entry has a
    // record iff idx >= 0 and w != 0); an entry with a record but no cotangent gets a zero row.
    extern __shared__ float red[];                               // [4 waves][5*hw + 1]
    const int LPE = hw / VEC, EPI = 64 / LPE;
    const int lane = threadIdx.x & 63, wave = dgg::wave_id(), c = lane % LPE, slot = lane / LPE;
    const int o0 = c * VEC;
    float wdu_r[VEC], wdv_r[VEC], wex_r[VEC], b1_r[VEC], w2_r[VEC];
#pragma unroll
    for (int q = 0; q < VEC; q++) {
        wdu_r[q] = deg ? wdu[o0 + q] : 0.0f; wdv_r[q] = deg ? wdv[o0 + q] : 0.0f; wex_r[q] = ex ? wex[o0 + q] : 0.0f;
        b1_r[q] = b1[o0 + q]; w2_r[q] = w2[o0 + q];
    }
    const float b2v = b2[0];
    float g_wdu[VEC] = {}, g_wdv[VEC] = {}, g_wex[VEC] = {}, g_b1[VEC] = {}, g_w2[VEC] = {}, g_b2 = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * 4 + wave; i < N; i += (int64_t)gridDim.x * 4) {
        float Ai[VEC], dA[VEC] = {};
#pragma unroll
        for (int q = 0; q < VEC; q++) Ai[q] = AB[i * 2 * hw + o0 + q];
        const float du = deg ? deg[i] : 0.0f;
        // entries of the row: ELL [i*K, i*K+K) or, with rowptr, the CSR range of a variable-width adjacency
        const int64_t base = rowptr ? rowptr[i] : i * K;
        int cnt = rowptr ? (int)(rowptr[i + 1] - base) : K;
        // ELL rows (K <= 64): one coalesced load of the row's columns and cotangents, then only the batches that hold an entry with
        // a non-zero cotangent are visited (a citation graph fills ~6 of the 64 slots: 2 batches of dependent gathers instead of 16)
        const bool ellrow = !rowptr && K <= 64;
        int32_t jrow = -1;
        float grow = 0.0f;
        const bool recs_on = VEC == 4 && dz_rec != nullptr && ellrow_ok(rowptr, K);
        if (ellrow) {
            if (lane < K) { jrow = idx[base + lane]; grow = dval[base + lane]; }
            const uint64_t am = __ballot(jrow >= 0 && grow != 0.0f);
            if (dex && lane < K) dex[base + lane] = 0.0f;           // (the visited batches overwrite their entries below)
            cnt = am ? 64 - __builtin_clzll(am) : 0;
            if (recs_on) {                                           // recorded entries without a cotangent: zero rows (rare)
                const float wl = lane < K ? wrow[base + lane] : 0.0f;
                uint64_t zm = __ballot(jrow >= 0 && wl != 0.0f && grow == 0.0f);
                while (zm != 0ull) {                                 // wave-uniform
                    const int r = __builtin_ctzll(zm);
                    zm &= zm - 1;
                    const int rp = recpos[i * 64 + r];
                    if (rp >= 0 && rp < dz_rows && lane < hw / 4)
                        reinterpret_cast<float4 *>(dz_rec + (int64_t)rp * hw)[lane] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                }
            }
        }
        for (int r0 = 0; r0 < cnt; r0 += EPI) {
            const int r = r0 + slot;
            const int64_t en = base + r;
            int32_t j;
            float g;
            if (ellrow) {
                j = __shfl(jrow, r & 63, 64);
                g = __shfl(grow, r & 63, 64);
                if (r >= cnt) { j = -1; g = 0.0f; }
            } else {
                j = r < cnt ? idx[en] : -1;
                g = r < cnt ? dval[en] : 0.0f;
            }
            const bool actv = j >= 0 && g != 0.0f;
            if (__ballot(actv) == 0ull) {
                if (!ellrow && dex && c == 0 && r < cnt) dex[en] = 0.0f;
                continue;
            }
            float z[VEC], hid[VEC], part = 0.0f;
            float dv = 0.0f, exv = 0.0f;
            if (actv) {
                dv = deg ? deg[j] : 0.0f;
                exv = ex ? ex[eid ? eid[en] : en] : 0.0f;
#pragma unroll
                for (int q = 0; q < VEC; q++) {
                    float zz = Ai[q] + AB[(int64_t)j * 2 * hw + hw + o0 + q];
                    zz = fmaf(du, wdu_r[q], zz); zz = fmaf(dv, wdv_r[q], zz); zz = fmaf(exv, wex_r[q], zz);
                    zz += b1_r[q];
                    z[q] = zz;
                    hid[q] = act_apply(zz, act);
                    part = fmaf(hid[q], w2_r[q], part);
                }
            } else {
#pragma unroll
                for (int q = 0; q < VEC; q++) { z[q] = 0.0f; hid[q] = 0.0f; }
            }
            for (int off = 1; off < LPE; off <<= 1) part += __shfl_xor(part, off, 64);
            float ds = 0.0f;
            if (actv) {
                const float p = 1.0f / (1.0f + c_exp(-(part + b2v)));
                const float dp = perturb ? g * val[en] / (p + 1e-8f) : g;
                ds = dp * p * (1.0f - p);
            }
            float de = 0.0f;
            float dzv[VEC];
#pragma unroll
            for (int q = 0; q < VEC; q++) {
                const float dh = ds * w2_r[q];
                const float dz = act == 1 ? (z[q] > 0.0f ? dh : 0.01f * dh) : dh;
                dzv[q] = dz;
                dA[q] += dz;
                g_wdu[q] = fmaf(dz, du, g_wdu[q]); g_wdv[q] = fmaf(dz, dv, g_wdv[q]); g_wex[q] = fmaf(dz, exv, g_wex[q]);
                g_b1[q] += dz;
                g_w2[q] = fmaf(ds, hid[q], g_w2[q]);
                de = fmaf(dz, wex_r[q], de);
            }
            bool stored = false;
            if constexpr (VEC == 4) {
                if (recs_on && actv) {                               // (an active entry has w != 0: it has a record)
                    const int rp = recpos[i * 64 + r];
                    if (rp >= 0 && rp < dz_rows) {
                        *reinterpret_cast<float4 *>(dz_rec + (int64_t)rp * hw + o0) = make_float4(dzv[0], dzv[1], dzv[2], dzv[3]);
                        stored = true;
                    }
                }
            }
            if (actv && !stored) {
#pragma unroll
                for (int q = 0; q < VEC; q++)
                    if (dzv[q] != 0.0f) atomicAdd(dAB + (int64_t)j * 2 * hw + hw + o0 + q, dzv[q]);
            }
            if (c == 0) g_b2 += ds;
            if (dex) {
                for (int off = 1; off < LPE; off <<= 1) de += __shfl_xor(de, off, 64);
                if (c == 0 && r < cnt) dex[en] = de;
            }
        }
        for (int off = LPE; off < 64; off <<= 1) {
#pragma unroll
            for (int q = 0; q < VEC; q++) dA[q] += __shfl_xor(dA[q], off, 64);
        }
        if (slot == 0) {
#pragma unroll
            for (int q = 0; q < VEC; q++) dAB[i * 2 * hw + o0 + q] = dA[q];
        }
    }
    // parameter gradients: slots -> lanes of slot 0 -> LDS across waves -> global
    for (int off = LPE; off < 64; off <<= 1) {
#pragma unroll
        for (int q = 0; q < VEC; q++) {
            g_wdu[q] += __shfl_xor(g_wdu[q], off, 64); g_wdv[q] += __shfl_xor(g_wdv[q], off, 64);
            g_wex[q] += __shfl_xor(g_wex[q], off, 64); g_b1[q] += __shfl_xor(g_b1[q], off, 64);
            g_w2[q] += __shfl_xor(g_w2[q], off, 64);
        }
        g_b2 += __shfl_xor(g_b2, off, 64);
    }
    const int NP = 5 * hw + 1;
    if (slot == 0) {
        float *mine = red + wave * NP;
#pragma unroll
        for (int q = 0; q < VEC; q++) {
            mine[o0 + q] = g_wdu[q]; mine[hw + o0 + q] = g_wdv[q]; mine[2 * hw + o0 + q] = g_wex[q];
            mine[3 * hw + o0 + q] = g_b1[q]; mine[4 * hw + o0 + q] = g_w2[q];
        }
        if (c == 0) mine[5 * hw] = g_b2;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < NP; e += 256) {
        const float t = (red[e] + red[NP + e]) + (red[2 * NP + e] + red[3 * NP + e]);
        if (t != 0.0f) atomicAdd(dpar + e, t);
    }
}

bool pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }

}  // namespace

extern "C" {

int dgg_edge_mlp_fwd(const float *AB, const float *xp, int64_t N, int h, int hw, const int32_t *erow, const int32_t *col,
                     int64_t E, const float *deg, const float *ex_in, int ex_mode, float t_ex, const float *wdu,
                     const float *wdv, const float *wex, const float *b1, const float *w2, const float *b2, int act,
                     float *p_edge, float *ex_out, void *stream) {
    (void)N;
    if (ex_mode < 0 || ex_mode > 2) return dgg_set_error(DGG_ERR_ARG, "edge_mlp_fwd: ex_mode must be 0 (none), 1 (array) or 2 (exp(t dist))");
    if ((ex_mode == 1 && !ex_in) || (ex_mode == 2 && !xp) || (ex_mode != 0 && !wex) || (deg && (!wdu || !wdv)))
        return dgg_set_error(DGG_ERR_ARG, "edge_mlp_fwd: missing extras / weights for the requested mode");
    if (E == 0) return 0;
    hipLaunchKernelGGL(edge_mlp_fwd_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, (hipStream_t)stream, AB, xp, h, hw,
                       erow, col, E, deg, ex_in, ex_mode, t_ex, wdu, wdv, wex, b1, w2, b2, act, p_edge, ex_out);
    return dgg_check_launch("edge_mlp_fwd");
}

int dgg_edgelist_topk_p(const float *p_edge, int64_t N, const int64_t *rowptr, const int32_t *col, int noise_mode,
                        const float *G, int64_t ldG, uint32_t s0, uint32_t s1, int K, int32_t *idx, float *val, int32_t *eid,
                        void *stream) {
    if (K < 1 || K > 64) return dgg_set_error(DGG_ERR_UNSUPPORTED, "ELL width K must be in [1,64]");
    if (noise_mode == 1 && !G) return dgg_set_error(DGG_ERR_ARG, "explicit noise requested but G is NULL");
    if (noise_mode < 0 || noise_mode > 3)
        return dgg_set_error(DGG_ERR_UNSUPPORTED, "edgelist_topk_p: noise_mode must be none / explicit / hash / symmetric hash");
    if (N == 0) return 0;
    hipLaunchKernelGGL(edgelist_topk_p_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p_edge, N,
                       rowptr, col, noise_mode, G, ldG, s0, s1, K, idx, val, eid);
    return dgg_check_launch("edgelist_topk_p");
}

// dAB [N, 2*hw] and dpar [5*hw + 1] = [dwdu | dwdv | dwex | db1 | dw2 | db2] are ACCUMULATED into (caller zeroes them);
// dex (nullable) [N,K] is overwritten.  rowptr == NULL: ELL adjacency (idx/eid/val/dval/dex are [N,K]); rowptr != NULL:
// CSR-valued adjacency (idx = col [E], val/dval/dex [E], eid NULL = identity, K ignored).
int dgg_edge_mlp_bwd(const float *AB, int64_t N, int hw, const int64_t *rowptr, const int32_t *idx, const int32_t *eid,
                     const float *val, const float *dval, int K, const float *deg, const float *ex, const float *wdu,
                     const float *wdv, const float *wex, const float *b1, const float *w2, const float *b2, int act,
                     int perturb, float *dAB, float *dpar, float *dex, void *stream) {
    if (!rowptr && (K < 1 || K > 64)) return dgg_set_error(DGG_ERR_UNSUPPORTED, "ELL width K must be in [1,64]");
    const int vec = hw % 4 == 0 ? 4 : 1;
    const int lpe = hw / vec;
    if (!pow2(lpe) || lpe > 64)
        return dgg_set_error(DGG_ERR_UNSUPPORTED, "edge_mlp_bwd: hidden width must be 1, 2 or 4 x a power of two (<= 256)");
    if ((ex && (!wex || (!eid && !rowptr))) || (deg && (!wdu || !wdv)))
        return dgg_set_error(DGG_ERR_ARG, "edge_mlp_bwd: missing extras / weights");
    if (N == 0) return 0;
    static const int64_t gmax = getenv("DGG_EMLP_GRID") ? atoll(getenv("DGG_EMLP_GRID")) : 1024;   // (Pubmed shape: 103 us at 2048 workgroups, 85-89 at 512-1024, 123 at 128: row latency against contended parameter sums)
    const unsigned grid = (unsigned)((N + 3) / 4 < gmax ? (N + 3) / 4 : gmax);
    const size_t lds = (size_t)4 * (5 * hw + 1) * sizeof(float);
    if (vec == 4)
        hipLaunchKernelGGL(edge_mlp_bwd_kernel<4>, dim3(grid), dim3(256), lds, (hipStream_t)stream, AB, N, hw, rowptr, idx, eid, val, dval, K,
                           deg, ex, wdu, wdv, wex, b1, w2, b2, act, perturb, dAB, dpar, dex);
    else
        hipLaunchKernelGGL(edge_mlp_bwd_kernel<1>, dim3(grid), dim3(256), lds, (hipStream_t)stream, AB, N, hw, rowptr, idx, eid, val, dval, K,
                           deg, ex, wdu, wdv, wex, b1, w2, b2, act, perturb, dAB, dpar, dex);
    return dgg_check_launch("edge_mlp_bwd");
}

// dgg_edge_mlp_bwd for an ELL block whose PAYLOAD PARTITION is at hand (dgg_partp_build of (idx, w) with the entry -> record map:
// dgg_partp_has_map(N)): the neighbour-side sums d B_j without float atomics -- the row kernel stores every selected entry's d z as a
// row of dz_rec in record order, edge_mlp_colsum adds each destination node's (consecutive) rows.  w [N,K]: the weights the partition
// was built from; dz_rec: dz_rows * hw floats of scratch, dz_rows >= the number of records (e.g. the number of candidate edges).
// hw a multiple of 4 (power-of-two lane groups), K <= 64; otherwise as dgg_edge_mlp_bwd.
int dgg_edge_mlp_bwd_partp(const float *AB, int64_t N, int hw, const int32_t *idx, const int32_t *eid, const float *val, const float *dval,
                           const float *w, int K, const float *deg, const float *ex, const float *wdu, const float *wdv, const float *wex,
                           const float *b1, const float *w2, const float *b2, int act, int perturb, const void *partp_ws, int64_t ncols,
                           float *dz_rec, int64_t dz_rows, float *dAB, float *dpar, float *dex, void *stream) {
    if (K < 1 || K > 64) return dgg_set_error(DGG_ERR_UNSUPPORTED, "ELL width K must be in [1,64]");
    const int lpe = hw / 4;
    if (hw % 4 != 0 || !pow2(lpe) || lpe > 64) return dgg_set_error(DGG_ERR_UNSUPPORTED, "edge_mlp_bwd_partp: hidden width must be 4 x a power of two (<= 256)");
    if ((ex && (!wex || !eid)) || (deg && (!wdu || !wdv)) || !w || !dz_rec || dz_rows <= 0)
        return dgg_set_error(DGG_ERR_ARG, "edge_mlp_bwd_partp: missing extras / weights / scratch");
    const int *nodeptr = nullptr, *recpos = nullptr;
    if (dgg_partp_internal_ptrs(partp_ws, N, K, ncols, &nodeptr, &recpos) != 0)
        return dgg_set_error(DGG_ERR_UNSUPPORTED, "edge_mlp_bwd_partp: no payload partition with an entry -> record map for this block");
    if (N == 0) return 0;
    static const int64_t gmax = getenv("DGG_EMLP_GRID") ? atoll(getenv("DGG_EMLP_GRID")) : 1024;
    const unsigned grid = (unsigned)((N + 3) / 4 < gmax ? (N + 3) / 4 : gmax);
    const size_t lds = (size_t)4 * (5 * hw + 1) * sizeof(float);
    hipLaunchKernelGGL(edge_mlp_bwd_kernel<4>, dim3(grid), dim3(256), lds, (hipStream_t)stream, AB, N, hw, (const int64_t *)nullptr, idx, eid, val, dval,
                       K, deg, ex, wdu, wdv, wex, b1, w2, b2, act, perturb, dAB, dpar, dex, w, recpos, dz_rec, dz_rows);
    hipLaunchKernelGGL(edge_mlp_colsum, dim3((unsigned)((ncols * lpe + 255) / 256)), dim3(256), 0, (hipStream_t)stream, nodeptr, ncols, dz_rec, dz_rows,
                       hw, dAB);
    return dgg_check_launch("edge_mlp_bwd_partp");
}

}  // extern "C"
