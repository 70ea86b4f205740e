// dgg_linear.hip -- dense layers of the hot path on the fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Replaces the ATen GEMMs behind
//   node_encode_for_edges / node_encode_for_k   nn.Linear + LeakyReLU     reference dgm.py:1097-1100, 1123-1126
//   GCNConv                                     relu(mm(mm(adj,x), W))    reference model.py:594-598
//   GraphConvolution                            mm(support, weight)       reference model.py:41
// and their autograd.  On gfx950 an fp32-input MFMA is an exact k-ordered fmaf chain, so the forward is
// bit-identical to  acc = 0; acc = fmaf(x[c], w[c], acc) (c ascending); acc + bias; activation.
//
// Forward tile: a workgroup (4 wavefronts) owns 128 rows x 128 output columns; each wavefront owns a
// 32-row strip with four 32x32 accumulators; K is walked in blocks of 32 staged through LDS (row stride 33
// floats -> conflict-free ds_read_b32 of the one-float-per-lane MFMA operands).
#include "dgg_common.h"
#include "dgg_api_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BM = 128, BK = 32, LDP = BK + 1;

__device__ __forceinline__ float act_apply(float v, int act) {
    if (act == 1) return v > 0.0f ? v : __fmul_rn(0.01f, v);
    if (act == 2) return v > 0.0f ? v : 0.0f;
    return v;
}

// y[N,out] = act(x[N,d] * W^T + b);  w_layout 0: W[out][d], 1: W[d][out].  NACC 32-column accumulators per wave
// (tile width BN = 32*NACC is matched to `out`, so narrow layers do not pay for 128 columns of MFMAs)
template <int NACC>
__global__ __launch_bounds__(256) void linear_fwd_mfma(const float *__restrict__ x, int64_t N, int d,
                                                       const float *__restrict__ W, const float *__restrict__ b,
                                                       int out, int w_layout, int act, float *__restrict__ y) {
    constexpr int BN = 32 * NACC;
    __shared__ float xs[BM * LDP];
    __shared__ float ws[BN * LDP];   // [j][k] (+pad)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    f32x16 acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; a++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[a][r] = 0.0f;
    const int li = lane & 31, hh = lane >> 5;
    const bool vec4 = (d % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);   // wave-uniform
    for (int k0 = 0; k0 < d; k0 += BK) {
        __syncthreads();
        if (vec4) {   // 16-byte loads: 8 lanes cover one 128-byte row segment
#pragma unroll
            for (int q = 0; q < BM * BK / 4 / 256; q++) {
                const int e = tid + q * 256, r = e >> 3, c4 = (e & 7) * 4;
                const int64_t gi = m0 + r;
                float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (gi < N && k0 + c4 < d) v = *reinterpret_cast<const float4 *>(x + gi * d + k0 + c4);
                float *dst = xs + r * LDP + c4;
                dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
            }
        } else {
            for (int e = tid; e < BM * BK; e += 256) {
                int r = e / BK, c = e % BK;
                int64_t gi = m0 + r;
                int gk = k0 + c;
                xs[r * LDP + c] = (gi < N && gk < d) ? x[gi * d + gk] : 0.0f;
            }
        }
        if (w_layout == 0) {
            for (int e = tid; e < BN * BK; e += 256) {
                int j = e / BK, c = e % BK;
                int gj = n0 + j, gk = k0 + c;
                ws[j * LDP + c] = (gj < out && gk < d) ? W[(int64_t)gj * d + gk] : 0.0f;
            }
        } else {
            for (int e = tid; e < BN * BK; e += 256) {
                int c = e / BN, j = e % BN;
                int gj = n0 + j, gk = k0 + c;
                ws[j * LDP + c] = (gj < out && gk < d) ? W[(int64_t)gk * out + gj] : 0.0f;
            }
        }
        __syncthreads();
        const float *xa = xs + (wave * 32 + li) * LDP + hh;
#pragma unroll
        for (int s = 0; s < BK / 2; s++) {
            float av = xa[2 * s];
#pragma unroll
            for (int a = 0; a < NACC; a++) {
                float bv = ws[(a * 32 + li) * LDP + 2 * s + hh];
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[a], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int a = 0; a < NACC; a++) {
        int gj = n0 + a * 32 + li;
        if (gj >= out) continue;
        float bj = b ? b[gj] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
            int64_t gi = m0 + wave * 32 + row;
            if (gi < N) {
                float v = acc[a][r];
                if (b) v = __fadd_rn(v, bj);
                y[gi * out + gj] = act_apply(v, act);
            }
        }
    }
}

// dp = dy * act'(y)
__global__ void act_bwd_kernel(const float *__restrict__ y, const float *__restrict__ dy, int64_t n, int act,
                               float *__restrict__ dp) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float g = dy[i], v = y[i];
        if (act == 1) g = v > 0.0f ? g : 0.01f * g;
        else if (act == 2) g = v > 0.0f ? g : 0.0f;
        dp[i] = g;
    }
}

// Weight-gradient GEMM: C[M1,M2] += A[N,M1]^T B[N,M2] with tiny M1, M2 (<= a few hundred) and huge N.
// Stage 1: one wavefront owns a 32 x (32*NBLK) block of C over a chunk of GT_ROWS node rows.  The fp32 MFMA operands are
// one float per lane (A[n][o0+lane%32], B[n][c0+..+lane%32], n = k-step*2 + lane/32), i.e. 128-byte coalesced row
// segments, so they are loaded straight from global memory: no LDS, no barriers.  Partial blocks go to a slab
// [chunk][M1p][M2p] with plain stores -- NOT atomics: every chunk would hit the same few KB of C, and same-address fp32
// atomics run ~14x below the streaming atomic rate (MI355X_MICROARCH.md, "Global float atomics").
// Stage 2: gemm_tn_reduce sums the slab over chunks (32-way split, 32 atomics per output) into C / colsum.
constexpr int GT_ROWS = 256;
template <int NBLK>
__global__ __launch_bounds__(64) void gemm_tn_partial(const float *__restrict__ A, const float *__restrict__ B, int64_t N,
                                                      int M1, int M2, int M1p, int M2p, float *__restrict__ slab,
                                                      float *__restrict__ cs_slab) {
    const int lane = threadIdx.x, li = lane & 31, hh = lane >> 5;
    const int64_t r0 = (int64_t)blockIdx.x * GT_ROWS;
    const int64_t r1 = r0 + GT_ROWS < N ? r0 + GT_ROWS : N;
    const int o0 = blockIdx.y * 32, c0 = blockIdx.z * (NBLK * 32);
    const bool ov = o0 + li < M1;
    bool cv[NBLK];
#pragma unroll
    for (int a = 0; a < NBLK; a++) cv[a] = c0 + a * 32 + li < M2;
    f32x16 acc[NBLK];
#pragma unroll
    for (int a = 0; a < NBLK; a++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[a][r] = 0.0f;
    float csum = 0.0f;
    const float *ap = A + o0 + li;
    const float *bp = B + c0 + li;
    // k-step = 2 node rows (one per lane half); PF k-steps are loaded together so that their latencies overlap
    constexpr int PF = 8;
    for (int64_t nb = r0; nb < r1; nb += 2 * PF) {
        float av[PF], bv[PF][NBLK];
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const int64_t n = nb + 2 * u + hh;
            const bool nv = n < r1;
            av[u] = (nv && ov) ? ap[n * M1] : 0.0f;
#pragma unroll
            for (int a = 0; a < NBLK; a++) bv[u][a] = (nv && cv[a]) ? bp[n * M2 + a * 32] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < PF; u++) {
            csum += av[u];
#pragma unroll
            for (int a = 0; a < NBLK; a++) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u][a], acc[a], 0, 0, 0);
        }
    }
    float *sl = slab + (int64_t)blockIdx.x * M1p * M2p;
#pragma unroll
    for (int a = 0; a < NBLK; a++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            int go = o0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            sl[(int64_t)go * M2p + c0 + a * 32 + li] = acc[a][r];
        }
    }
    if (cs_slab && blockIdx.z == 0) {
        csum += __uint_as_float(dgg::xor_shfl<32>(__float_as_uint(csum), lane));
        if (hh == 0) cs_slab[(int64_t)blockIdx.x * M1p + o0 + li] = csum;
    }
}

constexpr int GT_SPLIT = 32;
__global__ __launch_bounds__(256) void gemm_tn_reduce(const float *__restrict__ slab, const float *__restrict__ cs_slab,
                                                      int nchunks, int M1, int M2, int M1p, int M2p,
                                                      float *__restrict__ Cout, int c_layout, float *__restrict__ colsum) {
    const int e = blockIdx.x * 256 + threadIdx.x;                // element of the padded [M1p][M2p] block
    const int per = (nchunks + GT_SPLIT - 1) / GT_SPLIT;
    const int k0 = blockIdx.y * per, k1 = k0 + per < nchunks ? k0 + per : nchunks;
    if (e < M1p * M2p) {
        const int o = e / M2p, c = e % M2p;
        if (o < M1 && c < M2) {
            float s = 0.0f;
            for (int k = k0; k < k1; k++) s += slab[(int64_t)k * M1p * M2p + e];
            float *dst = c_layout == 0 ? Cout + (int64_t)o * M2 + c : Cout + (int64_t)c * M1 + o;
            atomicAdd(dst, s);
        }
    }
    if (colsum && cs_slab && e < M1) {
        float s = 0.0f;
        for (int k = k0; k < k1; k++) s += cs_slab[(int64_t)k * M1p + e];
        atomicAdd(colsum + e, s);
    }
}

int launch_linear_fwd(const float *x, int64_t N, int d, const float *W, const float *b, int out, int w_layout, int act,
                      float *y, hipStream_t st) {
    const unsigned gx = (unsigned)((N + BM - 1) / BM);
    if (out <= 32)
        hipLaunchKernelGGL(linear_fwd_mfma<1>, dim3(gx, (unsigned)((out + 31) / 32)), dim3(256), 0, st, x, N, d, W, b, out, w_layout, act, y);
    else if (out <= 64)
        hipLaunchKernelGGL(linear_fwd_mfma<2>, dim3(gx, (unsigned)((out + 63) / 64)), dim3(256), 0, st, x, N, d, W, b, out, w_layout, act, y);
    else
        hipLaunchKernelGGL(linear_fwd_mfma<4>, dim3(gx, (unsigned)((out + 127) / 128)), dim3(256), 0, st, x, N, d, W, b, out, w_layout, act, y);
    return dgg_check_launch("linear_fwd");
}

size_t gemm_tn_ws_floats(int64_t N, int M1, int M2) {
    const size_t nch = (size_t)((N + GT_ROWS - 1) / GT_ROWS), M1p = (size_t)(M1 + 31) / 32 * 32, M2p = (size_t)(M2 + 31) / 32 * 32;
    return nch * M1p * M2p + nch * M1p;
}

int launch_gemm_tn(const float *A, const float *B, int64_t N, int M1, int M2, float *C, int c_layout, float *colsum,
                   float *ws, hipStream_t st) {
    if (!ws) return dgg_set_error(DGG_ERR_ARG, "gemm_tn: workspace is NULL (dgg_gemm_tn_ws_floats)");
    const int nch = (int)((N + GT_ROWS - 1) / GT_ROWS), M1p = (M1 + 31) / 32 * 32, M2p = (M2 + 31) / 32 * 32;
    float *slab = ws, *cs_slab = ws + (size_t)nch * M1p * M2p;
    const unsigned gy = (unsigned)(M1p / 32);
    if (M2p % 64 == 0)
        hipLaunchKernelGGL(gemm_tn_partial<2>, dim3(nch, gy, (unsigned)(M2p / 64)), dim3(64), 0, st, A, B, N, M1, M2, M1p, M2p, slab, colsum ? cs_slab : nullptr);
    else
        hipLaunchKernelGGL(gemm_tn_partial<1>, dim3(nch, gy, (unsigned)(M2p / 32)), dim3(64), 0, st, A, B, N, M1, M2, M1p, M2p, slab, colsum ? cs_slab : nullptr);
    hipLaunchKernelGGL(gemm_tn_reduce, dim3((unsigned)((M1p * M2p + 255) / 256), GT_SPLIT), dim3(256), 0, st, slab,
                       colsum ? cs_slab : nullptr, nch, M1, M2, M1p, M2p, C, c_layout, colsum);
    return dgg_check_launch("gemm_tn");
}

}  // namespace

extern "C" {

int dgg_linear_fwd(const float *x, int64_t N, int d, const float *W, const float *b, int out, int w_layout, int act,
                   float *y, void *stream) {
    if (N < 0 || d < 1 || out < 1) return dgg_set_error(DGG_ERR_ARG, "linear_fwd: bad shape");
    if (N == 0) return 0;
    return launch_linear_fwd(x, N, d, W, b, out, w_layout, act, y, (hipStream_t)stream);
}

// floats of workspace dgg_linear_bwd needs: N*out for the activation backward + the weight-gradient slab
size_t dgg_linear_bwd_ws_floats(int64_t N, int d, int out) { return (size_t)N * out + gemm_tn_ws_floats(N, out, d); }
size_t dgg_gemm_tn_ws_floats(int64_t N, int M1, int M2) { return gemm_tn_ws_floats(N, M1, M2); }

// backward of y = act(x W^T + b).  ws: dgg_linear_bwd_ws_floats(N, d, out) floats.  dx (nullable) is OVERWRITTEN;
// dW (layout of W) and db (nullable) are ACCUMULATED into (caller zeroes them), so that several uses of one
// weight add up.
int dgg_linear_bwd(const float *x, int64_t N, int d, const float *W, int out, int w_layout, int act, const float *y,
                   const float *dy, float *dx, float *dW, float *db, float *ws, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    if (N == 0) return 0;
    if (!ws) return dgg_set_error(DGG_ERR_ARG, "linear_bwd: workspace is NULL (dgg_linear_bwd_ws_floats)");
    const float *dp = dy;
    if (act != 0) {
        int64_t n = N * out;
        unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
        hipLaunchKernelGGL(act_bwd_kernel, dim3(blocks), dim3(256), 0, st, y, dy, n, act, ws);
        dp = ws;
    }
    if (dx) {
        // dx[N,d] = dp[N,out] * W  ==  linear_fwd with the weight read in the transposed layout
        int rc = launch_linear_fwd(dp, N, out, W, nullptr, d, w_layout == 0 ? 1 : 0, 0, dx, st);
        if (rc != 0) return rc;
    }
    if (dW) return launch_gemm_tn(dp, x, N, out, d, dW, w_layout == 0 ? 0 : 1, db, ws + (size_t)N * out, st);
    return dgg_check_launch("linear_bwd");
}

// C[M1,M2] += A[N,M1]^T B[N,M2]  (c_layout 1: C stored [M2][M1]); colsum (nullable, [M1]) += column sums of A;
// ws: dgg_gemm_tn_ws_floats(N, M1, M2) floats
int dgg_gemm_tn_acc(const float *A, const float *B, int64_t N, int M1, int M2, float *C, int c_layout, float *colsum,
                    float *ws, void *stream) {
    if (N == 0) return 0;
    return launch_gemm_tn(A, B, N, M1, M2, C, c_layout, colsum, ws, (hipStream_t)stream);
}

}  // extern "C"
