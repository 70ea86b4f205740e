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

constexpr int BM = 128, BN = 128, BK = 32, LDP = BK + 1;

__device__ __forceinline__ float act_apply(float v, int act) {
    if (act == 1) return v > 0.0f ? v : __fmul_rn(0.01f, v);
    if (act == 2) return v > 0.0f ? v : 0.0f;
    return v;
}

// y[N,out] = act(x[N,d] * W^T + b);  w_layout 0: W[out][d], 1: W[d][out]
__global__ __launch_bounds__(256) void linear_fwd_mfma(const float *__restrict__ x, int64_t N, int d,
                                                       const float *__restrict__ W, const float *__restrict__ b,
                                                       int out, int w_layout, int act, float *__restrict__ y) {
    __shared__ float xs[BM * LDP];
    __shared__ float ws[BN * LDP];   // [j][k] (+pad)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    f32x16 acc[4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[a][r] = 0.0f;
    const int li = lane & 31, hh = lane >> 5;
    for (int k0 = 0; k0 < d; k0 += BK) {
        __syncthreads();
        for (int e = tid; e < BM * BK; e += 256) {
            int r = e / BK, c = e % BK;
            int64_t gi = m0 + r;
            int gk = k0 + c;
            xs[r * LDP + c] = (gi < N && gk < d) ? x[gi * d + gk] : 0.0f;
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
            for (int a = 0; a < 4; a++) {
                float bv = ws[(a * 32 + li) * LDP + 2 * s + hh];
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[a], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 4; a++) {
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

// C[M1,M2] += A[N,M1]^T * B[N,M2]   (weight gradient: reduction over the node dimension; fp32 atomics)
// grid: x = row chunks of RCH rows, y = M1 blocks of 128, z = M2 blocks of 128.
// c_layout 0: C[M1][M2] row-major; 1: C stored transposed C[M2][M1].  colsum (optional, M1 floats) += column sums of A.
constexpr int RCH = 256;   // rows per workgroup: N/256 workgroups keep all 256 CUs busy at N = 100k
__global__ __launch_bounds__(256) void gemm_tn_reduce_mfma(const float *__restrict__ A, const float *__restrict__ B,
                                                           int64_t N, int M1, int M2, float *__restrict__ Cout,
                                                           int c_layout, float *__restrict__ colsum) {
    __shared__ float as[BK * (BM + 1)];   // [n][o]
    __shared__ float bs[BK * (BN + 1)];   // [n][c]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * RCH;
    const int64_t r1 = r0 + RCH < N ? r0 + RCH : N;
    const int o0 = blockIdx.y * BM, c0 = blockIdx.z * BN;
    const int li = lane & 31, hh = lane >> 5;
    f32x16 acc[4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[a][r] = 0.0f;
    float csum = 0.0f;   // thread tid < 128 accumulates column o0+tid of A
    for (int64_t nb = r0; nb < r1; nb += BK) {
        __syncthreads();
        for (int e = tid; e < BK * BM; e += 256) {
            int n = e / BM, o = e % BM;
            int64_t gn = nb + n;
            int go = o0 + o;
            as[n * (BM + 1) + o] = (gn < r1 && go < M1) ? A[gn * M1 + go] : 0.0f;
        }
        for (int e = tid; e < BK * BN; e += 256) {
            int n = e / BN, c = e % BN;
            int64_t gn = nb + n;
            int gc = c0 + c;
            bs[n * (BN + 1) + c] = (gn < r1 && gc < M2) ? B[gn * M2 + gc] : 0.0f;
        }
        __syncthreads();
        if (colsum && blockIdx.z == 0 && tid < BM) {
#pragma unroll 8
            for (int n = 0; n < BK; n++) csum += as[n * (BM + 1) + tid];
        }
#pragma unroll
        for (int s = 0; s < BK / 2; s++) {
            float av = as[(2 * s + hh) * (BM + 1) + wave * 32 + li];
#pragma unroll
            for (int a = 0; a < 4; a++) {
                float bv = bs[(2 * s + hh) * (BN + 1) + a * 32 + li];
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[a], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 4; a++) {
        int gc = c0 + a * 32 + li;
        if (gc >= M2) continue;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            int go = o0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (go < M1) {
                float *dst = c_layout == 0 ? Cout + (int64_t)go * M2 + gc : Cout + (int64_t)gc * M1 + go;
                atomicAdd(dst, acc[a][r]);
            }
        }
    }
    if (colsum && blockIdx.z == 0 && tid < BM && o0 + tid < M1) atomicAdd(colsum + o0 + tid, csum);
}

}  // namespace

extern "C" {

int dgg_linear_fwd(const float *x, int64_t N, int d, const float *W, const float *b, int out, int w_layout, int act,
                   float *y, void *stream) {
    if (N < 0 || d < 1 || out < 1) return dgg_set_error(DGG_ERR_ARG, "linear_fwd: bad shape");
    if (N == 0) return 0;
    dim3 grid((unsigned)((N + BM - 1) / BM), (unsigned)((out + BN - 1) / BN));
    hipLaunchKernelGGL(linear_fwd_mfma, grid, dim3(256), 0, (hipStream_t)stream, x, N, d, W, b, out, w_layout, act, y);
    return dgg_check_launch("linear_fwd");
}

// backward of y = act(x W^T + b).  dp_ws: workspace of N*out floats.  dx (nullable) is OVERWRITTEN;
// dW (layout of W) and db (nullable) are ACCUMULATED into (caller zeroes them), so that several uses of one
// weight add up.
int dgg_linear_bwd(const float *x, int64_t N, int d, const float *W, int out, int w_layout, int act, const float *y,
                   const float *dy, float *dx, float *dW, float *db, float *dp_ws, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    if (N == 0) return 0;
    const float *dp = dy;
    if (act != 0) {
        int64_t n = N * out;
        unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
        hipLaunchKernelGGL(act_bwd_kernel, dim3(blocks), dim3(256), 0, st, y, dy, n, act, dp_ws);
        dp = dp_ws;
    }
    if (dx) {
        // dx[N,d] = dp[N,out] * W  ==  linear_fwd with the weight read in the transposed layout
        dim3 grid((unsigned)((N + BM - 1) / BM), (unsigned)((d + BN - 1) / BN));
        hipLaunchKernelGGL(linear_fwd_mfma, grid, dim3(256), 0, st, dp, N, out, W, (const float *)nullptr, d,
                           w_layout == 0 ? 1 : 0, 0, dx);
    }
    if (dW) {
        dim3 grid((unsigned)((N + RCH - 1) / RCH), (unsigned)((out + BM - 1) / BM), (unsigned)((d + BN - 1) / BN));
        hipLaunchKernelGGL(gemm_tn_reduce_mfma, grid, dim3(256), 0, st, dp, x, N, out, d, dW, w_layout == 0 ? 0 : 1, db);
    }
    return dgg_check_launch("linear_bwd");
}

// C[M1,M2] += A[N,M1]^T B[N,M2]  (c_layout 1: C stored [M2][M1]); colsum (nullable, [M1]) += column sums of A
int dgg_gemm_tn_acc(const float *A, const float *B, int64_t N, int M1, int M2, float *C, int c_layout, float *colsum,
                    void *stream) {
    if (N == 0) return 0;
    dim3 grid((unsigned)((N + RCH - 1) / RCH), (unsigned)((M1 + BM - 1) / BM), (unsigned)((M2 + BN - 1) / BN));
    hipLaunchKernelGGL(gemm_tn_reduce_mfma, grid, dim3(256), 0, (hipStream_t)stream, A, B, N, M1, M2, C, c_layout, colsum);
    return dgg_check_launch("gemm_tn_acc");
}

}  // extern "C"
