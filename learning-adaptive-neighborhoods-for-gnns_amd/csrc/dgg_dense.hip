// dgg_dense.hip -- the dense all-pairs alternates of the reference: DGG_LearnableK_SDD (dgm.py:259-351, dist_fn="metric") and
// DGG_StraightThrough (dgm.py:140-182 + 63-100), noise off.
//
// Both return a DENSE [B,N,N] adjacency whose rows are a softmax over ALL N columns:
//     prob = exp(-t dist);  log_p = log(prob);  y = softmax(log_p / temp)
//     SDD: sort the row, f = sigmoid((hs_start - interval pos) + interval (k_i - 1)), out = y f   (hard: (f - y f) + y f)
//     ST : out = y  (hard: (1[pos < k] - y) + y)
// so the backward couples every pair of a graph (d loss / d logit_ij = y_ij (dy_ij - <y_i, dy_i>) / temp for ALL j): the
// path is O(N^2) by definition and is written for what the reference uses it on -- batches of small graphs (B x N x N
// outputs) -- not for the 100k-node regime of the sparse path.  One wavefront per row (b,i), the row of logits / weights in
// LDS, positions by counting under (weight desc, column asc) like dgg_csr.hip.  Forward arithmetic is the canonical one
// (dgg_common.h), so out / y / pos equal the oracle bit-for-bit.
//   dense_rows_fwd   rows of out, y, pos
//   dense_rows_bwd   g = d out -> C_ij = coefficient of (x_i - x_j) from row i, dk_i (SDD), per-row terms of dt
//   dense_pairs_dx   d xq_i = sum_j (C_ij + C_ji)(xq_i - xq_j)      (a row's own terms + its appearances as a column)
//   feat_softmax     nn.Softmax(dim=-1) of the SDD input projection (dgm.py:217-221), forward / backward
#include "dgg_common.h"
#include "dgg_api_internal.h"

using namespace dgg;

namespace {

constexpr int WPB = 4;

__device__ __forceinline__ float wave_max(float m) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    return m;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float c_sigmoid(float z) { return 1.0f / (1.0f + c_exp(-z)); }
__device__ __forceinline__ float sdd_ramp_z(float hs_start, float interval, int pos, float k) {
    const float xs = __fadd_rn(hs_start, -__fmul_rn(interval, (float)pos));
    const float sh = __fmul_rn(__fadd_rn(k, -1.0f), interval);
    return __fadd_rn(xs, sh);
}

__global__ __launch_bounds__(WPB * 64) void dense_rows_fwd_kernel(const float *__restrict__ xq, int64_t rows, int64_t N, int h,
                                                                 const float *__restrict__ tp, float temp, int ramp,
                                                                 const float *__restrict__ k, int kfix, float hs_start,
                                                                 float interval, int hard, float *__restrict__ out,
                                                                 float *__restrict__ y, int32_t *__restrict__ pos) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wv = dgg::wave_id();
    float *ys = lds + (size_t)wv * N;
    int64_t bi = (int64_t)blockIdx.x * WPB + wv;
    const bool live = bi < rows;
    if (!live) bi = rows - 1;                                     // keep every wave in the block-wide barriers
    const float *X = xq + (bi / N) * N * h, *xi = xq + bi * h;
    const float nt = -tp[0];
    float m = -INFINITY;
    for (int64_t j = lane; j < N; j += 64) {
        const float d = c_sqrt(pair_d2_thread(xi, X + j * h, h));
        const float p = c_exp(__fmul_rn(nt, d));
        const float lp = c_log(p) / temp;
        ys[j] = lp;
        m = fmaxf(m, lp);
    }
    m = wave_max(m);
    float s = 0.0f;
    for (int64_t j = lane; j < N; j += 64) {
        const float e = c_exp(__fadd_rn(ys[j], -m));
        ys[j] = e;
        s = __fadd_rn(s, e);
    }
    const float Z = wave_sum_butterfly(s);
    for (int64_t j = lane; j < N; j += 64) ys[j] = ys[j] / Z;
    __syncthreads();
    const float ki = (ramp == 0) ? k[bi] : 0.0f;
    for (int64_t j = lane; j < N; j += 64) {
        const float yj = ys[j];
        int cnt = 0;
        for (int64_t q = 0; q < N; q++) {
            const float yq = ys[q];
            cnt += (yq > yj || (yq == yj && q < j)) ? 1 : 0;
        }
        float o;
        if (ramp == 0) {
            const float f = c_sigmoid(sdd_ramp_z(hs_start, interval, cnt, ki));
            const float a = __fmul_rn(yj, f);
            o = hard ? __fadd_rn(__fadd_rn(f, -a), a) : a;
        } else {
            const float ind = cnt < kfix ? 1.0f : 0.0f;
            o = hard ? __fadd_rn(__fadd_rn(ind, -yj), yj) : yj;
        }
        if (live) {
            out[bi * N + j] = o;
            y[bi * N + j] = yj;
            pos[bi * N + j] = cnt;
        }
    }
}

__global__ __launch_bounds__(WPB * 64) void dense_rows_bwd_kernel(const float *__restrict__ xq, int64_t rows, int64_t N, int h,
                                                                 const float *__restrict__ tp, float temp, int ramp,
                                                                 const float *__restrict__ k, float hs_start, float interval,
                                                                 const float *__restrict__ y, const int32_t *__restrict__ pos,
                                                                 const float *__restrict__ g, float *__restrict__ Cm,
                                                                 float *__restrict__ dk, float *__restrict__ dt_rows) {
    const int lane = threadIdx.x & 63;
    const int64_t bi = (int64_t)blockIdx.x * WPB + dgg::wave_id();
    if (bi >= rows) return;
    const float *X = xq + (bi / N) * N * h, *xi = xq + bi * h;
    const float *yr = y + bi * N, *gr = g + bi * N;
    const int32_t *pr = pos + bi * N;
    const float t = tp[0];
    const float ki = (ramp == 0) ? k[bi] : 0.0f;
    float S = 0.0f, dkk = 0.0f;
    for (int64_t j = lane; j < N; j += 64) {
        float dy = gr[j];
        if (ramp == 0) {
            const float f = c_sigmoid(sdd_ramp_z(hs_start, interval, pr[j], ki));
            dkk += gr[j] * yr[j] * f * (1.0f - f) * interval;
            dy *= f;
        }
        S += yr[j] * dy;
    }
    S = wave_sum(S);
    dkk = wave_sum(dkk);
    float dt = 0.0f;
    for (int64_t j = lane; j < N; j += 64) {
        float dy = gr[j];
        if (ramp == 0) dy *= c_sigmoid(sdd_ramp_z(hs_start, interval, pr[j], ki));
        const float dlp = yr[j] * (dy - S) / temp;                // d loss / d (-t d_ij): log(exp(.)) is the identity
        const float d = c_sqrt(pair_d2_thread(xi, X + j * h, h));
        dt -= dlp * d;
        Cm[bi * N + j] = d > 0.0f ? -t * dlp / d : 0.0f;
    }
    dt = wave_sum(dt);
    if (lane == 0) {
        dt_rows[bi] = dt;
        if (dk) dk[bi] = dkk;
    }
}

// lane = feature: d xq_i[c] = sum_j (C_ij + C_ji)(xq_i[c] - xq_j[c]); the N coefficients of the row staged in LDS
__global__ __launch_bounds__(WPB * 64) void dense_pairs_dx_kernel(const float *__restrict__ xq, int64_t rows, int64_t N, int h,
                                                                 const float *__restrict__ Cm, float *__restrict__ dxq) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wv = dgg::wave_id();
    float *cf = lds + (size_t)wv * N;
    int64_t bi = (int64_t)blockIdx.x * WPB + wv;
    const bool live = bi < rows;
    if (!live) bi = rows - 1;
    const int64_t b = bi / N, i = bi % N;
    const float *X = xq + b * N * h, *C0 = Cm + b * N * N;
    for (int64_t j = lane; j < N; j += 64) cf[j] = C0[i * N + j] + C0[j * N + i];
    __syncthreads();
    for (int c = lane; c < h; c += 64) {
        const float xic = X[i * h + c];
        float acc = 0.0f;
        for (int64_t j = 0; j < N; j++) acc += cf[j] * (xic - X[j * h + c]);
        if (live) dxq[bi * h + c] = acc;
    }
}

__global__ __launch_bounds__(WPB * 64) void feat_softmax_fwd_kernel(const float *__restrict__ z, int64_t rows, int h,
                                                                   float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * WPB + dgg::wave_id();
    if (r >= rows) return;
    const float *zr = z + r * h;
    float *o = out + r * h;
    float m = -INFINITY;
    for (int c = lane; c < h; c += 64) m = fmaxf(m, zr[c]);
    m = wave_max(m);
    float s = 0.0f;
    for (int c = lane; c < h; c += 64) {
        const float e = c_exp(__fadd_rn(zr[c], -m));
        o[c] = e;                                                 // each lane re-reads only its own entries
        s = __fadd_rn(s, e);
    }
    const float Z = wave_sum_butterfly(s);
    for (int c = lane; c < h; c += 64) o[c] = o[c] / Z;
}

__global__ __launch_bounds__(WPB * 64) void feat_softmax_bwd_kernel(const float *__restrict__ out, const float *__restrict__ g,
                                                                   int64_t rows, int h, float *__restrict__ dz) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * WPB + dgg::wave_id();
    if (r >= rows) return;
    float S = 0.0f;
    for (int c = lane; c < h; c += 64) S += out[r * h + c] * g[r * h + c];
    S = wave_sum(S);
    for (int c = lane; c < h; c += 64) dz[r * h + c] = out[r * h + c] * (g[r * h + c] - S);
}

constexpr int64_t DENSE_MAX_N = 8192;                             // WPB rows of N floats in LDS (128 KB of the CU's 160 KB)

}  // namespace

extern "C" {

int dgg_dense_rows_fwd(const float *xq, int B, int64_t N, int h, const float *t, float temp, int ramp, const float *k, int kfix,
                       float hs_start, float interval, int hard, float *out, float *y, int32_t *pos, void *stream) {
    if (B < 0 || N < 0 || h < 1) return dgg_set_error(DGG_ERR_ARG, "dense_rows_fwd: bad shape");
    if (ramp != 0 && ramp != 1) return dgg_set_error(DGG_ERR_ARG, "dense_rows_fwd: ramp must be 0 (SDD sigmoid) or 1 (fixed top-k)");
    if ((int64_t)B * N == 0) return 0;
    if (ramp == 0 && !k) return dgg_set_error(DGG_ERR_ARG, "dense_rows_fwd: the SDD ramp needs k");
    if (!(temp > 0.0f)) return dgg_set_error(DGG_ERR_ARG, "dense_rows_fwd: temp must be positive");
    if (N > DENSE_MAX_N) return dgg_set_error(DGG_ERR_UNSUPPORTED, "dense_rows: N > 8192 (dense [B,N,N] path; use the sparse all-pairs path)");
    const int64_t rows = (int64_t)B * N;
    if (rows == 0) return 0;
    const size_t shm = (size_t)WPB * N * sizeof(float);
    if (shm > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)dense_rows_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        if (e != hipSuccess) return dgg_check_hip(e, "dense_rows_fwd: LDS size");
    }
    hipLaunchKernelGGL(dense_rows_fwd_kernel, dim3((unsigned)((rows + WPB - 1) / WPB)), dim3(WPB * 64), shm, (hipStream_t)stream, xq, rows,
                       N, h, t, temp, ramp, k, kfix, hs_start, interval, hard, out, y, pos);
    return dgg_check_launch("dense_rows_fwd");
}

int dgg_dense_rows_bwd(const float *xq, int B, int64_t N, int h, const float *t, float temp, int ramp, const float *k, float hs_start,
                       float interval, const float *y, const int32_t *pos, const float *g, float *Cm, float *dk, float *dt_rows,
                       void *stream) {
    if (ramp != 0 && ramp != 1) return dgg_set_error(DGG_ERR_ARG, "dense_rows_bwd: ramp must be 0 or 1");
    const int64_t rows = (int64_t)B * N;
    if (rows == 0) return 0;
    if (ramp == 0 && (!k || !dk)) return dgg_set_error(DGG_ERR_ARG, "dense_rows_bwd: the SDD ramp needs k and dk");
    hipLaunchKernelGGL(dense_rows_bwd_kernel, dim3((unsigned)((rows + WPB - 1) / WPB)), dim3(WPB * 64), 0, (hipStream_t)stream, xq, rows, N,
                       h, t, temp, ramp, k, hs_start, interval, y, pos, g, Cm, ramp == 0 ? dk : nullptr, dt_rows);
    return dgg_check_launch("dense_rows_bwd");
}

int dgg_dense_pairs_dx(const float *xq, int B, int64_t N, int h, const float *Cm, float *dxq, void *stream) {
    if (N > DENSE_MAX_N) return dgg_set_error(DGG_ERR_UNSUPPORTED, "dense_pairs_dx: N > 8192");
    const int64_t rows = (int64_t)B * N;
    if (rows == 0) return 0;
    const size_t shm = (size_t)WPB * N * sizeof(float);
    if (shm > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)dense_pairs_dx_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        if (e != hipSuccess) return dgg_check_hip(e, "dense_pairs_dx: LDS size");
    }
    hipLaunchKernelGGL(dense_pairs_dx_kernel, dim3((unsigned)((rows + WPB - 1) / WPB)), dim3(WPB * 64), shm, (hipStream_t)stream, xq, rows,
                       N, h, Cm, dxq);
    return dgg_check_launch("dense_pairs_dx");
}

int dgg_feat_softmax_fwd(const float *z, int64_t rows, int h, float *out, void *stream) {
    if (rows == 0) return 0;
    if (h < 1) return dgg_set_error(DGG_ERR_ARG, "feat_softmax_fwd: h must be positive");
    hipLaunchKernelGGL(feat_softmax_fwd_kernel, dim3((unsigned)((rows + WPB - 1) / WPB)), dim3(WPB * 64), 0, (hipStream_t)stream, z, rows, h,
                       out);
    return dgg_check_launch("feat_softmax_fwd");
}

int dgg_feat_softmax_bwd(const float *out, const float *g, int64_t rows, int h, float *dz, void *stream) {
    if (rows == 0) return 0;
    hipLaunchKernelGGL(feat_softmax_bwd_kernel, dim3((unsigned)((rows + WPB - 1) / WPB)), dim3(WPB * 64), 0, (hipStream_t)stream, out, g,
                       rows, h, dz);
    return dgg_check_launch("feat_softmax_bwd");
}

}  // extern "C"
