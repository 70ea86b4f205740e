// MFMA + LDS operand stream probe (pattern of linear_fwd_reg's inner loop)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC, int PD, int MODE>
__global__ __launch_bounds__(256, 1) void k(float *o, int blocks) {
    constexpr int S = 64;
    __shared__ float wsf[S * NACC * 64];
    for (int e = threadIdx.x; e < S * NACC * 64; e += 256) wsf[e] = e * 1e-6f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    float xc[64];
    for (int q = 0; q < 64; q++) xc[q] = lane * 1e-3f + q;
    float tot = 0.f;
    for (int blk = 0; blk < blocks; blk++) {
        asm volatile("" ::: "memory");
        f32x16 acc[NACC];
#pragma unroll
        for (int a = 0; a < NACC; a++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][r] = 0.0f;
        constexpr int RING = PD + 1;
        float bw[RING][NACC];
        if (MODE == 0) {
#pragma unroll
            for (int s = 0; s < PD; s++)
#pragma unroll
                for (int a = 0; a < NACC; a++) bw[s][a] = wsf[(s * NACC + a) * 64 + lane];
#pragma unroll
            for (int s = 0; s < S; s++) {
                if (s + PD < S) {
#pragma unroll
                    for (int a = 0; a < NACC; a++) bw[(s + PD) % RING][a] = wsf[((s + PD) * NACC + a) * 64 + lane];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int a = 0; a < NACC; a++) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(xc[s], bw[s % RING][a], acc[a], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (MODE == 1) {   // no LDS: B from registers
#pragma unroll
            for (int s = 0; s < S; s++)
#pragma unroll
                for (int a = 0; a < NACC; a++) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(xc[s], xc[(s + a) & 63], acc[a], 0, 0, 0);
        } else {                  // LDS, compiler-scheduled
#pragma unroll
            for (int s = 0; s < S; s++)
#pragma unroll
                for (int a = 0; a < NACC; a++) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(xc[s], wsf[(s * NACC + a) * 64 + lane], acc[a], 0, 0, 0);
        }
#pragma unroll
        for (int a = 0; a < NACC; a++) tot += acc[a][0] + acc[a][15];
    }
    o[blockIdx.x * 256 + threadIdx.x] = tot;
}
template <int NACC, int PD, int MODE>
void run(int blocks) {
    float *o; hipMalloc(&o, 4 * 1024 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, PD, MODE>), dim3(256), dim3(256), 0, 0, o, 2);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, PD, MODE>), dim3(256), dim3(256), 0, 0, o, blocks);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double clk = ms * 1e-3 * 2.4e9 / ((double)blocks * 64 * NACC);
    printf("NACC=%d PD=%d MODE=%d: %.3f ms  %.1f clk@2.4GHz per MFMA\n", NACC, PD, MODE, ms, clk);
    hipFree(o);
}
int main() {
    run<6, 1, 0>(200); run<6, 2, 0>(200); run<6, 1, 1>(200); run<6, 1, 2>(200);
    run<4, 1, 0>(200); run<4, 2, 0>(200); run<4, 1, 1>(200); run<4, 1, 2>(200);
    run<2, 2, 0>(200); run<2, 4, 0>(200); run<2, 1, 1>(200);
    return 0;
}
