// pure MFMA throughput probe: fp32 32x32x2, NACC independent accumulators per wavefront
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float *o, int iters) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; a++) for (int r = 0; r < 16; r++) acc[a][r] = 0.f;
    float av = threadIdx.x * 1e-3f, bv = threadIdx.x * 2e-3f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int a = 0; a < NACC; a++) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[a], 0, 0, 0);
    }
    float s = 0;
    for (int a = 0; a < NACC; a++) for (int r = 0; r < 16; r++) s += acc[a][r];
    o[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int wgs, int threads, int iters) {
    float *o; hipMalloc(&o, 4 * 1024 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(wgs), dim3(threads), 0, 0, o, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(wgs), dim3(threads), 0, 0, o, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = (double)wgs * (threads / 64) * iters * 8.0 * NACC * 4096.0;
    double clk_per_mfma = ms * 1e-3 * 2.4e9 / ((double)iters * 8 * NACC * ((double)wgs * (threads / 64) / 1024.0));
    printf("NACC=%d wgs=%d thr=%d: %.3f ms  %.1f TF/s  (%.1f clk@2.4GHz per MFMA per SIMD)\n", NACC, wgs, threads, ms, fl / ms / 1e9, clk_per_mfma);
    hipFree(o);
}
int main() {
    run<6>(256, 256, 20000);
    run<6>(256, 256, 2000);
    run<6>(256, 256, 200);
    run<2>(256, 256, 20000);
    run<1>(256, 256, 20000);
    run<6>(512, 256, 20000);
    run<4>(256, 512, 20000);
    return 0;
}
