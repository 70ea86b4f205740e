// cost of "4 LDS writes + __syncthreads" per iteration, 782 WGs x 256 threads, 37 KB LDS per WG
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256, 4) void k(float *o, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char colA[2][128 * 144];
    __shared__ float nbt[2][128];
    const int tid = threadIdx.x;
    uint4 v[4];
    for (int q = 0; q < 4; q++) v[q] = make_uint4(tid, q, 1, 2);
    float acc = 0.f;
    for (int tl = 0; tl < iters; tl++) {
        const int buf = tl & 1;
        if (MODE >= 1) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int ch = q * 256 + tid;
                *reinterpret_cast<uint4 *>(&colA[buf ^ 1][(ch / 8) * 144 + (ch % 8) * 16]) = v[q];
            }
            if (tid < 128) nbt[buf ^ 1][tid] = (float)tl;
        }
        if (MODE >= 2) acc += *reinterpret_cast<const float *>(&colA[buf][(tid & 127) * 144]);
        __syncthreads();
    }
    o[blockIdx.x * 256 + tid] = acc + nbt[0][tid & 127];
}
template <int MODE>
void run(const char *name) {
    float *o; (void)hipMalloc(&o, 782 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(782), dim3(256), 0, 0, o, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(782), dim3(256), 0, 0, o, 782);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-32s %.3f ms\n", name, ms);
    (void)hipFree(o);
}
int main() { run<0>("barrier only"); run<1>("4 ds_write_b128 + barrier"); run<2>("+ 1 ds_read"); return 0; }
