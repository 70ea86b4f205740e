// issue cost of integer VALU ops (cycles per wave64 instruction per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(64) void k(unsigned *o, int iters, unsigned c1, unsigned c2) {
    unsigned x[16];
    for (int u = 0; u < 16; u++) x[u] = threadIdx.x * 2654435761u + u;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (MODE == 0) x[u] = x[u] * c1;                                   // v_mul_lo_u32
            else if (MODE == 1) x[u] = __umul24(x[u], c1);     // v_mul_u32_u24
            else if (MODE == 2) x[u] = x[u] ^ (x[u] >> 15);                    // shift + xor
            else if (MODE == 3) x[u] = x[u] + c1;                              // add
            else if (MODE == 4) { unsigned v = x[u] ^ c2; v *= 0x7feb352dU; v ^= v >> 15; v += c1; v *= 0x846ca68bU; x[u] = v; }
            else if (MODE == 5) { unsigned v = x[u] ^ c2; v = __umul24(v, 0xeb352dU); v ^= v >> 15; v += c1; v = __umul24(v, 0x6ca68bU); x[u] = v; }
        }
#pragma unroll
        for (int u = 0; u < 16; u++) asm volatile("" : "+v"(x[u]));
    }
    unsigned s = 0;
    for (int u = 0; u < 16; u++) s ^= x[u];
    o[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int MODE>
void run(const char *name, int nops) {
    unsigned *o; hipMalloc(&o, 1 << 22);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, wgs = 1024 * 4;    // 4 waves per SIMD
    hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(64), 0, 0, o, 10, 0x7feb352dU, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(64), 0, 0, o, iters, 0x7feb352dU, 12345u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * 16 * nops * 4);   // 4 waves share a SIMD
    printf("%-28s %.3f ms  %.2f cycles@2.4GHz per wave-instruction\n", name, ms, cyc);
    hipFree(o);
}
int main() {
    run<0>("v_mul_lo_u32", 1); run<1>("v_mul_u32_u24", 1); run<2>("shift+xor (2 ops)", 2); run<3>("v_add_u32", 1);
    run<4>("hash (2 mul_lo + 4)", 6); run<5>("hash24 (2 mul_u24 + 4)", 6);
    return 0;
}
