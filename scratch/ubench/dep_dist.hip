// How far apart must dependent VALU instructions be?  C interleaved dependent v_fma_f32 chains per lane (inline asm keeps the order).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int C>
__global__ void __launch_bounds__(64) k(float *out, int iters, float a, float b) {
    float x[8];
    for (int c = 0; c < 8; c++) x[c] = threadIdx.x + c;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 64 / C; u++) {
#pragma unroll
            for (int c = 0; c < C; c++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
        }
    }
    float s = 0; for (int c = 0; c < 8; c++) s += x[c];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int C> void run() {
    float *out; (void) hipMalloc(&out, 1 << 24);
    hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
    const int iters = 20000;
    for (int w : {1, 2, 5}) {
        const int blocks = 256 * 4 * w;
        k<C><<<blocks, 64>>>(out, 10, 1.0001f, 0.5f);
        (void) hipEventRecord(e0); k<C><<<blocks, 64>>>(out, iters, 1.0001f, 0.5f); (void) hipEventRecord(e1); (void) hipEventSynchronize(e1);
        float ms; (void) hipEventElapsedTime(&ms, e0, e1);
        printf("chains=%d waves/SIMD=%d  %.3f ms -> %.2f cycles per wave-instr per SIMD @2.4GHz\n", C, w, ms, ms * 1e-3 * 2.4e9 / ((double) iters * (64 / C) * C * w));
    }
}
int main() { run<1>(); run<2>(); run<3>(); run<4>(); run<8>(); return 0; }
