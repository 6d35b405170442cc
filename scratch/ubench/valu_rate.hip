// VALU issue-rate microbenchmark: N independent v_fma_f32 chains per lane, W waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ void __launch_bounds__(64) k(float *out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
                x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
            }
        } else if (MODE == 1) {   // dependent chain
#pragma unroll
            for (int u = 0; u < 64; u++) x0 = __builtin_fmaf(x0, a, b);
        } else if (MODE == 2) {   // packed
            typedef float v2f __attribute__((ext_vector_type(2)));
            v2f p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}; const v2f A = {a, a}, B = {b, b};
#pragma unroll
            for (int u = 0; u < 16; u++) {
                p0 = __builtin_elementwise_fma(p0, A, B); p1 = __builtin_elementwise_fma(p1, A, B);
                p2 = __builtin_elementwise_fma(p2, A, B); p3 = __builtin_elementwise_fma(p3, A, B);
            }
            x0 = p0.x; x1 = p0.y; x2 = p1.x; x3 = p1.y; x4 = p2.x; x5 = p2.y; x6 = p3.x; x7 = p3.y;
        } else {                  // integer / compare mix: v_max_i32, v_min_i32, v_cvt, v_floor
            int i0 = (int) x0, i1 = (int) x1, i2 = (int) x2, i3 = (int) x3;
#pragma unroll
            for (int u = 0; u < 16; u++) {
                i0 = min(max(i0 + 3, 1), 1 << 20) ^ u; i1 = min(max(i1 + 5, 1), 1 << 20) ^ u; i2 = min(max(i2 + 7, 1), 1 << 20) ^ u; i3 = min(max(i3 + 9, 1), 1 << 20) ^ u;
            }
            x0 = i0; x1 = i1; x2 = i2; x3 = i3;
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
template <int MODE> void run(const char *name, int per_iter) {
    float *out; hipMalloc(&out, 1 << 24);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int w : {1, 2, 3, 4, 5, 8}) {
        const int blocks = 256 * 4 * w;            // w waves per SIMD (1-wave blocks)
        k<MODE><<<blocks, 64>>>(out, 10, 1.0001f, 0.5f);
        hipEventRecord(e0); k<MODE><<<blocks, 64>>>(out, iters, 1.0001f, 0.5f); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double inst_per_simd = (double) iters * per_iter * w;
        printf("%s waves/SIMD=%d  %.3f ms  -> %.2f cycles per wave-instr per SIMD @2.4GHz\n", name, w, ms, ms * 1e-3 * 2.4e9 / inst_per_simd);
    }
}
int main() { run<0>("fma indep x8 ", 64); run<1>("fma dependent", 64); run<2>("pk_fma indep ", 64); run<3>("int min/max  ", 16 * 4 * 4); return 0; }
