// Does the random-line rate fall with the footprint (address translation reach)?  Random 32-byte records, one 128-byte line each beyond L2,
// over the first F bytes of one allocation, F = 256 MiB ... 128 GiB; global loads (64-bit addresses), 8 waves per SIMD, and the same at
// 2 waves per SIMD (K_march's occupancy).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint64_t hash64(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
__global__ void __launch_bounds__(256) gather(const u32x4 *base, uint64_t nrec, int iters, float *out) {
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x; float acc = 0; uint64_t c = hash64(tid) % nrec;
    for (int it = 0; it < iters; it++) {
        const u32x4 a = __builtin_nontemporal_load(base + c * 2);
        const float s = __uint_as_float(a.x); acc += s;
        c = hash64(c + tid + (uint64_t) (s * 0.0f)) % nrec;
    }
    out[tid] = acc;
}
static void run(const u32x4 *buf, float *out, size_t bytes, int blocks) {
    const uint64_t nrec = bytes / 32; const int iters = 256;
    hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1); float ms;
    gather<<<blocks, 256>>>(buf, nrec, 8, out);
    (void) hipEventRecord(e0); gather<<<blocks, 256>>>(buf, nrec, iters, out); (void) hipEventRecord(e1); (void) hipEventSynchronize(e1); (void) hipEventElapsedTime(&ms, e0, e1);
    printf("footprint %7.2f GiB, %4d blocks: %8.3f ms, %6.2f G lines/s\n", (double) bytes / (1 << 30), blocks, ms, (double) blocks * 256 * iters / ms * 1e-6);
    fflush(stdout);
}
int main() {
    const size_t total = (size_t) 128 << 30;
    u32x4 *buf; float *out;
    if (hipMalloc(&buf, total) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    (void) hipMalloc(&out, 1 << 26); (void) hipMemset(buf, 0, total); (void) hipDeviceSynchronize();
    const size_t sizes[] = {(size_t) 256 << 20, (size_t) 1 << 30, (size_t) 4 << 30, (size_t) 16 << 30, (size_t) 32 << 30, (size_t) 64 << 30, (size_t) 128 << 30};
    for (size_t s : sizes) run(buf, out, s, 256 * 8);
    for (size_t s : sizes) run(buf, out, s, 256 * 2);
    return 0;
}
