// Calibration of the memory-side read counters (TCC_EA0_RDREQ by request size, FETCH_SIZE) on access patterns of KNOWN byte count,
// over a 4 GiB buffer (16 x the Infinity Cache):  (1) stream: every byte once, 16 B per lane, coalesced;  (2) gather32: random 32-byte
// records, two 16-byte loads per lane (the CELL8 fetch of K_march);  (3) gather8x4: four 8-byte loads inside one random 128-byte
// record (the BRICK27 fetch);  (4) gather4x8: eight 4-byte loads in four rows of two z-planes (the dense trilinear fetch).
// Run under `rocprofv3 --pmc ...` (scratch/pmc_calib.sh); the kernel names carry the pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

__global__ void __launch_bounds__(256) calib_stream(const uint4 *src, size_t n16, float *out) {
    float acc = 0;
    for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t) gridDim.x * 256) { const uint4 v = src[i]; acc += __uint_as_float(v.x ^ v.y ^ v.z ^ v.w); }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ void __launch_bounds__(256) calib_gather32(const float *base, uint32_t nrec, int iters, float *out) {
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x; float acc = 0; uint32_t c = hash(tid) % nrec;
    for (int it = 0; it < iters; it++) {
        const uint4 *q = (const uint4 *) (base + (size_t) c * 8);
        const uint4 a = q[0], b = q[1];
        const float s = __uint_as_float(a.x ^ b.w); acc += s;
        c = hash(c + tid + (uint32_t) (s * 0.0f)) % nrec;
    }
    out[tid] = acc;
}
__global__ void __launch_bounds__(256) calib_gather8x4(const float *base, uint32_t nrec128, int iters, float *out) {
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x; float acc = 0; uint32_t c = hash(tid) % nrec128;
    for (int it = 0; it < iters; it++) {
        const float *q = base + (size_t) c * 32 + (hash(c) & 3u);           // cell corner inside the 3x3x3 record
        const u32x2 r0 = *(const u32x2 *) (q), r1 = *(const u32x2 *) (q + 4), r2 = *(const u32x2 *) (q + 10), r3 = *(const u32x2 *) (q + 14);
        const float s = __uint_as_float(r0.x ^ r1.y ^ r2.x ^ r3.y); acc += s;
        c = hash(c + tid + (uint32_t) (s * 0.0f)) % nrec128;
    }
    out[tid] = acc;
}
__global__ void __launch_bounds__(256) calib_gather4x8(const float *base, uint32_t N, int iters, float *out) {
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x; float acc = 0; uint32_t c = hash(tid);
    for (int it = 0; it < iters; it++) {
        const uint32_t x = c % (N - 1), y = (c / N) % (N - 1), z = (c / (N * N)) % (N - 1);
        const float *q = base + ((size_t) z * N + y) * N + x; const size_t sy = N, sz = (size_t) N * N;
        const float s = q[0] + q[1] + q[sy] + q[sy + 1] + q[sz] + q[sz + 1] + q[sz + sy] + q[sz + sy + 1]; acc += s;
        c = hash(c + tid + (uint32_t) (s * 0.0f));
    }
    out[tid] = acc;
}
int main() {
    const size_t bytes = (size_t) 4 << 30;
    float *buf, *out; (void) hipMalloc(&buf, bytes); (void) hipMalloc(&out, 1 << 26); (void) hipMemset(buf, 0, bytes);
    const int blocks = 256 * 8, iters = 512;
    hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1); float ms;
    const double lanes = (double) blocks * 256;
    (void) hipEventRecord(e0); calib_stream<<<blocks, 256>>>((const uint4 *) buf, bytes / 16, out); (void) hipEventRecord(e1); (void) hipEventSynchronize(e1); (void) hipEventElapsedTime(&ms, e0, e1);
    printf("calib_stream    : %.0f bytes read once, %.3f ms, %.1f GB/s\n", (double) bytes, ms, bytes / ms * 1e-6);
    (void) hipEventRecord(e0); calib_gather32<<<blocks, 256>>>(buf, (uint32_t) (bytes / 32), iters, out); (void) hipEventRecord(e1); (void) hipEventSynchronize(e1); (void) hipEventElapsedTime(&ms, e0, e1);
    printf("calib_gather32  : %.0f gathers of one 32-byte record = %.0f useful bytes, %.3f ms, %.2f G gathers/s\n", lanes * iters, lanes * iters * 32, ms, lanes * iters / ms * 1e-6);
    (void) hipEventRecord(e0); calib_gather8x4<<<blocks, 256>>>(buf, (uint32_t) (bytes / 128), iters, out); (void) hipEventRecord(e1); (void) hipEventSynchronize(e1); (void) hipEventElapsedTime(&ms, e0, e1);
    printf("calib_gather8x4 : %.0f gathers of 4 x 8 bytes inside one 128-byte record = %.0f useful bytes, %.3f ms, %.2f G gathers/s\n", lanes * iters, lanes * iters * 32, ms, lanes * iters / ms * 1e-6);
    (void) hipEventRecord(e0); calib_gather4x8<<<blocks, 256>>>(buf, 1024u, iters, out); (void) hipEventRecord(e1); (void) hipEventSynchronize(e1); (void) hipEventElapsedTime(&ms, e0, e1);
    printf("calib_gather4x8 : %.0f gathers of 8 x 4 bytes in 4 rows of a 1024^3 grid = %.0f useful bytes, %.3f ms, %.2f G gathers/s\n", lanes * iters, lanes * iters * 32, ms, lanes * iters / ms * 1e-6);
    return 0;
}
