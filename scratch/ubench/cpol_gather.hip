// Which cache-policy bits make a random 32-byte gather cost less than a 128-byte line beyond L2?  Random 32-byte records of a 4 GiB
// buffer, two 16-byte buffer loads per lane, with the aux (cache policy) operand of raw_buffer_load = 0, sc0 (1), nt (2), sc0|nt (3),
// sc1 (16), sc1|sc0 (17), sc1|nt (18), all (19).  Kernel names carry the policy: run under rocprofv3 --pmc TCC_EA0_RDREQ_{32B,64B,128B}_sum.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
template <int AUX>
__global__ void __launch_bounds__(256) cpol_gather32(const float *base, uint32_t nrec, uint32_t bytes_lo, int iters, float *out) {
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x; float acc = 0; uint32_t c = hash(tid) % nrec;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *) base, 0, (int) bytes_lo, 0x00020000);
    for (int it = 0; it < iters; it++) {
        const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rsrc, c * 32, 0, AUX);
        const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rsrc, c * 32 + 16, 0, AUX);
        const float s = __uint_as_float(a.x ^ b.w); acc += s;
        c = hash(c + tid + (uint32_t) (s * 0.0f)) % nrec;
    }
    out[tid] = acc;
}
template <int AUX> static void run(const float *buf, float *out, const char *name) {
    const uint32_t bytes = 0xFFFFFF00u; const uint32_t nrec = bytes / 32; const int blocks = 256 * 8, iters = 256;
    hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1); float ms;
    cpol_gather32<AUX><<<blocks, 256>>>(buf, nrec, bytes, 8, out);
    (void) hipEventRecord(e0); cpol_gather32<AUX><<<blocks, 256>>>(buf, nrec, bytes, iters, out); (void) hipEventRecord(e1); (void) hipEventSynchronize(e1); (void) hipEventElapsedTime(&ms, e0, e1);
    printf("aux %2d (%s): %.3f ms, %.2f G gathers/s\n", AUX, name, ms, (double) blocks * 256 * iters / ms * 1e-6);
}
int main() {
    float *buf, *out; (void) hipMalloc(&buf, (size_t) 4 << 30); (void) hipMalloc(&out, 1 << 26); (void) hipMemset(buf, 0, (size_t) 4 << 30);
    run<0>(buf, out, "default"); run<1>(buf, out, "sc0"); run<2>(buf, out, "nt"); run<3>(buf, out, "sc0 nt");
    run<16>(buf, out, "sc1"); run<17>(buf, out, "sc1 sc0"); run<18>(buf, out, "sc1 nt"); run<19>(buf, out, "sc1 sc0 nt");
    return 0;
}
