// Divergent 32-byte cell gathers: (0) every lane loads both 16-byte halves of its own cell (2 instructions, 64 distinct lines each)
// vs (1) adjacent lanes cooperate: lanes 2i,2i+1 load the two halves of cell(2i), then of cell(2i+1) (2 instructions, 32 distinct
// 32-byte chunks each) and swap by DPP.  Same data ends up in the same lanes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
template <int MODE>
__global__ void __launch_bounds__(256) k(const float *cells, uint32_t ncells, int iters, float *out) {
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *) cells, 0, (int) (ncells * 32u), 0x00020000);
    float acc = 0; uint32_t c = hash(tid) % ncells;
    const bool odd = threadIdx.x & 1;
    for (int it = 0; it < iters; it++) {
        u32x4 a, b;
        if (MODE == 0) {
            a = __builtin_amdgcn_raw_buffer_load_b128(rsrc, c * 32, 0, 0);
            b = __builtin_amdgcn_raw_buffer_load_b128(rsrc, c * 32 + 16, 0, 0);
        } else {
            const uint32_t cn = (uint32_t) __builtin_amdgcn_mov_dpp((int) c, 0xB1, 0xF, 0xF, true);     // quad_perm [1,0,3,2]: neighbour's cell
            const uint32_t c0 = odd ? cn : c, c1 = odd ? c : cn;                                        // cell of the even / odd lane of the pair
            const u32x4 r1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, c0 * 32 + (odd ? 16 : 0), 0, 0);
            const u32x4 r2 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, c1 * 32 + (odd ? 16 : 0), 0, 0);
            u32x4 t1, t2;
            for (int q = 0; q < 4; q++) { t1[q] = (uint32_t) __builtin_amdgcn_mov_dpp((int) r1[q], 0xB1, 0xF, 0xF, true); t2[q] = (uint32_t) __builtin_amdgcn_mov_dpp((int) r2[q], 0xB1, 0xF, 0xF, true); }
            for (int q = 0; q < 4; q++) { a[q] = odd ? t2[q] : r1[q]; b[q] = odd ? r2[q] : t1[q]; }
        }
        float s = 0;
        for (int q = 0; q < 4; q++) s += __uint_as_float(a[q]) * (q + 1) + __uint_as_float(b[q]) * (q + 5);
        acc += s;
        c = hash(c + tid + (uint32_t) (s * 0.0f)) % ncells;           // dependent on the data, like a marching ray
    }
    out[tid] = acc;
}
int main() {
    for (uint32_t ncells : {1u << 16, 1u << 24}) {              // 2 MiB (L2) and 512 MiB
        float *cells, *out; (void) hipMalloc(&cells, (size_t) ncells * 32); (void) hipMalloc(&out, 1 << 26);
        (void) hipMemset(cells, 0, (size_t) ncells * 32);
        hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
        const int blocks = 256 * 8 * 2, iters = 2000;
        float res[2][1];
        for (int mode = 0; mode < 2; mode++) {
            if (mode == 0) k<0><<<blocks, 256>>>(cells, ncells, 10, out); else k<1><<<blocks, 256>>>(cells, ncells, 10, out);
            (void) hipEventRecord(e0);
            if (mode == 0) k<0><<<blocks, 256>>>(cells, ncells, iters, out); else k<1><<<blocks, 256>>>(cells, ncells, iters, out);
            (void) hipEventRecord(e1); (void) hipEventSynchronize(e1);
            float ms; (void) hipEventElapsedTime(&ms, e0, e1); res[mode][0] = ms;
            printf("cells=%u (%u MiB) mode=%d: %.3f ms, %.2f G cell-fetches/s, %.2f cycles/lane-fetch/CU\n", ncells, ncells / 32768, mode, ms,
                   (double) blocks * 256 * iters / ms * 1e-6, ms * 1e-3 * 2.4e9 * 256 / ((double) blocks * 256 * iters));
        }
        // check both modes produce identical sums
        (void) hipFree(cells); (void) hipFree(out);
    }
    return 0;
}
