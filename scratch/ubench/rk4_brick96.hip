// Exploration for the next round: the K_march core (RK4 eikonal step on a CELL8 trilinear field, register cell cache) with ONE ray per
// lane (the shipped form: er_step from mer_device.hpp) against TWO rays per lane whose evaluations share one control flow (the slow
// path runs when either ray leaves its cell; loads predicated per ray), so that the scheduler has two independent dependency chains.
// No queues, no events: rays that leave the unit cube are reflected back.   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "mer_kernels.hpp"
using namespace mer;

__device__ __forceinline__ void load_cell(const DGrid &g, int cell, CellCache &cc) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *) g.cell8, 0, (int) g.buf_bytes, 0x00020000);
    const u32x4 ua = __builtin_amdgcn_raw_buffer_load_b128(rsrc, cell * 32, 0, 0);
    const u32x4 ub = __builtin_amdgcn_raw_buffer_load_b128(rsrc, cell * 32 + 16, 0, 0);
    cc.d000 = __uint_as_float(ua.x); cc.d001 = __uint_as_float(ua.y); cc.d010 = __uint_as_float(ua.z); cc.d011 = __uint_as_float(ua.w);
    cc.d100 = __uint_as_float(ub.x); cc.d101 = __uint_as_float(ub.y); cc.d110 = __uint_as_float(ub.z); cc.d111 = __uint_as_float(ub.w);
}
struct MCache { int cell; float cx, cy, cz, a0, a1, a2, a3, a4, a5, a6, a7; __device__ void reset() { cell = -1; cx = cy = cz = -1e30f; a0 = a1 = a2 = a3 = a4 = a5 = a6 = a7 = 0; } };
__device__ __forceinline__ void mono_eval(const DGrid &g, MCache &cc, f3 p, float &val, f3 &grad) {
    const float px = __builtin_fmaf(g.s[0], p.x, g.t[0]), py = __builtin_fmaf(g.s[1], p.y, g.t[1]), pz = __builtin_fmaf(g.s[2], p.z, g.t[2]);
    float fx = px - cc.cx, fy = py - cc.cy, fz = pz - cc.cz;
    if (max(__float_as_uint(fx), max(__float_as_uint(fy), __float_as_uint(fz))) >= 0x3F800000u) {
        cc.cx = __builtin_amdgcn_fmed3f(floorf(px), 0.0f, (float) (g.res[0] - 2));
        cc.cy = __builtin_amdgcn_fmed3f(floorf(py), 0.0f, (float) (g.res[1] - 2));
        cc.cz = __builtin_amdgcn_fmed3f(floorf(pz), 0.0f, (float) (g.res[2] - 2));
        const int x1 = (int) cc.cx, y1 = (int) cc.cy, z1 = (int) cc.cz;
        fx = px - cc.cx; fy = py - cc.cy; fz = pz - cc.cz;
        const int base = (int) (__umul24(__umul24(z1, g.res[1]) + y1, g.res[0]) + x1);
        if (base != cc.cell) {
            cc.cell = base;
            const int cell = (int) (__umul24(__umul24(z1, g.res[1] - 1) + y1, g.res[0] - 1) + x1);
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *) g.cell8, 0, (int) g.buf_bytes, 0x00020000);
            const u32x4 ua = __builtin_amdgcn_raw_buffer_load_b128(rsrc, cell * 32, 0, 0);
            const u32x4 ub = __builtin_amdgcn_raw_buffer_load_b128(rsrc, cell * 32 + 16, 0, 0);
            cc.a0 = __uint_as_float(ua.x); cc.a1 = __uint_as_float(ua.y); cc.a2 = __uint_as_float(ua.z); cc.a3 = __uint_as_float(ua.w);
            cc.a4 = __uint_as_float(ub.x); cc.a5 = __uint_as_float(ub.y); cc.a6 = __uint_as_float(ub.z); cc.a7 = __uint_as_float(ub.w);
        }
    }
    const float X0 = __builtin_fmaf(fx, cc.a1, cc.a0), X2 = __builtin_fmaf(fx, cc.a4, cc.a2), X3 = __builtin_fmaf(fx, cc.a6, cc.a3), X5 = __builtin_fmaf(fx, cc.a7, cc.a5);
    const float gz = __builtin_fmaf(fy, X5, X3);
    val = __builtin_fmaf(fz, gz, __builtin_fmaf(fy, X2, X0));
    const float gy = __builtin_fmaf(fz, X5, X2);
    const float gx = __builtin_fmaf(fz, __builtin_fmaf(fy, cc.a7, cc.a6), __builtin_fmaf(fy, cc.a4, cc.a1));
    grad = f3(gx * g.s[0], gy * g.s[1], gz * g.s[2]);
}
__device__ __forceinline__ void er_step_mono(const DGrid &g, MCache &cc, f3 &p, f3 &v, float h) {
    float n; f3 gr; const float hh = 0.5f * h;
    mono_eval(g, cc, p, n, gr);
    f3 kp = v * MER_RCP(n); f3 ps = kp, vs = gr; f3 vv = fma3(hh, gr, v);
    mono_eval(g, cc, fma3(hh, kp, p), n, gr);
    kp = vv * MER_RCP(n); ps = fma3(2.0f, kp, ps); vs = fma3(2.0f, gr, vs); vv = fma3(hh, gr, v);
    mono_eval(g, cc, fma3(hh, kp, p), n, gr);
    kp = vv * MER_RCP(n); ps = fma3(2.0f, kp, ps); vs = fma3(2.0f, gr, vs); vv = fma3(h, gr, v);
    mono_eval(g, cc, fma3(h, kp, p), n, gr);
    kp = vv * MER_RCP(n); ps = ps + kp; vs = vs + gr;
    const float h6 = h * (1.0f / 6.0f);
    p = fma3(h6, ps, p); v = fma3(h6, vs, v);
}
// BRICK27: 2x2x2 cells = 3x3x3 corners in one 128-byte record; two-level register cache (brick: 27 words, current cell: 8 words)
struct BCache { int brick, cell; float cx, cy, cz; float b[27]; float d000, d001, d010, d011, d100, d101, d110, d111;
    __device__ void reset() { brick = -1; cell = -1; cx = cy = cz = -1e30f; for (int i = 0; i < 27; i++) b[i] = 0; d000 = d001 = d010 = d011 = d100 = d101 = d110 = d111 = 0; } };
__device__ __forceinline__ void brick_eval(const DGrid &g, BCache &cc, f3 p, float &val, f3 &grad) {
    const float px = __builtin_fmaf(g.s[0], p.x, g.t[0]), py = __builtin_fmaf(g.s[1], p.y, g.t[1]), pz = __builtin_fmaf(g.s[2], p.z, g.t[2]);
    float fx = px - cc.cx, fy = py - cc.cy, fz = pz - cc.cz;
    if (max(__float_as_uint(fx), max(__float_as_uint(fy), __float_as_uint(fz))) >= 0x3F800000u) {
        cc.cx = __builtin_amdgcn_fmed3f(floorf(px), 0.0f, (float) (g.res[0] - 2));
        cc.cy = __builtin_amdgcn_fmed3f(floorf(py), 0.0f, (float) (g.res[1] - 2));
        cc.cz = __builtin_amdgcn_fmed3f(floorf(pz), 0.0f, (float) (g.res[2] - 2));
        const int x1 = (int) cc.cx, y1 = (int) cc.cy, z1 = (int) cc.cz;
        fx = px - cc.cx; fy = py - cc.cy; fz = pz - cc.cz;
        const int nb = (g.res[0] - 1) >> 1;
        const int bx = x1 >> 1, by = y1 >> 1, bz = z1 >> 1;
        const int brick = (bz * nb + by) * nb + bx;
        if (brick != cc.brick) {
            cc.brick = brick;
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *) g.cell8, 0, (int) g.buf_bytes, 0x00020000);
#pragma unroll
            for (int q = 0; q < 7; q++) {
                const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(rsrc, brick * 128 + q * 16, 0, 0);
                cc.b[4 * q] = __uint_as_float(u.x); if (4 * q + 1 < 27) cc.b[4 * q + 1] = __uint_as_float(u.y);
                if (4 * q + 2 < 27) cc.b[4 * q + 2] = __uint_as_float(u.z); if (4 * q + 3 < 27) cc.b[4 * q + 3] = __uint_as_float(u.w);
            }
        }
        // select the cell's 8 corners out of the 27 (sx, sy, sz in {0,1}): rows along x, then y, then z
        const bool sx = x1 & 1, sy = y1 & 1, sz = z1 & 1;
        float lo[9], hi[9];
#pragma unroll
        for (int r = 0; r < 9; r++) { lo[r] = sx ? cc.b[3 * r + 1] : cc.b[3 * r]; hi[r] = sx ? cc.b[3 * r + 2] : cc.b[3 * r + 1]; }
        float l2[6], h2[6];           // [z][ylo/yhi]
#pragma unroll
        for (int z = 0; z < 3; z++) { l2[2 * z] = sy ? lo[3 * z + 1] : lo[3 * z]; l2[2 * z + 1] = sy ? lo[3 * z + 2] : lo[3 * z + 1];
                                      h2[2 * z] = sy ? hi[3 * z + 1] : hi[3 * z]; h2[2 * z + 1] = sy ? hi[3 * z + 2] : hi[3 * z + 1]; }
        cc.d000 = sz ? l2[2] : l2[0]; cc.d010 = sz ? l2[3] : l2[1]; cc.d100 = sz ? l2[4] : l2[2]; cc.d110 = sz ? l2[5] : l2[3];
        cc.d001 = sz ? h2[2] : h2[0]; cc.d011 = sz ? h2[3] : h2[1]; cc.d101 = sz ? h2[4] : h2[2]; cc.d111 = sz ? h2[5] : h2[3];
    }
    const float dx00 = cc.d001 - cc.d000, dx01 = cc.d011 - cc.d010, dx10 = cc.d101 - cc.d100, dx11 = cc.d111 - cc.d110;
    const float c00 = __builtin_fmaf(fx, dx00, cc.d000), c01 = __builtin_fmaf(fx, dx01, cc.d010), c10 = __builtin_fmaf(fx, dx10, cc.d100), c11 = __builtin_fmaf(fx, dx11, cc.d110);
    const float dy0 = c01 - c00, dy1 = c11 - c10;
    const float c0 = __builtin_fmaf(fy, dy0, c00), c1 = __builtin_fmaf(fy, dy1, c10);
    const float gz = c1 - c0;
    val = __builtin_fmaf(fz, gz, c0);
    const float gy = __builtin_fmaf(fz, dy1 - dy0, dy0);
    const float gxa = __builtin_fmaf(fy, dx01 - dx00, dx00), gxb = __builtin_fmaf(fy, dx11 - dx10, dx10);
    const float gx = __builtin_fmaf(fz, gxb - gxa, gxa);
    grad = f3(gx * g.s[0], gy * g.s[1], gz * g.s[2]);
}
__device__ __forceinline__ void er_step_brick(const DGrid &g, BCache &cc, f3 &p, f3 &v, float h) {
    float n; f3 gr; const float hh = 0.5f * h;
    brick_eval(g, cc, p, n, gr);
    f3 kp = v * MER_RCP(n); f3 ps = kp, vs = gr; f3 vv = fma3(hh, gr, v);
    brick_eval(g, cc, fma3(hh, kp, p), n, gr);
    kp = vv * MER_RCP(n); ps = fma3(2.0f, kp, ps); vs = fma3(2.0f, gr, vs); vv = fma3(hh, gr, v);
    brick_eval(g, cc, fma3(hh, kp, p), n, gr);
    kp = vv * MER_RCP(n); ps = fma3(2.0f, kp, ps); vs = fma3(2.0f, gr, vs); vv = fma3(h, gr, v);
    brick_eval(g, cc, fma3(h, kp, p), n, gr);
    kp = vv * MER_RCP(n); ps = ps + kp; vs = vs + gr;
    const float h6 = h * (1.0f / 6.0f);
    p = fma3(h6, ps, p); v = fma3(h6, vs, v);
}
struct Ev { float fx, fy, fz, px, py, pz; bool miss; };
__device__ __forceinline__ void pre(const DGrid &g, const CellCache &cc, f3 p, Ev &e) {
    e.px = __builtin_fmaf(g.s[0], p.x, g.t[0]); e.py = __builtin_fmaf(g.s[1], p.y, g.t[1]); e.pz = __builtin_fmaf(g.s[2], p.z, g.t[2]);
    e.fx = e.px - cc.cx; e.fy = e.py - cc.cy; e.fz = e.pz - cc.cz;
    e.miss = max(__float_as_uint(e.fx), max(__float_as_uint(e.fy), __float_as_uint(e.fz))) >= 0x3F800000u;
}
__device__ __forceinline__ void slow(const DGrid &g, CellCache &cc, Ev &e) {
    cc.cx = __builtin_amdgcn_fmed3f(floorf(e.px), 0.0f, (float) (g.res[0] - 2));
    cc.cy = __builtin_amdgcn_fmed3f(floorf(e.py), 0.0f, (float) (g.res[1] - 2));
    cc.cz = __builtin_amdgcn_fmed3f(floorf(e.pz), 0.0f, (float) (g.res[2] - 2));
    const int x1 = (int) cc.cx, y1 = (int) cc.cy, z1 = (int) cc.cz;
    e.fx = e.px - cc.cx; e.fy = e.py - cc.cy; e.fz = e.pz - cc.cz;
    const int base = (int) (__umul24(__umul24(z1, g.res[1]) + y1, g.res[0]) + x1);
    if (base != cc.cell) { cc.cell = base; load_cell(g, (int) (__umul24(__umul24(z1, g.res[1] - 1) + y1, g.res[0] - 1) + x1), cc); }
}
__device__ __forceinline__ void post(const DGrid &g, const CellCache &cc, const Ev &e, float &val, f3 &grad) {
    const float fx = e.fx, fy = e.fy, fz = e.fz;
    const float dx00 = cc.d001 - cc.d000, dx01 = cc.d011 - cc.d010, dx10 = cc.d101 - cc.d100, dx11 = cc.d111 - cc.d110;
    const float c00 = __builtin_fmaf(fx, dx00, cc.d000), c01 = __builtin_fmaf(fx, dx01, cc.d010), c10 = __builtin_fmaf(fx, dx10, cc.d100), c11 = __builtin_fmaf(fx, dx11, cc.d110);
    const float dy0 = c01 - c00, dy1 = c11 - c10;
    const float c0 = __builtin_fmaf(fy, dy0, c00), c1 = __builtin_fmaf(fy, dy1, c10);
    const float gz = c1 - c0;
    val = __builtin_fmaf(fz, gz, c0);
    const float gy = __builtin_fmaf(fz, dy1 - dy0, dy0);
    const float gxa = __builtin_fmaf(fy, dx01 - dx00, dx00), gxb = __builtin_fmaf(fy, dx11 - dx10, dx10);
    const float gx = __builtin_fmaf(fz, gxb - gxa, gxa);
    grad = f3(gx * g.s[0], gy * g.s[1], gz * g.s[2]);
}
// two evaluations with one control flow
__device__ __forceinline__ void eval2(const DGrid &g, CellCache &ca, CellCache &cb, f3 pa, f3 pb, float &na, f3 &ga, float &nb, f3 &gb) {
    Ev ea, eb; pre(g, ca, pa, ea); pre(g, cb, pb, eb);
    if (ea.miss || eb.miss) { if (ea.miss) slow(g, ca, ea); if (eb.miss) slow(g, cb, eb); }
    post(g, ca, ea, na, ga); post(g, cb, eb, nb, gb);
}
__device__ __forceinline__ void er_step2(const DGrid &g, CellCache &ca, CellCache &cb, f3 &pa, f3 &va, f3 &pb, f3 &vb, float h) {
    float na, nb; f3 ga, gb; const float hh = 0.5f * h;
    eval2(g, ca, cb, pa, pb, na, ga, nb, gb);
    f3 ka = va * MER_RCP(na), kb = vb * MER_RCP(nb);
    f3 psa = ka, vsa = ga, psb = kb, vsb = gb;
    f3 wa = fma3(hh, ga, va), wb = fma3(hh, gb, vb);
    eval2(g, ca, cb, fma3(hh, ka, pa), fma3(hh, kb, pb), na, ga, nb, gb);
    ka = wa * MER_RCP(na); kb = wb * MER_RCP(nb);
    psa = fma3(2.0f, ka, psa); vsa = fma3(2.0f, ga, vsa); psb = fma3(2.0f, kb, psb); vsb = fma3(2.0f, gb, vsb);
    wa = fma3(hh, ga, va); wb = fma3(hh, gb, vb);
    eval2(g, ca, cb, fma3(hh, ka, pa), fma3(hh, kb, pb), na, ga, nb, gb);
    ka = wa * MER_RCP(na); kb = wb * MER_RCP(nb);
    psa = fma3(2.0f, ka, psa); vsa = fma3(2.0f, ga, vsa); psb = fma3(2.0f, kb, psb); vsb = fma3(2.0f, gb, vsb);
    wa = fma3(h, ga, va); wb = fma3(h, gb, vb);
    eval2(g, ca, cb, fma3(h, ka, pa), fma3(h, kb, pb), na, ga, nb, gb);
    ka = wa * MER_RCP(na); kb = wb * MER_RCP(nb);
    psa = psa + ka; vsa = vsa + ga; psb = psb + kb; vsb = vsb + gb;
    const float h6 = h * (1.0f / 6.0f);
    pa = fma3(h6, psa, pa); va = fma3(h6, vsa, va); pb = fma3(h6, psb, pb); vb = fma3(h6, vsb, vb);
}
__device__ __forceinline__ void bounce(f3 &p, f3 &v) {      // keep the ray inside (-1,1)^3
    if (fabsf(p.x) > 0.999f) { v.x = -v.x; p.x = copysignf(0.999f, p.x); }
    if (fabsf(p.y) > 0.999f) { v.y = -v.y; p.y = copysignf(0.999f, p.y); }
    if (fabsf(p.z) > 0.999f) { v.z = -v.z; p.z = copysignf(0.999f, p.z); }
}
__device__ __forceinline__ void init_ray(uint32_t id, f3 &p, f3 &v) {
    Rng r; r.seed(7, id, 0);
    p = f3(1.9f * r.next1D() - 0.95f, 1.9f * r.next1D() - 0.95f, 1.9f * r.next1D() - 0.95f);
    v = normalize(f3(r.next1D() - 0.5f, r.next1D() - 0.5f, r.next1D() - 0.5f)) * 1.45f;
}
template <int R>
__global__ void __launch_bounds__(256) march(const DGrid g, float h, int steps, float *out) {
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    if (R == 1) {
        f3 p, v; init_ray(tid, p, v); CellCache cc; cc.reset(); float opt = 0;
        for (int k = 0; k < steps; k++) { er_step<RIFK_CELL8_BUF, MER_STEP_RK4>(g, cc, p, v, h, opt); bounce(p, v); }
        out[tid] = p.x + p.y + p.z + v.x;
    } else if (R == 4) {
        f3 p, v; init_ray(tid, p, v); BCache cc; cc.reset();
        for (int k = 0; k < steps; k++) { er_step_brick(g, cc, p, v, h); bounce(p, v); }
        out[tid] = p.x + p.y + p.z + v.x;
    } else if (R == 3) {
        f3 p, v; init_ray(tid, p, v); MCache cc; cc.reset();
        for (int k = 0; k < steps; k++) { er_step_mono(g, cc, p, v, h); bounce(p, v); }
        out[tid] = p.x + p.y + p.z + v.x;
    } else {
        f3 pa, va, pb, vb; init_ray(2 * tid, pa, va); init_ray(2 * tid + 1, pb, vb); CellCache ca, cb; ca.reset(); cb.reset();
        for (int k = 0; k < steps; k++) { er_step2(g, ca, cb, pa, va, pb, vb, h); bounce(pa, va); bounce(pb, vb); }
        out[2 * tid] = pa.x + pa.y + pa.z + va.x; out[2 * tid + 1] = pb.x + pb.y + pb.z + vb.x;
    }
}
__global__ void fill_brick(float *rec, int N) {            // 3x3x3 corners of every 2x2x2-cell brick, 32 words per record; b[(z*3+y)*3+x]
    const int nb = (N - 1) / 2; const int64_t c = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (c >= (int64_t) nb * nb * nb) return;
    const int by = (int) ((c / nb) % nb);
    float *q = rec + c * 32;
    for (int z = 0; z < 3; z++) for (int y = 0; y < 3; y++) for (int x = 0; x < 3; x++) q[(z * 3 + y) * 3 + x] = 1.3f + 0.3f * (2 * by + y) / (N - 1);
}
__global__ void fill_mono(float *cell8, int N) {           // the same field as monomial coefficients a0..a7 (a2 = coefficient of fy)
    const int64_t c = (int64_t) blockIdx.x * 256 + threadIdx.x, M = N - 1;
    if (c >= M * M * M) return;
    const int y = (int) ((c / M) % M);
    const float n0 = 1.3f + 0.3f * y / (N - 1), n1 = 1.3f + 0.3f * (y + 1) / (N - 1);
    float *q = cell8 + c * 8; q[0] = n0; q[1] = 0; q[2] = n1 - n0; q[3] = 0; q[4] = 0; q[5] = 0; q[6] = 0; q[7] = 0;
}
__global__ void fill(float *cell8, int N) {                  // linear RIF 1.3 -> 1.6 along y, CELL8 layout
    const int64_t c = (int64_t) blockIdx.x * 256 + threadIdx.x, M = N - 1;
    if (c >= M * M * M) return;
    const int y = (int) ((c / M) % M);
    const float n0 = 1.3f + 0.3f * y / (N - 1), n1 = 1.3f + 0.3f * (y + 1) / (N - 1);
    float *q = cell8 + c * 8; q[0] = n0; q[1] = n0; q[2] = n1; q[3] = n1; q[4] = n0; q[5] = n0; q[6] = n1; q[7] = n1;
}
int main() {
    const int N = 257; const int64_t M = N - 1, cells = M * M * M;
    float *cell8, *out; (void) hipMalloc(&cell8, cells * 32); (void) hipMalloc(&out, 64 << 20);
    fill<<<(unsigned) ((cells + 255) / 256), 256>>>(cell8, N);
    DGrid g{}; for (int i = 0; i < 3; i++) { g.res[i] = N; g.bmin[i] = -1; g.bmax[i] = 1; g.s[i] = (N - 1) / 2.0f; g.t[i] = (N - 1) / 2.0f; }
    g.cell8 = cell8; g.layout = MER_LAYOUT_CELL8; g.buf_bytes = (uint32_t) (cells * 32);
    const float h = 0.5f * 2.0f / (N - 1); const int steps = 96; const int64_t rays = 4 << 20;
    hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
    std::vector<float> a(rays), b(rays);
    for (int R = 1; R <= 4; R++) {
        if (R == 3) fill_mono<<<(unsigned) ((cells + 255) / 256), 256>>>(cell8, N);
        if (R == 4) { const int64_t nb = (N - 1) / 2; fill_brick<<<(unsigned) ((nb * nb * nb + 255) / 256), 256>>>(cell8, N); g.buf_bytes = (uint32_t) (nb * nb * nb * 128); }
        const unsigned blocks = (unsigned) (rays / (R == 2 ? 2 : 1) / 256);
        for (int rep = 0; rep < 2; rep++) {
            (void) hipEventRecord(e0);
            if (R == 1) march<1><<<blocks, 256>>>(g, h, steps, out); else if (R == 2) march<2><<<blocks, 256>>>(g, h, steps, out); else if (R == 3) march<3><<<blocks, 256>>>(g, h, steps, out); else march<4><<<blocks, 256>>>(g, h, steps, out);
            (void) hipEventRecord(e1); (void) hipEventSynchronize(e1);
        }
        float ms; (void) hipEventElapsedTime(&ms, e0, e1);
        (void) hipMemcpy(R == 1 ? a.data() : b.data(), out, rays * 4, hipMemcpyDeviceToHost);
        printf("rays/lane=%d: %.3f ms for %lld rays x %d steps -> %.1f Gsteps/s (err=%s)\n", R, ms, (long long) rays, steps, rays * (double) steps / ms * 1e-6, hipGetErrorString(hipGetLastError()));
    }
    int64_t same = 0; for (int64_t i = 0; i < rays; i++) same += a[i] == b[i];
    printf("identical results for %lld of %lld rays\n", (long long) same, (long long) rays);
    return 0;
}
