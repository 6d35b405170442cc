// mer_device.hpp -- gfx950 device functions of the refractive volumetric path-tracing hot path.
// Written for wave64 CDNA4; compiled with -ffp-contract=off so the arithmetic is the one written here.
// Reference citations (file:line) are into cmu-ci-lab/MitsubaER.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mer.h"

#define MER_EPSILON 1e-4f                 // include/mitsuba/core/constants.h:25-31 (single precision)
#define MER_PI 3.14159265358979323846f
#define MER_INV_FOURPI 0.07957747154594766788f
#define MER_INV_PI 0.31830988618379067154f
#define MER_INF __builtin_huge_valf()

namespace mer {

struct f3 {
    float x, y, z;
    __host__ __device__ __forceinline__ f3() {}
    __host__ __device__ __forceinline__ f3(float a, float b, float c) : x(a), y(b), z(c) {}
};
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator-(f3 a) { return f3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ f3 operator*(float s, f3 a) { return f3(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return f3(a.x * b.x, a.y * b.y, a.z * b.z); }
// TVector3::operator/(T): reciprocal then multiply (include/mitsuba/core/vector.h:548-557)
__device__ __forceinline__ f3 operator/(f3 a, float s) { float r = 1.0f / s; return f3(a.x * r, a.y * r, a.z * r); }
__device__ __forceinline__ float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ f3 cross(f3 a, f3 b) { return f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ f3 normalize(f3 a) { return a / sqrtf(dot(a, a)); }
__device__ __forceinline__ float max3(f3 a) { return fmaxf(a.x, fmaxf(a.y, a.z)); }
__device__ __forceinline__ bool is_zero(f3 a) { return a.x == 0.0f && a.y == 0.0f && a.z == 0.0f; }

// ------------------------------------------------------------------------------------------------
// -DMER_BOUNDS_CHECK (libmer_check.so): every index a kernel forms into a device buffer -- path-state slots, work-list segments and
// items, hit ring, film, per-path output, grid payloads and their re-laid-out records, spline coefficients -- is compared with the
// buffer's extent before the access.  The first violation is recorded (kind, index, limit) in a device word array and the access is
// redirected to element 0, so that an out-of-range index is REPORTED (mer_debug_bounds) instead of faulting or, worse, silently
// reading mapped memory.  The product build compiles the checks away.
enum { CHK_SLOT = 1, CHK_QUEUE_SEG, CHK_QUEUE_ITEM, CHK_HITQ, CHK_FILM, CHK_PATHOUT, CHK_GRID_DENSE, CHK_GRID_RECORD, CHK_GRID_COEFF,
       CHK_GRID_RGB, CHK_LIVE_ROW };
#ifdef MER_BOUNDS_CHECK
__device__ __forceinline__ uint64_t mer_chk(unsigned long long *chk, int kind, uint64_t idx, uint64_t limit) {
    if (idx < limit) return idx;
    if (chk && atomicAdd(chk, 1ULL) == 0ULL) { chk[1] = (unsigned long long) kind; chk[2] = idx; chk[3] = limit; }
    return 0;
}
#define MER_CHK(chkptr, kind, idx, limit) mer_chk((chkptr), (kind), (uint64_t) (idx), (uint64_t) (limit))
#else
#define MER_CHK(chkptr, kind, idx, limit) (idx)
#endif

// ------------------------------------------------------------------------------------------------
// Sampler: counter-based PCG32 stream per (pixel, sample); float conversion as Random::nextFloat
// (src/libcore/random.cpp:630-639: 23 mantissa bits in [1,2) minus 1).
struct Rng {
    uint64_t state, inc;
    __device__ __forceinline__ static uint64_t splitmix64(uint64_t x) {
        x += 0x9E3779B97F4A7C15ULL;
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
        x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
        return x ^ (x >> 31);
    }
    __device__ __forceinline__ void seed(uint64_t seedv, uint32_t pixel, uint32_t sample) {
        uint64_t initseq = ((uint64_t) sample << 32) | (uint64_t) pixel;
        state = 0; inc = (initseq << 1) | 1ULL;
        next();
        state += splitmix64(seedv);
        next();
    }
    __device__ __forceinline__ uint32_t next() {
        uint64_t old = state;
        state = old * 6364136223846793005ULL + inc;
        uint32_t xorshifted = (uint32_t) (((old >> 18u) ^ old) >> 27u);
        uint32_t rot = (uint32_t) (old >> 59u);
        return (xorshifted >> rot) | (xorshifted << ((32u - rot) & 31u));
    }
    __device__ __forceinline__ float next1D() { return __uint_as_float((next() >> 9) | 0x3f800000u) - 1.0f; }
    // The stream of a SIDE WALK (transmittance walk of a luminaire sample: kind 1; of an emitter look-up: kind 2): a child of the path's stream at
    // the point where the walk starts.  The path's own stream does not advance while the walk runs, which makes the walk an independent piece of work.
    __device__ __forceinline__ Rng fork(uint64_t kind) const { Rng c; c.state = splitmix64(state ^ (kind * 0xD1B54A32D192ED03ULL)); c.inc = inc; return c; }
};

// ------------------------------------------------------------------------------------------------
// Device view of an uploaded grid (GridDataSource / SplineDataSource).
struct DGrid {
    const void  *data;        // dense: x fastest [z][y][x][c]
    const float *cell8;       // MER_LAYOUT_CELL8: 8 corner values per cell, [z][y][x][8] over (res-1)^3 cells;
                              // MER_LAYOUT_BRICK27: 27 corners (+5 pad) per 2x2x2-cell brick, [bz][by][bx][32], b[(dz*3+dy)*3+dx]
    int32_t nbx, nby;         // BRICK layouts: bricks along x and y
    int32_t bshift, bw, recw; // BRICK layouts: log2(cells per brick axis), corners per axis (cells + 1), words per record
    const float *coeff;       // cubic B-spline coefficients (dense) or NULL
    int32_t res[3];
    int32_t channels, dtype, layout;
    float   s[3], t[3];       // volumeToGrid: diagonal + translation (gridvolume.cpp:188-195 without the toWorld factor)
    float   m[12];            // worldToGrid = scale((res-1)/extents) * translate(-min) * worldToVolume, row-major 3x4 (gridvolume.cpp:188-195)
    float   w2v[12];          // worldToVolume (inverse of the plugin's toWorld); `affine` != 0 when it is not the identity
    int32_t affine;
    float   wmin[3], wmax[3]; // m_aabb: the world-space bounding box of the transformed data box (gridvolume.cpp:199-203)
    float   bmin[3], bmax[3]; // the data box, in volume space
    float   lim_min[3], lim_max[3];   // spline interpolatable limits (splinevolume.cpp:280-281)
    uint32_t buf_bytes;               // byte size of data / cell8 when it fits a buffer descriptor (< 4 GiB), else 0
    float   ac_n_o, ac_n_max, ac_k_r; int32_t ac_mode;   // RIFK_ACOUSTIC: the analytic field of acousticrifvolume (no data)
    unsigned long long *chk;          // MER_BOUNDS_CHECK: violation record (NULL in the product build)
    uint64_t n_dense, n_record;       // element counts of data (all channels) and of cell8 (floats): the extents the checks use
};

// Record index of cell (x, y, z) in the CELL8 layout: cells in x-major order (four x-neighbours per 128-byte line).  Storing the eight cells of a
// 2x2x2 tile together instead measured +1 ... 4 % (profiles/round2/ab_cell8_tile_order.txt): not adopted.
__device__ __forceinline__ uint32_t cell8_record(const DGrid &g, int x, int y, int z) {
    return __umul24(__umul24(z, g.res[1] - 1) + y, g.res[0] - 1) + x;
}


// include/mitsuba/core/aabb.h:308-339 (dRcp = 1/d as Ray::setDirection)
__device__ __forceinline__ bool aabb_intersect(const float mn[3], const float mx[3], f3 o, f3 d, float &nearT, float &farT) {
    nearT = -MER_INF; farT = MER_INF;
    const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
#pragma unroll
    for (int i = 0; i < 3; i++) {
        if (dd[i] == 0.0f) {
            if (oo[i] < mn[i] || oo[i] > mx[i]) return false;
        } else {
            const float dRcp = 1.0f / dd[i];
            float t1 = (mn[i] - oo[i]) * dRcp, t2 = (mx[i] - oo[i]) * dRcp;
            if (t1 > t2) { float tmp = t1; t1 = t2; t2 = tmp; }
            nearT = fmaxf(t1, nearT);
            farT = fminf(t2, farT);
            if (!(nearT <= farT)) return false;
        }
    }
    return true;
}

__device__ __forceinline__ float grid_fetch(const DGrid &g, long long idx) {
    idx = (long long) MER_CHK(g.chk, CHK_GRID_DENSE, idx, g.n_dense);
    if (g.dtype == MER_VOL_F32) return ((const float *) g.data)[idx];
    return (float) ((const uint8_t *) g.data)[idx] / 255.0f;      // m_densityMap, gridvolume.cpp:204-214
}

// GridDataSource::lookupFloat (gridvolume.cpp:337-388).  The integer part (x1,y1,z1, bounds test, linear
// index) is the bit-exact contract; the blend keeps the reference's operation order.  Branch-free: the reference's early
// `return 0` for a point off the grid is a select at the end -- the cell index is clamped into the grid and the eight corners are
// always fetched (a point off the grid is rare on this path, and a divergent early return inside the marching loops is what this
// toolchain miscompiled in K_connect: see sdf_value).  The bounds test is written so that it cannot wrap: v_cvt_i32_f32 saturates,
// a coordinate of +inf (or >= 2^31) gives x1 = INT_MAX, and INT_MAX + 1 >= res would pass.
__device__ __forceinline__ float lookup_float(const DGrid &g, f3 p, int *idx4 = nullptr) {
    const float px = g.m[0] * p.x + g.m[1] * p.y + g.m[2] * p.z + g.m[3], py = g.m[4] * p.x + g.m[5] * p.y + g.m[6] * p.z + g.m[7],
                pz = g.m[8] * p.x + g.m[9] * p.y + g.m[10] * p.z + g.m[11];     // Transform::transformAffine (transform.h:147-155)
    const int x1 = (int) floorf(px), y1 = (int) floorf(py), z1 = (int) floorf(pz);
    const bool inside = !(x1 < 0 || y1 < 0 || z1 < 0 || x1 >= g.res[0] - 1 || y1 >= g.res[1] - 1 || z1 >= g.res[2] - 1) && px == px && py == py && pz == pz;   // NaN: (int) is INT_MIN on the reference's x86, 0 here
    const int xc = min(max(x1, 0), g.res[0] - 2), yc = min(max(y1, 0), g.res[1] - 2), zc = min(max(z1, 0), g.res[2] - 2);
    const float fx = px - (float) x1, fy = py - (float) y1, fz = pz - (float) z1,
                _fx = 1.0f - fx, _fy = 1.0f - fy, _fz = 1.0f - fz;
    const int base = (zc * g.res[1] + yc) * g.res[0] + xc;
    if (idx4) { idx4[0] = x1; idx4[1] = y1; idx4[2] = z1; idx4[3] = inside ? base : -1; }
    float d000, d001, d010, d011, d100, d101, d110, d111;
    if (g.layout == MER_LAYOUT_CELL8) {
        const int cell = (int) cell8_record(g, xc, yc, zc);
        const float4 *c = (const float4 *) (g.cell8 + (size_t) MER_CHK(g.chk, CHK_GRID_RECORD, (size_t) cell * 8, g.n_record - 7));
        const float4 a = c[0], b = c[1];
        d000 = a.x; d001 = a.y; d010 = a.z; d011 = a.w; d100 = b.x; d101 = b.y; d110 = b.z; d111 = b.w;
    } else {
        const int sy = g.res[0], sz = g.res[0] * g.res[1];
        d000 = grid_fetch(g, base);          d001 = grid_fetch(g, base + 1);
        d010 = grid_fetch(g, base + sy);     d011 = grid_fetch(g, base + sy + 1);
        d100 = grid_fetch(g, base + sz);     d101 = grid_fetch(g, base + sz + 1);
        d110 = grid_fetch(g, base + sz + sy); d111 = grid_fetch(g, base + sz + sy + 1);
    }
    const float v = ((d000 * _fx + d001 * fx) * _fy + (d010 * _fx + d011 * fx) * fy) * _fz +
                    ((d100 * _fx + d101 * fx) * _fy + (d110 * _fx + d111 * fx) * fy) * fz;
    return inside ? v : 0.0f;
}

// GridDataSource::lookupSpectrum (gridvolume.cpp:390-421), 3 channels; branch-free as lookup_float
__device__ __forceinline__ f3 lookup_spectrum(const DGrid &g, f3 p) {
    const float px = g.m[0] * p.x + g.m[1] * p.y + g.m[2] * p.z + g.m[3], py = g.m[4] * p.x + g.m[5] * p.y + g.m[6] * p.z + g.m[7],
                pz = g.m[8] * p.x + g.m[9] * p.y + g.m[10] * p.z + g.m[11];     // Transform::transformAffine (transform.h:147-155)
    const int x1 = (int) floorf(px), y1 = (int) floorf(py), z1 = (int) floorf(pz);
    const bool inside = !(x1 < 0 || y1 < 0 || z1 < 0 || x1 >= g.res[0] - 1 || y1 >= g.res[1] - 1 || z1 >= g.res[2] - 1) && px == px && py == py && pz == pz;   // NaN: (int) is INT_MIN on the reference's x86, 0 here
    const int xc = min(max(x1, 0), g.res[0] - 2), yc = min(max(y1, 0), g.res[1] - 2), zc = min(max(z1, 0), g.res[2] - 2);
    const float fx = px - (float) x1, fy = py - (float) y1, fz = pz - (float) z1,
                _fx = 1.0f - fx, _fy = 1.0f - fy, _fz = 1.0f - fz;
    const int base = (zc * g.res[1] + yc) * g.res[0] + xc, sy = g.res[0], sz = g.res[0] * g.res[1];
    float out[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const long long b3 = (long long) base * 3 + c, y3 = (long long) sy * 3, z3 = (long long) sz * 3;
        const float d000 = grid_fetch(g, b3), d001 = grid_fetch(g, b3 + 3),
                    d010 = grid_fetch(g, b3 + y3), d011 = grid_fetch(g, b3 + y3 + 3),
                    d100 = grid_fetch(g, b3 + z3), d101 = grid_fetch(g, b3 + z3 + 3),
                    d110 = grid_fetch(g, b3 + z3 + y3), d111 = grid_fetch(g, b3 + z3 + y3 + 3);
        const float v = ((d000 * _fx + d001 * fx) * _fy + (d010 * _fx + d011 * fx) * fy) * _fz +
                        ((d100 * _fx + d101 * fx) * _fy + (d110 * _fx + d111 * fx) * fy) * fz;
        out[c] = inside ? v : 0.0f;
    }
    return f3(out[0], out[1], out[2]);
}

// Trilinear RIF: value + analytic gradient of the interpolant; cell clamped to the grid (new --
// gridvolume has no value()/gradient(), src/librender/volume.cpp:57-80).  Being new functionality, its
// arithmetic is DEFINED here (and restated identically in the oracle): monomial coefficients per cell (CellCache::set), fused Horner evaluation.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// The 8 corner values of the last cell are kept in registers: the 4 RK4 stages of a half-voxel step land in
// the same cell most of the time, so the gather is re-issued only when the cell index changes.
struct CellCache {
    int cell;                 // linear index of the cached cell's base corner, -1 = empty
    float cx, cy, cz;         // the cached cell's base corner in grid coordinates (exact small integers)
    // the cell's trilinear interpolant in monomial form about its base corner, f = a0 + ax x + ay y + az z + axy xy + axz xz + ayz yz + axyz xyz
    // (x, y, z in [0,1)): computed ONCE per cell change from the 8 gathered corners (12 subtractions), so that each of the ~8 evaluations a
    // ray makes inside a cell (4 RK4 stages per half-voxel step) costs 11 fused multiply-adds for value + gradient instead of the 22
    // add / fma operations of nested lerps that re-derive the corner differences every time
    float a0, ax, ay, az, axy, axz, ayz, axyz;
    int brick;                // RIFK_BRICK27_LDS: the brick whose record this lane holds in LDS, -1 = none
    // predicted-cell fetch (K_march's RK4 step on BRICK27 buffer loads, trilinear_value_grad_pf): the corners of the cell the ray is expected to enter
    // next, requested half a step ahead and not waited for; ncell = its linear index, -1 = nothing pending
    int ncell; u32x2 n00, n01, n10, n11;
    __device__ __forceinline__ void reset() { cell = -1; brick = -1; ncell = -1; cx = cy = cz = -1.0e30f; a0 = ax = ay = az = axy = axz = ayz = axyz = 0.0f; }
    // corners d[z][y][x] -> coefficients; this operation order is part of the definition of the interpolant's arithmetic (oracle: TriCoeff)
    __device__ __forceinline__ void set(float d000, float d001, float d010, float d011, float d100, float d101, float d110, float d111) {
        a0 = d000; ax = d001 - d000; ay = d010 - d000; az = d100 - d000;
        const float x1 = d011 - d010, x2 = d101 - d100, x3 = d111 - d110;
        axy = x1 - ax; axz = x2 - ax; ayz = (d110 - d100) - ay;
        axyz = (x3 - x2) - axy;
    }
};

// Internal fetch kinds of the trilinear RIF (template parameter RIF of the kernels):
//   MER_RIF_TRILINEAR (1): dense grid, global loads (any size)      RIFK_DENSE_BUF (3): dense grid, buffer loads
//   RIFK_CELL8 (4): cell-major grid, global loads                   RIFK_CELL8_BUF (5): cell-major, buffer loads
// Buffer loads take a 32-bit byte offset against a wave-uniform descriptor: one VGPR of address arithmetic per
// fetch instead of eight 64-bit adds, and the +row / +slice strides ride in the scalar offset operand.
#define RIFK_ACOUSTIC 8          // == MER_RIF_ACOUSTIC: analytic Bessel-mode field, no fetch
#define MER_BLOCK 256            // threads per block of every wavefront kernel
#define RIFK_BRICK27_BUF 6
#define RIFK_BRICK27 7
#define RIFK_BRICK27_LDS 9       // K_march only (never a scene's fetch kind): BRICK27 records below 4 GiB, every lane keeps its current brick's record in LDS
#define RIFK_DENSE_BUF 3
#define RIFK_CELL8 4
#define RIFK_CELL8_BUF 5
// MER_ALWAYS_LOAD (experiment): no cell cache, every evaluation gathers its cell -- straight-line code, no exec-mask regions
#ifdef MER_ALWAYS_LOAD
#define MER_CELL_TEST(cond) true
#else
#define MER_CELL_TEST(cond) (cond)
#endif
// the slow path of the trilinear fetch: (px, py, pz) in grid coordinates has left the cached cell -- clamp, form the cell's index and gather its
// 8 corners into the register cache.  The loads are issued here and first used by the caller's evaluation.
template <int RIFK>
__device__ __forceinline__ void cell_fill(const DGrid &g, CellCache &cc, float px, float py, float pz) {
    // clamp in the float domain (exact: the operands are small integers), one v_med3_f32 per axis
    cc.cx = __builtin_amdgcn_fmed3f(floorf(px), 0.0f, (float) (g.res[0] - 2));
    cc.cy = __builtin_amdgcn_fmed3f(floorf(py), 0.0f, (float) (g.res[1] - 2));
    cc.cz = __builtin_amdgcn_fmed3f(floorf(pz), 0.0f, (float) (g.res[2] - 2));
    const int x1 = (int) cc.cx, y1 = (int) cc.cy, z1 = (int) cc.cz;
    const int base = (int) (__umul24(__umul24(z1, g.res[1]) + y1, g.res[0]) + x1);      // make_params: res[1] * res[2] <= 2^24
    if (RIFK == RIFK_BRICK27_LDS) {
        // Brick staging in LDS (option lds_bricks, off by default: measured 19 % slower than the register cell cache, profiles/round2/
        // ab_lds_brick_staging.txt).  The rays of a wave are incoherent (a few hundred thousand rays in flight over 2^21 ... 2^27 bricks: no
        // two share one), so what LDS can hold is each lane's OWN current brick: 27 corners = 7 x 16 bytes per lane, [chunk][lane] so that a
        // wave's writes are contiguous; 28 KiB per block of 256, five blocks per CU.  A ray crosses ~4 cells per brick: the first gathers
        // the record (7 loads of 16 bytes, one 128-byte line, DMA'd into LDS), the others read LDS instead of going back to L1 / L2, which
        // do not hold a line that long (L2 hit rate 0.3).  No barrier: a lane reads only what it wrote itself.  Why it loses: the line
        // leaves the fabric once either way, but 7 vector-memory instructions per brick change instead of 4 per cell change load the
        // texture-address / L1 pipeline (busy 50 % / 81 % of the time already) more than the saved L2 requests relieve it.
        __shared__ u32x4 brick_lds[MER_BLOCK / 64][7][64];
        if (base != cc.cell) {
            cc.cell = base;
            const int brick = (int) (__umul24(__umul24(z1 >> 1, g.nby) + (y1 >> 1), g.nbx) + (x1 >> 1));
            const int lane = threadIdx.x & 63; u32x4 (*rec)[64] = brick_lds[__builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6))];   // wave-uniform: the addresses stay in SGPRs
            if (brick != cc.brick) {
                cc.brick = brick;
                const int o = (int) MER_CHK(g.chk, CHK_GRID_RECORD, (uint32_t) brick * 32u, g.n_record - 31u);
                const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *) g.cell8, 0, (int) g.buf_bytes, 0x00020000);
                // straight into LDS (buffer_load_dwordx4 ... lds: lane L's 16 bytes land at M0 + 16 L, no registers in between; the chunk's
                // byte offset rides in the scalar offset, which moves the buffer address only -- the instruction's immediate offset would
                // move the LDS address too); the reads below are this lane's own, once the loads have landed (vmcnt)
#define MER_BRICK_DMA(k) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *) &rec[k][0], 16, o * 4, (k) * 16, 0, 0)
                MER_BRICK_DMA(0); MER_BRICK_DMA(1); MER_BRICK_DMA(2); MER_BRICK_DMA(3); MER_BRICK_DMA(4); MER_BRICK_DMA(5); MER_BRICK_DMA(6);
#undef MER_BRICK_DMA
                __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0)
            }
            const int w = ((z1 & 1) * 3 + (y1 & 1)) * 3 + (x1 & 1);            // word of the cell's corner (0,0,0) in the record
            const uint32_t *words = (const uint32_t *) rec;
#define MER_BRICK_WORD(i) __uint_as_float(words[(((w + (i)) >> 2) * 64 + lane) * 4 + ((w + (i)) & 3)])
            cc.set(MER_BRICK_WORD(0), MER_BRICK_WORD(1), MER_BRICK_WORD(3), MER_BRICK_WORD(4), MER_BRICK_WORD(9), MER_BRICK_WORD(10), MER_BRICK_WORD(12), MER_BRICK_WORD(13));
#undef MER_BRICK_WORD
        }
    } else
    if (RIFK == RIFK_BRICK27 || RIFK == RIFK_BRICK27_BUF) {
        if (base != cc.cell) {
            // the cell's 8 corners as four x-pairs out of its brick's record (2^3 cells: 27 corners, one 128-byte line; 4^3 cells: 125
            // corners, 512 bytes): a cell change inside the brick is an L1 hit, only a change of line goes to L2 / the fabric
            cc.cell = base;
            const int bs = g.bshift, bm = (1 << bs) - 1, bw = g.bw;
            const int brick = (int) (__umul24(__umul24(z1 >> bs, g.nby) + (y1 >> bs), g.nbx) + (x1 >> bs));
            // word index of the cell's corner (0,0,0): 64 bits, a 1024^3 field has 2^32 record words (below 4 GiB, the buffer form, 32 suffice)
            const size_t o64 = (size_t) MER_CHK(g.chk, CHK_GRID_RECORD, (size_t) (uint32_t) brick * (uint32_t) g.recw + (uint32_t) (((z1 & bm) * bw + (y1 & bm)) * bw + (x1 & bm)),
                                                g.n_record - (uint64_t) (bw * bw + bw) - 1u);
            const int o = (int) o64;
            u32x2 r00, r01, r10, r11;
            if (RIFK == RIFK_BRICK27_BUF) {
                const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *) g.cell8, 0, (int) g.buf_bytes, 0x00020000);
                r00 = __builtin_amdgcn_raw_buffer_load_b64(rsrc, o * 4, 0, 0);
                r01 = __builtin_amdgcn_raw_buffer_load_b64(rsrc, o * 4, bw * 4, 0);
                r10 = __builtin_amdgcn_raw_buffer_load_b64(rsrc, o * 4, bw * bw * 4, 0);
                r11 = __builtin_amdgcn_raw_buffer_load_b64(rsrc, o * 4, (bw * bw + bw) * 4, 0);
            } else {
                const float *q = g.cell8 + o64;
                r00 = u32x2{__float_as_uint(q[0]), __float_as_uint(q[1])}; r01 = u32x2{__float_as_uint(q[bw]), __float_as_uint(q[bw + 1])};
                r10 = u32x2{__float_as_uint(q[bw * bw]), __float_as_uint(q[bw * bw + 1])}; r11 = u32x2{__float_as_uint(q[bw * bw + bw]), __float_as_uint(q[bw * bw + bw + 1])};
            }
            cc.set(__uint_as_float(r00.x), __uint_as_float(r00.y), __uint_as_float(r01.x), __uint_as_float(r01.y),
                   __uint_as_float(r10.x), __uint_as_float(r10.y), __uint_as_float(r11.x), __uint_as_float(r11.y));
        }
    } else
    if (MER_CELL_TEST(base != cc.cell)) {
        cc.cell = base;
        const int dbase = (int) MER_CHK(g.chk, CHK_GRID_DENSE, (uint32_t) base, g.n_dense - (uint32_t) (g.res[0] * g.res[1] + g.res[0]) - 1u); (void) dbase;
        if (RIFK == RIFK_CELL8 || RIFK == RIFK_CELL8_BUF) {
            const int cell = (int) MER_CHK(g.chk, CHK_GRID_RECORD, cell8_record(g, x1, y1, z1), g.n_record >> 3);
            float4 a, b;
            if (RIFK == RIFK_CELL8_BUF) {
                const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *) g.cell8, 0, (int) g.buf_bytes, 0x00020000);
                const u32x4 ua = __builtin_amdgcn_raw_buffer_load_b128(rsrc, cell * 32, 0, 0);
                const u32x4 ub = __builtin_amdgcn_raw_buffer_load_b128(rsrc, cell * 32 + 16, 0, 0);
                a = make_float4(__uint_as_float(ua.x), __uint_as_float(ua.y), __uint_as_float(ua.z), __uint_as_float(ua.w));
                b = make_float4(__uint_as_float(ub.x), __uint_as_float(ub.y), __uint_as_float(ub.z), __uint_as_float(ub.w));
            } else {
                const float4 *c = (const float4 *) (g.cell8 + (size_t) cell * 8);
                a = c[0]; b = c[1];
            }
            cc.set(a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w);
        } else if (RIFK == RIFK_DENSE_BUF) {
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *) g.data, 0, (int) g.buf_bytes, 0x00020000);
            const int sy4 = g.res[0] * 4, sz4 = g.res[0] * g.res[1] * 4;
            const u32x2 r00 = __builtin_amdgcn_raw_buffer_load_b64(rsrc, dbase * 4, 0, 0);
            const u32x2 r01 = __builtin_amdgcn_raw_buffer_load_b64(rsrc, dbase * 4, sy4, 0);
            const u32x2 r10 = __builtin_amdgcn_raw_buffer_load_b64(rsrc, dbase * 4, sz4, 0);
            const u32x2 r11 = __builtin_amdgcn_raw_buffer_load_b64(rsrc, dbase * 4, sz4 + sy4, 0);
            cc.set(__uint_as_float(r00.x), __uint_as_float(r00.y), __uint_as_float(r01.x), __uint_as_float(r01.y),
                   __uint_as_float(r10.x), __uint_as_float(r10.y), __uint_as_float(r11.x), __uint_as_float(r11.y));
        } else {
            const float *D = (const float *) g.data;
            const int sy = g.res[0], sz = g.res[0] * g.res[1];
            cc.set(D[dbase], D[dbase + 1], D[dbase + sy], D[dbase + sy + 1], D[dbase + sz], D[dbase + sz + 1], D[dbase + sz + sy], D[dbase + sz + sy + 1]);
        }
    }

}
// all three of (q - corner) in [0,1)  <=>  the largest of the three bit patterns, compared as unsigned, is below that of 1.0f (a negative
// value has the sign bit set; NaN and -0.0f take the slow path, which is always correct)
__device__ __forceinline__ bool cell_left(float fx, float fy, float fz) {
    return MER_CELL_TEST(max(__float_as_uint(fx), max(__float_as_uint(fy), __float_as_uint(fz))) >= 0x3F800000u);
}
template <int RIFK>
__device__ __forceinline__ void trilinear_value_grad(const DGrid &g, CellCache &cc, f3 p, float &val, f3 &grad) {
    const float px = __builtin_fmaf(g.s[0], p.x, g.t[0]), py = __builtin_fmaf(g.s[1], p.y, g.t[1]), pz = __builtin_fmaf(g.s[2], p.z, g.t[2]);
    // fast path: the point is still in the cached cell  <=>  0 <= q - corner < 1 on every axis (the subtraction is
    // exact, so this is the same decision as floor(q) == corner); the four RK4 stages of a half-voxel step mostly are
    float fx = px - cc.cx, fy = py - cc.cy, fz = pz - cc.cz;
    if (cell_left(fx, fy, fz)) {
        cell_fill<RIFK>(g, cc, px, py, pz);
        fx = px - cc.cx; fy = py - cc.cy; fz = pz - cc.cz;
    }
    // value + gradient from the monomial coefficients (Horner in z, then y, then x): 11 fused multiply-adds
    const float A = __builtin_fmaf(cc.axyz, fz, cc.axy), B = __builtin_fmaf(cc.axz, fz, cc.ax),
                C = __builtin_fmaf(cc.ayz, fz, cc.ay), D = __builtin_fmaf(cc.az, fz, cc.a0);
    const float gx = __builtin_fmaf(A, fy, B);                 // df/dx = (ax + axz z) + (axy + axyz z) y
    val = __builtin_fmaf(gx, fx, __builtin_fmaf(C, fy, D));    // f = f(x = 0) + x df/dx   (f is linear in x)
    const float gy = __builtin_fmaf(A, fx, C);                 // df/dy = (ay + ayz z) + (axy + axyz z) x
    const float gz = __builtin_fmaf(__builtin_fmaf(cc.axyz, fx, cc.ayz), fy, __builtin_fmaf(cc.axz, fx, cc.az));    // df/dz = (az + axz x) + (ayz + axyz x) y
    grad = f3(gx * g.s[0], gy * g.s[1], gz * g.s[2]);
}

// ---- predicted-cell fetch -------------------------------------------------------------------------------------------------------------------
// K_march is latency-bound: a wave waits twice per RK4 step -- at stage 2 and at stage 4, where some of its lanes have crossed
// a cell face -- for the slowest of those lanes' misses, and fabric, L2 and VALU all have room.  The ray's next cell is predictable half a step
// ahead (the stage-4 point from the stage-1 slope; the next step's stage-2 point from the stage-3 slope), so the gather of that cell is ISSUED
// at the previous fetch point -- after that point's own loads, which return first -- into a second set of 8 registers that nobody waits for; when
// the ray arrives and the cell is the predicted one the corners are (usually) there.  A wrong prediction costs one wasted gather of a line that
// is needed a step later anyway; the corners are the same floats either way: bit-identical results.  BRICK27 with buffer loads only.
// MEASURED (profiles/round3/ab_predicted_cell_fetch.txt): bit-identical (186 parity tests), and SLOWER -- 106 VGPR / 4 waves per SIMD: 287 against 369
// Mpaths/s on the headline job; forced to 96 VGPR / 5 waves (10 spills): 316; 512^3: 153 / 163 against 179.  The two predictions add ~50 VALU
// instructions and four exec-mask regions to a 344-instruction step, and the time goes up in proportion: with 2-3 of 5 waves waiting on memory at
// any time the ready ones already keep the SIMD's issue port about as busy as their dependent chains allow, so the loop pays for instructions as
// well as for latency.  Kept behind this macro (off) as the record of the variant.
#ifndef MER_PREFETCH
#define MER_PREFETCH 0
#endif
__device__ __forceinline__ int brick27_word(const DGrid &g, int x1, int y1, int z1) {
    const int bs = g.bshift, bm = (1 << bs) - 1, bw = g.bw;
    const int brick = (int) (__umul24(__umul24(z1 >> bs, g.nby) + (y1 >> bs), g.nbx) + (x1 >> bs));
    return (int) MER_CHK(g.chk, CHK_GRID_RECORD, (size_t) (uint32_t) brick * (uint32_t) g.recw + (uint32_t) (((z1 & bm) * bw + (y1 & bm)) * bw + (x1 & bm)),
                         g.n_record - (uint64_t) (bw * bw + bw) - 1u);
}
// PRED: `pred` (world / volume coordinates like p) is where the ray is expected to need a cell next
template <bool PRED>
__device__ __forceinline__ void trilinear_value_grad_pf(const DGrid &g, CellCache &cc, f3 p, f3 pred, float &val, f3 &grad) {
    const float px = __builtin_fmaf(g.s[0], p.x, g.t[0]), py = __builtin_fmaf(g.s[1], p.y, g.t[1]), pz = __builtin_fmaf(g.s[2], p.z, g.t[2]);
    float fx = px - cc.cx, fy = py - cc.cy, fz = pz - cc.cz;
    const bool miss = cell_left(fx, fy, fz);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *) g.cell8, 0, (int) g.buf_bytes, 0x00020000);
    const int bw = g.bw;
    u32x2 r00 = {0u, 0u}, r01 = {0u, 0u}, r10 = {0u, 0u}, r11 = {0u, 0u};
    if (miss) {
        cc.cx = __builtin_amdgcn_fmed3f(floorf(px), 0.0f, (float) (g.res[0] - 2));
        cc.cy = __builtin_amdgcn_fmed3f(floorf(py), 0.0f, (float) (g.res[1] - 2));
        cc.cz = __builtin_amdgcn_fmed3f(floorf(pz), 0.0f, (float) (g.res[2] - 2));
        const int x1 = (int) cc.cx, y1 = (int) cc.cy, z1 = (int) cc.cz;
        const int base = (int) (__umul24(__umul24(z1, g.res[1]) + y1, g.res[0]) + x1);
        cc.cell = base;
        if (base == cc.ncell) { r00 = cc.n00; r01 = cc.n01; r10 = cc.n10; r11 = cc.n11; cc.ncell = -1; }      // the predicted cell: its corners were requested half a step ago
        else {
            const int o = brick27_word(g, x1, y1, z1);
            r00 = __builtin_amdgcn_raw_buffer_load_b64(rsrc, o * 4, 0, 0);
            r01 = __builtin_amdgcn_raw_buffer_load_b64(rsrc, o * 4, bw * 4, 0);
            r10 = __builtin_amdgcn_raw_buffer_load_b64(rsrc, o * 4, bw * bw * 4, 0);
            r11 = __builtin_amdgcn_raw_buffer_load_b64(rsrc, o * 4, (bw * bw + bw) * 4, 0);
        }
    }
    if (PRED) {                 // request the cell of the predicted point, behind this point's own loads
        const float qx = __builtin_amdgcn_fmed3f(floorf(__builtin_fmaf(g.s[0], pred.x, g.t[0])), 0.0f, (float) (g.res[0] - 2)),
                    qy = __builtin_amdgcn_fmed3f(floorf(__builtin_fmaf(g.s[1], pred.y, g.t[1])), 0.0f, (float) (g.res[1] - 2)),
                    qz = __builtin_amdgcn_fmed3f(floorf(__builtin_fmaf(g.s[2], pred.z, g.t[2])), 0.0f, (float) (g.res[2] - 2));
        const int x1 = (int) qx, y1 = (int) qy, z1 = (int) qz;
        const int nb = (int) (__umul24(__umul24(z1, g.res[1]) + y1, g.res[0]) + x1);
        if (nb != cc.cell && nb != cc.ncell) {
            const int o = brick27_word(g, x1, y1, z1);
            cc.n00 = __builtin_amdgcn_raw_buffer_load_b64(rsrc, o * 4, 0, 0);
            cc.n01 = __builtin_amdgcn_raw_buffer_load_b64(rsrc, o * 4, bw * 4, 0);
            cc.n10 = __builtin_amdgcn_raw_buffer_load_b64(rsrc, o * 4, bw * bw * 4, 0);
            cc.n11 = __builtin_amdgcn_raw_buffer_load_b64(rsrc, o * 4, (bw * bw + bw) * 4, 0);
            cc.ncell = nb;
        }
    }
    if (miss) {
        cc.set(__uint_as_float(r00.x), __uint_as_float(r00.y), __uint_as_float(r01.x), __uint_as_float(r01.y),
               __uint_as_float(r10.x), __uint_as_float(r10.y), __uint_as_float(r11.x), __uint_as_float(r11.y));
        fx = px - cc.cx; fy = py - cc.cy; fz = pz - cc.cz;
    }
    const float A = __builtin_fmaf(cc.axyz, fz, cc.axy), B = __builtin_fmaf(cc.axz, fz, cc.ax),
                C = __builtin_fmaf(cc.ayz, fz, cc.ay), D = __builtin_fmaf(cc.az, fz, cc.a0);
    const float gx = __builtin_fmaf(A, fy, B);
    val = __builtin_fmaf(gx, fx, __builtin_fmaf(C, fy, D));
    const float gy = __builtin_fmaf(A, fx, C);
    const float gz = __builtin_fmaf(__builtin_fmaf(cc.axyz, fx, cc.ayz), fy, __builtin_fmaf(cc.axz, fx, cc.az));
    grad = f3(gx * g.s[0], gy * g.s[1], gz * g.s[2]);
}

// Cubic B-spline basis and derivative at the 4 taps around x (include/mitsuba/core/basisspline.h:40-72).
// t = x - floor(x) in [0,1); tap distances t+1, t, t-1, t-2.
__device__ __forceinline__ void bspline_weights(float t, float w[4], float dw[4]) {
    const float a = 1.0f - t;              // 2 - (t+1)
    // |x| in (1,2]: (1/6)(2-|x|)^3 ; |x| <= 1: 2/3 - x^2 + x^3/2
    const float x0 = t + 1.0f, x3 = 2.0f - t, x2 = 1.0f - t;
    w[0] = (1.0f / 6.0f) * (2.0f - x0) * (2.0f - x0) * (2.0f - x0);
    w[1] = (2.0f / 3.0f) - t * t + 0.5f * t * t * t;
    w[2] = (2.0f / 3.0f) - x2 * x2 + 0.5f * x2 * x2 * x2;
    w[3] = (1.0f / 6.0f) * (2.0f - x3) * (2.0f - x3) * (2.0f - x3);
    // derivative w.r.t. x of beta(x - i): sign(x-i) * {(-1/2)(2-|.|)^2 | (1.5|.| - 2)|.|}
    dw[0] = -0.5f * a * a;                         // distance t+1 > 0, in (1,2]
    dw[1] = (1.5f * t - 2.0f) * t;                 // distance t >= 0
    dw[2] = -((1.5f * x2 - 2.0f) * x2);            // distance t-1 < 0
    dw[3] = 0.5f * t * t;                          // distance t-2 < 0, |.| = 2-t in (1,2]
}

// Spline<3>::valueAndGradient (basisspline.h:438-471) with the per-axis weights hoisted.
// Caller guarantees the point lies inside the interpolatable limits (splinevolume.cpp:319-324).
__device__ __forceinline__ void bspline_value_grad(const DGrid &g, f3 p, float &val, f3 &grad) {
    const float px = (p.x - g.bmin[0]) * g.s[0], py = (p.y - g.bmin[1]) * g.s[1], pz = (p.z - g.bmin[2]) * g.s[2];  // convertToX :655-658
    const float flx = floorf(px), fly = floorf(py), flz = floorf(pz);
    float wx[4], dwx[4], wy[4], dwy[4], wz[4], dwz[4];
    bspline_weights(px - flx, wx, dwx);
    bspline_weights(py - fly, wy, dwy);
    bspline_weights(pz - flz, wz, dwz);
    int ix = (int) flx - 1, iy = (int) fly - 1, iz = (int) flz - 1;
    ix = min(max(ix, 0), g.res[0] - 4); iy = min(max(iy, 0), g.res[1] - 4); iz = min(max(iz, 0), g.res[2] - 4);  // memory safety only
    const float *C = g.coeff + MER_CHK(g.chk, CHK_GRID_COEFF, ((size_t) iz * g.res[1] + iy) * g.res[0] + ix, g.n_dense - 3u * (uint32_t) (g.res[0] * g.res[1] + g.res[0] + 1));
    const int sy = g.res[0], sz = g.res[0] * g.res[1];
    float f = 0.0f, gx = 0.0f, gy = 0.0f, gz = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        float fk = 0.0f, gxk = 0.0f, gyk = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float *row = C + k * sz + j * sy;
            const float c0 = row[0], c1 = row[1], c2 = row[2], c3 = row[3];
            const float rx = c0 * wx[0] + c1 * wx[1] + c2 * wx[2] + c3 * wx[3];
            const float rdx = c0 * dwx[0] + c1 * dwx[1] + c2 * dwx[2] + c3 * dwx[3];
            fk += rx * wy[j]; gxk += rdx * wy[j]; gyk += rx * dwy[j];
        }
        f += fk * wz[k]; gx += gxk * wz[k]; gy += gyk * wz[k]; gz += fk * dwz[k];
    }
    val = f;
    grad = f3(gx * g.s[0], gy * g.s[1], gz * g.s[2]);      // * dxres, basisspline.h:467-469
}

// world -> volume space of a grid with a `toWorld` transform (splinevolume.cpp:320-376: m_worldToVolume(pc)), and the gradient back:
// m_worldToVolume_RotT * v (:343,359)
__device__ __forceinline__ f3 to_volume(const DGrid &g, f3 p) {
    return f3(g.w2v[0] * p.x + g.w2v[1] * p.y + g.w2v[2] * p.z + g.w2v[3], g.w2v[4] * p.x + g.w2v[5] * p.y + g.w2v[6] * p.z + g.w2v[7],
              g.w2v[8] * p.x + g.w2v[9] * p.y + g.w2v[10] * p.z + g.w2v[11]);
}
__device__ __forceinline__ f3 rot_t(const DGrid &g, f3 v) {
    return f3(g.w2v[0] * v.x + g.w2v[4] * v.y + g.w2v[8] * v.z, g.w2v[1] * v.x + g.w2v[5] * v.y + g.w2v[9] * v.z, g.w2v[2] * v.x + g.w2v[6] * v.y + g.w2v[10] * v.z);
}
__device__ __forceinline__ bool inside_volume_limits(const DGrid &g, f3 p) {   // splinevolume.cpp:319-324
    if (g.affine) p = to_volume(g, p);
    return p.x > g.lim_min[0] && p.x < g.lim_max[0] && p.y > g.lim_min[1] && p.y < g.lim_max[1] &&
           p.z > g.lim_min[2] && p.z < g.lim_max[2];
}

// AcousticRIFVolume::valueAndGradient / gradientAndHessian (src/volume/acousticrifvolume.cpp:224-342): n = n_o + n_max J_m(k_r r) cos(m phi)
// in the (y, z) plane, r clamped at EpsilonRIF = 1e-8 (:15, :235-239); single precision, jnf / atan2f of the device library.
#define MER_EPSILON_RIF 1e-8f
__device__ __forceinline__ void acoustic_value_grad(const DGrid &g, f3 pc, float &n, f3 &gr) {
    float py = pc.y, pz = pc.z;
    float r = sqrtf(py * py + pz * pz);
    const float phi = atan2f(py, pz);
    if (r < MER_EPSILON_RIF) { py = MER_EPSILON_RIF; pz = MER_EPSILON_RIF; r = MER_EPSILON_RIF; }
    const float kr = g.ac_k_r, krr = kr * r, m = (float) g.ac_mode;
    const float bj = jnf(g.ac_mode, krr), dbj = m / krr * bj - jnf(g.ac_mode + 1, krr);
    const float invr = 1.0f / r, invr2 = invr * invr;
    const float cosmp = cosf(m * phi), sinmp = sinf(m * phi);
    n = g.ac_n_o + g.ac_n_max * bj * cosmp;
    gr = f3(0.0f, g.ac_n_max * (dbj * kr * py * invr * cosmp - bj * m * sinmp * pz * invr2),
            g.ac_n_max * (dbj * kr * pz * invr * cosmp + bj * m * sinmp * py * invr2));
}
// A RIF volume with a `toWorld` is read in the dense layout or as a spline: the record layouts of the hot path (CELL8 / BRICK27) carry
// no transform, so that the bench kernels do not pay for the wave-uniform test (measured: 4 % of K_march) -- make_params refuses the
// combination.
template <int RIF> __device__ __forceinline__ constexpr bool rif_affine_capable() { return RIF == MER_RIF_TRILINEAR || RIF == RIFK_DENSE_BUF || RIF == MER_RIF_BSPLINE3; }
template <int RIF> __device__ __forceinline__ void rif_value_grad(const DGrid &g, CellCache &cc, f3 p, float &n, f3 &gr) {
    if (RIF == RIFK_ACOUSTIC) { acoustic_value_grad(g, p, n, gr); return; }
    const bool affine = rif_affine_capable<RIF>() && g.affine;          // wave-uniform
    if (affine) p = to_volume(g, p);
    if (RIF != MER_RIF_BSPLINE3) trilinear_value_grad<RIF>(g, cc, p, n, gr);
    else bspline_value_grad(g, p, n, gr);
    if (affine) gr = rot_t(g, gr);
}

// RK4 is new functionality: its 1/n is the hardware reciprocal (v_rcp_f32, <= 1 ulp) instead of the ~12-
// instruction IEEE division sequence; the Verlet form (the reference's er_step) keeps the correctly rounded division.
#ifdef MER_IEEE_RCP
#define MER_RCP(x) (1.0f / (x))
#else
#define MER_RCP(x) __builtin_amdgcn_rcpf(x)
#endif
__device__ __forceinline__ f3 fma3(float s, f3 a, f3 b) {           // s*a + b, fused per component
    return f3(__builtin_fmaf(s, a.x, b.x), __builtin_fmaf(s, a.y, b.y), __builtin_fmaf(s, a.z, b.z));
}

// ------------------------------------------------------------------------------------------------
// er_step: velocity-Verlet (heterogeneousrefractive.cpp:653-661) or classic RK4 on
// dp/ds = v/n, dv/ds = grad n, dopt/ds = n.  The Verlet form keeps the reference's operation
// order; the RK4 form is new and is defined with fused multiply-adds (restated identically in the oracle).
template <int RIF, int STEPPER>
__device__ __forceinline__ void er_step(const DGrid &g, CellCache &cc, f3 &p, f3 &v, float h, float &opt) {
    if (STEPPER == MER_STEP_VERLET) {
        float n, n2; f3 G, G2;
        rif_value_grad<RIF>(g, cc, p, n, G);
        v = v + 0.5f * h * G;
        p = p + h * v / n;
        rif_value_grad<RIF>(g, cc, p, n2, G2);
        v = v + 0.5f * h * G2;
        opt += h * n;
    } else {
        float n; f3 gr;
        const float hh = 0.5f * h;
        // predicted-cell fetch (trilinear_value_grad_pf): same evaluations at the same points; stages 2 and 4 also REQUEST the cell the ray is expected in at
        // the other of the two (stage 4 from the stage-1 slope, the next step's stage 2 from the stage-3 slope)
        constexpr bool PF = MER_PREFETCH && RIF == RIFK_BRICK27_BUF;
        const bool pf = PF && !g.affine;
        if (pf) trilinear_value_grad_pf<false>(g, cc, p, p, n, gr); else
        rif_value_grad<RIF>(g, cc, p, n, gr);                       // k1
        f3 kp = v * MER_RCP(n);
        f3 ps = kp, vs = gr; float ns = n;
        f3 vv = fma3(hh, gr, v);
        if (pf) trilinear_value_grad_pf<true>(g, cc, fma3(hh, kp, p), fma3(h, kp, p), n, gr); else
        rif_value_grad<RIF>(g, cc, fma3(hh, kp, p), n, gr);          // k2
        kp = vv * MER_RCP(n);
        ps = fma3(2.0f, kp, ps); vs = fma3(2.0f, gr, vs); ns = __builtin_fmaf(2.0f, n, ns);
        vv = fma3(hh, gr, v);
        if (pf) trilinear_value_grad_pf<false>(g, cc, fma3(hh, kp, p), p, n, gr); else
        rif_value_grad<RIF>(g, cc, fma3(hh, kp, p), n, gr);          // k3
        kp = vv * MER_RCP(n);
        ps = fma3(2.0f, kp, ps); vs = fma3(2.0f, gr, vs); ns = __builtin_fmaf(2.0f, n, ns);
        vv = fma3(h, gr, v);
        if (pf) trilinear_value_grad_pf<true>(g, cc, fma3(h, kp, p), fma3(1.5f * h, kp, p), n, gr); else
        rif_value_grad<RIF>(g, cc, fma3(h, kp, p), n, gr);           // k4
        kp = vv * MER_RCP(n);
        ps = ps + kp; vs = vs + gr; ns = ns + n;
        const float h6 = h * (1.0f / 6.0f);
        p = fma3(h6, ps, p);
        v = fma3(h6, vs, v);
        opt = __builtin_fmaf(h6, ns, opt);
    }
}
template <int STEPPER> __device__ __forceinline__ constexpr int evals_per_step() { return STEPPER == MER_STEP_VERLET ? 2 : 4; }

// ------------------------------------------------------------------------------------------------
// Phase functions (src/phase/hg.cpp:74-110, src/phase/isotropic.cpp:62-78); wi points away from the vertex.
__device__ __forceinline__ float safe_sqrt(float v) { return sqrtf(fmaxf(0.0f, v)); }      // math.h:260-267
__device__ __forceinline__ f3 square_to_uniform_sphere(float sx, float sy) {                  // warp.cpp:25-31
    const float z = 1.0f - 2.0f * sy;
    const float r = safe_sqrt(1.0f - z * z);
    const float phi = 2.0f * MER_PI * sx;
    return f3(r * cosf(phi), r * sinf(phi), z);
}
__device__ __forceinline__ void coordinate_system(f3 a, f3 &b, f3 &c) {                       // util.cpp:606-615
    if (fabsf(a.x) > fabsf(a.y)) {
        const float invLen = 1.0f / sqrtf(a.x * a.x + a.z * a.z);
        c = f3(a.z * invLen, 0.0f, -a.x * invLen);
    } else {
        const float invLen = 1.0f / sqrtf(a.y * a.y + a.z * a.z);
        c = f3(0.0f, a.z * invLen, -a.y * invLen);
    }
    b = cross(c, a);
}
__device__ __forceinline__ float phase_eval(int kind, float g, f3 wi, f3 wo) {
    if (kind == MER_PHASE_ISOTROPIC) return MER_INV_FOURPI;
    const float temp = 1.0f + g * g + 2.0f * g * dot(wi, wo);
    return MER_INV_FOURPI * (1 - g * g) / (temp * sqrtf(temp));
}
__device__ __forceinline__ float phase_sample(int kind, float g, f3 wi, float sx, float sy, f3 &wo, float &pdf) {
    if (kind == MER_PHASE_ISOTROPIC) { wo = square_to_uniform_sphere(sx, sy); pdf = MER_INV_FOURPI; return 1.0f; }
    float cosTheta;
    if (fabsf(g) < MER_EPSILON) cosTheta = 1 - 2 * sx;
    else {
        const float sqrTerm = (1 - g * g) / (1 - g + 2 * g * sx);
        cosTheta = (1 + g * g - sqrTerm * sqrTerm) / (2 * g);
    }
    const float sinTheta = safe_sqrt(1.0f - cosTheta * cosTheta);
    const float phi = 2 * MER_PI * sy, sinPhi = sinf(phi), cosPhi = cosf(phi);
    const f3 n = -wi; f3 s, t;
    coordinate_system(n, s, t);                                    // Frame(n), frame.h:55-57
    const f3 l(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta);
    wo = s * l.x + t * l.y + n * l.z;                              // frame.h:83-85
    pdf = phase_eval(kind, g, wi, wo);
    return 1.0f;
}

// ------------------------------------------------------------------------------------------------
// A segmented work list (mer_wavefront.hpp)
#define MER_NSEG 32
struct SegQueue {
    uint32_t *items;          // [MER_NSEG][segcap]
    uint32_t *counts;         // [MER_LIVE_SLOTS][MER_NSEG], row = pass & (MER_LIVE_SLOTS-1)
    uint32_t segcap;
    unsigned long long *chk;  // MER_BOUNDS_CHECK: violation record (NULL in the product build)
    uint16_t *keys;           // march lists under option march_sort: the cell of each item's position, same indexing as items (else NULL)
};

// One of a pair of ping-pong queues, by value (K_connect / the EXTRA K_event, whose kernel arguments must stay out of scratch).  Written with constant indices and a select: an address-taken `P.cq[row & 1]` can make the compiler copy the
// whole 2.3 KB kernel-argument struct into per-lane scratch at kernel entry (the address of a kernarg member indexed by a run-time
// value), and every later read of a Params field in that kernel then becomes a scratch load.
__device__ __forceinline__ SegQueue pick_queue(const SegQueue (&q)[2], uint32_t row) {
    SegQueue r = q[0]; const SegQueue b = q[1];
    if (row & 1u) r = b;
    return r;
}

// MaxExpDist (src/medium/maxexp.h:28-98): the distance distribution proportional to max_i sigma_i exp(-sigma_i t) over the three
// channels (`strategy = maximum` of homogeneous / heterogeneousrefractive).  Tables built on the host (mer_api.hip) as :30-58.
struct MaxExp {
    float sigmaT[3], cdf[4], intervalStart[3], normalization, invNormalization;
};
__device__ __forceinline__ int maxexp_lower_bound(const float *a, int n, float v) { int k = 0; while (k < n && a[k] < v) k++; return k; }
__device__ __forceinline__ float maxexp_sample(const MaxExp &m, float u, float &pdf) {                  // :60-75
    const int index = max(0, maxexp_lower_bound(m.cdf, 4, u) - 1);
    const float t = -logf(expf(-m.intervalStart[index] * m.sigmaT[index]) - m.normalization * (u - m.cdf[index])) / m.sigmaT[index];
    pdf = m.sigmaT[index] * expf(-m.sigmaT[index] * t) * m.invNormalization;
    return t;
}
__device__ __forceinline__ float maxexp_cdf(const MaxExp &m, float t) {                                  // :85-96
    const int index = max(0, maxexp_lower_bound(m.intervalStart, 3, t) - 1);
    const float lower = (index == 0) ? -1.0f : -powf(m.sigmaT[index] / m.sigmaT[max(index - 1, 0)], -m.sigmaT[index] / (m.sigmaT[index] - m.sigmaT[max(index - 1, 0)]));
    const float upper = -expf(-m.sigmaT[index] * t);
    return m.cdf[index] + (upper - lower) * m.invNormalization;
}

// Everything a render / leaf kernel needs, passed by value as the kernel argument.
struct Params {
    mer_scene_desc sc;
    DGrid density, albedo, rif;
    // derived
    f3    sigA, sigS, sigT;
    float medium_sampling_weight, sampling_density;
    MaxExp maxexp;                      // strategy = maximum
    float inv_max_density;
    float het_step;               // method = simpson: the heterogeneous medium's stepSize (given, or inferred from the grids)
    float cam[12], aspect, cot_half_fov, inv_res_x, inv_res_y;
    const float *ftable;                // reconstruction-filter table, 33 floats in device memory: a table INSIDE this struct, indexed per lane, makes the
                                        // compiler copy the whole 2.3 KB kernel-argument struct into scratch (K_connect, the EXTRA K_event)
    float fradius, fscale;
    // work
    uint64_t seed;
    int32_t spp_begin, spp_count, spp_stride, tile_rank, tile_count;
    int32_t tiles_x, tiles_y, ntiles_mine;
    int32_t tile_skew;                  // tile dealing: row ty of the tile grid is rotated by tile_skew * ty columns before the round-robin deal (0 = plain)
    uint64_t total_work;
    float *film;                        // float[H][W][film_ch]: RGB per frame, alpha, weight
    int32_t frames, film_ch;            // frames = 1 and film_ch = 5 in steady state
    float mod_phase;                    // path-length modulation phase in radians
    float *path_out;                    // per-path radiance (mer_render_paths) or NULL
    unsigned long long *counters;       // MER_C_COUNT
    unsigned long long *work_counter;
    int32_t dbg_pixel;
    // wavefront path-state slots (struct of arrays, word k of slot i at slots[k*nslots + i])
    uint32_t *slots; uint32_t nslots; int32_t ksteps;
    uint32_t nslots_all;                // nslots path slots + the side-walk slots behind them (4 per path when walks are spawned, else 0): extent of `slots` and of the work lists' item ids
    int32_t spawn;                      // 1: K_event hands luminaire-sample / look-up transmittance walks to side-walk slots and goes on with the path (mer_wavefront.hpp)
    uint32_t *live;                     // live[0]: number of finished slots
    SegQueue eq, mq[2], sq[2];          // event queue, march lists (by pass parity), starved lists (by pass parity)
    DGrid sdf; float sdf_eps;           // boundary = MER_BOUNDARY_SDF: signed-distance grid (negative inside), 1e-4 x its diagonal
    SegQueue cq[2];                     // pending curved-ray connections (K_connect): launch l reads cq[l & 1] row l, re-queues into cq[(l+1) & 1] row l+1
    uint32_t *cstate;                   // parked solver state of the pending connections, MER_CSTATE_WORDS per slot (mer_connect.hpp)
    unsigned long long *hitq; unsigned long long hitq_cap;     // ring of work ids that will march (power-of-two capacity)
    unsigned long long *hitq_ctr;       // [0] produced (tail), [1] consumed (head)
    int32_t gen_iters, gen_all;
    uint32_t cq_row;                    // index l of the next K_connect launch: K_event appends its requests to cq[l & 1] row l
    int32_t mq_sort;                    // 1: march lists sorted by estimated steps to the boundary (option mq_sort = 0 turns it off for A/B runs)
    // spatial sort of the march list (option march_sort; mer_wavefront.hpp, msort_* kernels): bits per axis of the cell grid over the RIF's world box (0 = off),
    // bin order (0: cell-major, 1: class-major), the sorted list K_march sweeps, and the counting sort's scratch
    int32_t msort, msort_major; float msort_o[3], msort_s[3];
    uint32_t *msorted, *msort_hist, *msort_cursor, *msort_count;
    unsigned long long *chk;            // MER_BOUNDS_CHECK: violation record (NULL in the product build)
    uint64_t n_film, n_path_out;        // float counts of film / path_out: the extents the checks use
    // emitter `area` on a `rectangle` (EXTRA kernels, straight rays): objectToWorld, its inverse, the frame normal, 1 / area (make_params)
    int32_t has_area; float rect_o2w[12], rect_w2o[12], rect_n[3], rect_inv_area;
};
#define MER_LIVE_SLOTS 4096
#define MER_COUNTER_REPLICAS 64        // counters are flushed into one of this many copies (summed on the host)

__device__ __forceinline__ bool inside_shape(const mer_scene_desc &s, f3 p) {   // heterogeneousrefractive.cpp:707-726 as data (D5)
    if (s.boundary == MER_BOUNDARY_SPHERE) {
        const f3 q(p.x - s.sph_center[0], p.y - s.sph_center[1], p.z - s.sph_center[2]);
        return dot(q, q) < s.sph_radius * s.sph_radius;
    }
    return p.x >= s.bmin[0] && p.x <= s.bmax[0] && p.y >= s.bmin[1] && p.y <= s.bmax[1] && p.z >= s.bmin[2] && p.z <= s.bmax[2];
}

// ray / boundary-shape intersection restricted to [mint,maxt]; returns t or -1
__device__ __forceinline__ float intersect_shape(const mer_scene_desc &s, f3 o, f3 d, float mint, float maxt) {
    float nearT, farT;
    if (s.boundary == MER_BOUNDARY_SPHERE) {
        // src/shapes/sphere.cpp rayIntersect: double-precision quadratic
        const double ox = (double) o.x - s.sph_center[0], oy = (double) o.y - s.sph_center[1], oz = (double) o.z - s.sph_center[2];
        const double dx = d.x, dy = d.y, dz = d.z;
        const double A = dx * dx + dy * dy + dz * dz, B = 2 * (dx * ox + dy * oy + dz * oz),
                     C = ox * ox + oy * oy + oz * oz - (double) s.sph_radius * s.sph_radius;
        const double disc = B * B - 4 * A * C;
        if (disc < 0) return -1.0f;
        const double root = sqrt(disc);
        const double q = B < 0 ? -0.5 * (B - root) : -0.5 * (B + root);
        double t0 = q / A, t1 = C / q;
        if (t0 > t1) { double tmp = t0; t0 = t1; t1 = tmp; }
        nearT = (float) t0; farT = (float) t1;
    } else {
        if (!aabb_intersect(s.bmin, s.bmax, o, d, nearT, farT)) return -1.0f;
    }
    if (!(nearT <= maxt && farT >= mint)) return -1.0f;
    if (nearT >= mint) return nearT;
    if (farT <= maxt) return farT;
    return -1.0f;
}

// PerspectiveCamera::sampleRay (src/sensors/perspective.cpp:247-269) with the analytic inverse of
// cameraToSample at the near plane (perspective.cpp:150-155, transform.cpp:99-123)
__device__ __forceinline__ void sample_ray(const Params &P, float px, float py, f3 &o, f3 &d, float &mint, float &maxt) {
    const float sx = px * P.inv_res_x, sy = py * P.inv_res_y;
    const f3 nearP((1.0f - 2.0f * sx) * P.sc.near_clip / P.cot_half_fov,
                   (1.0f - 2.0f * sy) / P.aspect * P.sc.near_clip / P.cot_half_fov, P.sc.near_clip);
    const f3 dl = normalize(nearP);
    const float invZ = 1.0f / dl.z;
    mint = P.sc.near_clip * invZ; maxt = P.sc.far_clip * invZ;
    o = f3(P.cam[3], P.cam[7], P.cam[11]);
    d = f3(P.cam[0] * dl.x + P.cam[1] * dl.y + P.cam[2] * dl.z,
           P.cam[4] * dl.x + P.cam[5] * dl.y + P.cam[6] * dl.z,
           P.cam[8] * dl.x + P.cam[9] * dl.y + P.cam[10] * dl.z);
}

__device__ __forceinline__ f3 albedo_at(const Params &P, f3 p) {
    if (P.sc.albedo_mode == MER_ALBEDO_GRID) return lookup_spectrum(P.albedo, p);
    return f3(P.sc.albedo[0], P.sc.albedo[1], P.sc.albedo[2]);      // constvolume.cpp:57-64
}

__device__ __forceinline__ float mi_weight(float a, float b) { a *= a; b *= b; return a / (a + b); }   // volpath.cpp:430-433

// outward geometric normal of the boundary shape at a surface point (cube: the face whose plane the point is closest to)
__device__ __forceinline__ f3 shape_normal(const mer_scene_desc &s, f3 x) {
    if (s.boundary == MER_BOUNDARY_SPHERE) return normalize(f3(x.x - s.sph_center[0], x.y - s.sph_center[1], x.z - s.sph_center[2]));
    float best = -1.0f, sign = 1.0f; int axis = 0;
    const float xx[3] = {x.x, x.y, x.z};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float c = 0.5f * (s.bmin[i] + s.bmax[i]), hsz = 0.5f * (s.bmax[i] - s.bmin[i]);
        const float r = fabsf(xx[i] - c) / hsz;
        if (r > best) { best = r; axis = i; sign = xx[i] >= c ? 1.0f : -1.0f; }
    }
    return f3(axis == 0 ? sign : 0.0f, axis == 1 ? sign : 0.0f, axis == 2 ? sign : 0.0f);
}
// ---- boundary kind as a template switch (BND = 1: the negative region of a signed-distance grid, the reference's `sdf` child,
// src/medium/heterogeneousrefractive.cpp:366-375,481).  Kept out of the BND = 0 kernels: the hot loop must not carry a second gather.
// lookupFloat (gridvolume.cpp:337-388) of the signed-distance grid, "far outside" (1e30) where lookupFloat returns 0 for a point off
// the grid.  Written without a branch: the cell index is clamped into the grid, the eight corners are always fetched and the bounds
// test selects the result (same integer contract and blend order as lookup_float).  The branching form -- an early return inside
// the marching loops of K_connect -- was miscompiled by this toolchain at -O2 and above when the lanes of a wave diverge
// (scratch/miscompile/: 1 of 64 connections found against 61 at -O1).
#ifdef MER_SDF_BRANCHING          // scratch/miscompile/pl.hip only: round 1's form, kept to reproduce the miscompile
__device__ __forceinline__ float lookup_float_branching(const DGrid &g, f3 p, int *idx4 = nullptr) {
    const float px = g.m[0] * p.x + g.m[1] * p.y + g.m[2] * p.z + g.m[3], py = g.m[4] * p.x + g.m[5] * p.y + g.m[6] * p.z + g.m[7],
                pz = g.m[8] * p.x + g.m[9] * p.y + g.m[10] * p.z + g.m[11];     // Transform::transformAffine (transform.h:147-155)
    const int x1 = (int) floorf(px), y1 = (int) floorf(py), z1 = (int) floorf(pz);
    if (idx4) { idx4[0] = x1; idx4[1] = y1; idx4[2] = z1; idx4[3] = -1; }
    // x2 = x1 + 1 >= res, written so that it cannot wrap: v_cvt_i32_f32 saturates, a coordinate of +inf (or >= 2^31) gives
    // x1 = INT_MAX, and INT_MAX + 1 would pass the test and fetch from a wild address
    if (x1 < 0 || y1 < 0 || z1 < 0 || x1 >= g.res[0] - 1 || y1 >= g.res[1] - 1 || z1 >= g.res[2] - 1) return 0.0f;
    const float fx = px - (float) x1, fy = py - (float) y1, fz = pz - (float) z1,
                _fx = 1.0f - fx, _fy = 1.0f - fy, _fz = 1.0f - fz;
    const int base = (z1 * g.res[1] + y1) * g.res[0] + x1;
    if (idx4) idx4[3] = base;
    float d000, d001, d010, d011, d100, d101, d110, d111;
    if (g.layout == MER_LAYOUT_CELL8) {
        const int cell = (int) cell8_record(g, x1, y1, z1);
        const float4 *c = (const float4 *) (g.cell8 + (size_t) MER_CHK(g.chk, CHK_GRID_RECORD, (size_t) cell * 8, g.n_record - 7));
        const float4 a = c[0], b = c[1];
        d000 = a.x; d001 = a.y; d010 = a.z; d011 = a.w; d100 = b.x; d101 = b.y; d110 = b.z; d111 = b.w;
    } else {
        const int sy = g.res[0], sz = g.res[0] * g.res[1];
        d000 = grid_fetch(g, base);          d001 = grid_fetch(g, base + 1);
        d010 = grid_fetch(g, base + sy);     d011 = grid_fetch(g, base + sy + 1);
        d100 = grid_fetch(g, base + sz);     d101 = grid_fetch(g, base + sz + 1);
        d110 = grid_fetch(g, base + sz + sy); d111 = grid_fetch(g, base + sz + sy + 1);
    }
    return ((d000 * _fx + d001 * fx) * _fy + (d010 * _fx + d011 * fx) * fy) * _fz +
           ((d100 * _fx + d101 * fx) * _fy + (d110 * _fx + d111 * fx) * fy) * fz;
}
__device__ __forceinline__ float sdf_value(const Params &P, f3 p) {
    int idx4[4];
    const float v = lookup_float_branching(P.sdf, p, idx4);
    return idx4[3] >= 0 ? v : 1e30f;
}
#else
__device__ __forceinline__ float sdf_value(const Params &P, f3 p) {
    const DGrid &g = P.sdf;
    const float px = g.m[0] * p.x + g.m[1] * p.y + g.m[2] * p.z + g.m[3], py = g.m[4] * p.x + g.m[5] * p.y + g.m[6] * p.z + g.m[7],
                pz = g.m[8] * p.x + g.m[9] * p.y + g.m[10] * p.z + g.m[11];     // Transform::transformAffine (transform.h:147-155)
    const int x1 = (int) floorf(px), y1 = (int) floorf(py), z1 = (int) floorf(pz);
    const bool inside = !(x1 < 0 || y1 < 0 || z1 < 0 || x1 >= g.res[0] - 1 || y1 >= g.res[1] - 1 || z1 >= g.res[2] - 1) && px == px && py == py && pz == pz;   // NaN: (int) is INT_MIN on the reference's x86, 0 here
    const int xc = min(max(x1, 0), g.res[0] - 2), yc = min(max(y1, 0), g.res[1] - 2), zc = min(max(z1, 0), g.res[2] - 2);
    const float fx = px - (float) x1, fy = py - (float) y1, fz = pz - (float) z1, _fx = 1.0f - fx, _fy = 1.0f - fy, _fz = 1.0f - fz;
    const int base = (zc * g.res[1] + yc) * g.res[0] + xc, sy = g.res[0], sz = g.res[0] * g.res[1];
    const float d000 = grid_fetch(g, base), d001 = grid_fetch(g, base + 1), d010 = grid_fetch(g, base + sy), d011 = grid_fetch(g, base + sy + 1),
                d100 = grid_fetch(g, base + sz), d101 = grid_fetch(g, base + sz + 1), d110 = grid_fetch(g, base + sz + sy), d111 = grid_fetch(g, base + sz + sy + 1);
    const float v = ((d000 * _fx + d001 * fx) * _fy + (d010 * _fx + d011 * fx) * fy) * _fz +
                    ((d100 * _fx + d101 * fx) * _fy + (d110 * _fx + d111 * fx) * fy) * fz;
    return inside ? v : 1e30f;
}
#endif
template <int BND> __device__ __forceinline__ bool inside_shape_b(const Params &P, f3 p) {
    if (BND == 0) return inside_shape(P.sc, p);
    return sdf_value(P, p) < 0.0f;
}
// sphere tracing on the grid; entry points come back with sdf < eps/2, exit points with sdf in [eps, ~2 eps); a start within 4 eps of the
// surface is a start from the inside side and must get below 0 ("armed") before an exit counts
template <int BND> __device__ __forceinline__ float intersect_shape_b(const Params &P, f3 o, f3 d, float mint, float maxt) {
    if (BND == 0) return intersect_shape(P.sc, o, d, mint, maxt);
    float nearT, farT;
    if (!aabb_intersect(P.sdf.wmin, P.sdf.wmax, o, d, nearT, farT)) return -1.0f;
    const float t0 = fmaxf(nearT, mint), t1 = fminf(farT, maxt), eps = P.sdf_eps;
    if (!(t0 <= t1)) return -1.0f;
    float t = t0, v = sdf_value(P, o + d * t);
    const bool insideStart = v < 4 * eps;
    bool armed = false;
    for (int it = 0; it < 1024; ++it) {
        if (insideStart) {
            if (v < 0) armed = true;
            else if (armed && v >= eps) return t;
            else if (!armed && it >= 16) return t;
        } else if (v < 0.5f * eps) return t;
        t += v > 1e29f ? eps : fmaxf(fabsf(v), eps);
        if (t > t1) return insideStart ? (farT <= maxt ? farT : -1.0f) : -1.0f;
        v = sdf_value(P, o + d * t);
    }
    return -1.0f;
}
template <int BND> __device__ __forceinline__ f3 shape_normal_b(const Params &P, f3 x) {
    if (BND == 0) return shape_normal(P.sc, x);
    float v; f3 g; CellCache cc; cc.reset();
    rif_value_grad<MER_RIF_TRILINEAR>(P.sdf, cc, x, v, g);                 // normalized SDF gradient (heterogeneousrefractive.cpp:980-984)
    return normalize(g);
}

// fresnelDielectricExt (src/libcore/util.cpp:665-695)
__device__ __forceinline__ float fresnel_dielectric_ext(float cosThetaI_, float &cosThetaT_, float eta) {
    if (eta == 1.0f) { cosThetaT_ = -cosThetaI_; return 0.0f; }
    const float scale = (cosThetaI_ > 0) ? 1 / eta : eta, cosThetaTSqr = 1 - (1 - cosThetaI_ * cosThetaI_) * (scale * scale);
    if (cosThetaTSqr <= 0.0f) { cosThetaT_ = 0.0f; return 1.0f; }
    const float cosThetaI = fabsf(cosThetaI_), cosThetaT = sqrtf(cosThetaTSqr);
    const float Rs = (cosThetaI - eta * cosThetaT) / (cosThetaI + eta * cosThetaT);
    const float Rp = (eta * cosThetaI - cosThetaT) / (eta * cosThetaI + cosThetaT);
    cosThetaT_ = (cosThetaI_ > 0) ? -cosThetaT : cosThetaT;
    return 0.5f * (Rs * Rs + Rp * Rp);
}
// HDielectric::sample (src/bsdfs/hdielectric.cpp:183-242, ERadiance) at the boundary point ro + rd*t of the medium shape; eta is the
// RIF there (:115-118).  Returns true when the sampled direction wo stays in / enters the medium.
template <bool CURVED, int RIF, int BND = 0>
__device__ __forceinline__ bool dielectric_event(const Params &P, Rng &rng, f3 ro, f3 rd, float t, bool from_inside, f3 &T, float &etaPath,
                                                 f3 &x, f3 &wo) {
    const mer_scene_desc &S = P.sc;
    const float u1 = rng.next1D(); (void) rng.next1D();          // only sample.x is used (:196)
    x = ro + rd * t;
    const f3 n = shape_normal_b<BND>(P, x);
    const float cosI = dot(-rd, n);                              // Frame::cosTheta(wi), wi = -ray.d
    float etaB = S.rif_const;
    if (CURVED) {
        f3 q = x; f3 g; CellCache cc; cc.reset();
        if (RIF != RIFK_ACOUSTIC && !P.rif.affine) {             // the analytic field has no grid to stay inside of; a transformed one clamps its cell
            q.x = fminf(fmaxf(q.x, P.rif.bmin[0]), P.rif.bmax[0]); q.y = fminf(fmaxf(q.y, P.rif.bmin[1]), P.rif.bmax[1]);
            q.z = fminf(fmaxf(q.z, P.rif.bmin[2]), P.rif.bmax[2]);
        }
        rif_value_grad<RIF>(P.rif, cc, q, etaB, g);
    }
    const float invEtaB = 1 / etaB;
    float cosT; const float F = fresnel_dielectric_ext(cosI, cosT, etaB);
    if (u1 <= F) { wo = rd + n * (2 * cosI); return from_inside; }                   // reflect(wi): 2 (wi.n) n - wi
    const float scale = -(cosT < 0 ? invEtaB : etaB);                                // refract (:121-126)
    const f3 wi = -rd;
    wo = (wi - n * cosI) * scale + n * cosT;
    const float factor = cosT < 0 ? invEtaB : etaB;                                  // solid-angle compression (:213-216)
    T = T * (factor * factor);
    etaPath *= (cosT < 0 ? etaB : invEtaB);                                          // bRec.eta (:208)
    return cosT < 0;
}

// ---- emitter `area` on a `rectangle` shape (src/emitters/area.cpp:67-187, src/shapes/rectangle.cpp:99-222, src/librender/shape.cpp:102-126)
// Rectangle::rayIntersect (:125-148): t in [mint, maxt] or -1
__device__ __forceinline__ float rect_intersect(const Params &P, f3 o, f3 d, float mint, float maxt) {
    const float *W = P.rect_w2o;
    const float oz = W[8] * o.x + W[9] * o.y + W[10] * o.z + W[11], dz = W[8] * d.x + W[9] * d.y + W[10] * d.z;
    const float hit = -oz / dz;
    if (!(hit >= mint && hit <= maxt)) return -1.0f;
    const float lx = (W[0] * o.x + W[1] * o.y + W[2] * o.z + W[3]) + hit * (W[0] * d.x + W[1] * d.y + W[2] * d.z),
                ly = (W[4] * o.x + W[5] * o.y + W[6] * o.z + W[7]) + hit * (W[4] * d.x + W[5] * d.y + W[6] * d.z);
    return (fabsf(lx) <= 1 && fabsf(ly) <= 1) ? hit : -1.0f;
}
// AreaLight::eval (area.cpp:102-107): the radiance a ray travelling along d picks up on the rectangle (one-sided)
__device__ __forceinline__ f3 rect_le(const Params &P, f3 d) {
    const f3 n(P.rect_n[0], P.rect_n[1], P.rect_n[2]);
    return dot(n, -d) <= 0 ? f3(0, 0, 0) : f3(P.sc.area_radiance[0], P.sc.area_radiance[1], P.sc.area_radiance[2]);
}
// Shape::sampleDirect + AreaLight::sampleDirect (shape.cpp:102-115, area.cpp:162-177) for a reference point inside a medium (refN = 0):
// radiance / pdf (0 on the back side), direction, distance, solid-angle pdf
__device__ __forceinline__ f3 rect_sample_direct(const Params &P, f3 ref, float sx, float sy, f3 &d, float &dist, float &pdf) {
    const float *M = P.rect_o2w; const float lx = sx * 2 - 1, ly = sy * 2 - 1;
    const f3 p(M[0] * lx + M[1] * ly + M[3], M[4] * lx + M[5] * ly + M[7], M[8] * lx + M[9] * ly + M[11]);
    const f3 n(P.rect_n[0], P.rect_n[1], P.rect_n[2]);
    d = p - ref;
    const float distSquared = dot(d, d);
    dist = sqrtf(distSquared);
    d = d / dist;
    const float dp = fabsf(dot(d, n));
    pdf = P.rect_inv_area * (dp != 0 ? (distSquared / dp) : 0.0f);
    if (dot(d, n) < 0 && pdf != 0) return f3(P.sc.area_radiance[0], P.sc.area_radiance[1], P.sc.area_radiance[2]) / pdf;
    pdf = 0.0f;
    return f3(0, 0, 0);
}
// AreaLight::pdfDirect (area.cpp:179-187) for a hit at distance dist along d
__device__ __forceinline__ float rect_pdf_direct(const Params &P, f3 d, float dist) {
    const f3 n(P.rect_n[0], P.rect_n[1], P.rect_n[2]);
    return dot(d, n) < 0 ? P.rect_inv_area * (dist * dist) / fabsf(dot(d, n)) : 0.0f;
}
// what a ray sees that has left the convex medium shape for good (or never meets it): the rectangle if it is hit -- front side: its radiance, back
// side: black, and either way it hides the environment (all-absorbing BSDF, shape.cpp:48-56) -- else the environment.  extra = the optical length of
// the free-space leg to the rectangle (transient films).  AREA is a compile-time switch: the plain kernels carry none of this.
template <bool AREA>
__device__ __forceinline__ f3 escape_radiance(const Params &P, f3 env, f3 o, f3 d, float mint, float &extra) {
    extra = 0.0f;
    if (AREA && P.has_area) {
        const float t = rect_intersect(P, o, d, mint, MER_INF);
        if (t >= 0) { extra = t * P.sc.rif_const; return rect_le(P, d); }
    }
    return env;
}

// ImageBlock::put (include/mitsuba/render/imageblock.h:124-205) with one block = the whole image;
// accumulation by float atomics (replaces film->put under a mutex, renderproc.cpp:142-149)
// what = 1: RGB into frame `bin`; what = 2: alpha + weight; what = 3: both (steady state: bin 0)
__device__ __forceinline__ void film_splat(const Params &P, float px, float py, f3 L, float alpha, int bin, int what) {
    const float temp[5] = {L.x, L.y, L.z, alpha, 1.0f};
#pragma unroll
    for (int i = 0; i < 5; ++i) if (!isfinite(temp[i])) return;          // imageblock.h:148-152
    const int W = P.sc.width, H = P.sc.height;
    const float posx = px - 0.5f, posy = py - 0.5f, r = P.fradius;
    const int minx = max((int) ceilf(posx - r), 0), miny = max((int) ceilf(posy - r), 0),
              maxx = min((int) floorf(posx + r), W - 1), maxy = min((int) floorf(posy + r), H - 1);
    for (int y = miny; y <= maxy; ++y) {
        const float wy = P.ftable[min((int) fabsf(((float) y - posy) * P.fscale), 31)];   // rfilter.h:76-77
        for (int x = minx; x <= maxx; ++x) {
            const float wx = P.ftable[min((int) fabsf(((float) x - posx) * P.fscale), 31)];
            const float weight = wx * wy;
            float *dest = P.film + MER_CHK(P.chk, CHK_FILM, ((size_t) y * W + x) * P.film_ch, P.n_film - (uint32_t) P.film_ch + 1u);
            if (what & 1) {
#pragma unroll
                for (int k = 0; k < 3; ++k) atomicAdd(dest + bin * 3 + k, weight * temp[k]);
            }
            if (what & 2) { atomicAdd(dest + P.film_ch - 2, weight * temp[3]); atomicAdd(dest + P.film_ch - 1, weight * temp[4]); }
        }
    }
}
__device__ __forceinline__ void film_put(const Params &P, float px, float py, f3 L, float alpha) {
    film_splat(P, px, py, L, alpha, 0, (P.sc.decomposition && !P.sc.modulation) ? 2 : 3);     // transient: the RGB went out per contribution
}
// PathLengthSampler::mSeq / correlationFunction (include/mitsuba/render/pathlengthsampler.h:32-42,
// src/librender/pathlengthsampler.cpp:68-114), float / double mix as written there
// The function is out of line (inlined at every contribution site it bloats K_event), so it takes what it needs BY VALUE: a reference to
// the kernel-argument struct handed to an out-of-line function forces the compiler to copy the whole 2.3 KB struct into per-lane scratch at
// kernel entry and to read every Params field from scratch afterwards (K_connect and the EXTRA K_event carried 2.3 KB of scratch for it).
struct ModDesc { int modulation, mP, neighbors; float lambda, phase; };
__device__ __forceinline__ ModDesc mod_desc(const Params &P) { return ModDesc{P.sc.modulation, P.sc.mod_P, P.sc.mod_neighbors, P.sc.mod_lambda, P.mod_phase}; }
__device__ __forceinline__ float mseq(float lambda, int mP, float t, float phase) {
    float pathLength = t;
    pathLength = pathLength + phase * lambda * MER_INV_PI / 2;
    pathLength = fmodf(pathLength, lambda);
    if (pathLength < lambda / mP) return 1 - pathLength * (mP - 1) / lambda;
    else if (pathLength > (1 - 1.0 / mP) * lambda) return 1 - (lambda - pathLength) * (mP - 1) / lambda;
    else return (float) (1.0 / mP);
}
static __device__ __noinline__ float correlation_function(const ModDesc m, float t) {
    const float lambda = m.lambda, modPhase = m.phase;
    float pathLength = t;
    switch (m.modulation) {
    case MER_MODULATION_SINE: pathLength = pathLength + modPhase * lambda * MER_INV_PI / 2; return (float) cos(pathLength * 2 * M_PI / lambda);
    case MER_MODULATION_SQUARE: pathLength = pathLength + modPhase * lambda * MER_INV_PI / 2;
        return 4 / lambda * (fabsf(fmodf(pathLength, lambda) - lambda / 2) - lambda / 4);
    case MER_MODULATION_HAMILTONIAN: pathLength = pathLength + modPhase * lambda * MER_INV_PI / 2;
        pathLength = fmodf(pathLength, lambda);
        if (pathLength < lambda / 6) return 6 * pathLength / lambda;
        else if (pathLength < lambda / 2 && pathLength >= lambda / 6) return 1.0f;
        else if (pathLength < 2 * lambda / 3 && pathLength >= lambda / 2) return 1 - (pathLength - lambda / 2) * 6 / lambda;
        else return 0;
    case MER_MODULATION_MSEQ: return mseq(lambda, m.mP, pathLength, modPhase);
    case MER_MODULATION_DEPTHSELECTIVE: { float value = 0;
        for (int i = 0; i < m.neighbors; i++) value += mseq(lambda, m.mP, pathLength, (float) (modPhase - i * (2 * M_PI) / m.mP));
        value -= (float) (m.neighbors - 1) / m.mP;
        return value; }
    }
    return 1.0f;
}
// a radiance contribution as it enters the per-path sum: weighted by the correlation function under a modulation (bdpt_proc.cpp:446-447)
// MOD is a compile-time switch: the out-of-line correlation function must not appear in the kernels of unmodulated renders (a call
// site costs them their register allocation: K_event went from 50 to 296 ms per bench step with it)
template <bool MOD>
__device__ __forceinline__ f3 mod_weight(const Params &P, f3 value, float pathLength) {
    if (!MOD) return value;
    return P.sc.modulation ? value * correlation_function(mod_desc(P), pathLength) : value;
}
// Transient film: one radiance contribution binned by its optical path length (bdpt_proc.cpp:449-470)
__device__ __forceinline__ void film_contribute(const Params &P, float px, float py, f3 value, float pathLength) {
    if (!P.sc.decomposition || P.sc.modulation || P.path_out || is_zero(value)) return;
    const float b = floorf((pathLength - P.sc.min_bound) / P.sc.bin_width);
    if (!(b >= 0.0f) || !(b < (float) P.frames)) return;
    film_splat(P, px, py, value, 0.0f, (int) b, 1);
}

// ---------------------------------------------------------------------------------------------------
// Work decode: w -> (pixel, sample).  Sample-major; inside a pass pixels go by 32x32 image tiles (the
// reference's block size, src/mitsuba/mitsuba.cpp:80-81) and by 8x8 sub-tiles so that the 64 lanes of a
// fresh wavefront start on one 8x8 pixel patch (coherent camera rays, distinct film pixels per lane).
// Tile dealing (mer_shard.tile_rank / tile_count): the tiles are dealt round-robin in row-major order AFTER row ty has been
// rotated by tile_skew * ty columns.  A plain deal hands rank r whole tile COLUMNS whenever tile_count divides tiles_x (16 or 32
// columns, 8 ranks): the columns through the medium carry all the work, the outer ones none (measured: profiles/round3/
// shard_balance_*.txt).  With the rotation (a skew coprime to tiles_x) a rank's tiles lie on diagonals and visit every column
// and every row equally often -- the role of the reference's spiral block order (src/librender/renderproc.cpp:79).  A
// permutation inside each row: the shards still partition the tiles exactly.
#define MER_TILE 32
__device__ __forceinline__ bool decode_work(const Params &P, uint64_t w, int &x, int &y, uint32_t &sample) {
    const uint32_t npix = (uint32_t) P.ntiles_mine * (MER_TILE * MER_TILE);
    const uint32_t s_local = (uint32_t) (w / npix);
    const uint32_t r = (uint32_t) (w - (uint64_t) s_local * npix);
    const uint32_t tile_local = r >> 10, q = r & 1023u, sub = q >> 6, lane = q & 63u;
    const uint32_t tile = (uint32_t) P.tile_rank + tile_local * (uint32_t) P.tile_count;
    const uint32_t ty = tile / (uint32_t) P.tiles_x, tx = (tile - ty * (uint32_t) P.tiles_x + (uint32_t) P.tile_skew * ty) % (uint32_t) P.tiles_x;
    x = (int) (tx * MER_TILE + (sub & 3u) * 8u + (lane & 7u));
    y = (int) (ty * MER_TILE + (sub >> 2) * 8u + (lane >> 3));
    sample = (uint32_t) P.spp_begin + s_local * (uint32_t) P.spp_stride;
    return x < P.sc.width && y < P.sc.height;
}

// what one path edge adds to the quantity a decomposed film bins by: its optical length (transient) or 1 (bounce: bdpt_proc.cpp:179-187)
__device__ __forceinline__ float edge_length(const Params &P, float optical_length) {
    return P.sc.decomposition == MER_DECOMPOSITION_BOUNCE ? 1.0f : optical_length;
}

// `calibratedTransient` drops the camera edge from a TRANSIENT film's path length (bdpt_proc.cpp:163-170: startIndex 3); a BOUNCE film counts
// every edge whatever the flag says (:179-187 loops from i = 2 unconditionally)
__device__ __forceinline__ bool camera_edge_counts(const Params &P) {
    return !(P.sc.calibrated_transient && P.sc.decomposition == MER_DECOMPOSITION_TRANSIENT);
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

}  // namespace mer
