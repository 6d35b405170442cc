// mer_kernels.hpp -- gfx950 kernels behind the leaf entry points (parity tests), grid re-layout, B-spline prefilter
// (K_prefilter), synthetic fields.  The render kernels are in mer_wavefront.hpp.
#pragma once
#include "mer_walk.hpp"
#include "mer_connect.hpp"

namespace mer {

// ---------------------------------------------------------------------------------------------------
// Leaf kernels (parity entry points).  One thread per item; divergence is irrelevant here.
__global__ void lookup_trilinear_kernel(DGrid g, const float *pts, int64_t n, float *out_val, int32_t *out_idx) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int idx[4];
    out_val[i] = lookup_float(g, f3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]), idx);
    if (out_idx) { out_idx[4 * i] = idx[0]; out_idx[4 * i + 1] = idx[1]; out_idx[4 * i + 2] = idx[2]; out_idx[4 * i + 3] = idx[3]; }
}
__global__ void lookup_rgb_kernel(DGrid g, const float *pts, int64_t n, float *out) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const f3 v = lookup_spectrum(g, f3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]));
    out[3 * i] = v.x; out[3 * i + 1] = v.y; out[3 * i + 2] = v.z;
}
__global__ void rif_value_grad_kernel(DGrid g, int interp, const float *pts, int64_t n, float *val, float *grad) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v; f3 gr;
    f3 p(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
    if (g.affine) p = to_volume(g, p);                    // a `toWorld` on the volume plugin: splinevolume.cpp:320-376
    CellCache cc; cc.reset();
    if (interp == MER_RIF_BSPLINE3) bspline_value_grad(g, p, v, gr);
    else if (g.layout == MER_LAYOUT_BRICK27 || g.layout == MER_LAYOUT_BRICK125) { if (g.buf_bytes) trilinear_value_grad<RIFK_BRICK27_BUF>(g, cc, p, v, gr); else trilinear_value_grad<RIFK_BRICK27>(g, cc, p, v, gr); }
    else if (g.layout == MER_LAYOUT_CELL8) { if (g.buf_bytes) trilinear_value_grad<RIFK_CELL8_BUF>(g, cc, p, v, gr); else trilinear_value_grad<RIFK_CELL8>(g, cc, p, v, gr); }
    else { if (g.buf_bytes) trilinear_value_grad<RIFK_DENSE_BUF>(g, cc, p, v, gr); else trilinear_value_grad<MER_RIF_TRILINEAR>(g, cc, p, v, gr); }
    if (g.affine) gr = rot_t(g, gr);
    val[i] = v; grad[3 * i] = gr.x; grad[3 * i + 1] = gr.y; grad[3 * i + 2] = gr.z;
}

template <int RIF, int STEPPER>
__global__ void er_trace_kernel(const Params P, const float *p0, const float *d0, const float *dist, int64_t n,
                                float *out_p, float *out_v, float *out_ds, float *out_opt, int32_t *out_ok) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Walk<true, RIF, STEPPER, MER_SIGMA_HOMOGENEOUS> W;
    LaneCounters C; C.clear();
    W.kind = K_FREE; W.trsum = 0; W.walk = 0; W.Tr = 1; W.dist = 0; W.opt = 0; W.sdens = 0; W.t = 0; W.tmin = 0; W.tmax = 0;
    W.backstep = 0; W.hprev = 0; W.cc.reset();
    W.p = f3(p0[3 * i], p0[3 * i + 1], p0[3 * i + 2]);
    const f3 d(d0[3 * i], d0[3 * i + 1], d0[3 * i + 2]);
    float n0; f3 g;
    rif_value_grad<RIF>(P.rif, W.cc, W.p, n0, g);
    W.n0 = n0; W.v = d * n0;
    if (isfinite(dist[i])) W.set_segment(P, dist[i]);
    else { W.seg_inf = 1; W.steps_left = 100000; W.rem = 0.0f; }
    Rng rng; rng.state = 0; rng.inc = 1;
    int ev = EV_NONE;
    while (ev == EV_NONE) ev = W.advance(P, rng, C);
    out_p[3 * i] = W.p.x; out_p[3 * i + 1] = W.p.y; out_p[3 * i + 2] = W.p.z;
    out_v[3 * i] = W.v.x; out_v[3 * i + 1] = W.v.y; out_v[3 * i + 2] = W.v.z;
    out_ds[i] = W.dist; out_opt[i] = W.opt; out_ok[i] = (ev == EV_ARRIVED) ? 1 : 0;
}

// Medium::sampleDistance for item i with RNG stream (seed, pixel=i, sample=0); rec stride 20
template <bool CURVED, int RIF, int STEPPER, int SIGMA>
__global__ void sample_distance_kernel(const Params P, const float *o, const float *d, const float *maxt, int64_t n, float *rec) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Walk<CURVED, RIF, STEPPER, SIGMA> W;
    LaneCounters C; C.clear();
    Rng rng; rng.seed(P.seed, (uint32_t) i, 0);
    const f3 oo(o[3 * i], o[3 * i + 1], o[3 * i + 2]), dd(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
    W.t = 0; W.tmin = 0; W.tmax = 0; W.n0 = 1; W.rem = 0; W.steps_left = 0; W.seg_inf = 0; W.sdens = 0; W.backstep = 0; W.hprev = 0; W.cc.reset();
    int ev = W.begin(P, rng, C, K_FREE, oo, dd, maxt[i]);
    float sigma = 0.0f;
    for (;;) {
        if (ev == EV_NONE) ev = W.advance(P, rng, C);
        else if (ev == EV_ARRIVED) ev = W.on_arrived(P, rng, C, sigma);
        else if (ev == EV_EXITED) ev = EV_FAIL;
        else break;
    }
    float *r = rec + 20 * i;
    MRec m;
    bool success = ev == EV_REAL;
    if (ev == EV_GATE_FAIL) {
        m.p = oo; m.d = dd; m.t = 0; m.sigmaS = f3(0, 0, 0); m.transmittance = f3(0, 0, 0);
        m.pdfSuccess = 1; m.pdfFailure = 1; m.refRatioSq = 1; success = false;
        if (SIGMA == MER_SIGMA_GRID) m.transmittance = f3(0, 0, 0);
    } else {
        finish_free_flight(P, C, W, success, sigma, m);
        if (SIGMA == MER_SIGMA_HOMOGENEOUS && success && m.p.x == oo.x && m.p.y == oo.y && m.p.z == oo.z) success = false;
    }
    r[0] = success ? 1.0f : 0.0f; r[1] = m.t; r[2] = m.p.x; r[3] = m.p.y; r[4] = m.p.z;
    r[5] = success ? m.sigmaS.x : 0.0f; r[6] = success ? m.sigmaS.y : 0.0f; r[7] = success ? m.sigmaS.z : 0.0f;
    r[8] = m.transmittance.x; r[9] = m.transmittance.y; r[10] = m.transmittance.z;
    r[11] = m.pdfSuccess; r[12] = m.pdfFailure; r[13] = m.refRatioSq; r[14] = m.d.x; r[15] = m.d.y; r[16] = m.d.z;
    r[17] = r[18] = r[19] = 0.0f;
}

// Medium::evalTransmittance over [0,maxt] (straight) or to the boundary (curved)
template <bool CURVED, int RIF, int STEPPER, int SIGMA>
__global__ void eval_transmittance_kernel(const Params P, const float *o, const float *d, const float *maxt, int64_t n, float *out) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Walk<CURVED, RIF, STEPPER, SIGMA> W;
    LaneCounters C; C.clear();
    Rng rng; rng.seed(P.seed, (uint32_t) i, 0);
    const f3 oo(o[3 * i], o[3 * i + 1], o[3 * i + 2]), dd(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
    const int nwalks = (SIGMA == MER_SIGMA_GRID && P.sc.tr_estimator == MER_TR_WOODCOCK2) ? 2 : 1;
    W.t = 0; W.tmin = 0; W.tmax = 0; W.n0 = 1; W.rem = 0; W.steps_left = 0; W.seg_inf = 0; W.sdens = 0; W.backstep = 0; W.hprev = 0; W.cc.reset();
    int ev = W.begin(P, rng, C, K_NEE, oo, dd, maxt[i]);
    float sigma = 0.0f; f3 tr(1, 1, 1);
    bool gate = false, closed = (ev == EV_TR_DONE);
    for (;;) {
        if (ev == EV_NONE) ev = W.advance(P, rng, C);
        else if (ev == EV_ARRIVED) ev = W.on_arrived(P, rng, C, sigma);
        else if (ev == EV_EXITED) ev = EV_WALK_END;
        else if (ev == EV_GATE_FAIL) { gate = true; break; }
        else if (ev == EV_WALK_END) {
            W.trsum += W.Tr; W.walk++;
            if (W.walk < nwalks) ev = W.begin(P, rng, C, K_NEE, oo, dd, maxt[i], false); else break;
        } else break;
    }
    if (gate) tr = f3(0, 0, 0);
    else if (SIGMA == MER_SIGMA_GRID) { const float v = closed ? 1.0f : W.trsum / (float) nwalks; tr = f3(v, v, v); }
    else if (CURVED) tr = homogeneous_transmittance(P, -W.dist);
    else tr = homogeneous_transmittance(P, 0.0f - maxt[i]);
    out[3 * i] = tr.x; out[3 * i + 1] = tr.y; out[3 * i + 2] = tr.z;
}

__global__ void phase_sample_kernel(int kind, float g, const float *wi, const float *u2, int64_t n, float *wo, float *pdf) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    f3 o; float p;
    phase_sample(kind, g, f3(wi[3 * i], wi[3 * i + 1], wi[3 * i + 2]), u2[2 * i], u2[2 * i + 1], o, p);
    wo[3 * i] = o.x; wo[3 * i + 1] = o.y; wo[3 * i + 2] = o.z; pdf[i] = p;
}
__global__ void phase_eval_kernel(int kind, float g, const float *wi, const float *wo, int64_t n, float *val) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    val[i] = phase_eval(kind, g, f3(wi[3 * i], wi[3 * i + 1], wi[3 * i + 2]), f3(wo[3 * i], wo[3 * i + 1], wo[3 * i + 2]));
}
__global__ void camera_rays_kernel(const Params P, const float *pos2, int64_t n, float *o, float *d) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    f3 oo, dd; float a, b;
    sample_ray(P, pos2[2 * i], pos2[2 * i + 1], oo, dd, a, b);
    o[3 * i] = oo.x; o[3 * i + 1] = oo.y; o[3 * i + 2] = oo.z; d[3 * i] = dd.x; d[3 * i + 1] = dd.y; d[3 * i + 2] = dd.z;
}
__global__ void correlation_kernel(const Params P, const float *t, int64_t n, float *out) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = correlation_function(mod_desc(P), t[i]);
}
__global__ void rng_kernel(uint64_t seed, uint32_t pixel, uint32_t sample, int n, float *out) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    Rng r; r.seed(seed, pixel, sample);
    for (int i = 0; i < n; i++) out[i] = r.next1D();
}

// ---------------------------------------------------------------------------------------------------
// Grid re-layout DENSE -> CELL8: the 8 corners of every cell stored contiguously (32 B, one sector):
// a trilinear fetch becomes two 16-B loads from one cache line instead of eight 4-B loads from four.
// The integer index contract stays (x,y,z); the cell address is a pure function of it.
__global__ void relayout_cell8_kernel(const float *dense, float *cell8, int rx, int ry, int rz) {
    const int64_t ncell = (int64_t) (rx - 1) * (ry - 1) * (rz - 1);
    for (int64_t c = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; c < ncell; c += (int64_t) gridDim.x * blockDim.x) {
        const int x = (int) (c % (rx - 1)), y = (int) ((c / (rx - 1)) % (ry - 1)), z = (int) (c / ((int64_t) (rx - 1) * (ry - 1)));
        const int64_t base = ((int64_t) z * ry + y) * rx + x, sy = rx, sz = (int64_t) rx * ry;
        float4 a, b;
        a.x = dense[base]; a.y = dense[base + 1]; a.z = dense[base + sy]; a.w = dense[base + sy + 1];
        b.x = dense[base + sz]; b.y = dense[base + sz + 1]; b.z = dense[base + sz + sy]; b.w = dense[base + sz + sy + 1];
        float4 *dst = (float4 *) (cell8 + c * 8);
        dst[0] = a; dst[1] = b;
    }
}

// Grid re-layout DENSE -> BRICK27: the 3x3x3 corners of every 2x2x2-cell brick in one 128-byte record (27 words + 5 of padding).  A ray
// crosses a brick face half as often as a cell face, and a cell change inside the brick is served from registers: half the memory
// requests of CELL8 and half its footprint.  Corners past the last node (odd cell counts) replicate the last node and are never selected.
__global__ void relayout_brick_kernel(const float *dense, float *rec, int rx, int ry, int rz, int nbx, int nby, int nbz, int bshift, int recw) {
    const int64_t nbrick = (int64_t) nbx * nby * nbz; const int bw = (1 << bshift) + 1;
    for (int64_t c = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; c < nbrick; c += (int64_t) gridDim.x * blockDim.x) {
        const int bx = (int) (c % nbx), by = (int) ((c / nbx) % nby), bz = (int) (c / ((int64_t) nbx * nby));
        float *q = rec + c * recw;
        for (int dz = 0; dz < bw; dz++) for (int dy = 0; dy < bw; dy++) for (int dx = 0; dx < bw; dx++) {
            const int x = min((bx << bshift) + dx, rx - 1), y = min((by << bshift) + dy, ry - 1), z = min((bz << bshift) + dz, rz - 1);
            q[(dz * bw + dy) * bw + dx] = dense[((int64_t) z * ry + y) * rx + x];
        }
        for (int k = bw * bw * bw; k < recw; k++) q[k] = 0.0f;
    }
}

// ---------------------------------------------------------------------------------------------------
// K_prefilter: cubic-B-spline coefficients, Spline<3>::build1d / build3d
// (include/mitsuba/core/basisspline.h:812-890): per line a causal + anti-causal 1-pole IIR with pole
// z1 = sqrt(3)-2 and the full mirror-sum initialisation, applied along y, then x, then z.  One thread per line.
__global__ void bspline_pass_kernel(const float *src, float *dst, int nlines_a, int nlines_b, int64_t stride_a, int64_t stride_b,
                                    int64_t stride_line, int size) {
    const int64_t id = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (int64_t) nlines_a * nlines_b) return;
    const int64_t a = id % nlines_a, b = id / nlines_a;
    const int64_t offset = a * stride_a + b * stride_b;
    const float z1 = -2.0f + sqrtf(3.0f);
    // cp[0]: the reference evaluates pow(z1, i) in double (float,int overload promotes) and sums in float
    float cp0 = 0.0f;
    double zp = 1.0;
    for (int i = 0; i < size; i++) { cp0 += (float) ((double) src[offset + i * stride_line] * zp); zp *= (double) z1; }
    for (int i = size - 2; i > 0; i--) cp0 += (float) ((double) src[offset + i * stride_line] * pow((double) z1, (double) (2 * size - 2 - i)));
    cp0 = (float) ((double) cp0 / (1.0 - pow((double) z1, (double) (2 * size - 2))));
    // causal pass written into dst, then the anti-causal pass in place
    float prev = cp0;
    dst[offset] = cp0;
    float cp_last2 = cp0;
    for (int i = 1; i < size; i++) {
        const float cur = src[offset + i * stride_line] + z1 * prev;
        cp_last2 = prev; prev = cur;
        dst[offset + i * stride_line] = cur;
    }
    float cn = z1 / (z1 * z1 - 1) * (prev + z1 * cp_last2);
    float cpi = prev;
    dst[offset + (int64_t) (size - 1) * stride_line] = 6 * cn;
    for (int i = size - 2; i >= 0; i--) {
        cpi = dst[offset + i * stride_line];
        cn = z1 * (cn - cpi);
        dst[offset + i * stride_line] = 6 * cn;
    }
}

// K_prefilter, parallel form.  The filter's pole is z1 = sqrt(3)-2 = -0.268: a coefficient depends on samples k positions away with
// weight z1^k, below float resolution (2^-24 relative) after 13 samples: a warm-up of 16 samples (7e-10) is exact in float.  Rounds 1-2 used
// 40 samples and segments of 32 outputs -- 3.5 reads per output; 16 + 64 + 16 makes it 1.5 (1024^3: 13.4 -> 8.7 ms with the wave-uniform
// addressing of bspline_win2_kernel, profiles/round3/prefilter_variants.txt).  So a line is cut into segments that
// are filtered independently after a warm-up of MER_PF_WARM samples (the first segment starts from the exact mirror sum, the last
// anti-causal one from the exact end condition): same arithmetic per sample as the sequential recursion, N^3 / SEG threads instead
// of N^2, every global access coalesced.  Two kernels per axis (causal -> tmp, anti-causal -> out): the anti-causal warm-up of one
// segment reads causal values that a neighbouring segment would otherwise already have overwritten.
#ifndef MER_PF_WARM
#define MER_PF_WARM 16
#endif
#ifndef MER_PF_SEG
#define MER_PF_SEG 64
#endif
__device__ __forceinline__ float bspline_cp0(const float *src, int64_t offset, int64_t stride_line, int size, float z1) {
    // basisspline.h:826-838: pow(z1, i) in double, sum in float; terms beyond z1^64 (< 1e-36) cannot change a float sum
    const int nf = size < 64 ? size : 64;
    float cp0 = 0.0f; double zp = 1.0;
    for (int i = 0; i < nf; i++) { cp0 += (float) ((double) src[offset + i * stride_line] * zp); zp *= (double) z1; }
    if (size <= 64) {
        for (int i = size - 2; i > 0; i--) cp0 += (float) ((double) src[offset + i * stride_line] * pow((double) z1, (double) (2 * size - 2 - i)));
        cp0 = (float) ((double) cp0 / (1.0 - pow((double) z1, (double) (2 * size - 2))));
    }
    return cp0;
}
// thread <-> (a, segment, b); a is the fastest index: pass stride_a = 1 (y and z passes) for coalesced accesses
__global__ void bspline_causal_kernel(const float *src, float *tmp, int na, int nb, int64_t stride_a, int64_t stride_b, int64_t stride_line, int size) {
    const int nseg = (size + MER_PF_SEG - 1) / MER_PF_SEG;
    const int64_t id = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (int64_t) na * nseg * nb) return;
    const int64_t a = id % na, r = id / na; const int seg = (int) (r % nseg); const int64_t b = r / nseg;
    const int64_t offset = a * stride_a + b * stride_b;
    const float z1 = -2.0f + sqrtf(3.0f);
    const int i0 = seg * MER_PF_SEG, i1 = min(i0 + MER_PF_SEG, size);
    int w0 = max(i0 - MER_PF_WARM, 0);
    float c;
    if (w0 == 0) { c = bspline_cp0(src, offset, stride_line, size, z1); if (i0 == 0) tmp[offset] = c; }
    else c = src[offset + (int64_t) w0 * stride_line];
    for (int i = w0 + 1; i < i0; i++) c = src[offset + (int64_t) i * stride_line] + z1 * c;
    for (int i = max(i0, 1); i < i1; i++) { c = src[offset + (int64_t) i * stride_line] + z1 * c; tmp[offset + (int64_t) i * stride_line] = c; }
}
__global__ void bspline_anticausal_kernel(const float *tmp, float *out, int na, int nb, int64_t stride_a, int64_t stride_b, int64_t stride_line, int size) {
    const int nseg = (size + MER_PF_SEG - 1) / MER_PF_SEG;
    const int64_t id = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (int64_t) na * nseg * nb) return;
    const int64_t a = id % na, r = id / na; const int seg = (int) (r % nseg); const int64_t b = r / nseg;
    const int64_t offset = a * stride_a + b * stride_b;
    const float z1 = -2.0f + sqrtf(3.0f);
    const int i0 = seg * MER_PF_SEG, i1 = min(i0 + MER_PF_SEG, size);
    const int w1 = min(i1 + MER_PF_WARM, size);
    float cn; int i;
    if (w1 == size) {            // basisspline.h:850-853: exact end condition
        cn = z1 / (z1 * z1 - 1) * (tmp[offset + (int64_t) (size - 1) * stride_line] + z1 * tmp[offset + (int64_t) (size - 2) * stride_line]);
        if (i1 == size) out[offset + (int64_t) (size - 1) * stride_line] = 6 * cn;
        i = size - 2;
    } else { cn = 0.0f; i = w1 - 1; }
    for (; i >= i1; i--) cn = z1 * (cn - tmp[offset + (int64_t) i * stride_line]);
    for (; i >= i0; i--) { cn = z1 * (cn - tmp[offset + (int64_t) i * stride_line]); out[offset + (int64_t) i * stride_line] = 6 * cn; }
}

// The pass along x (lines contiguous in memory), any line length: a block stages 256 lines x (SEG outputs + WARM warm-up samples on either side) in LDS
// with row-contiguous (coalesced) loads, each thread filters one line of the tile causally and anti-causally in place (odd row pitch:
// conflict-free), and the 32 outputs per line go back with coalesced stores -- one read and one write of the volume.
#define MER_PFX_ROWS 256
#define MER_PFX_COLS MER_PF_SEG
#define MER_PFX_W (MER_PFX_COLS + 2 * MER_PF_WARM)
#define MER_PFX_LD (MER_PFX_W + 1)
// tile_lo / tile_hi: the kernel covers the column tiles [0, tile_lo) and [tile_hi, ntile) only (the border tiles beside bspline_x_reg_kernel's
// interior segments); tile_lo = tile_hi = ntile: all of them
__global__ void __launch_bounds__(256) bspline_x_kernel(const float *src, float *out, int64_t nlines, int n, int tile_lo, int tile_hi) {
    __shared__ float lds[MER_PFX_ROWS * MER_PFX_LD];
    const int ntile_all = (n + MER_PFX_COLS - 1) / MER_PFX_COLS, ntile = tile_lo + (ntile_all - tile_hi);
    const int tidx = (int) (blockIdx.x % (unsigned) ntile);              // neighbouring column tiles run together: the warm-up overlap is an L2 hit
    const int tile = tidx < tile_lo ? tidx : tile_hi + (tidx - tile_lo);
    const int64_t row0 = (int64_t) (blockIdx.x / (unsigned) ntile) * MER_PFX_ROWS;
    const int c0 = tile * MER_PFX_COLS, c1 = min(c0 + MER_PFX_COLS, n);
    const int lo = max(c0 - MER_PF_WARM, 0), hi = min(c1 + MER_PF_WARM, n), w = hi - lo;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int r = wave; r < MER_PFX_ROWS; r += 4) {
        const int64_t row = row0 + r;
        if (row < nlines) for (int i = lane; i < w; i += 64) lds[r * MER_PFX_LD + i] = src[row * n + lo + i];
    }
    __syncthreads();
    const float z1 = -2.0f + sqrtf(3.0f);
    if (row0 + threadIdx.x < nlines) {
        float *L = lds + threadIdx.x * MER_PFX_LD;
        float c = (lo == 0) ? bspline_cp0(src + (row0 + threadIdx.x) * n, 0, 1, n, z1) : L[0];     // the mirror sum reads up to 64 samples: from the line itself (the tile may hold fewer)
        L[0] = c;
        for (int i = 1; i < w; i++) { c = L[i] + z1 * c; L[i] = c; }
        float cn; int i;
        if (hi == n) { cn = z1 / (z1 * z1 - 1) * (L[w - 1] + z1 * L[w - 2]); if (c1 == n) L[w - 1] = 6 * cn; i = w - 2; }
        else { cn = 0.0f; i = w - 1; }
        for (; i >= c1 - lo; i--) cn = z1 * (cn - L[i]);
        for (; i >= c0 - lo; i--) { cn = z1 * (cn - L[i]); L[i] = 6 * cn; }
    }
    __syncthreads();
    for (int r = wave; r < MER_PFX_ROWS; r += 4) {
        const int64_t row = row0 + r;
        if (row < nlines) for (int i = lane; i < c1 - c0; i += 64) out[row * n + c0 + i] = lds[r * MER_PFX_LD + (c0 - lo) + i];
    }
}

// x pass, register form (n % 4 == 0), INTERIOR segments only (the window lies inside the line; the border tiles run in bspline_x_kernel):
// thread <-> (segment, line); the window (WARM + SEG + WARM samples) is fetched with independent 16-byte loads at immediate offsets from one
// address -- adjacent lanes own adjacent segments of one line, so a wave reads one contiguous span and every cache line it touches is shared
// through L1 --, filtered in registers without any per-sample condition, and written with 16-byte stores.
__global__ void __launch_bounds__(256) bspline_x_reg_kernel(const float *src, float *out, int64_t nlines, int n, int seg_first, int nseg) {
    const int64_t id = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nlines * nseg) return;
    const int seg = seg_first + (int) (id % nseg); const int64_t row = id / nseg;
    const int c0 = seg * MER_PF_SEG, g0 = c0 - MER_PF_WARM;
    const float4 *S = (const float4 *) (src + row * n + g0); float4 *O = (float4 *) (out + row * n + c0);
    const float z1 = -2.0f + sqrtf(3.0f);
    float v[MER_PFX_W];
#pragma unroll
    for (int q = 0; q < MER_PFX_W / 4; q++) { const float4 t = S[q]; v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w; }
    float c = v[0];
#pragma unroll
    for (int j = 1; j < MER_PFX_W; j++) { c = v[j] + z1 * c; v[j] = c; }
    float cn = 0.0f;
#pragma unroll
    for (int j = MER_PFX_W - 1; j >= MER_PF_WARM; j--) { cn = z1 * (cn - v[j]); v[j] = 6 * cn; }
#pragma unroll
    for (int q = 0; q < MER_PF_SEG / 4; q++) O[q] = make_float4(v[MER_PF_WARM + 4 * q], v[MER_PF_WARM + 4 * q + 1], v[MER_PF_WARM + 4 * q + 2], v[MER_PF_WARM + 4 * q + 3]);
}

// x pass, register form, the BORDER segments [0, seg_lo) and [seg_hi, nseg) (n % 4 == 0): every sample guarded (line ends, mirror-sum start)
__global__ void __launch_bounds__(256) bspline_x_reg_border_kernel(const float *src, float *out, int64_t nlines, int n, int seg_lo, int seg_hi) {
    const int nseg_all = (n + MER_PF_SEG - 1) / MER_PF_SEG, nseg = seg_lo + (nseg_all - seg_hi);
    const int64_t id = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nlines * nseg) return;
    const int sidx = (int) (id % nseg); const int seg = sidx < seg_lo ? sidx : seg_hi + (sidx - seg_lo); const int64_t row = id / nseg;
    const float *S = src + row * n; float *O = out + row * n;
    const int c0 = seg * MER_PF_SEG, g0 = c0 - MER_PF_WARM;
    const float z1 = -2.0f + sqrtf(3.0f);
    float v[MER_PFX_W];
#pragma unroll
    for (int q = 0; q < MER_PFX_W / 4; q++) {
        const int g = g0 + 4 * q;
        float4 t = make_float4(0, 0, 0, 0);
        if (g >= 0 && g < n) t = *(const float4 *) (S + g);
        v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    }
    float c = 0.0f;
    if (g0 <= 0) c = bspline_cp0(S, 0, 1, n, z1);                  // the window reaches the line start: exact mirror-sum initialisation
#pragma unroll
    for (int j = 0; j < MER_PFX_W; j++) {
        const int g = g0 + j;
        if (g == 0 || (g0 > 0 && j == 0)) { if (g0 > 0) c = v[0]; }     // start: cp0 at the line start, the first sample otherwise
        else if (g > 0 && g < n) c = v[j] + z1 * c;
        v[j] = c;
    }
    float cn = 0.0f;
#pragma unroll
    for (int j = MER_PFX_W - 1; j >= MER_PF_WARM; j--) {
        const int g = g0 + j;
        if (g == n - 1) { cn = z1 / (z1 * z1 - 1) * (v[j] + z1 * v[j - 1]); v[j] = 6 * cn; }    // basisspline.h:850-853 (v[j-1]: n >= 16)
        else if (g < n - 1) { cn = z1 * (cn - v[j]); v[j] = 6 * cn; }
    }
#pragma unroll
    for (int q = 0; q < MER_PF_SEG / 4; q++) {
        const int g = c0 + 4 * q;
        if (g < n) *(float4 *) (O + g) = make_float4(v[MER_PF_WARM + 4 * q], v[MER_PF_WARM + 4 * q + 1], v[MER_PF_WARM + 4 * q + 2], v[MER_PF_WARM + 4 * q + 3]);
    }
}

// y / z passes, register-window form: thread <-> (a = x index, segment, b); the window is strided by the line pitch, every
// load and store is coalesced across the wave (adjacent lanes = adjacent x); causal and anti-causal sweeps fused: one read, one write.
// seg_lo / seg_hi: the kernel covers the segments [0, seg_lo) and [seg_hi, nseg) only (the border segments beside bspline_win2_kernel's
// interior ones); seg_lo = nseg, seg_hi = nseg: all of them
__global__ void __launch_bounds__(256) bspline_win_kernel(const float *src, float *out, int na, int nb, int64_t stride_b, int64_t stride_line, int n, int seg_lo, int seg_hi) {
    const int nseg_all = (n + MER_PF_SEG - 1) / MER_PF_SEG, nseg = seg_lo + (nseg_all - seg_hi);
    const int64_t id = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (int64_t) na * nseg * nb) return;
    const int64_t a = id % na, r = id / na; const int sidx = (int) (r % nseg); const int seg = sidx < seg_lo ? sidx : seg_hi + (sidx - seg_lo); const int64_t b = r / nseg;
    const float *S = src + a + b * stride_b; float *O = out + a + b * stride_b;
    const int c0 = seg * MER_PF_SEG, g0 = c0 - MER_PF_WARM;
    const float z1 = -2.0f + sqrtf(3.0f);
    float v[MER_PFX_W];
#pragma unroll
    for (int j = 0; j < MER_PFX_W; j++) { const int g = g0 + j; v[j] = (g >= 0 && g < n) ? S[(int64_t) g * stride_line] : 0.0f; }
    float c = 0.0f;
    if (g0 <= 0) c = bspline_cp0(S, 0, stride_line, n, z1);
#pragma unroll
    for (int j = 0; j < MER_PFX_W; j++) {
        const int g = g0 + j;
        if (g == 0 || (g0 > 0 && j == 0)) { if (g0 > 0) c = v[0]; }
        else if (g > 0 && g < n) c = v[j] + z1 * c;
        v[j] = c;
    }
    float cn = 0.0f;
#pragma unroll
    for (int j = MER_PFX_W - 1; j >= MER_PF_WARM; j--) {
        const int g = g0 + j;
        if (g == n - 1) { cn = z1 / (z1 * z1 - 1) * (v[j] + z1 * v[j - 1]); v[j] = 6 * cn; }
        else if (g < n - 1) { cn = z1 * (cn - v[j]); v[j] = 6 * cn; }
    }
#pragma unroll
    for (int j = 0; j < MER_PF_SEG; j++) { const int g = c0 + j; if (g < n) O[(int64_t) g * stride_line] = v[MER_PF_WARM + j]; }
}

// y / z passes, register-window form with WAVE-UNIFORM line addresses (option prefilter = 0, the default): blockIdx = (x block, segment, b),
// threadIdx = x.  Segment and b are block-uniform, so the address of sample j is one SCALAR base (advanced by the line pitch with scalar adds)
// plus the lane's 32-bit byte offset (global_load ... v_off, s[base]) -- bspline_win_kernel forms a 64-bit vector address per sample, which
// put 2 x (window) address registers beside the window itself (183 - 256 VGPRs, 1 - 2 waves per SIMD).  Interior segments (the window lies
// inside the line: all but the first and the last one or two) run here, without any per-sample condition; the border segments run in
// bspline_win_kernel.
__global__ void __launch_bounds__(256) bspline_win2_kernel(const float *src, float *out, int na, int64_t stride_b, int64_t stride_line, int n, int seg_first) {
    const int a = (int) (blockIdx.x * 256 + threadIdx.x);
    if (a >= na) return;
    const int seg = seg_first + (int) blockIdx.y;                       // interior segments only (host: g0 > 0 and g0 + window < n)
    const int c0 = seg * MER_PF_SEG, g0 = c0 - MER_PF_WARM;
    const float *S = src + (int64_t) blockIdx.z * stride_b; float *O = out + (int64_t) blockIdx.z * stride_b;        // uniform
    const uint32_t off = (uint32_t) a * 4u;                                                                           // the lane's byte offset
    const float z1 = -2.0f + sqrtf(3.0f);
    float v[MER_PFX_W];
    // one buffer descriptor per block (base = the window's first line: wave-uniform, 4 SGPRs); sample j sits at scalar offset j x pitch, the
    // lane adds its 32-bit x offset: no vector address arithmetic at all.  pitch x window < 2^31 bytes (host-checked).
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *) (S + (int64_t) g0 * stride_line), 0, 0x7FFFFFFF, 0x00020000);
    const int pitch = (int) stride_line * 4;
#pragma unroll
    for (int j = 0; j < MER_PFX_W; j++) v[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int) off, j * pitch, 0));
    // causal sweep started from the window's first sample (its error has decayed by z1^WARM at the first output), anti-causal sweep started
    // from 0 at the window's end; the same arithmetic per sample as the sequential recursion
    float c = v[0];
#pragma unroll
    for (int j = 1; j < MER_PFX_W; j++) { c = v[j] + z1 * c; v[j] = c; }
    float cn = 0.0f;
#pragma unroll
    for (int j = MER_PFX_W - 1; j >= MER_PF_WARM; j--) { cn = z1 * (cn - v[j]); v[j] = 6 * cn; }
    const __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc((void *) (O + (int64_t) c0 * stride_line), 0, 0x7FFFFFFF, 0x00020000);
#pragma unroll
    for (int j = 0; j < MER_PF_SEG; j++) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[MER_PF_WARM + j]), ws, (int) off, j * pitch, 0);
}

// ---------------------------------------------------------------------------------------------------
// Synthetic fields of BASELINE.json's configs, generated in HBM
__device__ __forceinline__ uint32_t lowbias32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x;
}
__global__ void synth_field_kernel(int kind, int N, float *out) {
    const int64_t total = (int64_t) N * N * N;
    for (int64_t c = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; c < total; c += (int64_t) gridDim.x * blockDim.x) {
        const int i = (int) (c % N), j = (int) ((c / N) % N), k = (int) (c / ((int64_t) N * N));
        const double x = -1.0 + 2.0 * i / (N - 1), y = -1.0 + 2.0 * j / (N - 1), z = -1.0 + 2.0 * k / (N - 1);
        float v;
        if (kind == 0) {
            const uint32_t lin = ((uint32_t) i + (uint32_t) N * ((uint32_t) j + (uint32_t) N * (uint32_t) k)) ^ 0x5EEDu;
            const double h = (double) lowbias32(lin) / 4294967296.0;
            const double pi = 3.14159265358979323846;
            double r = 0.5 + 0.35 * (sin(3.0 * pi * x) * sin(3.0 * pi * y) * sin(3.0 * pi * z)) + 0.15 * (h - 0.5);
            r = r < 0.0 ? 0.0 : (r > 1.0 ? 1.0 : r);
            v = (float) r;
        } else if (kind == 1) {
            v = (float) (1.3 + (1.6 - 1.3) / (N - 1) * j);          // mfiles/createLinearRIFWithBox.m:6-20
        } else {
            const double R2 = 3.0;                                    // half diagonal of [-1,1]^3, squared
            v = (float) (2.0 - (x * x + y * y + z * z) / R2);         // mfiles/createRadialRIFWithBox.m:15-23
        }
        out[c] = v;
    }
}

}  // namespace mer
