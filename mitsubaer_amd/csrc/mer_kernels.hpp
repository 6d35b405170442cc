// mer_kernels.hpp -- gfx950 kernels: the render megakernel (K_trace + K_film), leaf kernels for the
// parity entry points, grid re-layout, B-spline prefilter, synthetic fields.
#pragma once
#include "mer_walk.hpp"

namespace mer {

#define MER_BLOCK 256
#define MER_TILE 32

// ---------------------------------------------------------------------------------------------------
// Work decode: w -> (pixel, sample).  Sample-major; inside a pass pixels go by 32x32 image tiles (the
// reference's block size, src/mitsuba/mitsuba.cpp:80-81) and by 8x8 sub-tiles so that the 64 lanes of a
// fresh wavefront start on one 8x8 pixel patch (coherent camera rays, distinct film pixels per lane).
__device__ __forceinline__ bool decode_work(const Params &P, uint64_t w, int &x, int &y, uint32_t &sample) {
    const uint32_t npix = (uint32_t) P.ntiles_mine * (MER_TILE * MER_TILE);
    const uint32_t s_local = (uint32_t) (w / npix);
    const uint32_t r = (uint32_t) (w - (uint64_t) s_local * npix);
    const uint32_t tile_local = r >> 10, q = r & 1023u, sub = q >> 6, lane = q & 63u;
    const uint32_t tile = (uint32_t) P.tile_rank + tile_local * (uint32_t) P.tile_count;
    const uint32_t tx = tile % (uint32_t) P.tiles_x, ty = tile / (uint32_t) P.tiles_x;
    x = (int) (tx * MER_TILE + (sub & 3u) * 8u + (lane & 7u));
    y = (int) (ty * MER_TILE + (sub >> 2) * 8u + (lane >> 3));
    sample = (uint32_t) P.spp_begin + s_local * (uint32_t) P.spp_stride;
    return x < P.sc.width && y < P.sc.height;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

}  // namespace mer
#include "mer_wavefront.hpp"
namespace mer {

// ---------------------------------------------------------------------------------------------------
// Megakernel form of the same state machine (kept for A/B timing: MER_MODE=mega).
// K_trace: VolumetricPathTracer::Li (src/integrators/path/volpath.cpp:84-343) restricted to one convex
// index-matched shape + interior medium + constant environment emitter, with the refractive hooks of
// src/libbidir/edge.cpp:45-60,91-93 and src/libbidir/vertex.cpp:251-255, as a lane-persistent state machine.
template <bool CURVED, int RIF, int STEPPER, int SIGMA>
__global__ void __launch_bounds__(MER_BLOCK) render_kernel(const Params P) {
    typedef Walk<CURVED, RIF, STEPPER, SIGMA> WalkT;
    const mer_scene_desc &S = P.sc;
    const f3 env(S.env_radiance[0], S.env_radiance[1], S.env_radiance[2]);
    const bool hasEnv = !is_zero(env);
    const bool hasEmission = S.emission[0] != 0 || S.emission[1] != 0 || S.emission[2] != 0;
    const int maxDepth = S.max_depth;
    const int nwalks = (SIGMA == MER_SIGMA_GRID && S.tr_estimator == MER_TR_WOODCOCK2) ? 2 : 1;

    Rng rng; rng.state = 0; rng.inc = 1;
    LaneCounters C; C.clear();
    WalkT W;
    W.kind = K_FREE; W.steps_left = 0; W.rem = 0; W.seg_inf = 0; W.t = 0; W.tmin = 0; W.tmax = 0; W.n0 = 1;
    W.dist = 0; W.opt = 0; W.sdens = 0; W.Tr = 1; W.trsum = 0; W.walk = 0; W.p = f3(0, 0, 0); W.v = f3(0, 0, 1);
    W.backstep = 0; W.hprev = 0; W.cc.reset();
    int st = ST_NEW;
    int px_i = 0, py_i = 0; float px = 0, py = 0;
    f3 L(0, 0, 0), T(1, 1, 1);
    // path flags live in one VGPR word (lane-mask booleans spilled through SGPRs proved fragile here)
    enum { F_SCATTERED = 1, F_EMITTED = 2, F_ITSVALID = 4 };
    int depth = 1, flags = F_EMITTED;
#define scattered ((flags & F_SCATTERED) != 0)
#define emitted ((flags & F_EMITTED) != 0)
#define itsValid ((flags & F_ITSVALID) != 0)
#define SET_FLAG(f, v) flags = (v) ? (flags | (f)) : (flags & ~(f))
    f3 ps(0, 0, 0), dsave(0, 0, 1), dd(0, 0, 1), wi(0, 0, 1);
    f3 trv(1, 1, 1);                   // transmittance of the walk that just finished
    float phasePdf = 0, itsT = 0;
    uint32_t wave_iters = 0;

    for (;;) {
        int ev = EV_NONE;
        float sigma = 0.0f;
        // ------------------------------------------------------------------ regeneration (integrator.cpp:162-187)
        if (st == ST_NEW) {
            const uint64_t w = atomicAdd(P.work_counter, 1ULL);
            if (w >= P.total_work) st = ST_DONE;
            else {
                uint32_t sample;
                if (!decode_work(P, w, px_i, py_i, sample)) { /* pixel of a partial edge tile */ }
                else {
                    rng.seed(P.seed, (uint32_t) (py_i * S.width + px_i), sample);
                    const float sx = rng.next1D(), sy = rng.next1D();
                    px = (float) px_i + sx; py = (float) py_i + sy;
                    f3 o, d; float mint, maxt;
                    sample_ray(P, px, py, o, d, mint, maxt);
                    L = f3(0, 0, 0); T = f3(1, 1, 1); depth = 1; flags = F_EMITTED;
                    C.paths++;
                    itsT = intersect_shape(S, o, d, mint, maxt);                       // rRec.rayIntersect(ray)
                    if (itsT < 0) {
                        if (!S.hide_emitters) L = L + T * env;                         // volpath.cpp:194-201
                        ev = EV_PATH_DONE;
                    } else if (depth >= maxDepth && maxDepth != -1) ev = EV_PATH_DONE;
                    else {
                        (void) rng.next1D(); (void) rng.next1D();                      // null bsdf->sample(..., nextSample2D())
                        const f3 ro = o + d * itsT;
                        bool medium = true;
                        if (CURVED) { itsT = 0; SET_FLAG(F_ITSVALID, true); }
                        else { itsT = intersect_shape(S, ro, d, MER_EPSILON, MER_INF); SET_FLAG(F_ITSVALID, itsT >= 0); if (!itsValid) medium = false; }
                        depth++;
                        if (!(depth <= maxDepth || maxDepth < 0)) ev = EV_PATH_DONE;
                        else if (!medium) { if (!S.hide_emitters) L = L + T * env; ev = EV_PATH_DONE; }
                        else { C.segments++; ps = ro; dsave = d; ev = W.begin(P, rng, C, K_FREE, ro, d, itsT); st = ST_MARCH; }
                    }
                }
            }
        } else if (st == ST_MARCH) {
            ev = W.advance(P, rng, C);
        }
        wave_iters++;
        if (st == ST_DONE) break;

        // ------------------------------------------------------------------ events
        while (ev != EV_NONE) {
#ifdef MER_DEBUG
            if (P.dbg_pixel == py_i * S.width + px_i && ev != EV_ARRIVED)
                printf("gpu ev=%d kind=%d depth=%d T=%g L=%g Tr=%g trsum=%g walk=%d trv=%g t=%g tmax=%g rng=%llu\n", ev, W.kind, depth, T.x, L.x, W.Tr, W.trsum, W.walk, trv.x, W.t, W.tmax, (unsigned long long) rng.state);
#endif
            if (ev == EV_ARRIVED) {
                ev = W.on_arrived(P, rng, C, sigma);
            } else if (ev == EV_EXITED) {
                ev = (W.kind == K_FREE) ? EV_FAIL : EV_WALK_END;
            } else if (ev == EV_GATE_FAIL) {
                if (W.kind == K_FREE) ev = EV_PATH_DONE;          // transmittance 0 => nothing further contributes
                else { trv = f3(0, 0, 0); ev = EV_TR_DONE; }
            } else if (ev == EV_WALK_END) {
                W.trsum += W.Tr; W.walk++;
                if (W.walk < nwalks) ev = W.begin(P, rng, C, W.kind, ps, W.kind == K_NEE ? dd : dsave, itsT, false);
                else {
                    if (SIGMA == MER_SIGMA_GRID) { const float tv = W.trsum / (float) nwalks; trv = f3(tv, tv, tv); }
                    else trv = homogeneous_transmittance(P, -W.dist);                    // heterogeneousrefractive.cpp:393-400
                    ev = EV_TR_DONE;
                }
            } else if (ev == EV_REAL) {
                // ---- medium interaction: volpath.cpp:104-118
                MRec m;
                finish_free_flight(P, C, W, true, sigma, m);
                bool success = true;
                if (SIGMA == MER_SIGMA_HOMOGENEOUS) {
                    const f3 o0 = ps;
                    if (m.p.x == o0.x && m.p.y == o0.y && m.p.z == o0.z) success = false;   // no forward progress
                }
                if (!success) { ev = EV_FAIL; continue; }
                C.real++;
                if (depth >= maxDepth && maxDepth != -1) { ev = EV_PATH_DONE; continue; }
                if (hasEmission && SIGMA == MER_SIGMA_GRID)
                    L = L + T * f3(S.emission[0], S.emission[1], S.emission[2]) * m.refRatioSq;
                T = T * (m.sigmaS * m.transmittance / m.pdfSuccess);
                if (CURVED) T = T * m.refRatioSq;                                         // edge.cpp:91-93
                wi = CURVED ? normalize(-m.d) : -W.v;                                     // vertex.cpp:251-255
                ps = m.p;
                if (hasEnv) {
                    // ---- luminaire sampling: scene.cpp:854-874, constant.cpp:179-214
                    C.nee++;
                    const int interactions = maxDepth - depth - 1;
                    const float s2x = rng.next1D(), s2y = rng.next1D();
                    dd = square_to_uniform_sphere(s2x, s2y);
                    W.kind = K_NEE;
                    if (interactions != 0) {                                              // scene.cpp:619-678: one null crossing
                        float tExit = 0.0f;
                        if (!CURVED) tExit = intersect_shape(S, ps, dd, 0.0f, MER_INF);
                        if (tExit >= 0) {
                            itsT = tExit;
                            ev = W.begin(P, rng, C, K_NEE, ps, dd, tExit);
                            if (ev == EV_TR_DONE) trv = (SIGMA == MER_SIGMA_GRID) ? f3(1, 1, 1) : homogeneous_transmittance(P, 0.0f - tExit);
                        } else { trv = f3(1, 1, 1); ev = EV_TR_DONE; }
                    } else { trv = f3(0, 0, 0); ev = EV_TR_DONE; }
                } else ev = EV_PHASE;
            } else if (ev == EV_TR_DONE) {
                const f3 tr = trv;
                if (W.kind == K_NEE) {
                    const float dpdf = MER_INV_FOURPI;
                    f3 value = env / dpdf;
                    value = value * tr;
                    if (!is_zero(value)) {
                        const float phaseVal = phase_eval(S.phase, S.g, wi, dd);
                        if (phaseVal != 0) {
                            const float weight = mi_weight(dpdf, phaseVal);              // env emitter is "on surface": constant.cpp:47
                            L = L + T * value * phaseVal * weight;
                        }
                    }
                    ev = EV_PHASE;
                } else {
                    // emitter look-up along the phase-sampled direction: volpath.cpp:162-173,370-428
                    const int maxInteractions = maxDepth - depth - 1;
                    const bool blocked = (maxInteractions == 0) && (CURVED || itsValid);
                    if (!blocked && !is_zero(tr)) {
                        const f3 value = tr * env;
                        L = L + T * value * mi_weight(phasePdf, MER_INV_FOURPI);
                    }
                    ev = EV_AFTER_LOOKUP;
                }
            } else if (ev == EV_PHASE) {
                // ---- phase function sampling: volpath.cpp:149-160
                const float p2x = rng.next1D(), p2y = rng.next1D();
                f3 wo;
                phase_sample(S.phase, S.g, wi, p2x, p2y, wo, phasePdf);
                dsave = wo;
                if (CURVED) { itsT = 0; SET_FLAG(F_ITSVALID, true); }
                else { itsT = intersect_shape(S, ps, wo, 0.0f, MER_INF); SET_FLAG(F_ITSVALID, itsT >= 0); }
                if (hasEnv) {
                    W.kind = K_LOOKUP;
                    if (!CURVED && !itsValid) { trv = f3(1, 1, 1); ev = EV_TR_DONE; }
                    else {
                        ev = W.begin(P, rng, C, K_LOOKUP, ps, wo, itsT);
                        if (ev == EV_TR_DONE) trv = (SIGMA == MER_SIGMA_GRID) ? f3(1, 1, 1) : homogeneous_transmittance(P, 0.0f - itsT);
                    }
                } else ev = EV_AFTER_LOOKUP;
            } else if (ev == EV_AFTER_LOOKUP) {
                SET_FLAG(F_EMITTED, false);                                               // ERadianceNoEmission
                ev = EV_NONE;
                if (depth++ >= S.rr_depth) {                                              // volpath.cpp:326-336
                    const float q = fminf(max3(T) * 1.0f * 1.0f, 0.95f);
                    if (rng.next1D() >= q) ev = EV_PATH_DONE;
                    else T = T / q;
                }
                if (ev == EV_NONE) {
                    SET_FLAG(F_SCATTERED, true);
                    if (!(depth <= maxDepth || maxDepth < 0)) ev = EV_PATH_DONE;
                    else { C.segments++; ev = W.begin(P, rng, C, K_FREE, ps, dsave, itsT); }
                }
            } else if (ev == EV_FAIL) {
                // ---- no medium interaction: volpath.cpp:183-201,289-301
                MRec m;
                finish_free_flight(P, C, W, false, 0.0f, m);
                T = T * (m.transmittance / m.pdfFailure);
                if (CURVED) { T = T * m.refRatioSq; SET_FLAG(F_ITSVALID, true); }         // edge.cpp:45-60
                ev = EV_PATH_DONE;
                if (!itsValid) {
                    if (emitted && (!S.hide_emitters || scattered)) L = L + T * env;
                } else if (!(depth >= maxDepth && maxDepth != -1)) {
                    (void) rng.next1D(); (void) rng.next1D();                             // null BSDF sample
                    SET_FLAG(F_EMITTED, !scattered);
                    depth++;
                    if (depth <= maxDepth || maxDepth < 0)
                        if (emitted && (!S.hide_emitters || scattered)) L = L + T * env;
                }
            } else {  // EV_PATH_DONE: ImageBlock::put (imageblock.h:124-205)
                if (P.path_out) {
                    float *q = P.path_out + ((size_t) py_i * S.width + px_i) * 3;
                    q[0] = L.x; q[1] = L.y; q[2] = L.z;
                } else film_put(P, px, py, L, 1.0f);
                st = ST_NEW;
                ev = EV_NONE;
            }
        }
    }
#undef scattered
#undef emitted
#undef itsValid
#undef SET_FLAG
    // ---- counters (StatsCounter analogue; inputs of the roofline formula)
    const uint32_t sums[8] = {wave_sum(C.paths), wave_sum(C.steps), wave_sum(C.rif_evals), wave_sum(C.tentative),
                              wave_sum(C.real), wave_sum(C.segments), wave_sum(C.nee), wave_sum(C.marched)};
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int i = 0; i < 7; i++) if (sums[i]) atomicAdd(P.counters + i, (unsigned long long) sums[i]);
        atomicAdd(P.counters + MER_C_LOOP_ITERS, (unsigned long long) wave_iters * 64ULL);
        atomicAdd(P.counters + MER_C_ACTIVE_LANES, (unsigned long long) sums[7]);
    }
}

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
    const f3 p(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
    CellCache cc; cc.reset();
    if (interp == MER_RIF_BSPLINE3) bspline_value_grad(g, p, v, gr);
    else if (g.layout == MER_LAYOUT_BRICK27 || g.layout == MER_LAYOUT_BRICK125) { if (g.buf_bytes) trilinear_value_grad<RIFK_BRICK27_BUF>(g, cc, p, v, gr); else trilinear_value_grad<RIFK_BRICK27>(g, cc, p, v, gr); }
    else if (g.layout == MER_LAYOUT_CELL8) { if (g.buf_bytes) trilinear_value_grad<RIFK_CELL8_BUF>(g, cc, p, v, gr); else trilinear_value_grad<RIFK_CELL8>(g, cc, p, v, gr); }
    else { if (g.buf_bytes) trilinear_value_grad<RIFK_DENSE_BUF>(g, cc, p, v, gr); else trilinear_value_grad<MER_RIF_TRILINEAR>(g, cc, p, v, gr); }
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
    out[i] = correlation_function(P, t[i]);
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
// weight z1^k, below float resolution (2^-24 relative) after 13 samples and below 1e-23 after 40.  So a line is cut into segments that
// are filtered independently after a warm-up of MER_PF_WARM samples (the first segment starts from the exact mirror sum, the last
// anti-causal one from the exact end condition): same arithmetic per sample as the sequential recursion, N^3 / SEG threads instead
// of N^2, every global access coalesced.  Two kernels per axis (causal -> tmp, anti-causal -> out): the anti-causal warm-up of one
// segment reads causal values that a neighbouring segment would otherwise already have overwritten.
#define MER_PF_WARM 40
#define MER_PF_SEG 32
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

// The pass along x (lines contiguous in memory): a block stages 256 lines x (32 outputs + 40 warm-up samples on either side) in LDS
// with row-contiguous (coalesced) loads, each thread filters one line of the tile causally and anti-causally in place (row pitch
// 113 words: conflict-free), and the 32 outputs per line go back with coalesced stores -- one read and one write of the volume.
#define MER_PFX_ROWS 256
#define MER_PFX_COLS 32
#define MER_PFX_W (MER_PFX_COLS + 2 * MER_PF_WARM)
#define MER_PFX_LD (MER_PFX_W + 1)
__global__ void __launch_bounds__(256) bspline_x_kernel(const float *src, float *out, int64_t nlines, int n) {
    __shared__ float lds[MER_PFX_ROWS * MER_PFX_LD];
    const int ntile = (n + MER_PFX_COLS - 1) / MER_PFX_COLS;
    const int tile = (int) (blockIdx.x % (unsigned) ntile);              // neighbouring column tiles run together: the warm-up overlap is an L2 hit
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
        float c = (lo == 0) ? bspline_cp0(L, 0, 1, n, z1) : L[0];
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

// x pass, register form (n % 4 == 0): thread <-> (segment of 32 outputs, line); the 112-sample window (40 + 32 + 40) is fetched with
// 28 independent 16-byte loads -- adjacent lanes own adjacent 128-byte segments of one line, so a wave reads one contiguous 8 KB
// span (plus halo) and every line it touches is shared through L1 -- filtered in registers, and written with 8 16-byte stores.
__global__ void __launch_bounds__(256) bspline_x_reg_kernel(const float *src, float *out, int64_t nlines, int n) {
    const int nseg = (n + MER_PF_SEG - 1) / MER_PF_SEG;
    const int64_t id = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nlines * nseg) return;
    const int seg = (int) (id % nseg); const int64_t row = id / nseg;
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

// y / z passes, register-window form: thread <-> (a = x index, segment, b); the 112-sample window is strided by the line pitch, every
// load and store is coalesced across the wave (adjacent lanes = adjacent x); causal and anti-causal sweeps fused: one read, one write.
__global__ void __launch_bounds__(256) bspline_win_kernel(const float *src, float *out, int na, int nb, int64_t stride_b, int64_t stride_line, int n) {
    const int nseg = (n + MER_PF_SEG - 1) / MER_PF_SEG;
    const int64_t id = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (int64_t) na * nseg * nb) return;
    const int64_t a = id % na, r = id / na; const int seg = (int) (r % nseg); const int64_t b = r / nseg;
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
