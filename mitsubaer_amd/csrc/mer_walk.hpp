// mer_walk.hpp -- per-lane path state machine (one lane = one path, regenerated on death).
//
// The reference runs one pixel sample to completion per thread (src/librender/integrator.cpp:162-187 ->
// volpath.cpp:84-343 -> Medium::sampleDistance).  On a 64-wide wavefront that nests three data-dependent
// loops and leaves lanes idle until the slowest path of the wave has finished.  Here the only loop is the
// march loop: every iteration each live lane takes ONE eikonal step (curved) or ONE tentative-collision
// jump (straight) of whichever ray it is on -- free flight, NEE transmittance or emitter look-up -- and the
// rare events (collision test, scattering, exit, regeneration) are handled under exec masks in between.
#pragma once
#include "mer_device.hpp"

namespace mer {

enum { ST_NEW = 0, ST_MARCH = 1, ST_DONE = 2 };
enum { K_FREE = 0, K_NEE = 1, K_LOOKUP = 2 };
enum { EV_NONE = 0, EV_ARRIVED, EV_EXITED, EV_REAL, EV_FAIL, EV_WALK_END, EV_TR_DONE, EV_PHASE, EV_AFTER_LOOKUP,
       EV_PATH_DONE, EV_GATE_FAIL, EV_PHASE2 /* phase sampling after K_connect's luminaire sample */ };

struct LaneCounters {
    uint32_t steps, rif_evals, tentative, real, segments, nee, paths, marched;
    __device__ __forceinline__ void clear() { steps = rif_evals = tentative = real = segments = nee = paths = marched = 0; }
};

// ---------------------------------------------------------------------------------------------------
// method = simpson of the heterogeneous medium (src/medium/heterogeneous.cpp:301-544): deterministic composite Simpson quadrature of
// the density along a straight ray, and its inversion for the free flight.  Not a wavefront walk: the whole march runs where the
// walk would have begun (Walk::begin), one lane per ray -- the north star's estimators are Woodcock / ratio tracking, this is the
// reference's other `method`, kept for completeness and as the deterministic cross-check of the trackers.  ray.mint = 0 as
// everywhere on this path; HETVOL_EARLY_EXIT as the reference defines it (:31).
__device__ __forceinline__ float simpson_lookup(const Params &P, LaneCounters &C, f3 p) { C.tentative++; return lookup_float(P.density, p); }
__device__ inline float simpson_integrate(const Params &P, LaneCounters &C, f3 o, f3 d, float rayMaxt) {       // integrateDensity, :301-376
    float mint, maxt;
    if (!aabb_intersect(P.density.wmin, P.density.wmax, o, d, mint, maxt)) return 0.0f;
    mint = fmaxf(mint, 0.0f); maxt = fminf(maxt, rayMaxt);
    const float length = maxt - mint;
    f3 p = o + d * mint; const f3 pLast = o + d * maxt;
    const float maxComp = fmaxf(fmaxf(fmaxf(fmaxf(fmaxf(fmaxf(0.0f, fabsf(p.x)), fabsf(pLast.x)), fabsf(p.y)), fabsf(pLast.y)), fabsf(p.z)), fabsf(pLast.z));
    if (length < 1e-6f * maxComp) return 0.0f;
    uint32_t nSteps = (uint32_t) ceilf(length / P.het_step);
    nSteps += nSteps % 2;
    const float stepSize = length / nSteps;
    const f3 increment = d * stepSize;
    float integratedDensity = simpson_lookup(P, C, p) + simpson_lookup(P, C, pLast);
    const float stopAfterDensity = -logf(1e-4f);
    const float stopValue = stopAfterDensity * 3.0f / (stepSize * P.sc.density_scale);
    p = p + increment;
    float m = 4;
    for (uint32_t i = 1; i < nSteps; ++i) {
        integratedDensity += m * simpson_lookup(P, C, p);
        m = 6 - m;
        if (integratedDensity > stopValue) return MER_INF;
        const f3 next = p + increment;
        if (p.x == next.x && p.y == next.y && p.z == next.z) break;
        p = next;
    }
    return integratedDensity * P.sc.density_scale * stepSize * (1.0f / 3.0f);
}
__device__ inline bool simpson_invert(const Params &P, LaneCounters &C, f3 o, f3 d, float rayMaxt, float desiredDensity,
                                      float &integratedDensity, float &t, float &densityAtT) {                      // invertDensityIntegral, :419-544
    integratedDensity = densityAtT = 0.0f; t = 0.0f;
    float mint, maxt;
    if (!aabb_intersect(P.density.wmin, P.density.wmax, o, d, mint, maxt)) return false;
    mint = fmaxf(mint, 0.0f); maxt = fminf(maxt, rayMaxt);
    const float length = maxt - mint;
    f3 p = o + d * mint; const f3 pLast = o + d * maxt;
    const float maxComp = fmaxf(fmaxf(fmaxf(fmaxf(fmaxf(fmaxf(0.0f, fabsf(p.x)), fabsf(pLast.x)), fabsf(p.y)), fabsf(pLast.y)), fabsf(p.z)), fabsf(pLast.z));
    if (length < 1e-6f * maxComp) return false;
    const uint32_t nSteps = (uint32_t) ceilf(length / (2 * P.het_step));
    const float stepSize = length / nSteps, multiplier = (1.0f / 6.0f) * stepSize * P.sc.density_scale;
    const f3 fullStep = d * stepSize, halfStep = fullStep * .5f;
    float node1 = simpson_lookup(P, C, p);
    for (uint32_t i = 0; i < nSteps; ++i) {
        const float node2 = simpson_lookup(P, C, p + halfStep), node3 = simpson_lookup(P, C, p + fullStep),
                    newDensity = integratedDensity + multiplier * (node1 + node2 * 4 + node3);
        if (newDensity >= desiredDensity) {
            // Newton-bisection on the quadratic fitted to the last three look-ups (:476-528): no further density queries
            float a = 0, b = stepSize, x = a, fx = integratedDensity - desiredDensity;
            const float stepSizeSqr = stepSize * stepSize, temp = P.sc.density_scale / stepSizeSqr;
            int it = 1;
            for (;;) {
                const float dfx = temp * (node1 * stepSizeSqr - (3 * node1 - 4 * node2 + node3) * stepSize * x + 2 * (node1 - 2 * node2 + node3) * x * x);
                x -= fx / dfx;
                if (x <= a || x >= b || dfx == 0) x = 0.5f * (b + a);
                const float intval = integratedDensity + temp * (1.0f / 6.0f) * (x * (6 * node1 * stepSizeSqr - 3 * (3 * node1 - 4 * node2 + node3) * stepSize * x
                                     + 4 * (node1 - 2 * node2 + node3) * x * x));
                fx = intval - desiredDensity;
                if (fabsf(fx) < 1e-6f) {
                    t = mint + stepSize * i + x;
                    integratedDensity = intval;
                    densityAtT = temp * (node1 * stepSizeSqr - (3 * node1 - 4 * node2 + node3) * stepSize * x + 2 * (node1 - 2 * node2 + node3) * x * x);
                    return true;
                } else if (++it > 30) return false;
                if (fx > 0) b = x; else a = x;
            }
        }
        const f3 next = p + fullStep;
        if (p.x == next.x && p.y == next.y && p.z == next.z) break;
        integratedDensity = newDensity;
        node1 = node3;
        p = next;
    }
    return false;
}

// State of the ray a lane is currently marching.
template <bool CURVED, int RIF, int STEPPER, int SIGMA, int BND = 0>
struct Walk {
    // ray: curved -> p = position, v = optical momentum n*d ; straight -> p = origin, v = direction
    f3 p, v;
    float t, tmin, tmax;         // straight: current parameter and clipped segment
    float n0;                    // refStart (heterogeneousrefractive.cpp:468)
    float rem;                   // remainder step of the current trace()
    int   steps_left;            // full steps left; -1 => remainder taken; INT_MAX/1e5 for traceTillBoundary
    int   seg_inf;               // traceTillBoundary (int, not bool: bool flags living across big loops were corrupted under SGPR pressure, DESIGN.md section 4 compiler note 1)
    float dist;                  // distSurf accumulated (curved) / sampled distance (homogeneous)
    float opt;                   // optical length
    float sdens;                 // sampling density chosen by the strategy (homogeneous)
    float Tr, trsum; int walk;   // transmittance estimator
    int   kind;
    int   backstep;              // 1 => the next step is the step back after leaving the shape (trace(): :678-681)
    float hprev;
    int   agg;                   // `aggressivetracing` (BND = 1 only): 1 => the current leg takes its steps without inside tests
    float dleft;                 //   ... and this much of the segment is left after it
    CellCache cc;                // RIF cell cache (trilinear)
    static constexpr int kBND = BND;

    __device__ __forceinline__ f3 pos() const { return CURVED ? p : p + v * t; }

    // homogeneous.cpp:277-296 == heterogeneousrefractive.cpp:435-451
    __device__ __forceinline__ float sample_exp_distance(const Params &P, Rng &rng) {
        float rand = rng.next1D(), sampledDistance;
        sdens = P.sampling_density;
        if (rand < P.medium_sampling_weight) {
            rand /= P.medium_sampling_weight;
            if (P.sc.strategy == MER_STRATEGY_MAXIMUM) return maxexp_sample(P.maxexp, 1 - rand, sdens);    // :291: the pdf rides in sdens
            if (P.sc.strategy == MER_STRATEGY_BALANCE) {
                const int channel = min((int) (rng.next1D() * 3), 2);
                sdens = channel == 0 ? P.sigT.x : (channel == 1 ? P.sigT.y : P.sigT.z);
            }
            sampledDistance = -logf(1 - rand) / sdens;
        } else sampledDistance = MER_INF;
        return sampledDistance;
    }

    // trace(): steps = int(d/h), remainder = d - steps*h (heterogeneousrefractive.cpp:671-675)
    __device__ __forceinline__ void set_segment(const Params &P, float s) {
        const float h = P.sc.stepsize;
        const int steps = (int) (s / h);
        rem = s - steps * h;
        steps_left = steps;
        seg_inf = 0;
        agg = 0;
        if (BND == 1 && P.sc.aggressive_tracing) agg_next(P, s);
    }
    // aggressivetracing (heterogeneousrefractive.cpp:476-493): while the point is at least Epsilon below the surface (less the SDF's
    // error bound), the next leg of min(depth, distance left) is walked by aggressive_trace (:697-704: int(d/h) full steps + the
    // remainder step, no inside tests); what is left afterwards is an ordinary tested trace().
    __device__ __forceinline__ void agg_next(const Params &P, float dist_left) {
        const float h = P.sc.stepsize;
        if (dist_left > MER_EPSILON) {
            const float depth = -sdf_value(P, p) - P.sc.sdf_max_error;
            if (!(depth < MER_EPSILON)) {
                const float d = fminf(depth, dist_left);
                const int steps = (int) (d / h);
                rem = d - steps * h; steps_left = steps; agg = 1; dleft = dist_left - d;
                return;
            }
        }
        const int steps = (int) (dist_left / h);
        rem = dist_left - steps * h; steps_left = steps; agg = 0; dleft = 0.0f;
    }
    __device__ __forceinline__ void draw_segment(const Params &P, Rng &rng) {      // heterogeneous.cpp:634
        set_segment(P, -logf(1 - rng.next1D()) * P.inv_max_density);
    }

    // Start marching a ray of the given kind from o along d.  Returns EV_NONE when marching started,
    // otherwise the event that finishes the walk immediately.
    __device__ __forceinline__ int begin(const Params &P, Rng &rng, LaneCounters &C, int k, f3 o, f3 d, float rayMaxt, bool first_walk = true) {
        kind = k;
        if (first_walk) { trsum = 0.0f; walk = 0; }
        Tr = 1.0f; dist = 0.0f; opt = 0.0f; backstep = 0; agg = 0; dleft = 0.0f;
        if (CURVED) {
            p = o; v = d;
            if (RIF == MER_RIF_BSPLINE3 && !inside_volume_limits(P.rif, p)) return EV_GATE_FAIL;   // heterogeneousrefractive.cpp:461-466
            float n; f3 g;
            rif_value_grad<RIF>(P.rif, cc, p, n, g); C.rif_evals++;
            n0 = n;
            v = d * n0;                                                       // :470-472
            if (SIGMA == MER_SIGMA_GRID) draw_segment(P, rng);
            else {
                const float s = (k == K_FREE) ? sample_exp_distance(P, rng) : MER_INF;
                if (isfinite(s)) { set_segment(P, s); dist = 0.0f; t = s; }
                else { seg_inf = 1; steps_left = 100000; rem = 0.0f; t = MER_INF; }   // traceTillBoundary :742-776
            }
            return EV_NONE;
        } else {
            p = o; v = d;
            if (SIGMA == MER_SIGMA_GRID && P.sc.method == MER_METHOD_SIMPSON) {          // heterogeneous.cpp:547-548, 594-612: the march runs here
                if (k == K_FREE) {
                    const float desired = -logf(1 - rng.next1D());
                    float integrated, dens;
                    const bool ok = simpson_invert(P, C, o, d, rayMaxt, desired, integrated, t, dens);
                    Tr = expf(-integrated); sdens = dens;                                 // expVal and densityAtT ride to finish_free_flight
                    return (ok && Tr * dens > 0) ? EV_REAL : EV_FAIL;
                }
                Tr = expf(-simpson_integrate(P, C, o, d, rayMaxt));
                return EV_WALK_END;
            }
            if (SIGMA == MER_SIGMA_GRID) {
                float mint, maxt;                                             // heterogeneous.cpp:626-630
                if (!aabb_intersect(P.density.wmin, P.density.wmax, o, d, mint, maxt)) {
                    if (k == K_FREE) return EV_FAIL;
                    return EV_TR_DONE;                                        // evalTransmittance returns 1 (:553-554)
                }
                tmin = fmaxf(mint, 0.0f);
                tmax = fminf(maxt, rayMaxt);
                t = tmin;
                return EV_NONE;
            } else {
                // homogeneous.cpp:275-352: closed form
                tmax = rayMaxt;
                if (k == K_FREE) {
                    const float s = sample_exp_distance(P, rng);
                    const float distSurf = rayMaxt - 0.0f;
                    if (s < distSurf) { t = s; dist = s; return EV_REAL; }
                    dist = distSurf; t = distSurf; return EV_FAIL;
                }
                return EV_TR_DONE;       // transmittance = exp(-sigma_t * maxt), evaluated by the caller
            }
        }
    }

    // One unit of marching work.  Curved: one er_step + insideShape test (trace(): :676-689).
    // Straight: one tentative-collision jump (heterogeneous.cpp:633-636).
    // COUNT = false: the caller keeps one trip counter and adds steps / rif_evals / marched itself (K_march's hot loop)
    template <bool COUNT = true>
    __device__ __forceinline__ int advance(const Params &P, Rng &rng, LaneCounters &C) {
        if (COUNT) C.marched++;
        if (CURVED) {
            // one er_step call site: the step back after an exit is one more trip through here with -h
            const bool full = steps_left > 0;
            const float h = backstep ? -hprev : (full ? P.sc.stepsize : rem);
            er_step<RIF, STEPPER>(P.rif, cc, p, v, h, opt);
            if (COUNT) { C.steps++; C.rif_evals += evals_per_step<STEPPER>(); }
            if (backstep) {
                backstep = 0;
                if (seg_inf) dist -= hprev;                                   // traceTillBoundary :757-759 (as shipped)
                return EV_EXITED;
            }
            if (BND == 1 && agg) {                                             // a leg of aggressive_trace: no inside test
                dist += h;
                if (full) { steps_left--; return EV_NONE; }
                agg_next(P, dleft);                                           // leg done: the next one, or the tested trace of the rest
                return EV_NONE;
            }
            if (!inside_shape_b<BND>(P, p)) { backstep = 1; hprev = h; return EV_NONE; }   // (:678-681)
            dist += h;
            if (full) {
                steps_left--;
                if (seg_inf && steps_left == 0) return EV_EXITED;             // 1e5 steps exhausted (:773-775)
                return EV_NONE;
            }
            return EV_ARRIVED;
        } else {
            t -= logf(1 - rng.next1D()) * P.inv_max_density;
            if (t >= tmax) return EV_EXITED;
            return EV_ARRIVED;
        }
    }

    // Tentative collision at pos(): free flight -> real/null test (heterogeneous.cpp:638-656);
    // transmittance walks -> ratio tracking or the reference's binary Woodcock estimator (:562-585).
    __device__ __forceinline__ int on_arrived(const Params &P, Rng &rng, LaneCounters &C, float &sigma_out) {
        if (SIGMA == MER_SIGMA_GRID) {
            const float sigma = lookup_float(P.density, pos()) * P.sc.density_scale;     // lookupDensity * m_scale
            C.tentative++;
            sigma_out = sigma;
            if (kind == K_FREE) {
                if (sigma * P.inv_max_density > rng.next1D()) return EV_REAL;
            } else if (P.sc.tr_estimator == MER_TR_RATIO) {
                Tr *= 1.0f - sigma * P.inv_max_density;
                if (Tr == 0.0f) return EV_WALK_END;
            } else {
                if (sigma * P.inv_max_density > rng.next1D()) { Tr = 0.0f; return EV_WALK_END; }
            }
            if (CURVED) draw_segment(P, rng);
            return EV_NONE;
        } else {
            sigma_out = 0.0f;
            return EV_REAL;          // homogeneous sigma: the sampled distance was reached
        }
    }
};

// strategy pdfs + transmittance for a homogeneous medium (homogeneous.cpp:317-349 == hetrefr.cpp:533-565)
__device__ __forceinline__ void strategy_pdfs(const Params &P, float sampledDistance, float samplingDensity,
                                              f3 &transmittance, float &pdfSuccess, float &pdfFailure) {
    if (P.sc.strategy == MER_STRATEGY_BALANCE) {
        pdfFailure = 0; pdfSuccess = 0;
        const float sT[3] = {P.sigT.x, P.sigT.y, P.sigT.z};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float tmp = expf(-sT[i] * sampledDistance);
            pdfFailure += tmp; pdfSuccess += sT[i] * tmp;
        }
        pdfFailure /= 3; pdfSuccess /= 3;
    } else if (P.sc.strategy == MER_STRATEGY_MAXIMUM) {       // :318-320; pdfSuccess was set by MaxExpDist::sample
        pdfFailure = 1 - maxexp_cdf(P.maxexp, sampledDistance);
        pdfSuccess = samplingDensity;
    } else {
        pdfFailure = expf(-samplingDensity * sampledDistance);
        pdfSuccess = samplingDensity * pdfFailure;
    }
    transmittance = f3(expf(P.sigT.x * (-sampledDistance)), expf(P.sigT.y * (-sampledDistance)), expf(P.sigT.z * (-sampledDistance)));
    pdfSuccess = pdfSuccess * P.medium_sampling_weight;
    pdfFailure = P.medium_sampling_weight * pdfFailure + (1 - P.medium_sampling_weight);
    if (max3(transmittance) < 1e-20f) transmittance = f3(0, 0, 0);
}

// homogeneous transmittance over a length (homogeneous.cpp:264-273, heterogeneousrefractive.cpp:393-400)
__device__ __forceinline__ f3 homogeneous_transmittance(const Params &P, float negLength) {
    return f3(P.sigT.x != 0 ? expf(P.sigT.x * negLength) : 1.0f,
              P.sigT.y != 0 ? expf(P.sigT.y * negLength) : 1.0f,
              P.sigT.z != 0 ? expf(P.sigT.z * negLength) : 1.0f);
}

// MediumSamplingRecord fields used by the integrator (include/mitsuba/render/medium.h:40-98)
struct MRec {
    f3 p, d, sigmaS, transmittance;
    float t, pdfSuccess, pdfFailure, refRatioSq, opticalLength;
};

// Fill the record at the end of a free-flight walk (success = real collision).
template <bool CURVED, int RIF, int STEPPER, int SIGMA, int BND>
__device__ __forceinline__ void finish_free_flight(const Params &P, LaneCounters &C, Walk<CURVED, RIF, STEPPER, SIGMA, BND> &W,
                                                   bool success, float sigma, MRec &m) {
    m.refRatioSq = 1.0f; m.opticalLength = W.opt;
    m.p = W.pos(); m.d = W.v; m.t = CURVED ? W.dist : W.t;
    if (CURVED) {
        float refEnd; f3 g;
        rif_value_grad<RIF>(P.rif, W.cc, W.p, refEnd, g); C.rif_evals++;                // :500-501
        m.refRatioSq = (1.0f / (W.n0 * W.n0)) * (refEnd * refEnd);
    }
    if (SIGMA == MER_SIGMA_GRID) {
        m.pdfSuccess = 1.0f; m.pdfFailure = 1.0f; m.transmittance = f3(1, 1, 1);         // heterogeneous.cpp:616-619
        if (!CURVED && P.sc.method == MER_METHOD_SIMPSON) {                              // :604-611: expVal in W.Tr, densityAtT in W.sdens
            m.transmittance = f3(W.Tr, W.Tr, W.Tr); m.pdfFailure = W.Tr; m.pdfSuccess = W.Tr * W.sdens;
            m.sigmaS = success ? albedo_at(P, m.p) * W.sdens : f3(0, 0, 0);
        } else
        if (success) {
            const f3 albedo = albedo_at(P, m.p);
            m.sigmaS = albedo * sigma;                                                   // :645-649
            float tr = sigma != 0.0f ? 1.0f / sigma : 0.0f;
            if (!isfinite(tr)) tr = 0.0f;
            m.transmittance = f3(tr, tr, tr);
        } else m.sigmaS = f3(0, 0, 0);
    } else {
        m.sigmaS = P.sigS;
        const float d = CURVED ? (success ? W.t : W.dist) : W.dist;
        if (CURVED) m.t = d;
        strategy_pdfs(P, d, W.sdens, m.transmittance, m.pdfSuccess, m.pdfFailure);
    }
}

}  // namespace mer
