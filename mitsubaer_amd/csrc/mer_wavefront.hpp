// mer_wavefront.hpp -- wavefront form of the hot path: K_march (hot, lean) + K_event (cold, fat).
//
// Why two kernels: the per-path state of volpath (throughput, radiance, scatter point, saved directions, MIS
// pdfs, sampler, film position ...) is ~45 registers that the marching loop never touches.  Kept live in one
// megakernel they cap the occupancy at 1-2 waves/SIMD, and the loop is latency-bound (4 dependent field
// gathers per RK4 step).  Here path state lives in HBM as struct-of-arrays slots; K_march loads only the ray it
// is marching (~20 words), takes up to `ksteps` eikonal steps / tentative-collision jumps -- null collisions are
// resolved in place -- and parks the lane at the first real event.  K_event then runs the volpath state machine
// (scatter, NEE, look-up, Russian roulette, exit, film splat, regeneration from the work counter) for the
// lanes that have an event.  Slot traffic is ~160 B per lane per pass against ksteps x 128 B of field fetches.
#pragma once
#include "mer_walk.hpp"
#include "mer_connect.hpp"

namespace mer {

// hot words (K_march reads/writes these)
enum { H_PX = 0, H_PY, H_PZ, H_VX, H_VY, H_VZ, H_OPT, H_DIST, H_REM, H_HPREV, H_TR, H_T, H_TMAX, H_STEPS, H_FLAGS,
       H_RNG_LO, H_RNG_HI, H_PIXEL, H_SAMPLE, H_SIGMA, H_COUNT };   // record order of load_hot / store_hot
// cold words (K_event only)
enum { CO_PXF = H_COUNT, CO_PYF, CO_LX, CO_LY, CO_LZ, CO_TX, CO_TY, CO_TZ, CO_DEPTH, CO_PFLAGS, CO_PSX, CO_PSY, CO_PSZ,
       CO_DSX, CO_DSY, CO_DSZ, CO_DDX, CO_DDY, CO_DDZ, CO_WIX, CO_WIY, CO_WIZ, CO_PHASEPDF, CO_ITST, CO_N0, CO_TRSUM,
       CO_SDENS, CO_TMIN, CO_WNEXT_LO, CO_WNEXT_HI, CO_WLEFT, CO_PLEN, CO_TROPT, CO_ETA, SLOT_WORDS };

// H_FLAGS: st[1:0] ev[5:2] kind[7:6] seg_inf[8] backstep[9] walk[11:10] agg[12] init[13] (a spawned side walk that K_march has yet to begin)
#define MER_FLAG_INIT (1u << 13)
#define MER_FLAG_CHILD (1u << 14)      // the record is a spawned side walk (hot copy of CO_PFLAGS' F_CHILD: K_march reads hot words only)
__device__ __forceinline__ uint32_t pack_flags(int st, int ev, int kind, int seg_inf, int backstep, int walk, int agg) {
    return (uint32_t) st | ((uint32_t) ev << 2) | ((uint32_t) kind << 6) | ((uint32_t) seg_inf << 8) | ((uint32_t) backstep << 9) |
           ((uint32_t) walk << 10) | ((uint32_t) agg << 12);
}

// Slot i is one 256-byte record: words [0,20) hot, [20,48) cold, rest padding.  Lanes reach their slot through the
// compacted march / event lists, i.e. by gather: a record-per-slot layout turns each lane's state access into a few
// whole 16-byte pieces of two cache lines instead of 20-48 scattered dwords of a struct-of-arrays.
#define MER_SLOT_WORDS 64
#define MER_HITQ_HEAD 32          // hitq_ctr[0] = produced (tail), hitq_ctr[32] = consumed (head): separate 256-B lines
#ifndef MER_WORK_BATCH
#define MER_WORK_BATCH 8
#endif
#define SLOT(k) P.slots[(size_t) MER_CHK(P.chk, CHK_SLOT, i, P.nslots_all) * MER_SLOT_WORDS + (k)]
#define SLOTF(k) __uint_as_float(SLOT(k))

template <class WalkT>
__device__ __forceinline__ void load_hot(const Params &P, uint32_t i, uint32_t &fl, WalkT &W, Rng &rng, uint32_t &pixel, uint32_t &sample, float &sigma) {
    const uint4 *r = (const uint4 *) (P.slots + (size_t) MER_CHK(P.chk, CHK_SLOT, i, P.nslots_all) * MER_SLOT_WORDS);
    const uint4 a = r[0], b = r[1], c = r[2], d = r[3], e = r[4];
    W.p = f3(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z));
    W.v = f3(__uint_as_float(a.w), __uint_as_float(b.x), __uint_as_float(b.y));
    W.opt = __uint_as_float(b.z); W.dist = __uint_as_float(b.w);
    W.rem = __uint_as_float(c.x); W.hprev = __uint_as_float(c.y); W.Tr = __uint_as_float(c.z); W.t = __uint_as_float(c.w);
    W.tmax = __uint_as_float(d.x); W.steps_left = (int) d.y; fl = d.z;
    rng.state = (uint64_t) d.w | ((uint64_t) e.x << 32);
    pixel = e.y; sample = e.z; sigma = __uint_as_float(e.w);
    W.kind = (fl >> 6) & 3; W.seg_inf = (fl >> 8) & 1; W.backstep = (fl >> 9) & 1; W.walk = (fl >> 10) & 3;
    W.agg = (fl >> 12) & 1; W.dleft = sigma;              // a marching lane keeps the rest of its aggressive segment in the sigma word
    rng.inc = (((((uint64_t) sample) << 32) | (uint64_t) pixel) << 1) | 1ULL;
}
template <class WalkT>
__device__ __forceinline__ void store_hot(const Params &P, uint32_t i, int st, int ev, const WalkT &W, const Rng &rng,
                                          uint32_t pixel, uint32_t sample, float sigma, uint32_t keep_bits = 0u) {
    uint4 *r = (uint4 *) (P.slots + (size_t) MER_CHK(P.chk, CHK_SLOT, i, P.nslots_all) * MER_SLOT_WORDS);
    r[0] = make_uint4(__float_as_uint(W.p.x), __float_as_uint(W.p.y), __float_as_uint(W.p.z), __float_as_uint(W.v.x));
    r[1] = make_uint4(__float_as_uint(W.v.y), __float_as_uint(W.v.z), __float_as_uint(W.opt), __float_as_uint(W.dist));
    r[2] = make_uint4(__float_as_uint(W.rem), __float_as_uint(W.hprev), __float_as_uint(W.Tr), __float_as_uint(W.t));
    const bool agg = WalkT::kBND == 1 && W.agg != 0;
    r[3] = make_uint4(__float_as_uint(W.tmax), (uint32_t) W.steps_left, pack_flags(st, ev, W.kind, W.seg_inf, W.backstep, W.walk, agg ? 1 : 0) | keep_bits,
                      (uint32_t) rng.state);
    r[4] = make_uint4((uint32_t) (rng.state >> 32), pixel, sample, __float_as_uint((WalkT::kBND == 1 && ev == EV_NONE) ? W.dleft : sigma));
}

// The cold words [H_COUNT, H_COUNT + 36) of a record as nine 16-byte pieces.  K_event used to read and write them one dword at a time: ~30 vector-memory
// instructions per visit, each of which sends 64 lanes to 64 different lines -- ~2 100 L2 requests per wave for 128 lines (the section profile of K_event,
// scratch/kevent_profile.py, had a third of a visit's time in the record load).
#define MER_COLD_WORDS 36
__device__ __forceinline__ void load_cold(const Params &P, uint32_t i, uint32_t (&cw)[MER_COLD_WORDS]) {
    const uint4 *r = (const uint4 *) (P.slots + (size_t) MER_CHK(P.chk, CHK_SLOT, i, P.nslots_all) * MER_SLOT_WORDS + H_COUNT);
#pragma unroll
    for (int k = 0; k < MER_COLD_WORDS / 4; k++) { const uint4 q = r[k]; cw[4 * k] = q.x; cw[4 * k + 1] = q.y; cw[4 * k + 2] = q.z; cw[4 * k + 3] = q.w; }
}
__device__ __forceinline__ void store_cold(const Params &P, uint32_t i, const uint32_t (&cw)[MER_COLD_WORDS]) {
    uint4 *r = (uint4 *) (P.slots + (size_t) MER_CHK(P.chk, CHK_SLOT, i, P.nslots_all) * MER_SLOT_WORDS + H_COUNT);
#pragma unroll
    for (int k = 0; k < MER_COLD_WORDS / 4; k++) r[k] = make_uint4(cw[4 * k], cw[4 * k + 1], cw[4 * k + 2], cw[4 * k + 3]);
}
#define CW(k) cw[(k) - H_COUNT]
#define CWF(k) __uint_as_float(cw[(k) - H_COUNT])

// ---- spawned side walks ---------------------------------------------------------------------------------------------------------
// The transmittance walk of a luminaire sample and the walk of an emitter look-up run to the BOUNDARY of the medium (hundreds of steps: 2 - 7
// passes each), and the path has to wait for neither: their result only scales a contribution that is already fully known when the walk starts
// (throughput x emitter x phase function x MIS weight), and since round 3 they draw from a forked sampler stream (Rng::fork), so the path's own
// draws do not depend on them.  K_event therefore hands such a walk to a SIDE-WALK SLOT -- a record of the same layout behind the path slots, 2 x MER_SIDE_PER_KIND
// per path (luminaire / look-up x three searched in rotation) -- puts it on the march list and goes straight on with the path (phase sample, Russian
// roulette, next free flight) in the same visit.  The side walk's lane marches in K_march like any other, and at the end of the walk K_event
// splats prefactor x transmittance into the film (RGB only; alpha and weight arrive once, with the path) and frees the slot.  A path thus
// advances one SCATTERING EVENT per pass instead of one walk per pass: the critical path of a render -- what its drain at the end, and a
// small render as a whole, wait for -- is ~6 x shorter.  If the side-walk slot is still busy the walk runs in the path's own lane as before:
// same streams, same contribution, so the film does not depend on which way a walk went (up to float summation order).  live[1] counts the
// side walks in flight: a render is over when every path slot is done AND that count is zero.
// writes the record of a side walk: hot words = (origin, direction, kind, forked sampler state) with the INIT flag -- K_march sets up its first
// segment (Walk::begin) --, cold words = what K_event needs when the walk has ended (film position, prefactor, the ray for a second Woodcock walk)
__device__ __forceinline__ void spawn_side_walk(const Params &P, uint32_t c, int kind, f3 o, f3 d, float rayT, f3 pref, float px, float py, uint64_t rng_state,
                                                uint32_t pixel, uint32_t sample) {
    uint32_t *rec = P.slots + (size_t) MER_CHK(P.chk, CHK_SLOT, c, P.nslots_all) * MER_SLOT_WORDS;
    uint4 *r = (uint4 *) rec;
    r[0] = make_uint4(__float_as_uint(o.x), __float_as_uint(o.y), __float_as_uint(o.z), __float_as_uint(d.x));
    r[1] = make_uint4(__float_as_uint(d.y), __float_as_uint(d.z), 0u, 0u);
    r[2] = make_uint4(0u, 0u, __float_as_uint(1.0f), __float_as_uint(rayT));
    r[3] = make_uint4(0u, 0u, pack_flags(ST_MARCH, EV_NONE, kind, 0, 0, 0, 0) | MER_FLAG_INIT | MER_FLAG_CHILD, (uint32_t) rng_state);
    r[4] = make_uint4((uint32_t) (rng_state >> 32), pixel, sample, 0u);
    uint32_t cw[MER_COLD_WORDS];
#pragma unroll
    for (int k = 0; k < MER_COLD_WORDS; k++) cw[k] = 0u;
    CW(CO_PXF) = __float_as_uint(px); CW(CO_PYF) = __float_as_uint(py);
    CW(CO_LX) = __float_as_uint(pref.x); CW(CO_LY) = __float_as_uint(pref.y); CW(CO_LZ) = __float_as_uint(pref.z);
    CW(CO_PFLAGS) = 64u /* F_CHILD */;
    CW(CO_PSX) = __float_as_uint(o.x); CW(CO_PSY) = __float_as_uint(o.y); CW(CO_PSZ) = __float_as_uint(o.z);
    CW(CO_DSX) = __float_as_uint(d.x); CW(CO_DSY) = __float_as_uint(d.y); CW(CO_DSZ) = __float_as_uint(d.z);
    CW(CO_DDX) = __float_as_uint(d.x); CW(CO_DDY) = __float_as_uint(d.y); CW(CO_DDZ) = __float_as_uint(d.z);
    CW(CO_ITST) = __float_as_uint(rayT); CW(CO_N0) = __float_as_uint(1.0f);
    store_cold(P, c, cw);           // (the words a side walk never reads are written as zeros: whole 16-byte pieces)
    (void) rec;
}
#define CSLOT(c, k) P.slots[(size_t) MER_CHK(P.chk, CHK_SLOT, c, P.nslots_all) * MER_SLOT_WORDS + (k)]
#ifndef MER_SIDE_PER_KIND
#define MER_SIDE_PER_KIND 3                                   // side-walk slots per path and kind (luminaire sample / look-up): 2 left 11 - 15 % of the walks in the path's lane, 3 leave ~2 %
#endif
enum { F_NEE_ROT = 16 /* bits 4-5 */, F_CHILD = 64, F_LK_ROT = 128 /* bits 7-8 */ };      // path flags (CO_PFLAGS) beside K_event's own: where the search for a free side-walk slot of each kind starts; the record is a side walk
// first free side-walk slot of (path i, kind k), searched from rotation r: its record id, or 0
// the state words of path i's 2 x MER_SIDE_PER_KIND side-walk slots, read in ONE round trip at the collision (the look-up's search, half a visit later, finds its
// three already in registers: a slot can only have become free in between, never busy -- the path itself is the only one that fills them)
__device__ __forceinline__ void side_flags_load(const Params &P, uint32_t i, uint32_t (&fl)[2 * MER_SIDE_PER_KIND]) {
    const uint32_t base = P.nslots + i * 2u * MER_SIDE_PER_KIND;
#pragma unroll
    for (uint32_t t = 0; t < 2u * MER_SIDE_PER_KIND; t++)
        fl[t] = P.slots[(size_t) MER_CHK(P.chk, CHK_SLOT, base + t, P.nslots_all) * MER_SLOT_WORDS + H_FLAGS];
}
__device__ __forceinline__ uint32_t free_side_slot(const Params &P, uint32_t i, uint32_t k, uint32_t r, uint32_t &r_next, const uint32_t (&fl)[2 * MER_SIDE_PER_KIND]) {
    const uint32_t base = P.nslots + (i * 2u + k) * MER_SIDE_PER_KIND;
    uint32_t found = 0u; r_next = r;
#pragma unroll
    for (uint32_t t = MER_SIDE_PER_KIND; t-- > 0u; ) {         // last to first, so that the first free slot in rotation order wins
        const uint32_t q = (r + t) % MER_SIDE_PER_KIND;
        uint32_t f = k ? fl[MER_SIDE_PER_KIND] : fl[0];
#pragma unroll
        for (uint32_t u = 1; u < MER_SIDE_PER_KIND; u++) f = q == u ? (k ? fl[MER_SIDE_PER_KIND + u] : fl[u]) : f;
        if ((f & 3u) == 0u) { found = base + q; r_next = (q + 1u) % MER_SIDE_PER_KIND; }
    }
    return found;
}

// Counter flush: one set of atomics per wave, spread over MER_COUNTER_REPLICAS copies (summed on the host) so that a
// pass of thousands of waves does not serialise on nine addresses (one word sustains ~88 atomics/us).  No barrier:
// a wave that is done must not wait for its block mates while holding registers.
__device__ __forceinline__ void flush_counters(const Params &P, const LaneCounters &C, uint32_t lane_slots) {
    const uint32_t sums[9] = {wave_sum(C.paths), wave_sum(C.steps), wave_sum(C.rif_evals), wave_sum(C.tentative),
                              wave_sum(C.real), wave_sum(C.segments), wave_sum(C.nee), wave_sum(C.marched), wave_sum(lane_slots)};
    if ((threadIdx.x & 63) == 0) {
        const uint32_t wave_id = (blockIdx.x * MER_BLOCK + threadIdx.x) >> 6;
        unsigned long long *dst = P.counters + (size_t) (wave_id % MER_COUNTER_REPLICAS) * MER_C_COUNT;
#pragma unroll
        for (int k = 0; k < 7; k++) if (sums[k]) atomicAdd(dst + k, (unsigned long long) sums[k]);
        if (sums[7]) atomicAdd(dst + MER_C_ACTIVE_LANES, (unsigned long long) sums[7]);
        if (sums[8]) atomicAdd(dst + MER_C_LOOP_ITERS, (unsigned long long) sums[8]);
    }
}

// ---------------------------------------------------------------------------------------------------
// K_gen: pixel-sample generation (SamplingIntegrator::renderBlock, src/librender/integrator.cpp:162-187) run in bulk at
// full SIMD efficiency.  A camera sample whose path ends before any marching -- it misses the medium shape, or maxDepth
// forbids entering it -- is retired here (environment radiance + film splat); a sample that will march is handed to
// K_event as a work id through the hit ring.  Without this, K_event's regeneration is a per-lane loop with a geometric
// trip count (80% of the bench scene's camera rays miss) that leaves most lanes of every wave idle.
template <bool CURVED, bool EXTRA, int BND = 0>
__global__ void __launch_bounds__(MER_BLOCK) gen_kernel(const Params P) {
    const mer_scene_desc &S = P.sc;
    const f3 env(S.env_radiance[0], S.env_radiance[1], S.env_radiance[2]);
    LaneCounters C; C.clear();
    const int maxDepth = S.max_depth;
    // one returning atomic per WAVE: it reserves 64 x gen_iters consecutive work ids (sample-major, 8x8 pixel patches)
    const int lane = threadIdx.x & 63;
    unsigned long long wave_base = 0; int go = 0;
    if (lane == 0) {
        // throttle: enough hits waiting for K_event already (the head is stable while K_gen runs)
        const unsigned long long tl = P.hitq_ctr[0], hd = P.hitq_ctr[MER_HITQ_HEAD];
        if (tl - min(tl, hd) <= P.hitq_cap / 2 && *P.work_counter < P.total_work) {
            wave_base = atomicAdd(P.work_counter, (unsigned long long) (64 * P.gen_iters));
            go = 1;
        }
    }
    go = __shfl(go, 0, 64);
    wave_base = ((unsigned long long) (uint32_t) __shfl((int) (wave_base >> 32), 0, 64) << 32) | (uint32_t) __shfl((int) (uint32_t) wave_base, 0, 64);
    for (int it = 0; go && it < P.gen_iters; ++it) {
        const uint64_t w = wave_base + (unsigned long long) it * 64ULL + (unsigned long long) lane;
        bool hit = false;
        if (w < P.total_work) {
            int x, y; uint32_t sample;
            if (P.gen_all) hit = true;
            else if (decode_work(P, w, x, y, sample)) {
                Rng rng; rng.seed(P.seed, (uint32_t) (y * S.width + x), sample);
                const float sx = rng.next1D(), sy = rng.next1D();
                const float px = (float) x + sx, py = (float) y + sy;
                f3 o, d; float mint, maxt;
                sample_ray(P, px, py, o, d, mint, maxt);
                f3 L(0, 0, 0);
                float plen = 0.0f;                                   // transient film: optical path length so far
                const float itsT = intersect_shape_b<BND>(P, o, d, mint, maxt);
                // the area emitter's rectangle in front of the medium shape (or hit instead of it): its.isEmitter() => Le, then the all-absorbing BSDF ends the path
                const float tRect = (EXTRA && P.has_area) ? rect_intersect(P, o, d, mint, maxt) : -1.0f;
                if (tRect >= 0 && (itsT < 0 || tRect < itsT)) { if (!S.hide_emitters) { L = rect_le(P, d); if (camera_edge_counts(P)) plen = edge_length(P, tRect * S.rif_const); } }
                else if (itsT < 0) { if (!S.hide_emitters) L = env; }
                else if (1 >= maxDepth && maxDepth != -1) { }
                else if (EXTRA && S.boundary_bsdf == MER_BSDF_HDIELECTRIC) hit = true;      // Fresnel sampling at the surface: K_event
                else {
                    bool medium = true;
                    if (!CURVED) { const f3 ro = o + d * itsT; medium = intersect_shape_b<BND>(P, ro, d, MER_EPSILON, MER_INF) >= 0; }
                    if (!(2 <= maxDepth || maxDepth < 0)) { }
                    else if (!medium) {
                        float extra = 0.0f;
                        if (!S.hide_emitters) L = escape_radiance<EXTRA>(P, env, o + d * itsT, d, 0.0f, extra);
                        if (camera_edge_counts(P)) plen = edge_length(P, itsT);
                        if (S.decomposition != MER_DECOMPOSITION_BOUNCE) plen += extra;
                    }
                    else hit = true;
                }
                if (!hit) {
                    C.paths++;
                    if (P.path_out) { const f3 Lm = mod_weight<EXTRA>(P, L, plen); float *q = P.path_out + MER_CHK(P.chk, CHK_PATHOUT, ((size_t) y * S.width + x) * 3, P.n_path_out - 2); q[0] = Lm.x; q[1] = Lm.y; q[2] = Lm.z; }
                    else { film_contribute(P, px, py, L, plen); film_put(P, px, py, mod_weight<EXTRA>(P, L, plen), 1.0f); }
                }
            }
        }
        // wave-aggregated push of the hits
        const unsigned long long mask = __ballot(hit);
        if (mask) {
            const int leader = __ffsll((long long) mask) - 1;
            unsigned long long base = 0;
            if (lane == leader) base = atomicAdd(P.hitq_ctr, (unsigned long long) __popcll(mask));
            base = ((unsigned long long) (uint32_t) __shfl((int) (base >> 32), leader, 64) << 32) | (uint32_t) __shfl((int) (uint32_t) base, leader, 64);
            // ring occupancy: launch_render sizes the ring for cap/2 (the throttle) + every id one launch can produce
            (void) MER_CHK(P.chk, CHK_HITQ, base + (unsigned long long) __popcll(mask) - min(base, P.hitq_ctr[MER_HITQ_HEAD]), P.hitq_cap + 1);
            if (hit) P.hitq[(base + (unsigned long long) __popcll(mask & ((1ULL << lane) - 1ULL))) & (P.hitq_cap - 1)] = w;
        }
        if (wave_base + (unsigned long long) (it + 1) * 64ULL >= P.total_work) break;
    }
    flush_counters(P, C, 0);
}

// Work lists (event queue, march lists, starved list) are MER_NSEG independent segments.  A wave appends to the segment
// picked by its own position in the grid with ONE wave-aggregated atomic -- no block barrier (a finished wave must not
// wait for its block mates while holding 90 VGPRs), and no hot word: a pass of 32 K waves spreads over 16 counters.
// Segment s only ever receives the lanes of the waves w with w % MER_NSEG == s, so its capacity is bounded.
__device__ __forceinline__ void queue_push(const SegQueue &q, uint32_t row, bool pred, uint32_t i) {
    const unsigned long long mask = __ballot(pred);
    if (mask) {
        const int lane = threadIdx.x & 63;
        const uint32_t seg = ((blockIdx.x * MER_BLOCK + threadIdx.x) >> 6) & (MER_NSEG - 1);
        const int leader = __ffsll((long long) mask) - 1;
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(q.counts + (size_t) (row & (MER_LIVE_SLOTS - 1)) * MER_NSEG + seg, (uint32_t) __popcll(mask));
        base = (uint32_t) __shfl((int) base, leader, 64);
        if (pred) q.items[(size_t) seg * q.segcap + MER_CHK(q.chk, CHK_QUEUE_SEG, base + (uint32_t) __popcll(mask & ((1ULL << lane) - 1ULL)), q.segcap)] = i;
    }
}
// Class-sorted lists: the same segments, grouped by a class of the item (MER_NSEG / NCLS segments per class), so that the
// concatenation the consumer sweeps is sorted by class.  One round trip: lane c issues the atomic of class c.
//  * event queue (NCLS = MER_EV_CLASSES): every wave but the few on a class boundary runs ONE branch of K_event's state machine
//    (collision / end of an NEE walk / end of a look-up walk / end of a free flight) instead of their union;
//  * march lists (NCLS = MER_MQ_CLASSES): lanes are grouped by the estimated number of steps until their ray leaves the shape, so
//    the lanes of a wave park at about the same trip of K_march's loop instead of idling until the slowest one has.
#define MER_EV_CLASSES 4
#define MER_MQ_CLASSES 8
// Pending connections are sorted by WHAT their next solver unit does and by HOW LONG it will run (one segment per class):
//   group 0 (classes 0 .. 19):  a shooting ray with its sensitivity matrices (Connector::computefdf: phases NEW / EVAL0 / TRIAL)
//   group 1 (classes 20 .. 25): the arc / optical length of the converged ray (path_lengths: PATHLEN, a plain Verlet march)
//   group 2 (classes 26 .. 31): the luminaire sample along the found ray (connection_value: OK -- RK4 steps + delta tracking)
// and inside a group by the predicted number of steps, longest first: the ray runs to its closest approach to the emitter, i.e. over the
// projection of the chord p1 -> p2 on its launch direction (a NEW connection draws that direction in its first unit: cos = the next number
// of the path's sampler stream, which K_event peeks without consuming).  Round 2 sorted by the chord alone: every wave then held all three
// kinds of unit and ran the three loops one after another with the other lanes masked (and a first shot, whose direction is random in the
// hemisphere about the chord, ran half as long on average as the trials beside it).
#define MER_CQ_CLASSES 32
#define MER_CQ_G0 20
#define MER_CQ_G1 6
#define MER_CQ_G2 6
__device__ __forceinline__ int connect_class(const Params &P, int group, float predicted_len) {
    const mer_scene_desc &S = P.sc;
    float diag2 = 4.0f * S.sph_radius * S.sph_radius;
    if (S.boundary != MER_BOUNDARY_SPHERE) { diag2 = 0; for (int k = 0; k < 3; k++) diag2 += (S.bmax[k] - S.bmin[k]) * (S.bmax[k] - S.bmin[k]); }
    const int n = group == 0 ? MER_CQ_G0 : (group == 1 ? MER_CQ_G1 : MER_CQ_G2), first = group == 0 ? 0 : (group == 1 ? MER_CQ_G0 : MER_CQ_G0 + MER_CQ_G1);
    const float f = fmaxf(predicted_len, 0.0f) * __builtin_amdgcn_rsqf(diag2) * (float) (2 * n);      // lengths beyond half the diagonal share the group's first class
    return first + max(0, n - 1 - (int) fminf(f, (float) n));
}
// group and predicted length of the unit a parked solver state will run next
__device__ __forceinline__ int connect_class_of(const Params &P, const ConnState &S, f3 ps) {
    const f3 d(P.sc.point_position[0] - ps.x, P.sc.point_position[1] - ps.y, P.sc.point_position[2] - ps.z);
    if (S.phase == CP_OK) return connect_class(P, 2, S.dist);
    const f3 dir = S.phase == CP_TRIAL ? S.xn : (S.phase == CP_PATHLEN ? S.dir : S.x);
    return connect_class(P, S.phase == CP_PATHLEN ? 1 : 0, dot(d, dir) * __builtin_amdgcn_rsqf(dot(dir, dir)));
}
template <int NCLS>
__device__ __forceinline__ void queue_push_class(const SegQueue &q, uint32_t row, bool pred, uint32_t i, int cls, uint32_t key = 0u) {
    constexpr uint32_t SPC = MER_NSEG / NCLS;
    static_assert(SPC >= 1 && SPC * NCLS == MER_NSEG, "classes must divide the segments");
    const unsigned long long any = __ballot(pred);
    if (any) {
        const int lane = threadIdx.x & 63;
        const uint32_t wave = (blockIdx.x * MER_BLOCK + threadIdx.x) >> 6;
        const int c = cls < 0 ? 0 : (cls >= NCLS ? NCLS - 1 : cls);
        unsigned long long mine = 0, lane_mask = 0;
#pragma unroll
        for (int k = 0; k < NCLS; k++) {
            const unsigned long long m = __ballot(pred && c == k);
            if (c == k) mine = m;
            if (lane == k) lane_mask = m;
        }
        uint32_t base = 0;
        if (lane < NCLS && lane_mask)
            base = atomicAdd(q.counts + (size_t) (row & (MER_LIVE_SLOTS - 1)) * MER_NSEG + (uint32_t) lane * SPC + (wave & (SPC - 1)), (uint32_t) __popcll(lane_mask));
        base = (uint32_t) __shfl((int) base, c, 64);
        if (pred) {
            const size_t at = (size_t) ((uint32_t) c * SPC + (wave & (SPC - 1))) * q.segcap + MER_CHK(q.chk, CHK_QUEUE_SEG, base + (uint32_t) __popcll(mine & ((1ULL << lane) - 1ULL)), q.segcap);
            q.items[at] = i;
            if (q.keys) q.keys[at] = (uint16_t) key;          // march lists under option march_sort: the cell of the lane's position (msort_cell)
        }
    }
}
// Class of a marching lane: 0 = at least one whole pass (ksteps trips) left before the ray can leave the shape, 1 .. 7 = sevenths
// of a pass, longest first (the long waves of a launch start first).  Curved rays: the chord to the boundary along the current
// direction over the step size (the bench scene's rays bend by ~0.1 rad per unit length: good to ~10 %); straight rays: the
// expected number of tentative collisions up to tmax.  The signed-distance boundary is not estimated (class 0).
// Three pushes into one class-sorted list with ONE returning atomic per class (K_event: the path and the two side walks it has just spawned join the
// same march list; three queue_push_class calls are three dependent round trips for the wave).  Inside a class segment the wave's items lie in the
// order (all of a, all of b, all of c).
template <int NCLS>
__device__ __forceinline__ void queue_push_class3(const SegQueue &q, uint32_t row, bool pa, uint32_t ia, int ca, uint32_t ka,
                                                  bool pb, uint32_t ib, int cb, uint32_t kb, bool pc, uint32_t ic, int cc, uint32_t kc) {
    constexpr uint32_t SPC = MER_NSEG / NCLS;
    if (!__ballot(pa || pb || pc)) return;
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * MER_BLOCK + threadIdx.x) >> 6, sub = wave & (SPC - 1);
    ca = ca < 0 ? 0 : (ca >= NCLS ? NCLS - 1 : ca); cb = cb < 0 ? 0 : (cb >= NCLS ? NCLS - 1 : cb); cc = cc < 0 ? 0 : (cc >= NCLS ? NCLS - 1 : cc);
    unsigned long long mine_a = 0, mine_b = 0, mine_c = 0; uint32_t na = 0, nb = 0, nc = 0;      // lane k < NCLS: the wave's counts of class k
#pragma unroll
    for (int k = 0; k < NCLS; k++) {
        const unsigned long long ma = __ballot(pa && ca == k), mb = __ballot(pb && cb == k), mc = __ballot(pc && cc == k);
        if (ca == k) mine_a = ma;
        if (cb == k) mine_b = mb;
        if (cc == k) mine_c = mc;
        if (lane == k) { na = (uint32_t) __popcll(ma); nb = (uint32_t) __popcll(mb); nc = (uint32_t) __popcll(mc); }
    }
    uint32_t base = 0;
    if (lane < NCLS && na + nb + nc) base = atomicAdd(q.counts + (size_t) (row & (MER_LIVE_SLOTS - 1)) * MER_NSEG + (uint32_t) lane * SPC + sub, na + nb + nc);
    const unsigned long long below = (1ULL << lane) - 1ULL;
    const uint32_t pos_a = (uint32_t) __shfl((int) base, ca, 64) + (uint32_t) __popcll(mine_a & below);
    const uint32_t pos_b = (uint32_t) __shfl((int) (base + na), cb, 64) + (uint32_t) __popcll(mine_b & below);
    const uint32_t pos_c = (uint32_t) __shfl((int) (base + na + nb), cc, 64) + (uint32_t) __popcll(mine_c & below);
    if (pa) { const size_t at = (size_t) ((uint32_t) ca * SPC + sub) * q.segcap + MER_CHK(q.chk, CHK_QUEUE_SEG, pos_a, q.segcap); q.items[at] = ia; if (q.keys) q.keys[at] = (uint16_t) ka; }
    if (pb) { const size_t at = (size_t) ((uint32_t) cb * SPC + sub) * q.segcap + MER_CHK(q.chk, CHK_QUEUE_SEG, pos_b, q.segcap); q.items[at] = ib; if (q.keys) q.keys[at] = (uint16_t) kb; }
    if (pc) { const size_t at = (size_t) ((uint32_t) cc * SPC + sub) * q.segcap + MER_CHK(q.chk, CHK_QUEUE_SEG, pos_c, q.segcap); q.items[at] = ic; if (q.keys) q.keys[at] = (uint16_t) kc; }
}
// cell of a position in the 2^b x 2^b x 2^b grid over the RIF's world box, in Morton order (option march_sort: b = P.msort bits per axis, at most 4)
__device__ __forceinline__ uint32_t msort_spread3(uint32_t v) { v &= 15u; return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4) | ((v & 8u) << 6); }
__device__ __forceinline__ uint32_t msort_cell(const Params &P, f3 p) {
    const int nc = 1 << P.msort;
    const int cx = min(max((int) ((p.x - P.msort_o[0]) * P.msort_s[0]), 0), nc - 1), cy = min(max((int) ((p.y - P.msort_o[1]) * P.msort_s[1]), 0), nc - 1),
              cz = min(max((int) ((p.z - P.msort_o[2]) * P.msort_s[2]), 0), nc - 1);
    return msort_spread3((uint32_t) cx) | (msort_spread3((uint32_t) cy) << 1) | (msort_spread3((uint32_t) cz) << 2);
}
template <bool CURVED, int BND, class WalkT> __device__ __forceinline__ int march_class_only(const Params &P, const WalkT &W);
/// what a lane joins the march list with: its exit-time class in bits 0-7 and, under option march_sort, the cell of its position above them
template <bool CURVED, int BND, class WalkT>
__device__ __forceinline__ int march_class(const Params &P, const WalkT &W) {
    const int c = march_class_only<CURVED, BND>(P, W);
    return (CURVED && P.msort) ? c | (int) (msort_cell(P, W.p) << 8) : c;
}
template <bool CURVED, int BND, class WalkT>
__device__ __forceinline__ int march_class_only(const Params &P, const WalkT &W) {
    if (BND != 0 || !P.mq_sort) return 0;
    const mer_scene_desc &S = P.sc;
    float r;
    if (CURVED) {
        const float vv = dot(W.v, W.v);
        float dist;
        if (S.boundary == MER_BOUNDARY_SPHERE) {
            const f3 q(W.p.x - S.sph_center[0], W.p.y - S.sph_center[1], W.p.z - S.sph_center[2]);
            const float b = dot(q, W.v) * __builtin_amdgcn_rsqf(vv), c0 = dot(q, q) - S.sph_radius * S.sph_radius;
            dist = sqrtf(fmaxf(b * b - c0, 0.0f)) - b;
        } else {
            const float tx = ((W.v.x > 0 ? S.bmax[0] : S.bmin[0]) - W.p.x) * __builtin_amdgcn_rcpf(W.v.x),
                        ty = ((W.v.y > 0 ? S.bmax[1] : S.bmin[1]) - W.p.y) * __builtin_amdgcn_rcpf(W.v.y),
                        tz = ((W.v.z > 0 ? S.bmax[2] : S.bmin[2]) - W.p.z) * __builtin_amdgcn_rcpf(W.v.z);
            dist = fminf(fminf(tx, ty), tz) * sqrtf(vv);
        }
        r = dist * __builtin_amdgcn_rcpf(S.stepsize);
    } else r = (W.tmax - W.t) * __builtin_amdgcn_rcpf(P.inv_max_density);
    // straight rays never come near ksteps trips (a crossing is ~8 tentative collisions): classes of two expected collisions each
    const float f = CURVED ? r * ((float) (MER_MQ_CLASSES - 1) * __builtin_amdgcn_rcpf((float) P.ksteps)) : 0.5f * r;
    return f >= (float) (MER_MQ_CLASSES - 1) ? 0 : (MER_MQ_CLASSES - 1) - (int) fmaxf(f, 0.0f);
}
__device__ __forceinline__ uint32_t queue_total(const SegQueue &q, uint32_t row) {
    const uint32_t *c = q.counts + (size_t) (row & (MER_LIVE_SLOTS - 1)) * MER_NSEG;
    uint32_t t = 0;
#pragma unroll
    for (int s = 0; s < MER_NSEG; s++) t += c[s];
    return t;
}
/// j-th item of the concatenated segments (j < queue_total)
__device__ __forceinline__ uint32_t queue_item(const SegQueue &q, uint32_t row, uint32_t j) {
    const uint32_t *c = q.counts + (size_t) (row & (MER_LIVE_SLOTS - 1)) * MER_NSEG;
    uint32_t seg = 0, off = j;
#pragma unroll
    for (int s = 0; s < MER_NSEG - 1; s++) { const uint32_t n = c[s]; if (seg == (uint32_t) s && off >= n) { off -= n; seg = s + 1; } }
    return q.items[(size_t) seg * q.segcap + MER_CHK(q.chk, CHK_QUEUE_ITEM, off, q.segcap)];
}
__device__ __forceinline__ void queue_clear_row(const SegQueue &q, uint32_t row, uint32_t j) {
    if (j < MER_NSEG) q.counts[(size_t) (row & (MER_LIVE_SLOTS - 1)) * MER_NSEG + j] = 0;
}

// ---------------------------------------------------------------------------------------------------
// Spatial sort of the march list (option march_sort).  K_march is bound by the rate at which the fabric serves random 128-byte lines (DESIGN
// section 4): the only way to take lines off it is to have them answered by the XCD's L2, and the L2 turns over every few microseconds, so two
// lanes share a line only if they are near one another in the volume AND resident on the same XCD at the same time.  Between K_event and K_march
// the list of a pass is therefore counting-sorted by (cell of a 2^b x 2^b x 2^b grid over the RIF's world box in Morton order, exit-time class), and
// K_march deals the sorted list to the XCDs in eight contiguous chunks (blocks b and b + 8 share an XCD): the ~41 K lanes an XCD holds at any time
// come from one neighbourhood of the volume.  Three small launches per pass: histogram (+ the gather of ids and keys), scan, scatter.
#define MER_SORT_CHUNK 2048u             // list items per block of the histogram / scatter kernels (256 threads x 8)
#define MER_SORT_MAXBINS 4096u           // 8 classes x 8^3 cells
__device__ __forceinline__ uint32_t msort_bin(const Params &P, uint32_t cell, uint32_t cls) {
    if (!P.mq_sort) return cell;                 // no exit-time classes (fields beyond 2^28 voxels): cells only
    return P.msort_major ? (cls << (3 * P.msort)) + cell : cell * MER_MQ_CLASSES + cls;
}
__device__ __forceinline__ uint32_t msort_bins(const Params &P) { return (P.mq_sort ? (uint32_t) MER_MQ_CLASSES : 1u) << (3 * P.msort); }
// (the three kernels of the sort live in mer_render.hip: one translation unit)

// ---------------------------------------------------------------------------------------------------
// K_march: the hot loop.  trace() / traceTillBoundary (heterogeneousrefractive.cpp:671-691,742-776) and the
// delta-tracking loop (heterogeneous.cpp:633-656, :562-585) for whichever ray the lane is on.
#ifndef MER_MARCH_WAVES
#define MER_MARCH_WAVES 4
#endif
#ifndef MER_ARRIVE_BATCH
#define MER_ARRIVE_BATCH 4        // power of two
#endif
template <bool CURVED, int RIF, int STEPPER, int SIGMA, int BND = 0>
__global__ void __launch_bounds__(MER_BLOCK, MER_MARCH_WAVES) march_kernel(const Params P, uint32_t pass) {
    const uint32_t j = blockIdx.x * MER_BLOCK + threadIdx.x;
    if (j >= P.nslots_all) return;
    // K_event pops the hit ring with a bare atomicAdd and overshoots its tail when it starves; nothing touches the ring
    // while K_march runs, so this is the race-free place to clamp the head before K_gen produces again
    if (j == 0) { const unsigned long long t = P.hitq_ctr[0], h = P.hitq_ctr[MER_HITQ_HEAD]; if (h > t) P.hitq_ctr[MER_HITQ_HEAD] = t; }
    // sweep the compacted list of marching slots: dense waves in the steady state and in the tail alike
    const uint32_t count = P.msort ? P.msort_count[0] : queue_total(P.mq[pass & 1u], pass);
    uint32_t jl = j;                         // position in the list
    if (P.msort) {                           // the spatially sorted list in eight contiguous chunks, one per XCD (blocks b, b + 8, ... share one): bijective over the T blocks with work
        const uint32_t T = (count + MER_BLOCK - 1) / MER_BLOCK, b = blockIdx.x, x = b & 7u, qd = T >> 3, r = T & 7u;
        jl = b < T ? ((x < r ? x * (qd + 1u) : r * (qd + 1u) + (x - r) * qd) + (b >> 3)) * MER_BLOCK + threadIdx.x : 0xffffffffu;
    }
    LaneCounters C; C.clear();
    uint32_t iters = 0, i = 0;
    bool has_event = false, still_marching = false, child_ended = false; int ev_class = 0, mq_class = 0;
    if (jl < count) {
        i = P.msort ? P.msorted[jl] : queue_item(P.mq[pass & 1u], pass, jl);
        uint32_t fl;
        Walk<CURVED, RIF, STEPPER, SIGMA, BND> W;
        Rng rng; uint32_t pixel, sample; float sigma;
        load_hot(P, i, fl, W, rng, pixel, sample, sigma);
        W.cc.reset(); W.n0 = 1.0f; W.tmin = 0.0f; W.trsum = 0.0f; W.sdens = 0.0f;
        int ev = EV_NONE; sigma = 0.0f;
        // a side walk spawned by K_event arrives as (origin, direction, kind, forked sampler): its first segment is set up here, in the lean kernel
        if (CURVED && (fl & MER_FLAG_INIT)) ev = W.begin(P, rng, C, W.kind, W.p, W.v, W.t);
        const int K = ev == EV_NONE ? P.ksteps : 0;          // (a side walk whose start failed -- outside the spline's limits -- goes straight to K_event)
        if (CURVED) {
            // A lane reaches a tentative collision about once in 65 steps, so at nearly every trip ONE lane of the wave would drag all
            // 64 through the collision code (sigma_t fetch, ratio / Woodcock test, next exponential segment: ~125 VALU against ~320 for
            // the step itself).  Arrived lanes wait instead, and the wave resolves them together every MER_ARRIVE_BATCH trips: ~2 % of
            // the lane-steps idle for 1/MER_ARRIVE_BATCH of the collision code.  Per lane the sequence of operations is unchanged.
            bool pend = false; uint32_t nadv = 0;
            for (int k = 0; k < K; ++k) {
                if (!pend) {
                    ev = W.template advance<false>(P, rng, C); nadv++;
                    if (ev == EV_ARRIVED) { pend = true; ev = EV_NONE; }
                }
                if (((k & (MER_ARRIVE_BATCH - 1)) == MER_ARRIVE_BATCH - 1 || k == K - 1) && pend) { ev = W.on_arrived(P, rng, C, sigma); pend = false; }
                iters++;
                if (ev != EV_NONE) break;
            }
            C.marched += nadv; C.steps += nadv; C.rif_evals += nadv * evals_per_step<STEPPER>();
        } else
        for (int k = 0; k < K; ++k) {
            ev = W.advance(P, rng, C);
            if (ev == EV_ARRIVED) ev = W.on_arrived(P, rng, C, sigma);
            iters++;
            if (ev != EV_NONE) break;
        }
        // A spawned side walk that has reached its end is finished HERE when one walk is the whole estimate (ratio tracking; homogeneous sigma_t):
        // contribution = prefactor x transmittance into the film, and the slot is idle again -- no trip through the event queue and K_event.
        const bool child = CURVED && BND == 0 && (fl & MER_FLAG_CHILD) != 0u;
        if (child && ev != EV_NONE && !(SIGMA == MER_SIGMA_GRID && P.sc.tr_estimator == MER_TR_WOODCOCK2)) {
            f3 tr = SIGMA == MER_SIGMA_GRID ? f3(W.Tr, W.Tr, W.Tr) : homogeneous_transmittance(P, -W.dist);
            if (ev == EV_GATE_FAIL) tr = f3(0, 0, 0);
            const uint4 q = *(const uint4 *) (P.slots + (size_t) MER_CHK(P.chk, CHK_SLOT, i, P.nslots_all) * MER_SLOT_WORDS + CO_PXF);      // film position, prefactor x y
            const f3 c = f3(__uint_as_float(q.z), __uint_as_float(q.w), SLOTF(CO_LZ)) * tr;
            if (!is_zero(c)) film_splat(P, __uint_as_float(q.x), __uint_as_float(q.y), c, 0.0f, 0, 1);
            SLOT(H_FLAGS) = 0u;
            child_ended = true; ev = EV_NONE;
        } else
        store_hot(P, i, ST_MARCH, ev, W, rng, pixel, sample, sigma, fl & MER_FLAG_CHILD);
        has_event = ev != EV_NONE;
        ev_class = ev == EV_REAL ? 0 : (W.kind == K_NEE ? 1 : (W.kind == K_LOOKUP ? 2 : 3));
        mq_class = march_class<CURVED, BND>(P, W);
        still_marching = !has_event && !child_ended;
    }
    // compaction: lanes parked on an event go to K_event's queue (by class), the others straight to the next march list
    queue_push_class<MER_EV_CLASSES>(P.eq, pass + 1, has_event, i, ev_class);
    queue_push_class<MER_MQ_CLASSES>(P.mq[(pass + 1) & 1u], pass + 1, still_marching, i, mq_class & 255, (uint32_t) mq_class >> 8);
    if (CURVED && BND == 0) {                    // side walks finished in this launch leave the in-flight count (one atomic per wave)
        const int ended = __popcll(__ballot(child_ended));
        if ((threadIdx.x & 63) == 0 && ended) atomicAdd(P.live + 1, (uint32_t) (-ended));
    }
    // lane-slot accounting: every lane of the wave is held for as many trips as its slowest lane
    uint32_t wave_iters = iters;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) wave_iters = max(wave_iters, (uint32_t) __shfl_xor((int) wave_iters, off, 64));
    flush_counters(P, C, wave_iters);
}

// ---------------------------------------------------------------------------------------------------
// K_event: VolumetricPathTracer::Li (src/integrators/path/volpath.cpp:84-343) restricted to one convex
// index-matched shape + interior medium + constant environment emitter, with the refractive hooks of
// src/libbidir/edge.cpp:45-60,91-93 and src/libbidir/vertex.cpp:251-255, plus pixel regeneration
// (src/librender/integrator.cpp:162-187) and ImageBlock::put (include/mitsuba/render/imageblock.h:124-205).
// EXTRA: the scene has a point emitter and / or a modulated film (the plain kernel carries neither)
#ifdef MER_EVENT_WAVES
#define MER_EVENT_BOUNDS __launch_bounds__(MER_BLOCK, MER_EVENT_WAVES)
#else
#define MER_EVENT_BOUNDS __launch_bounds__(MER_BLOCK)
#endif
// INLINE (straight rays in a gridded sigma_t only): a walk is run to its end HERE instead of parking the lane for K_march.  A straight walk is ~3
// tentative collisions (a log, a sampler draw and one 32-byte cell fetch each), far less than what the hand-over costs: 80 + 216 bytes of slot
// record out, two list appends, a launch boundary, 80 + 216 bytes back in.  A lane then carries its path from the camera to the film inside one
// launch, pops the next sample from the hit ring and goes on until the ring is empty -- a persistent-lane megakernel, which for curved rays
// (hundreds of 350-instruction steps per walk) was the slowest form of all (section 4's table) and for straight rays is the fastest.  Same
// sampler draws in the same order: per-path results do not change (tested).
// Section profile of K_event (compile with -DMER_PROFILE; scratch/kevent_profile.sh): wave time between marks (s_memtime) summed per section into
// mer_prof[], read back by mer_debug_prof (mer_render_brick.hip).  Off in the product build: PROF() expands to nothing.
#ifdef MER_PROFILE
static __device__ unsigned long long mer_prof[256 * 16];      // 256 replicas (by wave id), summed by mer_debug_prof: one word sustains ~90 atomics per microsecond
#define PROF_DECL unsigned long long prof_t = __builtin_readcyclecounter(), prof_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; int prof_cur = 0;
#define PROF(k) do { const unsigned long long t_ = __builtin_readcyclecounter(); _Pragma("unroll") for (int q_ = 0; q_ < 10; q_++) prof_acc[q_] += prof_cur == q_ ? t_ - prof_t : 0ull; prof_t = t_; prof_cur = (k); } while (0)
#define PROF_FLUSH(active) do { PROF(9); if ((threadIdx.x & 63) == 0 && __ballot(active)) { unsigned long long *d_ = mer_prof + (((blockIdx.x * MER_BLOCK + threadIdx.x) >> 6) & 255u) * 16u; _Pragma("unroll") for (int q_ = 0; q_ < 10; q_++) atomicAdd(d_ + q_, prof_acc[q_]); atomicAdd(d_ + 15, 1ull); } } while (0)
#else
#define PROF_DECL
#define PROF(k) do { } while (0)
#define PROF_FLUSH(active) do { } while (0)
#endif
#ifndef MER_EVENT_WAVES
#define MER_EVENT_WAVES 2              // waves per SIMD K_event is compiled for (256 VGPR): 3 and 4 measured, see DESIGN section 4
#endif
#ifndef MER_INLINE_EVENT_WAVES
#define MER_INLINE_EVENT_WAVES 1
#endif
template <bool CURVED, int RIF, int STEPPER, int SIGMA, bool EXTRA, int BND = 0, bool INLINE = false>
__global__ void __launch_bounds__(MER_BLOCK, INLINE ? MER_INLINE_EVENT_WAVES : MER_EVENT_WAVES) event_kernel(const Params P, uint32_t pass) {
    typedef Walk<CURVED, RIF, STEPPER, SIGMA, BND> WalkT;
    const uint32_t j = blockIdx.x * MER_BLOCK + threadIdx.x;
    if (j >= P.nslots_all) return;
    // pass 0: every slot is new; later passes: the compacted list of slots K_march parked on an event
    const uint32_t nq = pass == 0 ? P.nslots : queue_total(P.eq, pass);
    const uint32_t ns = pass == 0 ? 0u : queue_total(P.sq[pass & 1u], pass);      // slots left without work last pass
    const uint32_t count = nq + ns;
    // ring hygiene: the rows K_march / K_event of pass+1 will add to
    queue_clear_row(P.eq, pass + 2, j); queue_clear_row(P.mq[pass & 1u], pass + 2, j); queue_clear_row(P.sq[pass & 1u], pass + 2, j);
    LaneCounters C; C.clear();
    PROF_DECL
    bool marching = false, starved_out = false, connecting = false; uint32_t i = 0; int mq_class = 0, cq_class = 0;
    bool child_done = false; uint32_t side_inline = 0;
    uint32_t sfl[2 * MER_SIDE_PER_KIND] = {}; bool sfl_ok = false;      // state words of the path's side-walk slots (side_flags_load)
    uint32_t child0 = 0, child1 = 0; int c0class = 0, c1class = 0;           // side walks this lane has just spawned (0 = none): they join the march list below
    constexpr bool SPAWNABLE = CURVED && !EXTRA;                             // the plain curved kernels (the bench kernels) spawn side walks when P.spawn says so
    if (j < count) {
    i = pass == 0 ? j : (j < nq ? queue_item(P.eq, pass, j) : queue_item(P.sq[pass & 1u], pass, j - nq));
    int st, ev;                                // from the record's state word (below: it arrives with the rest of the record, in one round trip)

    const mer_scene_desc &S = P.sc;
    const f3 env(S.env_radiance[0], S.env_radiance[1], S.env_radiance[2]);
    const bool dielectric = EXTRA && S.boundary_bsdf == MER_BSDF_HDIELECTRIC;
    // a dielectric boundary blocks emitter sampling and look-ups from inside: the environment is reached by refracting out
    const bool hasEnv = !is_zero(env) && !dielectric;
    const bool hasEmission = S.emission[0] != 0 || S.emission[1] != 0 || S.emission[2] != 0;
    const bool hasPoint = EXTRA && (S.point_intensity[0] != 0 || S.point_intensity[1] != 0 || S.point_intensity[2] != 0);
    const int maxDepth = S.max_depth;
    const int nwalks = (SIGMA == MER_SIGMA_GRID && S.tr_estimator == MER_TR_WOODCOCK2) ? 2 : 1;
    enum { F_SCATTERED = 1, F_EMITTED = 2, F_ITSVALID = 4, F_FORKED = 8 };      // F_FORKED: the lane's sampler is a side walk's child stream; the path's own state waits in prng
#define scattered ((flags & F_SCATTERED) != 0)
#define emitted ((flags & F_EMITTED) != 0)
#define itsValid ((flags & F_ITSVALID) != 0)
#define SET_FLAG(f, v) flags = (v) ? (flags | (f)) : (flags & ~(f))

    WalkT W; Rng rng; uint32_t pixel = 0, sample = 0;
    uint64_t prng = 0;                         // the path's sampler state while a side walk runs on its forked stream (Rng::fork)
    W.cc.reset();
    float px = 0, py = 0, sigma = 0, phasePdf = 0, itsT = 0;
    float etaPath = 1.0f;                      // relative index along the path (volpath.cpp:276; Russian roulette)
    float plen = 0, trOpt = 0;                 // transient film: optical path length sensor -> vertex; length of the last NEE / look-up walk
    f3 L(0, 0, 0), T(1, 1, 1), ps(0, 0, 0), dsave(0, 0, 1), dd(0, 0, 1), wi(0, 0, 1), trv(1, 1, 1);
    int depth = 1, flags = F_EMITTED, px_i = 0, py_i = 0;
    bool starved = false;
    {   // The whole record is read before its state word is looked at: a lane whose slot holds no path (new, starved) reads a stale or zeroed record and
        // throws it away -- one round trip for the wave instead of two (state word, then the record of the lanes that have one).
        uint32_t fl;
        load_hot(P, i, fl, W, rng, pixel, sample, sigma);
        st = fl & 3u; ev = (fl >> 2) & 15u;
        uint32_t cw[MER_COLD_WORDS];
        load_cold(P, i, cw);                       // nine 16-byte loads (was ~30 dword loads)
        px = CWF(CO_PXF); py = CWF(CO_PYF);
        L = f3(CWF(CO_LX), CWF(CO_LY), CWF(CO_LZ)); T = f3(CWF(CO_TX), CWF(CO_TY), CWF(CO_TZ));
        depth = (int) CW(CO_DEPTH); flags = (int) CW(CO_PFLAGS);
        ps = f3(CWF(CO_PSX), CWF(CO_PSY), CWF(CO_PSZ)); dsave = f3(CWF(CO_DSX), CWF(CO_DSY), CWF(CO_DSZ));
        dd = f3(CWF(CO_DDX), CWF(CO_DDY), CWF(CO_DDZ)); wi = f3(CWF(CO_WIX), CWF(CO_WIY), CWF(CO_WIZ));
        phasePdf = CWF(CO_PHASEPDF); itsT = CWF(CO_ITST); W.n0 = CWF(CO_N0); W.trsum = CWF(CO_TRSUM);
        W.sdens = CWF(CO_SDENS); W.tmin = CWF(CO_TMIN);
        plen = CWF(CO_PLEN); trOpt = CWF(CO_TROPT); if (EXTRA) etaPath = CWF(CO_ETA);
        prng = (uint64_t) CW(CO_WNEXT_LO) | ((uint64_t) CW(CO_WNEXT_HI) << 32);
    }
    if (st == ST_MARCH) {
        px_i = (int) (pixel % (uint32_t) S.width); py_i = (int) (pixel / (uint32_t) S.width);
    } else {
        W.kind = K_FREE; W.steps_left = 0; W.rem = 0; W.seg_inf = 0; W.t = 0; W.tmin = 0; W.tmax = 0; W.n0 = 1; W.dist = 0;
        W.opt = 0; W.sdens = 0; W.Tr = 1; W.trsum = 0; W.walk = 0; W.p = f3(0, 0, 0); W.v = f3(0, 0, 1); W.backstep = 0; W.hprev = 0;
        W.agg = 0; W.dleft = 0;
        rng.state = 0; rng.inc = 1;
        pixel = 0; sample = 0; prng = 0; px = 0; py = 0; sigma = 0; phasePdf = 0; itsT = 0; etaPath = 1.0f; plen = 0; trOpt = 0;
        L = f3(0, 0, 0); T = f3(1, 1, 1); ps = f3(0, 0, 0); dsave = f3(0, 0, 1); dd = f3(0, 0, 1); wi = f3(0, 0, 1);
        depth = 1; flags = F_EMITTED;
    }

    // ring tail and work counter are stable while K_event runs (K_gen is not running): read them once, and keep them on
    // cache lines of their own -- the head's line is hammered by the pop atomics
    const unsigned long long tail = P.hitq_ctr[0], work_issued = *P.work_counter;
    for (;;) {
        // ---------------------------------------------------------------- regeneration (integrator.cpp:162-187)
        if (st == ST_NEW) {
            PROF(1);
            // K_gen has already retired the camera samples that never reach the medium; what is left to do here is to
            // pop one sample that does (a work id from the hit ring) and replay its deterministic prologue
            // one returning atomic per WAVE for the lanes that regenerate in this trip (64 per-lane atomics on the one head word serialise in L2, and every
            // wave of the launch pops from it)
            // (INLINE: lanes regenerate in different trips of the loop, a group is a few lanes -- per-lane atomics measured 2 % faster there)
            unsigned long long idx;
            if (INLINE) idx = atomicAdd(P.hitq_ctr + MER_HITQ_HEAD, 1ULL);
            else {
                const unsigned long long regen = __ballot(true);
                const int rl = threadIdx.x & 63, rleader = __ffsll((long long) regen) - 1;
                unsigned long long rbase = 0;
                if (rl == rleader) rbase = atomicAdd(P.hitq_ctr + MER_HITQ_HEAD, (unsigned long long) __popcll(regen));
                rbase = ((unsigned long long) (uint32_t) __shfl((int) (rbase >> 32), rleader, 64) << 32) | (uint32_t) __shfl((int) (uint32_t) rbase, rleader, 64);
                idx = rbase + (unsigned long long) __popcll(regen & ((1ULL << rl) - 1ULL));
            }
            if (idx >= tail) {
                // nothing to pop: finished if the work counter is exhausted too, otherwise wait for the next pass
                if (work_issued >= P.total_work) st = ST_DONE; else starved = true;
                break;
            }
            const uint64_t w = P.hitq[idx & (P.hitq_cap - 1)];
            int x, y;
            decode_work(P, w, x, y, sample);
            px_i = x; py_i = y; pixel = (uint32_t) (y * S.width + x);
            rng.seed(P.seed, pixel, sample);
            const float sx = rng.next1D(), sy = rng.next1D();
            px = (float) px_i + sx; py = (float) py_i + sy;
            f3 o, d; float mint, maxt;
            sample_ray(P, px, py, o, d, mint, maxt);
            L = f3(0, 0, 0); T = f3(1, 1, 1); depth = 1; flags = F_EMITTED;
            plen = 0.0f; trOpt = 0.0f; etaPath = 1.0f;
            C.paths++;
            ev = EV_NONE;
            itsT = intersect_shape_b<BND>(P, o, d, mint, maxt);                       // rRec.rayIntersect(ray)
            const float tRect = (EXTRA && P.has_area) ? rect_intersect(P, o, d, mint, maxt) : -1.0f;
            if (tRect >= 0 && (itsT < 0 || tRect < itsT)) {                              // the area emitter's rectangle is met first (K_gen retires these; gen_all hands them over)
                if (!S.hide_emitters) { if (camera_edge_counts(P)) plen = edge_length(P, tRect * S.rif_const); const f3 Le = rect_le(P, d); L = L + mod_weight<EXTRA>(P, Le, plen); film_contribute(P, px, py, Le, plen); }
                ev = EV_PATH_DONE;
            } else
            if (itsT < 0) {
                if (!S.hide_emitters) { L = L + mod_weight<EXTRA>(P, T * env, plen); film_contribute(P, px, py, T * env, plen); }   // volpath.cpp:194-201
                ev = EV_PATH_DONE;
            } else if (depth >= maxDepth && maxDepth != -1) ev = EV_PATH_DONE;
            else if (dielectric) {
                // hdielectric boundary (N2): reflect away (the ray escapes: environment, weight 1) or refract into the medium
                if (camera_edge_counts(P)) plen += edge_length(P, itsT);
                f3 x, wo;
                if (!dielectric_event<CURVED, RIF, BND>(P, rng, o, d, itsT, false, T, etaPath, x, wo)) {
                    L = L + mod_weight<EXTRA>(P, T * env, plen); film_contribute(P, px, py, T * env, plen);
                    ev = EV_PATH_DONE;
                } else {
                    ps = x; dsave = wo;
                    if (CURVED) { itsT = 0; SET_FLAG(F_ITSVALID, true); }
                    else { itsT = intersect_shape_b<BND>(P, x, wo, MER_EPSILON, MER_INF); SET_FLAG(F_ITSVALID, itsT >= 0); }
                    ev = (CURVED || itsValid) ? EV_AFTER_LOOKUP : EV_PATH_DONE;       // Russian roulette, scattered = true, next segment
                }
            } else {
                (void) rng.next1D(); (void) rng.next1D();                      // null bsdf->sample(..., nextSample2D())
                if (camera_edge_counts(P)) plen += edge_length(P, itsT);       // the camera edge (bdpt_proc.cpp:163-176)
                const f3 ro = o + d * itsT;
                bool medium = true;
                if (CURVED) { itsT = 0; SET_FLAG(F_ITSVALID, true); }
                else { itsT = intersect_shape_b<BND>(P, ro, d, MER_EPSILON, MER_INF); SET_FLAG(F_ITSVALID, itsT >= 0); if (!itsValid) medium = false; }
                depth++;
                if (!(depth <= maxDepth || maxDepth < 0)) ev = EV_PATH_DONE;
                else if (!medium) {
                    if (!S.hide_emitters) {
                        float extra; const f3 Le = escape_radiance<EXTRA>(P, env, ro, d, 0.0f, extra);
                        const float pl = plen + (S.decomposition != MER_DECOMPOSITION_BOUNCE ? extra : 0.0f);
                        L = L + mod_weight<EXTRA>(P, T * Le, pl); film_contribute(P, px, py, T * Le, pl);
                    }
                    ev = EV_PATH_DONE;
                }
                else { C.segments++; ps = ro; dsave = d; ev = W.begin(P, rng, C, K_FREE, ro, d, itsT); }
            }
            st = ST_MARCH;
        }
        if (ev == EV_NONE) {
            if (INLINE && !CURVED) {               // the walk, inline: K_march's loop for straight rays (heterogeneous.cpp:633-656, :562-585) without its pass limit
                sigma = 0.0f;
                do {
                    ev = W.advance(P, rng, C);
                    if (ev == EV_ARRIVED) ev = W.on_arrived(P, rng, C, sigma);
                } while (ev == EV_NONE);
                continue;
            }
            break;                                 // marching again: K_march takes over
        }

        // ---------------------------------------------------------------- one event
        PROF(6);
        if (ev == EV_ARRIVED) {
            ev = W.on_arrived(P, rng, C, sigma);
        } else if (ev == EV_EXITED) {
            if (CURVED && W.kind != K_FREE) trOpt = W.opt;     // this transmittance walk reached the boundary
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
            PROF(2);
            // ---- medium interaction: volpath.cpp:104-118
            MRec m;
            finish_free_flight(P, C, W, true, sigma, m);
            bool success = true;
            if (SIGMA == MER_SIGMA_HOMOGENEOUS) {
                if (m.p.x == ps.x && m.p.y == ps.y && m.p.z == ps.z) success = false;   // no forward progress
            }
            if (!success) { ev = EV_FAIL; continue; }
            C.real++;
            plen += edge_length(P, CURVED ? m.opticalLength : m.t * S.rif_const);                     // bdpt_proc.cpp:158-176
            if (depth >= maxDepth && maxDepth != -1) { ev = EV_PATH_DONE; continue; }
            if (hasEmission && SIGMA == MER_SIGMA_GRID) {
                const f3 c = T * f3(S.emission[0], S.emission[1], S.emission[2]) * m.refRatioSq;
                L = L + mod_weight<EXTRA>(P, c, plen);
                film_contribute(P, px, py, c, plen);
            }
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
                trOpt = 0.0f;
                if (interactions != 0) {                                              // scene.cpp:619-678: one null crossing
                    float tExit = 0.0f;
                    if (!CURVED) tExit = intersect_shape_b<BND>(P, ps, dd, 0.0f, MER_INF);
                    if (tExit >= 0) {
                        itsT = tExit;
                        if (!CURVED) trOpt = tExit * S.rif_const;
                        bool spawned = false;
                        if (SPAWNABLE && P.spawn) {
                            uint32_t rn; side_flags_load(P, i, sfl); sfl_ok = true; const uint32_t c = free_side_slot(P, i, 0u, ((uint32_t) flags >> 4) & 3u, rn, sfl);
                            if (c != 0u) {                                            // a side-walk slot is free: the walk goes there, the path goes on
                                const float phaseVal = phase_eval(S.phase, S.g, wi, dd);
                                const f3 pref = T * (env / MER_INV_FOURPI) * phaseVal * mi_weight(MER_INV_FOURPI, phaseVal);
                                if (!is_zero(pref)) {
                                    spawn_side_walk(P, c, K_NEE, ps, dd, tExit, pref, px, py, rng.fork(1).state, pixel, sample);
                                    W.p = ps; W.v = dd;                                // (W is idle between the collision and the next free flight: borrowed for the class estimate)
                                    child0 = c; c0class = march_class<CURVED, BND>(P, W);
                                    flags = (int) (((uint32_t) flags & ~(3u << 4)) | (rn << 4));
                                }
                                spawned = true; ev = EV_PHASE;
                            }
                        }
                        if (!spawned) {
                        if (SPAWNABLE && P.spawn) side_inline++;
                        prng = rng.state; rng = rng.fork(1); SET_FLAG(F_FORKED, true);       // the walk runs on a child stream (the oracle's sideTransmittance)
                        ev = W.begin(P, rng, C, K_NEE, ps, dd, tExit);
                        if (ev == EV_TR_DONE) trv = (SIGMA == MER_SIGMA_GRID) ? f3(1, 1, 1) : homogeneous_transmittance(P, 0.0f - tExit);
                        }
                    } else { trv = f3(1, 1, 1); ev = EV_TR_DONE; }
                } else { trv = f3(0, 0, 0); ev = EV_TR_DONE; }
            } else ev = EV_PHASE;
        } else if (ev == EV_TR_DONE) {
            PROF(3);
            if (SPAWNABLE && (flags & F_CHILD)) {                                            // a spawned side walk has ended: its contribution, and the slot is free again
                const f3 c = L * trv;                                                        // L holds the prefactor (throughput x emitter x phase x MIS weight)
                if (!is_zero(c)) film_splat(P, px, py, c, 0.0f, 0, 1);
                child_done = true;
                break;
            }
            if (flags & F_FORKED) { rng.state = prng; SET_FLAG(F_FORKED, false); }           // the side walk is over: back on the path's own stream
            f3 tr = trv;
            if (W.kind == K_NEE) {
                const float dpdf = MER_INV_FOURPI;
                f3 value = env / dpdf;
                // the rectangle shadows the environment (Scene::evalTransmittance stops at a non-null surface); tested after the walk so that the sampler draws stay the oracle's
                if (EXTRA && P.has_area && rect_intersect(P, ps, dd, 0.0f, MER_INF) >= 0) tr = f3(0, 0, 0);
                value = value * tr;
                if (!is_zero(value)) {
                    const float phaseVal = phase_eval(S.phase, S.g, wi, dd);
                    if (phaseVal != 0) {
                        const float weight = mi_weight(dpdf, phaseVal);              // env emitter is "on surface": constant.cpp:47
                        const f3 c = T * value * phaseVal * weight;
                        L = L + mod_weight<EXTRA>(P, c, plen + edge_length(P, trOpt));
                        film_contribute(P, px, py, c, plen + edge_length(P, trOpt));
                    }
                }
                ev = EV_PHASE;
            } else {
                // emitter look-up along the phase-sampled direction: volpath.cpp:162-173,370-428
                const int maxInteractions = maxDepth - depth - 1;
                const bool blocked = (maxInteractions == 0) && (CURVED || itsValid);
                if (!blocked && !is_zero(tr)) {
                    f3 value = tr * env;
                    float emitterPdf = MER_INV_FOURPI, extra = 0.0f;
                    if (EXTRA && P.has_area) {
                        // rayIntersectAndLookForEmitter (volpath.cpp:370-428): beyond the null boundary the ray meets the rectangle or the environment
                        const float tR = rect_intersect(P, ps, dsave, 0.0f, MER_INF);
                        if (tR >= 0) { value = tr * rect_le(P, dsave); emitterPdf = rect_pdf_direct(P, dsave, tR); extra = (tR - (itsValid ? itsT : 0.0f)) * S.rif_const; }
                        else if (!hasEnv) value = f3(0, 0, 0);
                    }
                    if (!is_zero(value)) {
                        const f3 c = T * value * mi_weight(phasePdf, emitterPdf);
                        L = L + mod_weight<EXTRA>(P, c, plen + edge_length(P, trOpt + extra));
                        film_contribute(P, px, py, c, plen + edge_length(P, trOpt + extra));
                    }
                }
                ev = EV_AFTER_LOOKUP;
            }
        } else if (ev == EV_PHASE || ev == EV_PHASE2) {
            PROF(4);
            // ---- luminaire sampling of the point emitter, if any (after the environment NEE, as in the oracle's draw order).
            // Curved rays: the connection is a shooting problem of hundreds of sensitivity steps -- it gets a kernel of its
            // own (K_connect) in which every lane solves one; the path resumes at EV_PHASE2 in the next pass.
            if (EXTRA && hasPoint && ev == EV_PHASE) {
                if (CURVED) {                        // park the slot: K_connect takes the connection from here (state CP_NEW)
                    connecting = true;
                    {   // its first unit shoots along a direction whose cosine to the chord is the NEXT sampler number (Connector::uniform_sample): peek
                        Rng peek = rng; const float cosChord = peek.next1D();
                        const f3 dch(S.point_position[0] - ps.x, S.point_position[1] - ps.y, S.point_position[2] - ps.z);
                        cq_class = connect_class(P, 0, cosChord * sqrtf(dot(dch, dch)));
                    }
                    P.cstate[(size_t) MER_CHK(P.chk, CHK_SLOT, i, P.nslots) * MER_CSTATE_WORDS + MER_CSTATE_WORDS - 1] = (uint32_t) CP_NEW << 15;
                    break;
                }
                float optLen = 0.0f;
                const f3 c = T * point_nee<false, RIF, STEPPER, SIGMA, BND>(P, rng, C, ps, wi, depth, optLen);
                L = L + mod_weight<EXTRA>(P, c, plen + edge_length(P, optLen));
                film_contribute(P, px, py, c, plen + edge_length(P, optLen));
            }
            // ---- luminaire sampling of the area emitter (scene.cpp:854-874, area.cpp:162-177, shape.cpp:102-115); its MIS partner is the look-up below
            if (EXTRA && !CURVED && P.has_area && ev == EV_PHASE) {
                C.nee++;
                const int interactions = maxDepth - depth - 1;
                const float sx = rng.next1D(), sy = rng.next1D();
                f3 dvec; float dist, dpdf;
                f3 value = rect_sample_direct(P, ps, sx, sy, dvec, dist, dpdf);
                if (!is_zero(value)) {
                    const float tExit = intersect_shape_b<BND>(P, ps, dvec, 0.0f, MER_INF);      // the segment crosses the (null) boundary once on its way out
                    const bool crosses = tExit >= 0 && tExit < dist;
                    f3 trA(1, 1, 1);
                    if (crosses && interactions == 0) trA = f3(0, 0, 0);
                    else trA = straight_transmittance<SIGMA>(P, rng, C, ps, dvec, crosses ? tExit : dist);
                    value = value * trA;
                    if (!is_zero(value)) {
                        const float phaseVal = phase_eval(S.phase, S.g, wi, dvec);
                        if (phaseVal != 0) {
                            const f3 c = T * value * phaseVal * mi_weight(dpdf, phaseVal);       // on-surface emitter, solid-angle measure: phasePdf = phase->pdf = its value
                            L = L + mod_weight<EXTRA>(P, c, plen + edge_length(P, dist * S.rif_const));
                            film_contribute(P, px, py, c, plen + edge_length(P, dist * S.rif_const));
                        }
                    }
                }
            }
            // ---- phase function sampling: volpath.cpp:149-160
            const float p2x = rng.next1D(), p2y = rng.next1D();
            f3 wo;
            phase_sample(S.phase, S.g, wi, p2x, p2y, wo, phasePdf);
            dsave = wo;
            if (CURVED) { itsT = 0; SET_FLAG(F_ITSVALID, true); }
            else { itsT = intersect_shape_b<BND>(P, ps, wo, 0.0f, MER_INF); SET_FLAG(F_ITSVALID, itsT >= 0); }
            if (hasEnv || (EXTRA && P.has_area)) {
                W.kind = K_LOOKUP;
                trOpt = (!CURVED && itsValid) ? itsT * S.rif_const : 0.0f;
                if (!CURVED && !itsValid) { trv = f3(1, 1, 1); ev = EV_TR_DONE; }
                else {
                    bool spawned = false;
                    if (SPAWNABLE && P.spawn) {
                        uint32_t rn; if (!sfl_ok) side_flags_load(P, i, sfl); const uint32_t c = free_side_slot(P, i, 1u, ((uint32_t) flags >> 7) & 3u, rn, sfl);
                        if (c != 0u) {
                            const bool blocked = (maxDepth - depth - 1 == 0);                    // curved rays: the look-up always crosses the boundary once
                            const f3 pref = blocked ? f3(0, 0, 0) : T * env * mi_weight(phasePdf, MER_INV_FOURPI);
                            if (!is_zero(pref)) {
                                spawn_side_walk(P, c, K_LOOKUP, ps, wo, itsT, pref, px, py, rng.fork(2).state, pixel, sample);
                                W.p = ps; W.v = wo;
                                child1 = c; c1class = march_class<CURVED, BND>(P, W);
                                flags = (int) (((uint32_t) flags & ~(3u << 7)) | (rn << 7));
                            }
                            spawned = true; ev = EV_AFTER_LOOKUP;
                        }
                    }
                    if (!spawned) {
                    if (SPAWNABLE && P.spawn) side_inline++;
                    prng = rng.state; rng = rng.fork(2); SET_FLAG(F_FORKED, true);
                    ev = W.begin(P, rng, C, K_LOOKUP, ps, wo, itsT);
                    if (ev == EV_TR_DONE) trv = (SIGMA == MER_SIGMA_GRID) ? f3(1, 1, 1) : homogeneous_transmittance(P, 0.0f - itsT);
                    }
                }
            } else ev = EV_AFTER_LOOKUP;
        } else if (ev == EV_AFTER_LOOKUP) {
            PROF(5);
            SET_FLAG(F_EMITTED, false);                                               // ERadianceNoEmission
            ev = EV_NONE;
            bool alive = true;
            if (depth++ >= S.rr_depth) {                                              // volpath.cpp:326-336
                const float q = fminf(max3(T) * etaPath * etaPath, 0.95f);
                if (rng.next1D() >= q) { ev = EV_PATH_DONE; alive = false; }
                else T = T / q;
            }
            if (alive) {
                SET_FLAG(F_SCATTERED, true);
                if (!(depth <= maxDepth || maxDepth < 0)) ev = EV_PATH_DONE;
                else { C.segments++; ev = W.begin(P, rng, C, K_FREE, ps, dsave, itsT); }
            }
        } else if (ev == EV_FAIL) {
            // ---- no medium interaction: volpath.cpp:183-201,289-301
            MRec m;
            finish_free_flight(P, C, W, false, 0.0f, m);
            plen += edge_length(P, CURVED ? m.opticalLength : itsT * S.rif_const);
            T = T * (m.transmittance / m.pdfFailure);
            if (CURVED) { T = T * m.refRatioSq; SET_FLAG(F_ITSVALID, true); }         // edge.cpp:45-60
            ev = EV_PATH_DONE;
            if (dielectric && itsValid && !(depth >= maxDepth && maxDepth != -1)) {
                // hdielectric boundary from inside: total internal / Fresnel reflection keeps the path in the medium
                f3 ro = ps, rd = dsave; float tHit = itsT;
                if (CURVED) { ro = m.p; rd = normalize(m.d); tHit = intersect_shape_b<BND>(P, ro, rd, 0.0f, MER_INF); if (!(tHit >= 0)) tHit = 0.0f; }   // re-hit (edge.cpp:45-60)
                f3 x, wo;
                if (!dielectric_event<CURVED, RIF, BND>(P, rng, ro, rd, tHit, true, T, etaPath, x, wo)) {
                    L = L + mod_weight<EXTRA>(P, T * env, plen); film_contribute(P, px, py, T * env, plen);
                } else {
                    ps = x; dsave = wo;
                    if (CURVED) itsT = 0;
                    else { itsT = intersect_shape_b<BND>(P, x, wo, MER_EPSILON, MER_INF); SET_FLAG(F_ITSVALID, itsT >= 0); }
                    if (CURVED || itsValid) ev = EV_AFTER_LOOKUP;
                }
            } else if (dielectric) {
                if (!itsValid && emitted && (!S.hide_emitters || scattered)) { L = L + mod_weight<EXTRA>(P, T * env, plen); film_contribute(P, px, py, T * env, plen); }
            } else
            if (!itsValid) {
                if (emitted && (!S.hide_emitters || scattered)) {
                    float extra; const f3 Le = escape_radiance<EXTRA && !CURVED>(P, env, ps, dsave, 0.0f, extra);
                    const float pl = plen + (S.decomposition != MER_DECOMPOSITION_BOUNCE ? extra : 0.0f);
                    L = L + mod_weight<EXTRA>(P, T * Le, pl); film_contribute(P, px, py, T * Le, pl);
                }
            } else if (!(depth >= maxDepth && maxDepth != -1)) {
                (void) rng.next1D(); (void) rng.next1D();                             // null BSDF sample
                SET_FLAG(F_EMITTED, !scattered);
                depth++;
                if (depth <= maxDepth || maxDepth < 0)
                    if (emitted && (!S.hide_emitters || scattered)) {
                        // outside the (convex) shape now: the environment, or the area emitter's rectangle (straight rays: from the exit point ps + itsT dsave)
                        float extra; const f3 Le = escape_radiance<EXTRA && !CURVED>(P, env, ps + dsave * itsT, dsave, 0.0f, extra);
                        const float pl = plen + (S.decomposition != MER_DECOMPOSITION_BOUNCE ? extra : 0.0f);
                        L = L + mod_weight<EXTRA>(P, T * Le, pl); film_contribute(P, px, py, T * Le, pl);
                    }
            }
        } else {  // EV_PATH_DONE: ImageBlock::put (imageblock.h:124-205)
            if (P.path_out) {
                float *q = P.path_out + MER_CHK(P.chk, CHK_PATHOUT, ((size_t) py_i * S.width + px_i) * 3, P.n_path_out - 2);
                q[0] = L.x; q[1] = L.y; q[2] = L.z;
            } else film_put(P, px, py, L, 1.0f);
            st = ST_NEW;
            ev = EV_NONE;
        }
    }
#undef scattered
#undef emitted
#undef itsValid
#undef SET_FLAG

    PROF(7);
    // ---- park the lane
    if (child_done) SLOT(H_FLAGS) = 0u;                                           // side-walk slot idle again
    else if (st == ST_DONE) { SLOT(H_FLAGS) = ST_DONE; atomicAdd(P.live, 1u); }      // live[0] counts finished slots
    else if (starved) { SLOT(H_FLAGS) = ST_NEW; starved_out = true; }
    else {
        store_hot(P, i, ST_MARCH, connecting ? EV_PHASE2 : EV_NONE, W, rng, pixel, sample, 0.0f, (SPAWNABLE && (flags & F_CHILD)) ? MER_FLAG_CHILD : 0u);
        uint32_t cw[MER_COLD_WORDS];
        CW(CO_PXF) = __float_as_uint(px); CW(CO_PYF) = __float_as_uint(py);
        CW(CO_LX) = __float_as_uint(L.x); CW(CO_LY) = __float_as_uint(L.y); CW(CO_LZ) = __float_as_uint(L.z);
        CW(CO_TX) = __float_as_uint(T.x); CW(CO_TY) = __float_as_uint(T.y); CW(CO_TZ) = __float_as_uint(T.z);
        CW(CO_DEPTH) = (uint32_t) depth; CW(CO_PFLAGS) = (uint32_t) flags;
        CW(CO_PSX) = __float_as_uint(ps.x); CW(CO_PSY) = __float_as_uint(ps.y); CW(CO_PSZ) = __float_as_uint(ps.z);
        CW(CO_DSX) = __float_as_uint(dsave.x); CW(CO_DSY) = __float_as_uint(dsave.y); CW(CO_DSZ) = __float_as_uint(dsave.z);
        CW(CO_DDX) = __float_as_uint(dd.x); CW(CO_DDY) = __float_as_uint(dd.y); CW(CO_DDZ) = __float_as_uint(dd.z);
        CW(CO_WIX) = __float_as_uint(wi.x); CW(CO_WIY) = __float_as_uint(wi.y); CW(CO_WIZ) = __float_as_uint(wi.z);
        CW(CO_PHASEPDF) = __float_as_uint(phasePdf); CW(CO_ITST) = __float_as_uint(itsT);
        CW(CO_N0) = __float_as_uint(W.n0); CW(CO_TRSUM) = __float_as_uint(W.trsum);
        CW(CO_SDENS) = __float_as_uint(W.sdens); CW(CO_TMIN) = __float_as_uint(W.tmin);
        CW(CO_WNEXT_LO) = (uint32_t) prng; CW(CO_WNEXT_HI) = (uint32_t) (prng >> 32); CW(CO_WLEFT) = 0u;
        CW(CO_PLEN) = __float_as_uint(plen); CW(CO_TROPT) = __float_as_uint(trOpt); CW(CO_ETA) = EXTRA ? __float_as_uint(etaPath) : 0u;
        cw[MER_COLD_WORDS - 2] = 0u; cw[MER_COLD_WORDS - 1] = 0u;
        store_cold(P, i, cw);                      // nine 16-byte stores
        marching = !connecting;
        mq_class = march_class<CURVED, BND>(P, W);

    }
    }   // j < count
    PROF(8);
    // a new connection joins the pending ones of the next K_connect launch, grouped by the length of the rays its solver will trace
    if (EXTRA && CURVED) queue_push_class<MER_CQ_CLASSES>(pick_queue(P.cq, P.cq_row), P.cq_row, connecting, i, cq_class);
    if (!SPAWNABLE) queue_push_class<MER_MQ_CLASSES>(P.mq[pass & 1u], pass, marching, i, mq_class & 255, (uint32_t) mq_class >> 8);
    if (SPAWNABLE) {                                                           // the side walks spawned in this visit march with everybody else
        queue_push_class3<MER_MQ_CLASSES>(P.mq[pass & 1u], pass, marching, i, mq_class & 255, (uint32_t) mq_class >> 8,
                                          child0 != 0u, child0, c0class & 255, (uint32_t) c0class >> 8, child1 != 0u, child1, c1class & 255, (uint32_t) c1class >> 8);
        // live[1] = side walks in flight: one atomic per wave (spawned minus ended), not one per walk -- a single word sustains ~90 atomics per microsecond
        const int delta = __popcll(__ballot(child0 != 0u)) + __popcll(__ballot(child1 != 0u)) - __popcll(__ballot(child_done));
        if ((threadIdx.x & 63) == 0 && delta != 0) atomicAdd(P.live + 1, (uint32_t) delta);
        const uint32_t nsp = (uint32_t) (__popcll(__ballot(child0 != 0u)) + __popcll(__ballot(child1 != 0u))), nin = wave_sum(side_inline);
        if ((threadIdx.x & 63) == 0 && (nsp | nin)) {
            unsigned long long *dst = P.counters + (size_t) ((j >> 6) % MER_COUNTER_REPLICAS) * MER_C_COUNT;
            if (nsp) atomicAdd(dst + MER_C_SIDE_SPAWNED, (unsigned long long) nsp);
            if (nin) atomicAdd(dst + MER_C_SIDE_INLINE, (unsigned long long) nin);
        }
    }
    queue_push(P.sq[(pass + 1) & 1u], pass + 1, starved_out, i);
    flush_counters(P, C, 0);
    PROF_FLUSH(j < count);
}

// ---------------------------------------------------------------------------------------------------
// K_connect: curved-ray luminaire sampling of the point emitter for the slots K_event parked on a scattering event
// (Medium::eval -> makeDirectConnections, src/medium/heterogeneousrefractive.cpp:571-640,1087-1163).  One lane per pending
// connection and ONE unit of its solver per launch (Connector::unit: one traced ray + the algebra up to the next one; then the
// transmittance walk along the found ray): a connection needs 3 ... 100+ units and which needs how many cannot be known in advance,
// so unfinished connections are re-queued -- grouped by the length of the rays they trace -- and the host launches the kernel
// several times per pass.  Every launch is a dense sweep over the connections still pending; run to completion per lane instead, a
// wave idled until its slowest solve had finished.  The sampler stream of the path continues through the units in order, so the
// draws are the oracle's.
// 4 waves per SIMD (128 VGPR, ~50 spilled): measured optimum -- configs[4] 25.0 / 27.7 / 23.1 / 22.6 Mpaths/s at 3 / 4 / 5 / 6 (profiles/round2/ab_connect_waves.txt)
#ifndef MER_CONNECT_WAVES
#define MER_CONNECT_WAVES 4
#endif
#define MER_CONNECT_BOUNDS __launch_bounds__(MER_BLOCK, MER_CONNECT_WAVES)
template <int RIF, int STEPPER, int SIGMA, int BND = 0, bool XC = false>
__global__ void MER_CONNECT_BOUNDS connect_stage_kernel(const Params P, uint32_t pass) {
    constexpr bool EXTRA = true;
    const uint32_t j = blockIdx.x * MER_BLOCK + threadIdx.x;
    if (j >= P.nslots) return;
    const uint32_t row = P.cq_row;
    const SegQueue &qin = pick_queue(P.cq, row), &qout = pick_queue(P.cq, row + 1u);
    const uint32_t count = queue_total(qin, row);
    queue_clear_row(qin, row + 2, j);                    // the row launch l+1 re-queues into for launch l+2 (same list as this one's input)
    LaneCounters C; C.clear();
    uint32_t i = 0, usteps = 0; bool again = false, finished = false; int cls = 0;
    if (j < count) {
        i = queue_item(qin, row, j);
        Rng rng;
        const uint32_t pixel = SLOT(H_PIXEL), sample = SLOT(H_SAMPLE);
        rng.state = (uint64_t) SLOT(H_RNG_LO) | ((uint64_t) SLOT(H_RNG_HI) << 32);
        rng.inc = (((((uint64_t) sample) << 32) | (uint64_t) pixel) << 1) | 1ULL;
        uint32_t *cs = P.cstate + (size_t) MER_CHK(P.chk, CHK_SLOT, i, P.nslots) * MER_CSTATE_WORDS;
        ConnState S; S.load(cs);
        const f3 ps(SLOTF(CO_PSX), SLOTF(CO_PSY), SLOTF(CO_PSZ)), pp(P.sc.point_position[0], P.sc.point_position[1], P.sc.point_position[2]);
        if (S.phase == CP_OK) {
            // the connecting ray is known: transmittance along it, emitter value, phase function -- the luminaire sample of this vertex
            const f3 T(SLOTF(CO_TX), SLOTF(CO_TY), SLOTF(CO_TZ)), wi(SLOTF(CO_WIX), SLOTF(CO_WIY), SLOTF(CO_WIZ));
            const f3 c0 = T * connection_value<RIF, STEPPER, SIGMA, BND>(P, rng, C, ps, wi, S.dir, S.dist, S.weight);
            const float plen = SLOTF(CO_PLEN) + edge_length(P, S.optDist);
            film_contribute(P, SLOTF(CO_PXF), SLOTF(CO_PYF), c0, plen);
            const f3 c = mod_weight<EXTRA>(P, c0, plen);
            SLOT(CO_LX) = __float_as_uint(SLOTF(CO_LX) + c.x); SLOT(CO_LY) = __float_as_uint(SLOTF(CO_LY) + c.y); SLOT(CO_LZ) = __float_as_uint(SLOTF(CO_LZ) + c.z);
            finished = true;
        } else {
            if (S.phase == CP_NEW) { S.weight = 1.0f; C.nee++; }
            Connector<RIF, BND, XC> K(P);
            f3 rev;
            K.unit(S, ps, pp, rng, rev);
            usteps = K.nsteps;
            if (S.phase == CP_FAIL) finished = true;            // no connection: the luminaire sample is zero
            else { S.store(cs); again = true; cls = connect_class_of(P, S, ps); }
        }
        SLOT(H_RNG_LO) = (uint32_t) rng.state; SLOT(H_RNG_HI) = (uint32_t) (rng.state >> 32);
    }
    queue_push_class<MER_CQ_CLASSES>(qout, row + 1u, again, i, cls);
    queue_push_class<MER_EV_CLASSES>(P.eq, pass + 1, finished, i, 1);   // resumes at EV_PHASE2 in K_event of the next pass (with the ends of NEE walks)
    flush_counters(P, C, 0);
    {   // K_connect's own counters: units, steps, and the lane slots its waves held (every lane is held for the longest unit of its wave)
        const uint32_t units = wave_sum(j < count ? 1u : 0u), steps = wave_sum(usteps);
        uint32_t longest = usteps;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) longest = max(longest, (uint32_t) __shfl_xor((int) longest, off, 64));
        if ((threadIdx.x & 63) == 0 && units) {
            unsigned long long *dst = P.counters + (size_t) ((j >> 6) % MER_COUNTER_REPLICAS) * MER_C_COUNT;
            atomicAdd(dst + MER_C_CONNECT_UNITS, (unsigned long long) units); atomicAdd(dst + MER_C_CONNECT_STEPS, (unsigned long long) steps);
            atomicAdd(dst + MER_C_CONNECT_LANE_SLOTS, 64ull * longest);
        }
    }
}

#undef SLOT
#undef SLOTF

}  // namespace mer
