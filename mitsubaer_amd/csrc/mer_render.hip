// mer_render.hip -- host loop of the wavefront render: K_gen, K_event, [K_connect,] K_march passes over the path-state slots until no
// lane is alive.  The kernels themselves are instantiated in mer_render_<group>.hip (compiled in parallel); this file picks the
// group's function pointers and launches them.
//
// The passes of ONE pipeline are a chain of dependent launches, and K_event -- fat, 2 waves per SIMD -- leaves most of the chip idle
// while it runs.  So a render is cut into `pipes` independent pipelines: pipeline q takes the sample indices q, q + pipes, ... of the
// shard (the sharding contract of SURVEY section 8e, applied inside one GPU), has its own slots, lists, hit ring and work counter, and
// runs on its own stream; the film is shared (atomics).  One pipeline's K_event then overlaps the others' K_march.  Per-path results do
// not depend on the number of pipelines.
#include "mer_internal.hpp"
#include "mer_wavefront.hpp"
#include <chrono>

namespace mer {

static bool pick_kernels(mer_context *ctx, const mer_scene_desc *sc, bool extra, KernelSet &k) {
    const bool curved = sc->rif_mode != MER_RIF_CONST;
    const int sigma = sc->sigma_mode == MER_SIGMA_GRID ? MER_SIGMA_GRID : MER_SIGMA_HOMOGENEOUS;
    const int bnd = sc->boundary == MER_BOUNDARY_SDF ? 1 : 0;
    if (!curved) return kernels_straight(sigma, bnd, extra, k);
    const int rifk = rif_fetch_kind(ctx, sc);
    if (bnd) return kernels_sdf_curved(rifk, sc->stepper, sigma, k) || kernels_sdf_curved_records(rifk, sc->stepper, sigma, k);
    switch (rifk) {
    case RIFK_ACOUSTIC: return kernels_acoustic(sc->stepper, sigma, extra, k);
    case MER_RIF_TRILINEAR: case RIFK_DENSE_BUF: return kernels_dense(rifk, sc->stepper, sigma, extra, k);
    case RIFK_CELL8: case RIFK_CELL8_BUF: return kernels_cell8(rifk, sc->stepper, sigma, extra, k);
    case RIFK_BRICK27: case RIFK_BRICK27_BUF: return kernels_brick(rifk, sc->stepper, sigma, extra, k);
    case MER_RIF_BSPLINE3: return kernels_bspline(sc->stepper, sigma, extra, k);
    }
    return false;
}

// column rotation per tile row of the tile deal (decode_work): 0 for an unsharded film (the single-GPU work order is unchanged) and
// for option tile_deal = 0; otherwise the smallest of 3, 5, 7, 11, 13 that is coprime to the number of tile columns
int tile_skew_for(int tiles_x, int tile_count, int tile_deal) {
    if (tile_count <= 1 || tiles_x <= 1 || !tile_deal) return 0;
    for (int s : {3, 5, 7, 11, 13}) {
        int a = s % tiles_x, b = tiles_x;
        while (a) { const int t = b % a; b = a; a = t; }
        if (b == 1) return s % tiles_x;
    }
    return 1 % tiles_x;
}

__global__ void clear_side_flags_kernel(uint32_t *slots, uint32_t first, uint32_t end) {
    const uint32_t c = first + blockIdx.x * blockDim.x + threadIdx.x;
    if (c < end) slots[(size_t) c * MER_SLOT_WORDS + H_FLAGS] = 0u;
}

// counting sort of a pass's march list by (cell, exit-time class): see "Spatial sort of the march list" in mer_wavefront.hpp
// where item j of the concatenated segments of a row lives
__device__ __forceinline__ size_t msort_locate(const SegQueue &q, uint32_t row, uint32_t j, uint32_t &cls) {
    const uint32_t *c = q.counts + (size_t) (row & (MER_LIVE_SLOTS - 1)) * MER_NSEG;
    uint32_t seg = 0, off = j;
#pragma unroll
    for (int s = 0; s < MER_NSEG - 1; s++) { const uint32_t n = c[s]; if (seg == (uint32_t) s && off >= n) { off -= n; seg = s + 1; } }
    cls = seg / (MER_NSEG / MER_MQ_CLASSES);
    return (size_t) seg * q.segcap + MER_CHK(q.chk, CHK_QUEUE_ITEM, off, q.segcap);
}
__global__ void __launch_bounds__(MER_BLOCK) msort_hist_kernel(const Params P, uint32_t row) {
    __shared__ uint32_t h[MER_SORT_MAXBINS];
    const SegQueue q = pick_queue(P.mq, row);
    const uint32_t count = queue_total(q, row), base = blockIdx.x * MER_SORT_CHUNK, nb = msort_bins(P);
    if (base >= count) return;
    for (uint32_t b = threadIdx.x; b < nb; b += MER_BLOCK) h[b] = 0u;
    __syncthreads();
    for (uint32_t k = 0; k < MER_SORT_CHUNK / MER_BLOCK; k++) {
        const uint32_t j = base + k * MER_BLOCK + threadIdx.x;
        if (j < count) { uint32_t cls; const size_t at = msort_locate(q, row, j, cls); atomicAdd(&h[msort_bin(P, q.keys[at], cls)], 1u); }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nb; b += MER_BLOCK) if (h[b]) atomicAdd(P.msort_hist + b, h[b]);
}
__global__ void __launch_bounds__(1024) msort_scan_kernel(const Params P) {
    __shared__ uint32_t part[1024];
    const uint32_t nb = msort_bins(P), t = threadIdx.x;
    uint32_t loc[MER_SORT_MAXBINS / 1024], sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < MER_SORT_MAXBINS / 1024; k++) { const uint32_t b = t * (MER_SORT_MAXBINS / 1024) + k; loc[k] = sum; sum += b < nb ? P.msort_hist[b] : 0u; }
    part[t] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        const uint32_t v = t >= off ? part[t - off] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    const uint32_t excl = part[t] - sum;
#pragma unroll
    for (uint32_t k = 0; k < MER_SORT_MAXBINS / 1024; k++) { const uint32_t b = t * (MER_SORT_MAXBINS / 1024) + k; if (b < nb) { P.msort_cursor[b] = excl + loc[k]; P.msort_hist[b] = 0u; } }
    if (t == 1023) P.msort_count[0] = part[1023];
}
__global__ void __launch_bounds__(MER_BLOCK) msort_scatter_kernel(const Params P, uint32_t row) {
    __shared__ uint32_t h[MER_SORT_MAXBINS];
    const SegQueue q = pick_queue(P.mq, row);
    const uint32_t count = P.msort_count[0], base = blockIdx.x * MER_SORT_CHUNK, nb = msort_bins(P);
    if (base >= count) return;
    for (uint32_t b = threadIdx.x; b < nb; b += MER_BLOCK) h[b] = 0u;
    __syncthreads();
    uint32_t item[MER_SORT_CHUNK / MER_BLOCK], key[MER_SORT_CHUNK / MER_BLOCK], rank[MER_SORT_CHUNK / MER_BLOCK];
#pragma unroll
    for (uint32_t k = 0; k < MER_SORT_CHUNK / MER_BLOCK; k++) {
        const uint32_t j = base + k * MER_BLOCK + threadIdx.x;
        key[k] = 0xffffffffu;
        if (j < count) { uint32_t cls; const size_t at = msort_locate(q, row, j, cls); item[k] = q.items[at]; key[k] = msort_bin(P, q.keys[at], cls); rank[k] = atomicAdd(&h[key[k]], 1u); }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nb; b += MER_BLOCK) { const uint32_t c = h[b]; if (c) h[b] = atomicAdd(P.msort_cursor + b, c); }     // the block's run inside bin b
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < MER_SORT_CHUNK / MER_BLOCK; k++)
        if (key[k] != 0xffffffffu) P.msorted[MER_CHK(P.chk, CHK_QUEUE_ITEM, h[key[k]] + rank[k], P.nslots_all)] = item[k];
}

int launch_render(mer_context *ctx, const mer_scene_desc *scene, const mer_shard *shard, uint64_t seed, float *film_dev, float *path_out_dev,
                  uint64_t n_film, uint64_t n_path_out) {
    Params P;
    if (make_params(ctx, scene, P, true)) return 1;
    if (!shard || shard->spp_count < 0 || shard->spp_stride <= 0 || shard->tile_count <= 0 || shard->tile_rank < 0 ||
        shard->tile_rank >= shard->tile_count || shard->spp_begin < 0)
        return fail(ctx, "invalid shard");
    const Options &opt = ctx->opt;
    P.seed = seed;
    P.spp_begin = shard->spp_begin; P.spp_count = shard->spp_count; P.spp_stride = shard->spp_stride;
    P.tile_rank = shard->tile_rank; P.tile_count = shard->tile_count;
    P.tiles_x = (scene->width + MER_TILE - 1) / MER_TILE; P.tiles_y = (scene->height + MER_TILE - 1) / MER_TILE;
    P.tile_skew = tile_skew_for(P.tiles_x, shard->tile_count, (int) ctx->opt.tile_deal);
    const int ntiles = P.tiles_x * P.tiles_y;
    P.ntiles_mine = (ntiles - shard->tile_rank + shard->tile_count - 1) / shard->tile_count;
    P.total_work = (uint64_t) P.ntiles_mine * MER_TILE * MER_TILE * (uint64_t) shard->spp_count;
    P.film = film_dev; P.path_out = path_out_dev; P.n_film = n_film; P.n_path_out = n_path_out;
    HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (P.total_work == 0) return 0;

    const bool has_point = scene->point_intensity[0] != 0 || scene->point_intensity[1] != 0 || scene->point_intensity[2] != 0;
    const bool curved = scene->rif_mode != MER_RIF_CONST;
    // EXTRA kernels carry the point emitter, the modulated film and the dielectric boundary; the signed-distance boundary exists in
    // the EXTRA kernels only
    const bool has_area = scene->area_radiance[0] != 0 || scene->area_radiance[1] != 0 || scene->area_radiance[2] != 0;
    const bool extra = scene->boundary == MER_BOUNDARY_SDF || has_point || has_area || scene->modulation != MER_MODULATION_NONE || scene->boundary_bsdf != MER_BSDF_NULL;
    KernelSet ks{};
    if (!pick_kernels(ctx, scene, extra, ks)) {
        if (scene->boundary == MER_BOUNDARY_SDF && curved) return fail(ctx, "signed-distance boundary: no kernel for this RIF layout (MER_LAYOUT_BRICK125 is not built with it)");
        return fail(ctx, "unsupported rif_mode / stepper combination");
    }
    if (opt.lds_bricks && ks.march_lds && curved) {      // LDS staging holds 27-corner records: BRICK27, not BRICK125
        auto it = ctx->volumes.find(scene->rif);
        if (it != ctx->volumes.end() && it->second.layout == MER_LAYOUT_BRICK27) ks.march = ks.march_lds;
    }
    // straight rays through a gridded sigma_t: K_event runs every walk itself (a walk is ~3 tentative collisions: the hand-over to K_march costs
    // more than the walk).  K_march is still launched: its list is empty, and it is where the hit ring's head is clamped between passes.
    if (opt.inline_walks && ks.event_inline && !curved && scene->method != MER_METHOD_SIMPSON) ks.event = ks.event_inline;
    const bool connect_stage = has_point && curved;
    if (connect_stage) {            // an emitter outside the shape is reached through the boundary: the kernel that carries the refraction code
        bool inside = false;        // (signed-distance shapes: the plain kernel carries it too, the side is tested per connection)
        if (scene->boundary == MER_BOUNDARY_AABB) { inside = true; for (int i = 0; i < 3; i++) inside = inside && scene->point_position[i] >= scene->bmin[i] && scene->point_position[i] <= scene->bmax[i]; }
        else if (scene->boundary == MER_BOUNDARY_SPHERE) { float d2 = 0; for (int i = 0; i < 3; i++) d2 += (scene->point_position[i] - scene->sph_center[i]) * (scene->point_position[i] - scene->sph_center[i]); inside = d2 < scene->sph_radius * scene->sph_radius; }
        if (!inside) ks.connect = ks.connect_cross;
    }

    // spawned side walks (mer_wavefront.hpp): the plain curved kernels hand luminaire-sample / look-up walks to side-walk slots -- four per path, behind
    // the path slots -- when the render is a steady-state film with an environment to reach (per-path output keeps every walk in the path's own lane:
    // one value per path, written once, bit-reproducible)
    const bool has_env = scene->env_radiance[0] != 0 || scene->env_radiance[1] != 0 || scene->env_radiance[2] != 0;
    const bool spawn = opt.spawn_walks && curved && !extra && has_env && !path_out_dev && scene->decomposition == MER_DECOMPOSITION_NONE;
    const uint32_t slot_mult = spawn ? 1u + 2u * MER_SIDE_PER_KIND : 1u;
    // spatial sort of the march lists (msort_* kernels): plain curved kernels on a gridded RIF inside a box or sphere
    const int msort_bits = (curved && !extra && scene->boundary != MER_BOUNDARY_SDF && P.rif.res[0] > 1 && P.rif.wmax[0] > P.rif.wmin[0]) ? (int) opt.march_sort : 0;
    int npipes = (int) opt.pipes;
    if (shard->spp_count < npipes) npipes = std::max(1, shard->spp_count);
    uint32_t want = opt.nslots > 0 ? (uint32_t) opt.nslots : (uint32_t) ctx->prop.multiProcessorCount * 2048u * 4u;   // 4 x the resident lanes of the chip, over all pipelines
    // a small render (a tile shard of a strong-scaled job, a preview): with fewer than ~8 paths per slot the slots are never refilled and the
    // lists only get sparser; a quarter of the slots keeps the wavefront denser while it drains (+3 ... 5 % at 8 ... 32 spp, ab_adaptive_k.txt)
    if (opt.nslots == 0 && opt.small_render_slots && P.total_work / 8 < want) want = (uint32_t) std::max<uint64_t>(want / 4, P.total_work / 8);
    want = (want / (uint32_t) npipes + MER_BLOCK - 1) / MER_BLOCK * MER_BLOCK;       // per pipeline
    const int ksteps0 = (int) opt.ksteps;
    // sorting the march lists by exit time scatters the lanes of a wave over the volume: a gain while the RIF sits near the caches
    // (256^3: +8 %, 512^3: +3 %), a loss once every fetch goes to HBM (1024^3: -4 %)
    P.mq_sort = opt.mq_sort >= 0 ? (opt.mq_sort != 0) : ((int64_t) P.rif.res[0] * P.rif.res[1] * P.rif.res[2] <= ((int64_t) 1 << 28) ? 1 : 0);
    P.gen_iters = 8; P.gen_all = opt.gen_all ? 1 : 0;
    // K_connect runs one solver unit (one traced ray) per pending connection per launch: several launches per pass
    const int connect_launches = (int) opt.connect_launches;
    // K_gen: one launch = gen_blocks x 4 waves x 64 x gen_iters work ids
    const unsigned gen_blocks_max = std::max(1u, std::min(want / MER_BLOCK, 1024u));
    const unsigned long long ids_per_launch = (unsigned long long) gen_blocks_max * (MER_BLOCK / 64) * 64ull * (unsigned long long) P.gen_iters;

    Run runs[MER_MAX_PIPES];
    const auto t_start = std::chrono::steady_clock::now();
    HIP_CHECK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    for (int q = 0; q < npipes; q++) {
        Pipe &pp = ctx->pipes[q]; Run &R = runs[q];
        if (q == 0) pp.stream = ctx->stream;
        else if (!pp.own_stream) { HIP_CHECK(ctx, hipStreamCreateWithFlags(&pp.own_stream, hipStreamNonBlocking)); }
        if (q > 0) { pp.stream = pp.own_stream; HIP_CHECK(ctx, hipStreamWaitEvent(pp.stream, ctx->ev0, 0)); }   // after what the caller queued (film zeroing ...)
        // The hit ring holds what K_gen's throttle lets wait (cap / 2: a wave produces while at most that many ids are waiting) plus,
        // in the worst case, every id of ONE launch (all its waves read the tail before any of them pushes, and every camera sample
        // may hit the medium): capacity >= 2 x max(slots, ids per launch), or unread ids would be overwritten.
        unsigned long long ring = 1; while (ring < 2ull * std::max<unsigned long long>(want, ids_per_launch)) ring <<= 1;
        // event-queue segments: a segment of class c receives the lanes of the waves w = s (mod segments per class) of EVERY producer launch of
        // the row -- K_march plus connect_launches K_connect launches, each of which re-packs its pending lanes into its first waves
        auto eq_segcap = [&](uint32_t cap) { return std::min<uint64_t>((uint64_t) cap, (uint64_t) ((connect_stage ? connect_launches : 0) + 2) * (cap / (MER_NSEG / MER_EV_CLASSES))) + 256u; };
        // pp.nslots counts RECORDS (path slots + side-walk slots); the lists hold record ids and are sized by it
        if (pp.nslots < want * slot_mult || pp.hitq_cap < ring || (pp.nslots && pp.eq.segcap < eq_segcap(pp.nslots))) {                 // capacity: grows, never shrinks
            const uint32_t cap = std::max(want * slot_mult, pp.nslots);
            if (pp.slots) (void) hipFree(pp.slots);
            if (pp.hitq) (void) hipFree(pp.hitq);
            pp.slots = nullptr; pp.hitq = nullptr; pp.nslots = 0;
            HIP_CHECK(ctx, hipMalloc((void **) &pp.slots, (size_t) cap * MER_SLOT_WORDS * sizeof(uint32_t)));
            for (SegQueue *sq : {&pp.eq, &pp.mq[0], &pp.mq[1], &pp.sq[0], &pp.sq[1], &pp.cq[0], &pp.cq[1]}) {
                if (sq->items) (void) hipFree(sq->items);
                if (sq->keys) (void) hipFree(sq->keys);
                sq->items = nullptr; sq->keys = nullptr;
                sq->segcap = 2u * (cap / MER_NSEG) + 256u;          // two producer kernels may feed one segment
                if (sq == &pp.eq) sq->segcap = (uint32_t) eq_segcap(cap);                        // every lane may be of one event class
                if (sq == &pp.mq[0] || sq == &pp.mq[1]) sq->segcap = 2u * (cap / (MER_NSEG / MER_MQ_CLASSES)) + 256u;   // ... or of one march class
                if (sq == &pp.cq[0] || sq == &pp.cq[1]) sq->segcap = cap + 256u;   // every slot may be pending, in one class
                HIP_CHECK(ctx, hipMalloc((void **) &sq->items, (size_t) sq->segcap * MER_NSEG * sizeof(uint32_t)));
                if (!sq->counts) HIP_CHECK(ctx, hipMalloc((void **) &sq->counts, (size_t) MER_LIVE_SLOTS * MER_NSEG * sizeof(uint32_t)));
                sq->chk = ctx->chk;
            }
            pp.hitq_cap = std::max(ring, pp.hitq_cap);
            HIP_CHECK(ctx, hipMalloc((void **) &pp.hitq, (size_t) pp.hitq_cap * sizeof(unsigned long long)));
            pp.nslots = cap;
        }
        if (!pp.live) {
            HIP_CHECK(ctx, hipMalloc((void **) &pp.live, MER_LIVE_SLOTS * sizeof(uint32_t)));
            HIP_CHECK(ctx, hipHostMalloc((void **) &pp.host_live, 8 * sizeof(uint32_t)));
            HIP_CHECK(ctx, hipMalloc((void **) &pp.hitq_ctr, 64 * sizeof(unsigned long long)));       // [0] tail, [32] head, [48] this pipeline's work counter
            HIP_CHECK(ctx, hipEventCreateWithFlags(&pp.readback[0], hipEventDisableTiming));
            HIP_CHECK(ctx, hipEventCreateWithFlags(&pp.readback[1], hipEventDisableTiming));
            HIP_CHECK(ctx, hipEventCreateWithFlags(&pp.finished, hipEventDisableTiming));
        }
        // pipeline q's part of the shard: sample indices spp_begin + (q + k npipes) spp_stride
        R.P = P;
        R.P.spp_begin = shard->spp_begin + q * shard->spp_stride; R.P.spp_stride = shard->spp_stride * npipes;
        R.P.spp_count = (shard->spp_count - q + npipes - 1) / npipes;
        R.P.total_work = (uint64_t) P.ntiles_mine * MER_TILE * MER_TILE * (uint64_t) R.P.spp_count;
        R.nslots = want;
        const uint64_t need_slots = (R.P.total_work + MER_BLOCK - 1) / MER_BLOCK * MER_BLOCK;
        if (need_slots < R.nslots) R.nslots = (uint32_t) need_slots;
        R.P.slots = pp.slots; R.P.nslots = R.nslots; R.P.nslots_all = R.nslots * slot_mult; R.P.spawn = spawn ? 1 : 0; R.P.live = pp.live; R.P.eq = pp.eq; R.P.mq[0] = pp.mq[0]; R.P.mq[1] = pp.mq[1];
        R.P.sq[0] = pp.sq[0]; R.P.sq[1] = pp.sq[1]; R.P.cq[0] = pp.cq[0]; R.P.cq[1] = pp.cq[1];
        if (connect_stage && pp.cstate_slots < pp.nslots) {      // (never together with spawned walks: a point emitter selects the EXTRA kernels)
            if (pp.cstate) (void) hipFree(pp.cstate);
            pp.cstate = nullptr; pp.cstate_slots = 0;
            HIP_CHECK(ctx, hipMalloc((void **) &pp.cstate, (size_t) pp.nslots * MER_CSTATE_WORDS * sizeof(uint32_t)));
            pp.cstate_slots = pp.nslots;
        }
        R.P.cstate = pp.cstate;
        R.P.msort = 0; R.P.mq[0].keys = nullptr; R.P.mq[1].keys = nullptr;
        if (msort_bits) {              // spatial sort of the march list: a cell per list entry (written with the entry), the sorted list, histogram, cursors, count
            if (pp.msort_cap < pp.nslots) {
                if (pp.msort) (void) hipFree(pp.msort);
                pp.msort = nullptr; pp.msort_cap = 0;
                HIP_CHECK(ctx, hipMalloc((void **) &pp.msort, ((size_t) pp.nslots + 2 * MER_SORT_MAXBINS + 16) * sizeof(uint32_t)));
                pp.msort_cap = pp.nslots;
            }
            for (int k = 0; k < 2; k++) {
                if (!pp.mq[k].keys) HIP_CHECK(ctx, hipMalloc((void **) &pp.mq[k].keys, (size_t) pp.mq[k].segcap * MER_NSEG * sizeof(uint16_t)));
                R.P.mq[k].keys = pp.mq[k].keys;
            }
            R.P.msort = std::min(msort_bits, P.mq_sort ? 3 : 4); R.P.msort_major = (int) opt.march_sort_major;
            R.P.msorted = pp.msort;
            R.P.msort_hist = pp.msort + pp.msort_cap; R.P.msort_cursor = R.P.msort_hist + MER_SORT_MAXBINS; R.P.msort_count = R.P.msort_cursor + MER_SORT_MAXBINS;
            for (int k = 0; k < 3; k++) { R.P.msort_o[k] = P.rif.wmin[k]; const float e = P.rif.wmax[k] - P.rif.wmin[k]; R.P.msort_s[k] = e > 0 ? (float) (1 << R.P.msort) / e : 0.0f; }
            HIP_CHECK(ctx, hipMemsetAsync(R.P.msort_hist, 0, (2 * MER_SORT_MAXBINS + 16) * sizeof(uint32_t), pp.stream));
        }
        R.P.hitq = pp.hitq; R.P.hitq_cap = pp.hitq_cap; R.P.hitq_ctr = pp.hitq_ctr; R.P.work_counter = pp.hitq_ctr + 48;
        R.P.ksteps = ksteps0; R.P.cq_row = 0;
        R.done = R.P.total_work == 0;
        if (R.done) continue;
        R.blocks = R.nslots * slot_mult / MER_BLOCK;                 // the lists may hold every record
        R.alive_bound = R.nslots; R.child_bound = R.nslots * (slot_mult - 1u);
        R.gen_blocks = std::max(1u, std::min(R.nslots / MER_BLOCK, gen_blocks_max));
        HIP_CHECK(ctx, hipMemsetAsync(pp.slots, 0, (size_t) R.nslots * MER_SLOT_WORDS * sizeof(uint32_t), pp.stream));
        // side-walk records: only their state word must read "idle" (the region may hold stale records of a render with another slot count)
        if (spawn) hipLaunchKernelGGL(clear_side_flags_kernel, dim3(nblocks((int64_t) R.nslots * (slot_mult - 1u))), dim3(256), 0, pp.stream, pp.slots, R.nslots, R.nslots * slot_mult);
        HIP_CHECK(ctx, hipMemsetAsync(pp.live, 0, MER_LIVE_SLOTS * sizeof(uint32_t), pp.stream));
        for (SegQueue *sq : {&pp.eq, &pp.mq[0], &pp.mq[1], &pp.sq[0], &pp.sq[1], &pp.cq[0], &pp.cq[1]})
            HIP_CHECK(ctx, hipMemsetAsync(sq->counts, 0, (size_t) MER_LIVE_SLOTS * MER_NSEG * sizeof(uint32_t), pp.stream));
        HIP_CHECK(ctx, hipMemsetAsync(pp.hitq_ctr, 0, 64 * sizeof(unsigned long long), pp.stream));
    }

    const uint32_t check_every = (uint32_t) opt.check_every;      // passes per batch (one read-back of the finished-slot count each)
    const bool pass_events = opt.pass_events != 0;          // per-kernel timing of every pass (mer_last_render_stats)
    // one batch = check_every passes of a pipeline followed by the read-back of its finished-slot count into slot `rb`.  Two batches
    // are kept in flight per pipeline, so that a pipeline never runs dry while the host waits for another one's read-back (a
    // finished render thus carries one batch of empty passes: ~0.5 ms)
    // Grid sizes.  A list can hold every record, but a launch of R.blocks blocks costs its dispatch whether or not they find work (~0.17 ms for the
    // 14 336 blocks of a 512 K-slot pipeline with side walks: the whole duration of a pass in the drain of a render).  What the lists CAN hold is
    // known: finished path slots never come back, so from the last read-back on there are at most `alive` paths, each with its side-walk slots, plus
    // the side walks that were in flight then; and a side walk returns to K_event only under the two-walk Woodcock estimator.
    const bool side_walks_return = scene->sigma_mode == MER_SIGMA_GRID && scene->tr_estimator == MER_TR_WOODCOCK2;
    auto enqueue_batch = [&](int q, int rb) -> int {
        Pipe &pp = ctx->pipes[q]; Run &R = runs[q];
        unsigned march_blocks = R.blocks, event_blocks = R.blocks;
        if (opt.grid_fit) {
            const uint64_t records = std::min<uint64_t>((uint64_t) R.nslots * slot_mult, (uint64_t) R.alive_bound * slot_mult + R.child_bound);
            march_blocks = (unsigned) std::max<uint64_t>(1, (records + MER_BLOCK - 1) / MER_BLOCK);
            event_blocks = side_walks_return ? march_blocks : (unsigned) std::max<uint64_t>(1, ((uint64_t) R.alive_bound + MER_BLOCK - 1) / MER_BLOCK);
        }
        for (uint32_t b = 0; b < check_every; b++) {
            const uint32_t pass = R.pass;
            while (pp.pass_events.size() < (size_t) (pass + 1) * 3) {
                hipEvent_t e; HIP_CHECK(ctx, hipEventCreate(&e)); pp.pass_events.push_back(e);
            }
            if (pass_events) HIP_CHECK(ctx, hipEventRecord(pp.pass_events[pass * 3 + 0], pp.stream));
            for (int g = 0; R.work_left && g < (pass == 0 ? 6 : 1); g++) hipLaunchKernelGGL(ks.gen, dim3(R.gen_blocks), dim3(MER_BLOCK), 0, pp.stream, R.P);
            hipLaunchKernelGGL(ks.event, dim3(event_blocks), dim3(MER_BLOCK), 0, pp.stream, R.P, pass);
            for (int l = 0; connect_stage && l < connect_launches; l++) {
                hipLaunchKernelGGL(ks.connect, dim3(event_blocks), dim3(MER_BLOCK), 0, pp.stream, R.P, pass);
                R.P.cq_row++;
            }
            if (pass_events) HIP_CHECK(ctx, hipEventRecord(pp.pass_events[pass * 3 + 1], pp.stream));
            if (R.P.msort) {
                const unsigned sb = (unsigned) (((uint64_t) march_blocks * MER_BLOCK + MER_SORT_CHUNK - 1) / MER_SORT_CHUNK);
                hipLaunchKernelGGL(msort_hist_kernel, dim3(sb), dim3(MER_BLOCK), 0, pp.stream, R.P, pass);
                hipLaunchKernelGGL(msort_scan_kernel, dim3(1), dim3(1024), 0, pp.stream, R.P);
                hipLaunchKernelGGL(msort_scatter_kernel, dim3(sb), dim3(MER_BLOCK), 0, pp.stream, R.P, pass);
            }
            hipLaunchKernelGGL(ks.march, dim3(march_blocks), dim3(MER_BLOCK), (size_t) opt.march_lds_kb * 1024, pp.stream, R.P, pass);   // dynamic LDS: an occupancy cap for A/B runs
            if (pass_events) HIP_CHECK(ctx, hipEventRecord(pp.pass_events[pass * 3 + 2], pp.stream));
            R.pass++;
        }
        HIP_CHECK(ctx, hipGetLastError());
        HIP_CHECK(ctx, hipMemcpyAsync(pp.host_live + 4 * rb, pp.live, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, pp.stream));     // finished path slots, side walks in flight
        HIP_CHECK(ctx, hipMemcpyAsync(pp.host_live + 4 * rb + 2, R.P.work_counter, sizeof(unsigned long long), hipMemcpyDeviceToHost, pp.stream));
        HIP_CHECK(ctx, hipEventRecord(pp.readback[rb], pp.stream));
        return 0;
    };
    // an error while other pipelines are running: wait for what was launched before handing control (and the film) back to the caller
    auto abort_render = [&]() -> int {
        const std::string msg = ctx->error;
        for (int q = 0; q < npipes; q++) if (ctx->pipes[q].stream || q == 0) (void) hipStreamSynchronize(q == 0 ? ctx->stream : ctx->pipes[q].stream);
        ctx->error = msg;
        return 1;
    };
    for (int q = 0; q < npipes; q++) if (!runs[q].done && enqueue_batch(q, 0)) return abort_render();
    for (int q = 0; q < npipes; q++) if (!runs[q].done && enqueue_batch(q, 1)) return abort_render();
    for (;;) {
        bool any = false;
        for (int q = 0; q < npipes; q++) {
            Pipe &pp = ctx->pipes[q]; Run &R = runs[q];
            if (R.done) continue;
            any = true;
            const int rb = R.cur; R.cur ^= 1;
            if (hipEventSynchronize(pp.readback[rb]) != hipSuccess) { ctx->error = "hipEventSynchronize(readback) failed"; return abort_render(); }
            const uint32_t finished_slots = pp.host_live[4 * rb];
            if (opt.verbose >= 2)        // drain timeline: host time, pipeline, passes issued, slots finished, work ids handed out
                fprintf(stderr, "[mer] t=%8.3f ms pipe %d passes %u finished %u / %u work %llu / %llu K=%d\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(),
                        q, R.pass, finished_slots, R.nslots, *(unsigned long long *) (pp.host_live + 4 * rb + 2), (unsigned long long) R.P.total_work, R.P.ksteps);
            if (finished_slots >= R.nslots && pp.host_live[4 * rb + 1] == 0u) { R.done = true; continue; }      // every path done and no side walk in flight
            R.alive_bound = R.nslots - std::min(finished_slots, R.nslots); R.child_bound = pp.host_live[4 * rb + 1];
            R.work_left = *(unsigned long long *) (pp.host_live + 4 * rb + 2) < R.P.total_work;
            // Pass length in the tail.  The tail of a render is the serial latency of its deepest paths, and a path advances by ONE walk (free
            // flight, NEE or look-up: ~65 steps) per pass however long the pass may be, so a pass should end when most lanes have parked.
            // Rounds 1-2 LENGTHENED the passes as lanes ran out (fewer launches); with K x 32 every pass lasts as long as the longest walk in
            // flight (~1 ms) and every other lane gets one event per millisecond.  Measured (profiles/round3/ab_adaptive_k.txt): fixed K
            // 307.7 vs 286.7 Mpaths/s on the headline job, 143.7 vs 116.8 on an eighth of it, 26.3 vs 22.3 at 1024^3 x 8 spp; halving K
            // in the tail instead (mode 2) is within 1-5 % of fixed K, below it.  Default: fixed.
            if (opt.adaptive_k == 1) {
                const uint32_t alive = R.nslots - finished_slots;
                R.P.ksteps = alive < R.nslots / 64 ? ksteps0 * 32 : (alive < R.nslots / 16 ? ksteps0 * 8 : (alive < R.nslots / 4 ? ksteps0 * 2 : ksteps0));
            } else if (opt.adaptive_k == 2) {   // the opposite: the tail of a render is the serial latency of its deepest paths -- one walk (free flight, NEE or look-up) per
                // pass -- so a pass should end as soon as most lanes have parked: shorter passes once few lanes are left
                const uint32_t alive = R.nslots - finished_slots;
                R.P.ksteps = alive < R.nslots / 16 ? std::max(16, ksteps0 / 2) : ksteps0;
            }
            if (R.pass > (1u << 24)) { ctx->error = "mer_render: pass limit exceeded"; return abort_render(); }
            if (enqueue_batch(q, rb)) return abort_render();
        }
        if (!any) break;
    }
    // join: the caller's stream continues after every pipeline
    for (int q = 1; q < npipes; q++) {
        if (runs[q].pass == 0) continue;
        HIP_CHECK(ctx, hipEventRecord(ctx->pipes[q].finished, ctx->pipes[q].stream));
        HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, ctx->pipes[q].finished, 0));
    }
    HIP_CHECK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    ctx->timed = true;
    {   // per-kernel device time of this render, from HIP events on the launch streams (summed over the pipelines: with several of
        // them running side by side the sum exceeds the wall time)
        HIP_CHECK(ctx, hipEventSynchronize(ctx->ev1));
        double em = 0, mm = 0; uint32_t passes = 0;
        for (int q = 0; q < npipes; q++) {
            for (uint32_t k = 0; pass_events && k < runs[q].pass; k++) {
                float a = 0, b = 0;
                (void) hipEventElapsedTime(&a, ctx->pipes[q].pass_events[k * 3 + 0], ctx->pipes[q].pass_events[k * 3 + 1]);
                (void) hipEventElapsedTime(&b, ctx->pipes[q].pass_events[k * 3 + 1], ctx->pipes[q].pass_events[k * 3 + 2]);
                em += a; mm += b;
            }
            passes += runs[q].pass;
        }
        ctx->last_event_ms = (float) em; ctx->last_march_ms = (float) mm; ctx->last_passes = (int) passes; ctx->last_pipes = npipes;
    }
    if (opt.verbose) { float ms = 0; (void) hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1); fprintf(stderr, "[mer] wavefront: %d pipelines, %u + %u passes, K=%d, nslots=%u each, %.3f ms\n", npipes, runs[0].pass, npipes > 1 ? runs[1].pass : 0u, ksteps0, runs[0].nslots, ms); }
    return 0;
}

}  // namespace mer
