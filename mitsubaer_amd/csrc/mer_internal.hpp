// mer_internal.hpp -- host-side state of libmer.so shared by its translation units (mer_api.hip: context, volumes, film, leaf entry
// points; mer_render.hip: the wavefront host loop; mer_render_*.hip: the kernel instantiations, one group per file so that they
// compile in parallel).  Nothing here is part of the C-ABI (include/mer.h).
#pragma once
#include "mer_device.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <string>
#include <vector>
#include <map>
#include <limits>

namespace mer {

struct Volume {
    mer_grid_desc desc;
    void  *dense = nullptr;     // device, dense layout
    float *cell8 = nullptr;     // device, CELL8 / BRICK records (optional)
    float *coeff = nullptr;     // device, B-spline coefficients (optional)
    int layout = MER_LAYOUT_DENSE;
    bool owns_dense = true;
    size_t bytes_dense = 0;
};

#define MER_MAX_PIPES 4
struct Pipe {
    hipStream_t stream = nullptr, own_stream = nullptr;      // pipeline 0 runs on the context stream
    uint32_t *slots = nullptr; uint32_t nslots = 0; uint32_t *live = nullptr; uint32_t *host_live = nullptr;
    SegQueue eq{}, mq[2]{}, sq[2]{}, cq[2]{};
    uint32_t *cstate = nullptr; uint32_t cstate_slots = 0;        // parked solver states of K_connect (allocated on first use)
    unsigned long long *hitq = nullptr, *hitq_ctr = nullptr; unsigned long long hitq_cap = 0;
    hipEvent_t readback[2] = {nullptr, nullptr}, finished = nullptr;     // two batches in flight per pipeline
    std::vector<hipEvent_t> pass_events;          // 3 per pass: before K_event, between, after K_march
    uint32_t *msort = nullptr; uint32_t msort_cap = 0;            // spatial sort of the march list: sorted ids, gathered ids, keys (cap each), histogram, cursors, count
};

// Tuning / A-B switches of a context (mer_context_set_option).  Defaults are the measured optima of DESIGN.md section 4.
struct Options {
    int64_t pipes = 4;            // concurrent pipelines per render (1..MER_MAX_PIPES)
    int64_t nslots = 0;           // path-state slots over all pipelines; 0 = 4 x the resident lanes of the chip
    int64_t ksteps = 128;         // eikonal steps / tentative collisions per lane per K_march launch
    int64_t mq_sort = -1;         // march lists sorted by steps-to-boundary class: -1 = by field size, 0 / 1 = off / on
    int64_t connect_launches = 2;    // K_connect launches per pass (one solver unit per pending connection per launch; 1-3 measured equal, 6+ slower)
    int64_t adaptive_k = 0;       // pass length in the tail of a render: 0 fixed (default), 1 longer (rounds 1-2: measured a loss), 2 shorter
    int64_t pass_events = 1;      // per-pass HIP events (mer_last_render_stats)
    int64_t buffer_loads = 1;     // 0: global loads even for fields below 4 GiB (the kernels a >= 4 GiB field selects)
    int64_t gen_all = 0;          // K_gen hands every camera sample to K_event (A/B)
    int64_t prefilter = 0;        // K_prefilter form: 0 register windows, 1 one thread per line, 2 two kernels per axis, 3 strided x pass, 4 LDS x pass
    int64_t verbose = 0;
    int64_t debug_pixel = -1;
    int64_t march_lds_kb = 0;     // KiB of (unused) dynamic LDS per K_march block: caps its occupancy (160 KiB per CU; 33 -> 4 blocks, 41 -> 3) for A/B runs
    int64_t tile_deal = 1;        // tile shards dealt on diagonals (1, default) or in plain row-major round robin (0): decode_work
    int64_t inline_walks = 1;     // straight rays in a gridded sigma_t: K_event runs the walks itself (persistent lanes) instead of handing them to K_march
    int64_t spawn_walks = 1;      // curved rays, steady-state film: luminaire-sample / look-up walks run in side-walk slots while the path goes on (mer_wavefront.hpp)
    int64_t small_render_slots = 1;   // a render with few paths per slot uses fewer slots / pipelines, so that the wavefront stays full while it drains
    int64_t check_every = 4;      // passes per batch: the host reads the finished-slot count back once per batch, two batches in flight per pipeline (8 until round 3: 4 is +4 % on configs[1], +1 % on small renders, neutral on long ones)
    int64_t grid_fit = 1;         // launch grids sized by what the lists can still hold (live path slots x records per path + side walks in flight at the last read-back) instead of by every record
    int64_t march_sort = 0;       // curved rays, record layouts: march list also sorted by position (bits per axis of the cell grid, 1..3; 0 = off) and swept in XCD-contiguous chunks
    int64_t march_sort_major = 0; // bin order of that sort: 0 = cell-major (all exit-time classes of a cell together), 1 = class-major
    int64_t lds_bricks = 0;       // K_march keeps every lane's current BRICK27 record in LDS (BRICK27 below 4 GiB only; measured slower)
};

}  // namespace mer

struct mer_context {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string error;
    std::map<int, mer::Volume> volumes;
    int next_handle = 1;
    unsigned long long *counters = nullptr;      // MER_C_COUNT x replicas + work counter
    float *ftable = nullptr; int ftable_kind = -1; float ftable_param = 0;   // reconstruction-filter table on the device (33 floats) and what it holds
    unsigned long long *chk = nullptr;           // MER_BOUNDS_CHECK build: violation record (count, kind, index, limit)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    hipDeviceProp_t prop;
    mer::Options opt;
    // wavefront pipelines: path-state slots, work lists, hit ring, work counter and stream of each (launch_render)
    mer::Pipe pipes[MER_MAX_PIPES];
    int last_passes = 0, last_pipes = 1;
    float last_march_ms = 0, last_event_ms = 0;
};

#define HIP_CHECK(ctx, call)                                                                              \
    do {                                                                                                  \
        hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess) {                                                                           \
            (ctx)->error = std::string(#call) + " failed: " + hipGetErrorString(e_);                      \
            return 1;                                                                                     \
        }                                                                                                 \
    } while (0)

namespace mer {

static inline int fail(mer_context *ctx, const std::string &msg) { ctx->error = msg; return 1; }
static inline unsigned nblocks(int64_t n, int bs = 256) { return (unsigned) std::max<int64_t>(1, (n + bs - 1) / bs); }

// mer_api.hip
void fill_dgrid(const mer_context *ctx, const Volume &v, DGrid &g);
int make_params(mer_context *ctx, const mer_scene_desc *sc, Params &P, bool allow_sdf = false);
// internal fetch kind (RIFK_*) of the scene's trilinear RIF volume, or the rif_mode itself for the other modes
int rif_fetch_kind(mer_context *ctx, const mer_scene_desc *sc);

// mer_render.hip: one render = a few independent pipelines of K_gen / K_event / [K_connect] / K_march passes
struct Run {
    Params P; uint32_t nslots = 0, pass = 0; unsigned blocks = 0, gen_blocks = 0;
    int cur = 0; bool work_left = true, done = false;
    uint32_t alive_bound = 0, child_bound = 0;      // from the last read-back: path slots not yet finished (never grows), side walks then in flight
};
// the kernels of one (CURVED, RIF, STEPPER, SIGMA, BND) combination, as launchable function pointers
typedef void (*GenKernel)(const Params);
typedef void (*PassKernel)(const Params, uint32_t);
struct KernelSet { GenKernel gen; PassKernel event, march, connect, connect_cross, march_lds, event_inline; };   // event_inline: K_event that runs straight walks itself (or null)   // connect_cross: the point emitter lies outside the medium shape;
                                                                                                    // march_lds: K_march with LDS-staged BRICK27 records (or null)
// each mer_render_<group>.hip answers for the combinations it instantiates (returns false if the combination is not in its group)
bool kernels_straight(int sigma, int bnd, bool extra, KernelSet &k);
bool kernels_acoustic(int stepper, int sigma, bool extra, KernelSet &k);
bool kernels_dense(int rifk, int stepper, int sigma, bool extra, KernelSet &k);
bool kernels_cell8(int rifk, int stepper, int sigma, bool extra, KernelSet &k);
bool kernels_brick(int rifk, int stepper, int sigma, bool extra, KernelSet &k);
bool kernels_bspline(int stepper, int sigma, bool extra, KernelSet &k);
bool kernels_sdf_curved(int rifk, int stepper, int sigma, KernelSet &k);
bool kernels_sdf_curved_records(int rifk, int stepper, int sigma, KernelSet &k);      // BRICK27 (both load kinds), CELL8 with global loads
int tile_skew_for(int tiles_x, int tile_count, int tile_deal);
int launch_render(mer_context *ctx, const mer_scene_desc *scene, const mer_shard *shard, uint64_t seed, float *film_dev, float *path_out_dev,
                  uint64_t n_film, uint64_t n_path_out);

}  // namespace mer
