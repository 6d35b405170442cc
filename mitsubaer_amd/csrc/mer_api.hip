// mer_api.hip -- libmer.so: C-ABI (include/mer.h) over the gfx950 kernels in mer_kernels.hpp.
// Host side is plain HIP runtime: device memory, one stream, HIP events.  No CPU compute path exists:
// every entry point that computes launches a kernel, and fails loudly when no device is present.
#include "mer_kernels.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <string>
#include <vector>
#include <map>
#include <limits>

using namespace mer;

namespace {
thread_local std::string g_create_error;

struct Volume {
    mer_grid_desc desc;
    void  *dense = nullptr;     // device, dense layout
    float *cell8 = nullptr;     // device, CELL8 layout (optional)
    float *coeff = nullptr;     // device, B-spline coefficients (optional)
    int layout = MER_LAYOUT_DENSE;
    bool owns_dense = true;
    size_t bytes_dense = 0;
};
}  // namespace

#define MER_MAX_PIPES 4
struct Pipe {
    hipStream_t stream = nullptr, own_stream = nullptr;      // pipeline 0 runs on the context stream
    uint32_t *slots = nullptr; uint32_t nslots = 0; uint32_t *live = nullptr; uint32_t *host_live = nullptr;
    SegQueue eq{}, mq[2]{}, sq[2]{}, cq{};
    unsigned long long *hitq = nullptr, *hitq_ctr = nullptr; unsigned long long hitq_cap = 0;
    hipEvent_t readback[2] = {nullptr, nullptr}, finished = nullptr;     // two batches in flight per pipeline
    std::vector<hipEvent_t> pass_events;          // 3 per pass: before K_event, between, after K_march
};

struct mer_context {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string error;
    std::map<int, Volume> volumes;
    int next_handle = 1;
    unsigned long long *counters = nullptr;      // MER_C_COUNT + 1 (work counter)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    hipDeviceProp_t prop;
    // wavefront pipelines: path-state slots, work lists, hit ring, work counter and stream of each (launch_render)
    Pipe pipes[MER_MAX_PIPES];
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

static int fail(mer_context *ctx, const std::string &msg) { ctx->error = msg; return 1; }

// worldToGrid = scale((res-1)/extents) * translate(-min) * toWorld^-1 with toWorld = identity
// (GridDataSource::configure, src/volume/gridvolume.cpp:188-195).  Float arithmetic as in the reference.
static void fill_dgrid(const Volume &v, DGrid &g) {
    std::memset(&g, 0, sizeof(g));
    g.data = v.dense; g.cell8 = v.cell8; g.coeff = v.coeff;
    g.layout = v.cell8 ? v.layout : MER_LAYOUT_DENSE;
    g.channels = v.desc.channels; g.dtype = v.desc.dtype;
    const bool brick = v.cell8 && (v.layout == MER_LAYOUT_BRICK27 || v.layout == MER_LAYOUT_BRICK125);
    g.bshift = v.layout == MER_LAYOUT_BRICK125 ? 2 : 1; g.bw = (1 << g.bshift) + 1; g.recw = v.layout == MER_LAYOUT_BRICK125 ? 128 : 32;
    const int bc = 1 << g.bshift;                                       // ceil((res-1)/bc) bricks per axis
    g.nbx = (v.desc.res[0] - 2) / bc + 1; g.nby = (v.desc.res[1] - 2) / bc + 1;
    {
        const uint64_t bytes = !v.cell8 ? (uint64_t) v.bytes_dense
                             : brick ? (uint64_t) g.nbx * g.nby * ((v.desc.res[2] - 2) / bc + 1) * (uint64_t) g.recw * 4ull
                             : (uint64_t) (v.desc.res[0] - 1) * (v.desc.res[1] - 1) * (v.desc.res[2] - 1) * 32ull;
        g.buf_bytes = bytes < 0xFFFFFFFFull && !getenv("MER_NO_BUFFER_LOADS") ? (uint32_t) bytes : 0u;
    }
    for (int i = 0; i < 3; i++) {
        g.res[i] = v.desc.res[i];
        g.bmin[i] = v.desc.aabb_min[i]; g.bmax[i] = v.desc.aabb_max[i];
        const float extent = g.bmax[i] - g.bmin[i];
        const float s = (float) (g.res[i] - 1) / extent;
        g.s[i] = s;
        g.t[i] = s * (-g.bmin[i]);
        // SplineDataSource interpolatable limits (src/volume/splinevolume.cpp:280-281): stride = 1/xres
        const float stride = (float) (1.0 / s);
        g.lim_min[i] = g.bmin[i] + (2.0f * stride + MER_EPSILON);
        g.lim_max[i] = g.bmax[i] + (-2.0f * stride - MER_EPSILON);
    }
}

static void filter_table(int kind, float param, float *values, float &radius, float &scale) {
    // ReconstructionFilter::configure (src/libcore/rfilter.cpp:40-55), MTS_FILTER_RESOLUTION = 31
    const int RES = 31;
    radius = kind == MER_FILTER_BOX ? param + 1e-5f : 4 * param;      // box.cpp:39, gaussian.cpp:42
    float sum = 0.0f;
    for (int i = 0; i < RES; ++i) {
        const float x = (radius * i) / RES;
        float v;
        if (kind == MER_FILTER_BOX) v = std::fabs(x) <= radius ? 1.0f : 0.0f;
        else {
            const float alpha = -1.0f / (2.0f * param * param);
            v = std::max(0.0f, std::exp(alpha * x * x) - std::exp(alpha * radius * radius));
        }
        values[i] = v; sum += v;
    }
    values[RES] = 0.0f; values[RES + 1] = 0.0f;
    scale = RES / radius;
    sum *= 2 * radius / RES;
    const float normalization = 1.0f / sum;
    for (int i = 0; i < RES; ++i) values[i] *= normalization;
}

// Validate the scene the way the reference plugins' constructors / configure() do, and flatten it.
static int film_frames(mer_context *ctx, const mer_scene_desc *sc, int &frames) {
    frames = 1;
    if (sc->modulation < MER_MODULATION_NONE || sc->modulation > MER_MODULATION_DEPTHSELECTIVE)            // pathlengthsampler.cpp:33-35
        return fail(ctx, "The \"modulation\" parameter must be equal toeither \"none\", \"square\", or \"hamiltonian\", or \"mseq\", or \"depthselective\"!");
    if (sc->modulation != MER_MODULATION_NONE && sc->decomposition != MER_DECOMPOSITION_TRANSIENT)
        return fail(ctx, "film: a path-length modulation needs decomposition = transient");
    if (sc->modulation != MER_MODULATION_NONE && (!(sc->mod_lambda > 0) || sc->mod_P < 1 || sc->mod_neighbors < 0))
        return fail(ctx, "film: modulation needs lambda > 0, P >= 1, neighbors >= 0");
    if (sc->decomposition == MER_DECOMPOSITION_NONE) return 0;
    if (sc->decomposition == MER_DECOMPOSITION_TRANSIENT && sc->modulation != MER_MODULATION_NONE) return 0;  // film.cpp:76-78: one frame
    if (sc->decomposition != MER_DECOMPOSITION_TRANSIENT)
        return fail(ctx, "The \"decomposition\" parameter must be equal toeither \"none\", \"transient\", or \"bounce\"!");   // film.cpp:66-68 (bounce: not built)
    const float f = std::ceil((sc->max_bound - sc->min_bound) / sc->bin_width);                                               // film.cpp:74
    if (!(f >= 1.0f) || f > 4096.0f) return fail(ctx, "film: transient decomposition needs 1 <= ceil((maxBound-minBound)/binWidth) <= 4096 frames");
    frames = (int) f;
    return 0;
}
static int make_params(mer_context *ctx, const mer_scene_desc *sc, Params &P, bool allow_sdf = false) {
    std::memset(&P, 0, sizeof(P));
    P.sc = *sc;
    if (sc->width <= 0 || sc->height <= 0) return fail(ctx, "film: width/height must be positive");
    if (sc->rr_depth <= 0) return fail(ctx, "'rrDepth' must be set to a value greater than zero!");                 // integrator.cpp:217
    if (sc->max_depth <= 0 && sc->max_depth != -1)
        return fail(ctx, "'maxDepth' must be set to -1 (infinite) or a value greater than zero!");                  // integrator.cpp:220
    if (sc->phase == MER_PHASE_HG && (sc->g >= 1 || sc->g <= -1))
        return fail(ctx, "The asymmetry parameter must lie in the interval (-1, 1)!");                              // hg.cpp:52-53
    if (sc->sigma_mode == MER_SIGMA_GRID) {
        auto it = ctx->volumes.find(sc->density);
        if (it == ctx->volumes.end()) return fail(ctx, "No density specified!");                                    // heterogeneous.cpp:229-230
        if (it->second.desc.channels != 1) return fail(ctx, "density volume must support float lookups");           // :270
        if (it->second.layout == MER_LAYOUT_BRICK27 || it->second.layout == MER_LAYOUT_BRICK125) return fail(ctx, "the BRICK layouts are for the refractive-index field only");
        fill_dgrid(it->second, P.density);
        if (!(sc->density_scale > 0)) return fail(ctx, "heterogeneous medium: 'scale' must be positive");
        // m_maxDensity = m_scale * getMaximumFloatValue() (= 1.0 for gridvolume): heterogeneous.cpp:239-242
        P.inv_max_density = 1.0f / (sc->density_scale * 1.0f);
    }
    if (sc->albedo_mode == MER_ALBEDO_GRID) {
        auto it = ctx->volumes.find(sc->albedo_grid);
        if (it == ctx->volumes.end()) return fail(ctx, "No albedo specified!");                                     // heterogeneous.cpp:231-232
        if (it->second.desc.channels != 3) return fail(ctx, "albedo volume must support spectrum lookups");
        Volume tmp = it->second; tmp.cell8 = nullptr;
        fill_dgrid(tmp, P.albedo);
    }
    if (sc->rif_mode == MER_RIF_ACOUSTIC) {
        // acousticrifvolume: analytic, no grid (src/volume/acousticrifvolume.cpp:101-106)
        if (!(sc->stepsize > 0)) return fail(ctx, "heterogeneousrefractive: 'stepsize' must be positive");
        if (!(sc->ac_k_r > 0) || !(sc->ac_n_o > 0) || sc->ac_mode < 0 || !std::isfinite(sc->ac_n_max)) return fail(ctx, "acousticrifvolume: n_o and k_r = 2 pi freq / speed must be positive, mode non-negative");
        std::memset(&P.rif, 0, sizeof(P.rif));
        P.rif.ac_n_o = sc->ac_n_o; P.rif.ac_n_max = sc->ac_n_max; P.rif.ac_k_r = sc->ac_k_r; P.rif.ac_mode = sc->ac_mode;
        P.rif.res[0] = P.rif.res[1] = P.rif.res[2] = 2;
    } else if (sc->rif_mode != MER_RIF_CONST) {
        if (sc->rif_mode != MER_RIF_TRILINEAR && sc->rif_mode != MER_RIF_BSPLINE3) return fail(ctx, "unknown rif_mode");
        auto it = ctx->volumes.find(sc->rif);
        if (it == ctx->volumes.end()) return fail(ctx, "No RIF specified!");                                        // heterogeneousrefractive.cpp:368-369
        if (it->second.desc.channels != 1 || it->second.desc.dtype != MER_VOL_F32)
            return fail(ctx, "RIF volume must be a 1-channel float32 grid");
        if (sc->rif_mode == MER_RIF_BSPLINE3 && !it->second.coeff)
            return fail(ctx, "RIF volume has no spline coefficients (call mer_volume_build_spline)");
        if (!(sc->stepsize > 0)) return fail(ctx, "heterogeneousrefractive: 'stepsize' must be positive");
        fill_dgrid(it->second, P.rif);
        if (sc->rif_mode == MER_RIF_BSPLINE3) {
            for (int i = 0; i < 3; i++) if (P.rif.res[i] < 5) return fail(ctx, "splinevolume needs at least 5 nodes per axis");
            // the medium must lie inside the spline-safe box (gate: heterogeneousrefractive.cpp:461-466)
        }
    }
    for (int i = 0; i < 3; i++) {
        if (sc->sigma_a[i] < 0 || sc->sigma_s[i] < 0) return fail(ctx, "sigmaA / sigmaS must be non-negative");
    }
    P.sigA = f3(sc->sigma_a[0], sc->sigma_a[1], sc->sigma_a[2]);
    P.sigS = f3(sc->sigma_s[0], sc->sigma_s[1], sc->sigma_s[2]);
    P.sigT = f3(sc->sigma_a[0] + sc->sigma_s[0], sc->sigma_a[1] + sc->sigma_s[1], sc->sigma_a[2] + sc->sigma_s[2]);
    const float sT[3] = {P.sigT.x, P.sigT.y, P.sigT.z}, sS[3] = {P.sigS.x, P.sigS.y, P.sigS.z};
    // mediumSamplingWeight: homogeneous.cpp:172-190 == heterogeneousrefractive.cpp:239-255
    float w = sc->medium_sampling_weight;
    if (w == -1) {
        for (int i = 0; i < 3; ++i) {
            const float albedo = sS[i] / sT[i];
            if (albedo > w && sT[i] != 0) w = albedo;
        }
        if (w > 0) w = std::max(w, 0.5f);
    }
    P.medium_sampling_weight = w;
    P.sampling_density = 0;
    if (sc->strategy == MER_STRATEGY_SINGLE) {
        int channel = 0; float smallest = std::numeric_limits<float>::infinity();
        for (int i = 0; i < 3; ++i) if (sT[i] < smallest) { smallest = sT[i]; channel = i; }
        if (sc->channel >= 0) { if (sc->channel > 2) return fail(ctx, "channel out of range"); channel = sc->channel; }
        P.sampling_density = sT[channel];
    } else if (sc->strategy == MER_STRATEGY_MANUAL) {
        P.sampling_density = sc->sampling_density;
    } else if (sc->strategy != MER_STRATEGY_BALANCE) {
        return fail(ctx, "Specified an unknown sampling strategy");                                                 // homogeneous.cpp:226
    }
    if (sc->sigma_mode == MER_SIGMA_HOMOGENEOUS && !(sT[0] > 0 && sT[1] > 0 && sT[2] > 0) && sc->strategy == MER_STRATEGY_BALANCE)
        return fail(ctx, "homogeneous medium: sigmaT must be positive in every channel for the balance strategy");
    for (int i = 0; i < 12; i++) P.cam[i] = sc->cam_to_world[i];
    P.aspect = (float) sc->width / (float) sc->height;
    P.cot_half_fov = 1.0f / std::tan((sc->fov_x_deg / 2.0f) * (MER_PI / 180.0f));
    P.inv_res_x = 1.0f / sc->width; P.inv_res_y = 1.0f / sc->height;
    if (sc->rfilter != MER_FILTER_BOX && sc->rfilter != MER_FILTER_GAUSSIAN) return fail(ctx, "unknown reconstruction filter");
    if (!(sc->rfilter_param > 0)) return fail(ctx, "reconstruction filter radius/stddev must be positive");
    filter_table(sc->rfilter, sc->rfilter_param, P.fvalues, P.fradius, P.fscale);
    if (P.fradius > 7.0f) return fail(ctx, "reconstruction filter radius too large");
    if (sc->boundary_bsdf != MER_BSDF_NULL && sc->boundary_bsdf != MER_BSDF_HDIELECTRIC) return fail(ctx, "boundary BSDF must be null or hdielectric");
    if (film_frames(ctx, sc, P.frames)) return 1;
    P.film_ch = P.frames * 3 + 2;
    P.mod_phase = (float) (sc->mod_phase_deg * M_PI / 180);                                                   // pathlengthsampler.cpp:15
    if (sc->boundary == MER_BOUNDARY_AABB) {
        for (int i = 0; i < 3; i++) if (!(sc->bmin[i] < sc->bmax[i])) return fail(ctx, "medium shape: empty bounding box");
    } else if (sc->boundary == MER_BOUNDARY_SPHERE) {
        if (!(sc->sph_radius > 0)) return fail(ctx, "medium shape: sphere radius must be positive");
    } else if (sc->boundary == MER_BOUNDARY_SDF) {
        if (!allow_sdf) return fail(ctx, "the signed-distance boundary is known to mer_render only (leaf entry points: cube / sphere)");
        auto it = ctx->volumes.find(sc->sdf);
        if (it == ctx->volumes.end()) return fail(ctx, "heterogeneousrefractive: no sdf volume (boundary = sdf)");
        if (it->second.desc.channels != 1 || it->second.desc.dtype != MER_VOL_F32) return fail(ctx, "heterogeneousrefractive: the sdf must be a 1-channel float32 grid");
        if (it->second.layout == MER_LAYOUT_BRICK27 || it->second.layout == MER_LAYOUT_BRICK125) return fail(ctx, "the BRICK layouts are for the refractive-index field only");
        fill_dgrid(it->second, P.sdf);
        float d2 = 0; for (int i = 0; i < 3; i++) d2 += (P.sdf.bmax[i] - P.sdf.bmin[i]) * (P.sdf.bmax[i] - P.sdf.bmin[i]);
        P.sdf_eps = 1e-4f * std::sqrt(d2);
    } else return fail(ctx, "unknown medium boundary");
    if (sc->aggressive_tracing) {
        if (sc->boundary != MER_BOUNDARY_SDF) return fail(ctx, "aggressivetracing needs the signed-distance boundary (the medium's sdf volume)");
        if (sc->rif_mode == MER_RIF_CONST) return fail(ctx, "aggressivetracing is a property of curved-ray tracing (heterogeneousrefractive)");
        if (!(sc->sdf_max_error >= 0)) return fail(ctx, "aggressivetracing: sdf_max_error must be non-negative");
    }
    {
        const bool has_point = sc->point_intensity[0] != 0 || sc->point_intensity[1] != 0 || sc->point_intensity[2] != 0;
        for (int i = 0; i < 3; i++) if (sc->point_intensity[i] < 0 || sc->env_radiance[i] < 0) return fail(ctx, "emitter radiance / intensity must be non-negative");
        if (has_point && sc->rif_mode != MER_RIF_CONST) {
            // curved-ray connections are solved for end points inside the medium shape only (boundary refraction = next row N2)
            bool inside = true;
            if (sc->boundary == MER_BOUNDARY_AABB) { for (int i = 0; i < 3; i++) inside = inside && sc->point_position[i] > sc->bmin[i] && sc->point_position[i] < sc->bmax[i]; }
            else if (sc->boundary == MER_BOUNDARY_SDF) inside = true;            // not checked on the host: a connection that leaves the shape is rejected per sample
            else { float d2 = 0; for (int i = 0; i < 3; i++) d2 += (sc->point_position[i] - sc->sph_center[i]) * (sc->point_position[i] - sc->sph_center[i]); inside = d2 < sc->sph_radius * sc->sph_radius; }
            if (!inside) return fail(ctx, "heterogeneousrefractive: a point emitter must lie inside the medium shape (boundary refraction of connections is not built yet)");
        }
    }
    P.counters = ctx->counters;
    P.work_counter = ctx->counters + MER_C_COUNT * MER_COUNTER_REPLICAS;
    return 0;
}

template <typename F> static int dispatch_modes(mer_context *ctx, const mer_scene_desc *sc, F &&f) {
    const bool curved = sc->rif_mode != MER_RIF_CONST;
    const bool grid = sc->sigma_mode == MER_SIGMA_GRID;
    if (!curved) {
        if (grid) return f(std::integral_constant<bool, false>(), std::integral_constant<int, MER_RIF_TRILINEAR>(),
                           std::integral_constant<int, MER_STEP_VERLET>(), std::integral_constant<int, MER_SIGMA_GRID>(), std::integral_constant<int, 0>());
        return f(std::integral_constant<bool, false>(), std::integral_constant<int, MER_RIF_TRILINEAR>(),
                 std::integral_constant<int, MER_STEP_VERLET>(), std::integral_constant<int, MER_SIGMA_HOMOGENEOUS>(), std::integral_constant<int, 0>());
    }
    // internal fetch kind of the trilinear RIF (mer_device.hpp): layout x {global, buffer} loads
    int rifk = sc->rif_mode;
    if (sc->rif_mode == MER_RIF_TRILINEAR) {
        const Volume &rv = ctx->volumes.find(sc->rif)->second;
        DGrid tmp; fill_dgrid(rv, tmp);
        if (tmp.layout == MER_LAYOUT_BRICK27 || tmp.layout == MER_LAYOUT_BRICK125) rifk = tmp.buf_bytes ? RIFK_BRICK27_BUF : RIFK_BRICK27;
        else if (tmp.layout == MER_LAYOUT_CELL8) rifk = tmp.buf_bytes ? RIFK_CELL8_BUF : RIFK_CELL8;
        else rifk = tmp.buf_bytes ? RIFK_DENSE_BUF : MER_RIF_TRILINEAR;
    }
#define MER_CASE(R, S, G)                                                                                         \
    if (rifk == R && sc->stepper == S && (int) grid == G)                                                         \
        return f(std::integral_constant<bool, true>(), std::integral_constant<int, R>(), std::integral_constant<int, S>(), \
                 std::integral_constant<int, G>(), std::integral_constant<int, 0>());
    MER_CASE(RIFK_ACOUSTIC, MER_STEP_VERLET, 1) MER_CASE(RIFK_ACOUSTIC, MER_STEP_RK4, 1)
    MER_CASE(RIFK_ACOUSTIC, MER_STEP_VERLET, 0) MER_CASE(RIFK_ACOUSTIC, MER_STEP_RK4, 0)
    MER_CASE(MER_RIF_TRILINEAR, MER_STEP_VERLET, 1) MER_CASE(MER_RIF_TRILINEAR, MER_STEP_RK4, 1)
    MER_CASE(RIFK_DENSE_BUF, MER_STEP_VERLET, 1) MER_CASE(RIFK_DENSE_BUF, MER_STEP_RK4, 1)
    MER_CASE(RIFK_CELL8, MER_STEP_VERLET, 1) MER_CASE(RIFK_CELL8, MER_STEP_RK4, 1)
    MER_CASE(RIFK_CELL8_BUF, MER_STEP_VERLET, 1) MER_CASE(RIFK_CELL8_BUF, MER_STEP_RK4, 1)
    MER_CASE(RIFK_BRICK27_BUF, MER_STEP_VERLET, 1) MER_CASE(RIFK_BRICK27_BUF, MER_STEP_RK4, 1)
    MER_CASE(RIFK_BRICK27, MER_STEP_VERLET, 1) MER_CASE(RIFK_BRICK27, MER_STEP_RK4, 1)
    MER_CASE(RIFK_BRICK27_BUF, MER_STEP_VERLET, 0) MER_CASE(RIFK_BRICK27_BUF, MER_STEP_RK4, 0)
    MER_CASE(RIFK_BRICK27, MER_STEP_VERLET, 0) MER_CASE(RIFK_BRICK27, MER_STEP_RK4, 0)
    MER_CASE(MER_RIF_BSPLINE3, MER_STEP_VERLET, 1) MER_CASE(MER_RIF_BSPLINE3, MER_STEP_RK4, 1)
    MER_CASE(MER_RIF_TRILINEAR, MER_STEP_VERLET, 0) MER_CASE(MER_RIF_TRILINEAR, MER_STEP_RK4, 0)
    MER_CASE(RIFK_DENSE_BUF, MER_STEP_VERLET, 0) MER_CASE(RIFK_DENSE_BUF, MER_STEP_RK4, 0)
    MER_CASE(RIFK_CELL8, MER_STEP_VERLET, 0) MER_CASE(RIFK_CELL8, MER_STEP_RK4, 0)
    MER_CASE(RIFK_CELL8_BUF, MER_STEP_VERLET, 0) MER_CASE(RIFK_CELL8_BUF, MER_STEP_RK4, 0)
    MER_CASE(MER_RIF_BSPLINE3, MER_STEP_VERLET, 0) MER_CASE(MER_RIF_BSPLINE3, MER_STEP_RK4, 0)
#undef MER_CASE
    return fail(ctx, "unsupported rif_mode / stepper combination");
}
// boundary = MER_BOUNDARY_SDF: a reduced set of kernels (straight rays; dense-global / cell8-buffer trilinear and B-spline RIFs)
template <typename F> static int dispatch_modes_sdf(mer_context *ctx, const mer_scene_desc *sc, F &&f) {
    const bool curved = sc->rif_mode != MER_RIF_CONST;
    const bool grid = sc->sigma_mode == MER_SIGMA_GRID;
    typedef std::integral_constant<int, 1> B1;
    if (!curved) {
        if (grid) return f(std::integral_constant<bool, false>(), std::integral_constant<int, MER_RIF_TRILINEAR>(),
                           std::integral_constant<int, MER_STEP_VERLET>(), std::integral_constant<int, MER_SIGMA_GRID>(), B1());
        return f(std::integral_constant<bool, false>(), std::integral_constant<int, MER_RIF_TRILINEAR>(),
                 std::integral_constant<int, MER_STEP_VERLET>(), std::integral_constant<int, MER_SIGMA_HOMOGENEOUS>(), B1());
    }
    int rifk = sc->rif_mode;
    if (sc->rif_mode == MER_RIF_TRILINEAR) {
        const Volume &rv = ctx->volumes.find(sc->rif)->second;
        DGrid tmp; fill_dgrid(rv, tmp);
        if (tmp.layout == MER_LAYOUT_BRICK27 || tmp.layout == MER_LAYOUT_BRICK125) return fail(ctx, "signed-distance boundary: upload the RIF dense or cell8 (the brick layouts are not instantiated for it)");
        if (tmp.layout == MER_LAYOUT_CELL8) rifk = tmp.buf_bytes ? RIFK_CELL8_BUF : RIFK_CELL8;
        else rifk = MER_RIF_TRILINEAR;                        // dense: global loads
    }
#define MER_CASE(R, S, G)                                                                                         \
    if (rifk == R && sc->stepper == S && (int) grid == G)                                                         \
        return f(std::integral_constant<bool, true>(), std::integral_constant<int, R>(), std::integral_constant<int, S>(), \
                 std::integral_constant<int, G>(), B1());
    MER_CASE(MER_RIF_TRILINEAR, MER_STEP_VERLET, 1) MER_CASE(MER_RIF_TRILINEAR, MER_STEP_RK4, 1)
    MER_CASE(RIFK_CELL8_BUF, MER_STEP_VERLET, 1) MER_CASE(RIFK_CELL8_BUF, MER_STEP_RK4, 1)
    MER_CASE(MER_RIF_BSPLINE3, MER_STEP_VERLET, 1) MER_CASE(MER_RIF_BSPLINE3, MER_STEP_RK4, 1)
    MER_CASE(MER_RIF_TRILINEAR, MER_STEP_VERLET, 0) MER_CASE(MER_RIF_TRILINEAR, MER_STEP_RK4, 0)
    MER_CASE(RIFK_CELL8_BUF, MER_STEP_VERLET, 0) MER_CASE(RIFK_CELL8_BUF, MER_STEP_RK4, 0)
    MER_CASE(MER_RIF_BSPLINE3, MER_STEP_VERLET, 0) MER_CASE(MER_RIF_BSPLINE3, MER_STEP_RK4, 0)
#undef MER_CASE
    return fail(ctx, "signed-distance boundary: the RIF must be dense, cell8 below 4 GiB, or a B-spline volume");
}

// staging helpers for the leaf entry points -------------------------------------------------------------
struct DevBuf {
    mer_context *ctx; void *p = nullptr;
    DevBuf(mer_context *c) : ctx(c) {}
    ~DevBuf() { if (p) (void) hipFree(p); }
    int alloc(size_t bytes) { HIP_CHECK(ctx, hipMalloc(&p, bytes ? bytes : 4)); return 0; }
    int upload(const void *host, size_t bytes) {
        if (alloc(bytes)) return 1;
        if (bytes) HIP_CHECK(ctx, hipMemcpyAsync(p, host, bytes, hipMemcpyHostToDevice, ctx->stream));
        return 0;
    }
    int download(void *host, size_t bytes) {
        if (bytes) HIP_CHECK(ctx, hipMemcpyAsync(host, p, bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        return 0;
    }
    template <typename T> T *as() { return (T *) p; }
};
static inline unsigned nblocks(int64_t n, int bs = 256) { return (unsigned) std::max<int64_t>(1, (n + bs - 1) / bs); }

extern "C" {

int mer_abi_version(void) { return MER_ABI_VERSION; }

int mer_context_create(int32_t device_id, mer_context **out) {
    if (!out) return 1;
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_create_error = "mer_context_create: no HIP device available (libmer has no CPU path)";
        return 1;
    }
    if (device_id < 0 || device_id >= ndev) { g_create_error = "mer_context_create: device id out of range"; return 1; }
    mer_context *ctx = new mer_context();
    ctx->device = device_id;
    if (hipSetDevice(device_id) != hipSuccess || hipGetDeviceProperties(&ctx->prop, device_id) != hipSuccess) {
        g_create_error = "mer_context_create: hipSetDevice failed"; delete ctx; return 1;
    }
    if (hipMalloc((void **) &ctx->counters, sizeof(unsigned long long) * (MER_C_COUNT * MER_COUNTER_REPLICAS + 8)) != hipSuccess ||
        hipMemset(ctx->counters, 0, sizeof(unsigned long long) * (MER_C_COUNT * MER_COUNTER_REPLICAS + 8)) != hipSuccess ||
        hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) {
        g_create_error = "mer_context_create: device allocation failed"; delete ctx; return 1;
    }
    *out = ctx;
    return 0;
}

void mer_context_destroy(mer_context *ctx) {
    if (!ctx) return;
    (void) hipSetDevice(ctx->device);
    for (auto &kv : ctx->volumes) {
        if (kv.second.dense && kv.second.owns_dense) (void) hipFree(kv.second.dense);
        if (kv.second.cell8) (void) hipFree(kv.second.cell8);
        if (kv.second.coeff) (void) hipFree(kv.second.coeff);
    }
    if (ctx->counters) (void) hipFree(ctx->counters);
    for (Pipe &pp : ctx->pipes) {
        if (pp.slots) (void) hipFree(pp.slots);
        if (pp.live) (void) hipFree(pp.live);
        for (SegQueue *q : {&pp.eq, &pp.mq[0], &pp.mq[1], &pp.sq[0], &pp.sq[1], &pp.cq}) { if (q->items) (void) hipFree(q->items); if (q->counts) (void) hipFree(q->counts); }
        if (pp.hitq) (void) hipFree(pp.hitq);
        if (pp.hitq_ctr) (void) hipFree(pp.hitq_ctr);
        if (pp.host_live) (void) hipHostFree(pp.host_live);
        for (hipEvent_t e : pp.readback) if (e) (void) hipEventDestroy(e);
        if (pp.finished) (void) hipEventDestroy(pp.finished);
        for (hipEvent_t e : pp.pass_events) (void) hipEventDestroy(e);
        if (pp.own_stream) (void) hipStreamDestroy(pp.own_stream);
    }
    if (ctx->ev0) (void) hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void) hipEventDestroy(ctx->ev1);
    delete ctx;
}

const char *mer_last_error(mer_context *ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

int mer_context_set_stream(mer_context *ctx, void *hip_stream) { ctx->stream = (hipStream_t) hip_stream; return 0; }

int mer_device_info(mer_context *ctx, char *name, int32_t name_len, int32_t *cu_count, int64_t *hbm_bytes) {
    if (name && name_len > 0) { std::strncpy(name, ctx->prop.name, name_len - 1); name[name_len - 1] = 0; }
    if (cu_count) *cu_count = ctx->prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t) ctx->prop.totalGlobalMem;
    return 0;
}

static int volume_finish(mer_context *ctx, Volume &v, int32_t layout, mer_volume *out) {
    if (layout == MER_LAYOUT_AUTO) {
        const int64_t nodes = (int64_t) v.desc.res[0] * v.desc.res[1] * v.desc.res[2];
        layout = (v.desc.channels != 1 || v.desc.dtype != MER_VOL_F32) ? MER_LAYOUT_DENSE : (nodes <= ((int64_t) 1 << 28) ? MER_LAYOUT_BRICK27 : MER_LAYOUT_CELL8);
    }
    if (layout == MER_LAYOUT_CELL8) {
        if (v.desc.channels != 1 || v.desc.dtype != MER_VOL_F32) return fail(ctx, "CELL8 layout needs a 1-channel float32 grid");
        const size_t ncell = (size_t) (v.desc.res[0] - 1) * (v.desc.res[1] - 1) * (v.desc.res[2] - 1);
        HIP_CHECK(ctx, hipMalloc((void **) &v.cell8, ncell * 8 * sizeof(float)));
        hipLaunchKernelGGL(relayout_cell8_kernel, dim3(4096), dim3(256), 0, ctx->stream, (const float *) v.dense, v.cell8,
                           v.desc.res[0], v.desc.res[1], v.desc.res[2]);
        HIP_CHECK(ctx, hipGetLastError());
        HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    } else if (layout == MER_LAYOUT_BRICK27 || layout == MER_LAYOUT_BRICK125) {
        if (v.desc.channels != 1 || v.desc.dtype != MER_VOL_F32) return fail(ctx, "the BRICK layouts need a 1-channel float32 grid");
        const int bshift = layout == MER_LAYOUT_BRICK125 ? 2 : 1, bc = 1 << bshift, recw = layout == MER_LAYOUT_BRICK125 ? 128 : 32;
        for (int i = 0; i < 3; i++) if (v.desc.res[i] < 2) return fail(ctx, "the BRICK layouts need at least 2 nodes per axis");
        const int nbx = (v.desc.res[0] - 2) / bc + 1, nby = (v.desc.res[1] - 2) / bc + 1, nbz = (v.desc.res[2] - 2) / bc + 1;
        HIP_CHECK(ctx, hipMalloc((void **) &v.cell8, (size_t) nbx * nby * nbz * recw * sizeof(float)));
        hipLaunchKernelGGL(relayout_brick_kernel, dim3(4096), dim3(256), 0, ctx->stream, (const float *) v.dense, v.cell8,
                           v.desc.res[0], v.desc.res[1], v.desc.res[2], nbx, nby, nbz, bshift, recw);
        HIP_CHECK(ctx, hipGetLastError());
        HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    } else if (layout != MER_LAYOUT_DENSE) return fail(ctx, "unknown volume layout");
    v.layout = layout;
    const int h = ctx->next_handle++;
    ctx->volumes[h] = v;
    *out = h;
    return 0;
}

static int check_desc(mer_context *ctx, const mer_grid_desc *d) {
    // GridDataSource::loadFromFile checks (src/volume/gridvolume.cpp:243-268)
    if (d->dtype != MER_VOL_F32 && d->dtype != MER_VOL_U8) {
        char buf[160];
        std::snprintf(buf, sizeof(buf), "Encountered a volume data file of unknown type (type=%i, channels=%i)!", d->dtype, d->channels);
        return fail(ctx, buf);
    }
    if (d->channels != 1 && d->channels != 3) {
        char buf[160];
        std::snprintf(buf, sizeof(buf), "Encountered an unsupported volume data file (%i channels, only 1 and 3 are supported)", d->channels);
        return fail(ctx, buf);
    }
    for (int i = 0; i < 3; i++) {
        if (d->res[i] < 2) return fail(ctx, "volume resolution must be at least 2 along every axis");
        if (!(d->aabb_min[i] < d->aabb_max[i])) return fail(ctx, "volume bounding box is empty");
    }
    if ((int64_t) d->res[0] * d->res[1] * d->res[2] > (int64_t) 1 << 31) return fail(ctx, "volume too large for the int32 index contract");
    return 0;
}

int mer_volume_upload(mer_context *ctx, const mer_grid_desc *desc, const void *host_data, int32_t layout, mer_volume *out) {
    if (!ctx || !desc || !host_data || !out) return 1;
    if (check_desc(ctx, desc)) return 1;
    HIP_CHECK(ctx, hipSetDevice(ctx->device));
    Volume v; v.desc = *desc;
    const size_t n = (size_t) desc->res[0] * desc->res[1] * desc->res[2] * desc->channels;
    v.bytes_dense = n * (desc->dtype == MER_VOL_F32 ? 4 : 1);
    HIP_CHECK(ctx, hipMalloc(&v.dense, v.bytes_dense));
    HIP_CHECK(ctx, hipMemcpy(v.dense, host_data, v.bytes_dense, hipMemcpyHostToDevice));
    return volume_finish(ctx, v, layout, out);
}

int mer_volume_upload_dev(mer_context *ctx, const mer_grid_desc *desc, const void *data_dev, int32_t layout, mer_volume *out) {
    if (!ctx || !desc || !data_dev || !out) return 1;
    if (check_desc(ctx, desc)) return 1;
    HIP_CHECK(ctx, hipSetDevice(ctx->device));
    Volume v; v.desc = *desc;
    const size_t n = (size_t) desc->res[0] * desc->res[1] * desc->res[2] * desc->channels;
    v.bytes_dense = n * (desc->dtype == MER_VOL_F32 ? 4 : 1);
    HIP_CHECK(ctx, hipMalloc(&v.dense, v.bytes_dense));
    HIP_CHECK(ctx, hipMemcpyAsync(v.dense, data_dev, v.bytes_dense, hipMemcpyDeviceToDevice, ctx->stream));
    HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return volume_finish(ctx, v, layout, out);
}

int mer_volume_build_spline(mer_context *ctx, mer_volume h) {
    auto it = ctx->volumes.find(h);
    if (it == ctx->volumes.end()) return fail(ctx, "invalid volume handle");
    Volume &v = it->second;
    if (v.desc.channels != 1 || v.desc.dtype != MER_VOL_F32) return fail(ctx, "splinevolume needs a 1-channel float32 grid");
    if (v.coeff) return 0;
    const int nx = v.desc.res[0], ny = v.desc.res[1], nz = v.desc.res[2];
    const size_t n = (size_t) nx * ny * nz;
    float *a = nullptr, *b = nullptr, *t = nullptr;
    HIP_CHECK(ctx, hipMalloc((void **) &a, n * 4));
    HIP_CHECK(ctx, hipMalloc((void **) &b, n * 4));
    // along y (lines indexed by x and z), then x (by y and z), then z (by x and y): basisspline.h:868-887
    const bool seq = getenv("MER_PREFILTER_SEQ") != nullptr || std::min(nx, std::min(ny, nz)) < 16;
    if (seq) {                 // one thread per line (reference order of operations; tiny grids)
        hipLaunchKernelGGL(bspline_pass_kernel, dim3(nblocks((int64_t) nx * nz)), dim3(256), 0, ctx->stream,
                           (const float *) v.dense, a, nx, nz, (int64_t) 1, (int64_t) nx * ny, (int64_t) nx, ny);
        hipLaunchKernelGGL(bspline_pass_kernel, dim3(nblocks((int64_t) ny * nz)), dim3(256), 0, ctx->stream,
                           (const float *) a, b, ny, nz, (int64_t) nx, (int64_t) nx * ny, (int64_t) 1, nx);
        hipLaunchKernelGGL(bspline_pass_kernel, dim3(nblocks((int64_t) nx * ny)), dim3(256), 0, ctx->stream,
                           (const float *) b, a, nx, ny, (int64_t) 1, (int64_t) nx, (int64_t) nx * ny, nz);
    } else {
        if (getenv("MER_PREFILTER_TWO_KERNEL") || getenv("MER_PREFILTER_NO_LDS")) HIP_CHECK(ctx, hipMalloc((void **) &t, n * 4));
        auto pass = [&](const float *src, float *dst, int na, int nb, int64_t sa, int64_t sb, int64_t sl, int size) {
            const int64_t threads = (int64_t) na * nb * ((size + MER_PF_SEG - 1) / MER_PF_SEG);
            hipLaunchKernelGGL(bspline_causal_kernel, dim3(nblocks(threads)), dim3(256), 0, ctx->stream, src, t, na, nb, sa, sb, sl, size);
            hipLaunchKernelGGL(bspline_anticausal_kernel, dim3(nblocks(threads)), dim3(256), 0, ctx->stream, (const float *) t, dst, na, nb, sa, sb, sl, size);
        };
        auto win = [&](const float *src, float *dst, int na, int nb, int64_t sb, int64_t sl, int size) {     // fused sweeps, unit stride in a
            const int64_t threads = (int64_t) na * nb * ((size + MER_PF_SEG - 1) / MER_PF_SEG);
            hipLaunchKernelGGL(bspline_win_kernel, dim3(nblocks(threads)), dim3(256), 0, ctx->stream, src, dst, na, nb, sb, sl, size);
        };
        const bool two_kernel = getenv("MER_PREFILTER_TWO_KERNEL") != nullptr;
        if (two_kernel) pass((const float *) v.dense, a, nx, nz, 1, (int64_t) nx * ny, nx, ny);          // y
        else win((const float *) v.dense, a, nx, nz, (int64_t) nx * ny, nx, ny);
        if (getenv("MER_PREFILTER_NO_LDS")) pass(a, b, ny, nz, nx, (int64_t) nx * ny, 1, nx);           // x, strided form
        else {                                                                                        // x: lines are contiguous -> LDS tiles
            const int64_t nlines = (int64_t) ny * nz;
            if (nx % 4 == 0 && !getenv("MER_PREFILTER_LDS")) {
                const int64_t threads = nlines * ((nx + MER_PF_SEG - 1) / MER_PF_SEG);
                hipLaunchKernelGGL(bspline_x_reg_kernel, dim3(nblocks(threads)), dim3(256), 0, ctx->stream, (const float *) a, b, nlines, nx);
            } else {
                const int64_t blocks = ((nlines + MER_PFX_ROWS - 1) / MER_PFX_ROWS) * ((nx + MER_PFX_COLS - 1) / MER_PFX_COLS);
                hipLaunchKernelGGL(bspline_x_kernel, dim3((unsigned) blocks), dim3(256), 0, ctx->stream, (const float *) a, b, nlines, nx);
            }
        }
        if (two_kernel) pass(b, a, nx, ny, 1, nx, (int64_t) nx * ny, nz);                                // z
        else win(b, a, nx, ny, nx, (int64_t) nx * ny, nz);
    }
    HIP_CHECK(ctx, hipGetLastError());
    HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    (void) hipFree(b);
    if (t) (void) hipFree(t);
    v.coeff = a;
    return 0;
}

int mer_volume_download_spline(mer_context *ctx, mer_volume h, float *coeff_host) {
    auto it = ctx->volumes.find(h);
    if (it == ctx->volumes.end() || !it->second.coeff) return fail(ctx, "volume has no spline coefficients");
    const size_t n = (size_t) it->second.desc.res[0] * it->second.desc.res[1] * it->second.desc.res[2];
    HIP_CHECK(ctx, hipMemcpy(coeff_host, it->second.coeff, n * 4, hipMemcpyDeviceToHost));
    return 0;
}

int mer_volume_destroy(mer_context *ctx, mer_volume h) {
    auto it = ctx->volumes.find(h);
    if (it == ctx->volumes.end()) return fail(ctx, "invalid volume handle");
    if (it->second.dense && it->second.owns_dense) (void) hipFree(it->second.dense);
    if (it->second.cell8) (void) hipFree(it->second.cell8);
    if (it->second.coeff) (void) hipFree(it->second.coeff);
    ctx->volumes.erase(it);
    return 0;
}

int mer_film_channels(mer_context *ctx, const mer_scene_desc *scene, int32_t *channels) {
    int frames;
    if (!scene || !channels) return 1;
    if (film_frames(ctx, scene, frames)) return 1;
    *channels = frames * 3 + 2;
    return 0;
}
int mer_film_alloc_n(mer_context *ctx, int32_t width, int32_t height, int32_t channels, float **film_dev) {
    HIP_CHECK(ctx, hipMalloc((void **) film_dev, (size_t) width * height * channels * sizeof(float)));
    HIP_CHECK(ctx, hipMemsetAsync(*film_dev, 0, (size_t) width * height * channels * sizeof(float), ctx->stream));
    return 0;
}
int mer_film_zero_n(mer_context *ctx, float *film_dev, int32_t width, int32_t height, int32_t channels) {
    HIP_CHECK(ctx, hipMemsetAsync(film_dev, 0, (size_t) width * height * channels * sizeof(float), ctx->stream));
    return 0;
}
int mer_film_download_n(mer_context *ctx, const float *film_dev, int32_t width, int32_t height, int32_t channels, float *film_host) {
    HIP_CHECK(ctx, hipMemcpyAsync(film_host, film_dev, (size_t) width * height * channels * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}
int mer_film_alloc(mer_context *ctx, int32_t width, int32_t height, float **film_dev) { return mer_film_alloc_n(ctx, width, height, 5, film_dev); }
int mer_film_zero(mer_context *ctx, float *film_dev, int32_t width, int32_t height) { return mer_film_zero_n(ctx, film_dev, width, height, 5); }
int mer_film_download(mer_context *ctx, const float *film_dev, int32_t width, int32_t height, float *film_host) {
    return mer_film_download_n(ctx, film_dev, width, height, 5, film_host);
}
int mer_film_free(mer_context *ctx, float *film_dev) { HIP_CHECK(ctx, hipFree(film_dev)); return 0; }
int mer_device_free(mer_context *ctx, void *p) { HIP_CHECK(ctx, hipFree(p)); return 0; }

static int launch_render(mer_context *ctx, const mer_scene_desc *scene, const mer_shard *shard, uint64_t seed,
                         float *film_dev, float *path_out_dev) {
    Params P;
    if (make_params(ctx, scene, P, true)) return 1;
    if (!shard || shard->spp_count < 0 || shard->spp_stride <= 0 || shard->tile_count <= 0 || shard->tile_rank < 0 ||
        shard->tile_rank >= shard->tile_count || shard->spp_begin < 0)
        return fail(ctx, "invalid shard");
    P.seed = seed;
    P.spp_begin = shard->spp_begin; P.spp_count = shard->spp_count; P.spp_stride = shard->spp_stride;
    P.tile_rank = shard->tile_rank; P.tile_count = shard->tile_count;
    P.tiles_x = (scene->width + MER_TILE - 1) / MER_TILE; P.tiles_y = (scene->height + MER_TILE - 1) / MER_TILE;
    const int ntiles = P.tiles_x * P.tiles_y;
    P.ntiles_mine = (ntiles - shard->tile_rank + shard->tile_count - 1) / shard->tile_count;
    P.total_work = (uint64_t) P.ntiles_mine * MER_TILE * MER_TILE * (uint64_t) shard->spp_count;
    P.film = film_dev; P.path_out = path_out_dev;
    { const char *e = getenv("MER_DEBUG_PIXEL"); P.dbg_pixel = e ? atoi(e) : -1; }
    HIP_CHECK(ctx, hipSetDevice(ctx->device));
    HIP_CHECK(ctx, hipMemsetAsync(P.work_counter, 0, sizeof(unsigned long long), ctx->stream));
    if (P.total_work == 0) return 0;
    const char *mode = getenv("MER_MODE");
    if (mode && std::strcmp(mode, "mega") == 0) {
        if (scene->decomposition != MER_DECOMPOSITION_NONE) return fail(ctx, "MER_MODE=mega renders steady-state films only; use the default wavefront mode");
        if (scene->boundary_bsdf != MER_BSDF_NULL) return fail(ctx, "MER_MODE=mega knows the index-matched boundary only; use the default wavefront mode");
        if (scene->boundary == MER_BOUNDARY_SDF) return fail(ctx, "MER_MODE=mega knows the cube / sphere boundaries only; use the default wavefront mode");
        if (scene->point_intensity[0] != 0 || scene->point_intensity[1] != 0 || scene->point_intensity[2] != 0)
            return fail(ctx, "MER_MODE=mega does not sample point emitters; use the default wavefront mode");
        return dispatch_modes(ctx, scene, [&](auto curved, auto rif, auto stepper, auto sigma, auto bnd) -> int {
            auto kern = render_kernel<decltype(curved)::value, decltype(rif)::value, decltype(stepper)::value, decltype(sigma)::value>;
            int per_cu = 0;
            HIP_CHECK(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, MER_BLOCK, 0));
            if (per_cu < 1) per_cu = 1;
            int64_t blocks = (int64_t) per_cu * ctx->prop.multiProcessorCount;
            const int64_t need = (int64_t) ((P.total_work + MER_BLOCK - 1) / MER_BLOCK);
            if (blocks > need) blocks = need;
            HIP_CHECK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
            hipLaunchKernelGGL(kern, dim3((unsigned) blocks), dim3(MER_BLOCK), 0, ctx->stream, P);
            HIP_CHECK(ctx, hipGetLastError());
            HIP_CHECK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
            ctx->timed = true;
            return 0;
        });
    }
    // ---- wavefront: K_event / K_march passes over the path-state slots until no lane is alive.
    // The passes of ONE pipeline are a chain of dependent launches (K_gen -> K_event -> K_march), and K_event -- fat, latency-bound,
    // 2 waves per SIMD -- leaves most of the chip idle while it runs.  So the render is cut into `npipes` independent pipelines: pipeline
    // q takes the sample indices q, q + npipes, ... of the shard (the sharding contract of section 8e, applied inside one GPU), has its
    // own slots, lists, hit ring and work counter, and runs on its own stream; the film is shared (atomics).  One pipeline's K_event then
    // overlaps the others' K_march.  Per-path results do not depend on npipes.  Measured on the bench line: 1 pipeline 260, 2: 277, 3: 282,
    // 4: 283 Mpaths/s (two PROCESSES on one GPU: 298); more slots per pipeline change nothing.
    int npipes = 4;
    { const char *e = getenv("MER_PIPES"); if (e && atoi(e) > 0) npipes = std::min(atoi(e), MER_MAX_PIPES); }
    if (shard->spp_count < npipes) npipes = std::max(1, shard->spp_count);
    uint32_t want = (uint32_t) ctx->prop.multiProcessorCount * 2048u * 4u;          // 4 x the resident lanes of the chip, over all pipelines
    { const char *e = getenv("MER_NSLOTS"); if (e && atoi(e) > 0) want = (uint32_t) atoi(e); }
    want = (want / (uint32_t) npipes + MER_BLOCK - 1) / MER_BLOCK * MER_BLOCK;       // per pipeline
    int ksteps0 = 128;        // eikonal steps per lane per pass (64..256 measured with class-sorted march lists: 128 is the flat optimum at 256^3 and 512^3)
    { const char *e = getenv("MER_KSTEPS"); if (e && atoi(e) > 0) ksteps0 = atoi(e); }
    // sorting the march lists by exit time scatters the lanes of a wave over the volume: a gain while the RIF sits near the caches
    // (256^3: +8 %, 512^3: +3 %), a loss once every fetch goes to HBM (1024^3: -4 %)
    P.mq_sort = (int64_t) P.rif.res[0] * P.rif.res[1] * P.rif.res[2] <= ((int64_t) 1 << 28) ? 1 : 0;
    { const char *e = getenv("MER_MQ_SORT"); if (e) P.mq_sort = atoi(e) != 0; }
    P.gen_iters = 8; P.gen_all = getenv("MER_GEN_ALL") ? 1 : 0;
    // Connection requests gather in one row of cq over connect_every passes (the parked slots wait, the others keep marching) and
    // K_connect drains the row at the end of the group: fuller launches, 5-8 % on configs[4]; in the tail it runs every pass.
    // (The stage is throughput-bound -- ~14 k sensitivity steps per connection, 22 G steps/s -- not launch-latency-bound: gathering
    // 16 passes gains no more than gathering 4.)
    int connect_every0 = 4;
    { const char *e = getenv("MER_CONNECT_EVERY"); if (e && atoi(e) > 0) connect_every0 = atoi(e); }

    struct Run { Params P; uint32_t nslots = 0, pass = 0, since_connect = 0; unsigned blocks = 0, gen_blocks = 0; int connect_every = 1, cur = 0; bool work_left = true, done = false; };
    Run runs[MER_MAX_PIPES];
    HIP_CHECK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    for (int q = 0; q < npipes; q++) {
        Pipe &pp = ctx->pipes[q]; Run &R = runs[q];
        if (q == 0) pp.stream = ctx->stream;
        else if (!pp.own_stream) { HIP_CHECK(ctx, hipStreamCreateWithFlags(&pp.own_stream, hipStreamNonBlocking)); }
        if (q > 0) { pp.stream = pp.own_stream; HIP_CHECK(ctx, hipStreamWaitEvent(pp.stream, ctx->ev0, 0)); }   // after what the caller queued (film zeroing ...)
        if (pp.nslots < want) {                 // capacity: grows, never shrinks (render_paths runs one pipeline, mer_render two)
            if (pp.slots) (void) hipFree(pp.slots);
            if (pp.hitq) (void) hipFree(pp.hitq);
            pp.slots = nullptr; pp.hitq = nullptr; pp.nslots = 0;
            HIP_CHECK(ctx, hipMalloc((void **) &pp.slots, (size_t) want * MER_SLOT_WORDS * sizeof(uint32_t)));
            for (SegQueue *sq : {&pp.eq, &pp.mq[0], &pp.mq[1], &pp.sq[0], &pp.sq[1], &pp.cq}) {
                if (sq->items) (void) hipFree(sq->items);
                sq->items = nullptr;
                sq->segcap = 2u * (want / MER_NSEG) + 256u;          // two producer kernels may feed one segment
                if (sq == &pp.eq) sq->segcap = 2u * (want / (MER_NSEG / MER_EV_CLASSES)) + 256u;   // every lane may be of one event class
                if (sq == &pp.mq[0] || sq == &pp.mq[1]) sq->segcap = 2u * (want / (MER_NSEG / MER_MQ_CLASSES)) + 256u;   // ... or of one march class
                if (sq == &pp.cq) sq->segcap = want + 256u;   // requests gather over several passes: a segment may see every slot once
                HIP_CHECK(ctx, hipMalloc((void **) &sq->items, (size_t) sq->segcap * MER_NSEG * sizeof(uint32_t)));
                if (!sq->counts) HIP_CHECK(ctx, hipMalloc((void **) &sq->counts, (size_t) MER_LIVE_SLOTS * MER_NSEG * sizeof(uint32_t)));
            }
            pp.hitq_cap = 1; while (pp.hitq_cap < (unsigned long long) want * 2) pp.hitq_cap <<= 1;
            HIP_CHECK(ctx, hipMalloc((void **) &pp.hitq, (size_t) pp.hitq_cap * sizeof(unsigned long long)));
            pp.nslots = want;
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
        R.P.slots = pp.slots; R.P.nslots = R.nslots; R.P.live = pp.live; R.P.eq = pp.eq; R.P.mq[0] = pp.mq[0]; R.P.mq[1] = pp.mq[1];
        R.P.sq[0] = pp.sq[0]; R.P.sq[1] = pp.sq[1]; R.P.cq = pp.cq;
        R.P.hitq = pp.hitq; R.P.hitq_cap = pp.hitq_cap; R.P.hitq_ctr = pp.hitq_ctr; R.P.work_counter = pp.hitq_ctr + 48;
        R.P.ksteps = ksteps0; R.P.cq_row = 0;
        R.connect_every = connect_every0;
        R.done = R.P.total_work == 0;
        if (R.done) continue;
        R.blocks = R.nslots / MER_BLOCK;
        R.gen_blocks = std::max(1u, std::min(R.nslots / MER_BLOCK, 1024u));      // 4096 waves x 512 ids per launch
        HIP_CHECK(ctx, hipMemsetAsync(pp.slots, 0, (size_t) R.nslots * MER_SLOT_WORDS * sizeof(uint32_t), pp.stream));
        HIP_CHECK(ctx, hipMemsetAsync(pp.live, 0, MER_LIVE_SLOTS * sizeof(uint32_t), pp.stream));
        for (SegQueue *sq : {&pp.eq, &pp.mq[0], &pp.mq[1], &pp.sq[0], &pp.sq[1], &pp.cq})
            HIP_CHECK(ctx, hipMemsetAsync(sq->counts, 0, (size_t) MER_LIVE_SLOTS * MER_NSEG * sizeof(uint32_t), pp.stream));
        HIP_CHECK(ctx, hipMemsetAsync(pp.hitq_ctr, 0, 64 * sizeof(unsigned long long), pp.stream));
    }
    auto body = [&](auto curved, auto rif, auto stepper, auto sigma, auto bnd) -> int {
        constexpr int BND = decltype(bnd)::value;
        const bool has_point = scene->point_intensity[0] != 0 || scene->point_intensity[1] != 0 || scene->point_intensity[2] != 0;
        // the signed-distance boundary exists in the EXTRA kernels only
        const bool extra = BND != 0 || has_point || scene->modulation != MER_MODULATION_NONE || scene->boundary_bsdf != MER_BSDF_NULL;
        auto kev = (extra || BND != 0) ? event_kernel<decltype(curved)::value, decltype(rif)::value, decltype(stepper)::value, decltype(sigma)::value, true, BND>
                                       : event_kernel<decltype(curved)::value, decltype(rif)::value, decltype(stepper)::value, decltype(sigma)::value, BND != 0, BND>;
        auto kma = march_kernel<decltype(curved)::value, decltype(rif)::value, decltype(stepper)::value, decltype(sigma)::value, BND>;
        auto kco = connect_stage_kernel<decltype(curved)::value ? decltype(rif)::value : MER_RIF_TRILINEAR, decltype(stepper)::value, decltype(sigma)::value, BND>;
        const bool connect_stage = has_point && decltype(curved)::value;
        auto kge = (extra || BND != 0) ? gen_kernel<decltype(curved)::value, true, BND> : gen_kernel<decltype(curved)::value, BND != 0, BND>;
        const uint32_t check_every = 8;
        const bool adaptive = getenv("MER_FIXED_K") == nullptr;
        const bool pass_events = getenv("MER_NO_PASS_EVENTS") == nullptr;          // per-kernel timing of every pass (mer_last_render_stats)
        // one batch = check_every passes of a pipeline followed by the read-back of its finished-slot count into slot `rb`.  Two batches
        // are kept in flight per pipeline, so that a pipeline never runs dry while the host waits for another one's read-back (a
        // finished render thus carries one batch of empty passes: ~0.5 ms)
        auto enqueue_batch = [&](int q, int rb) -> int {
            Pipe &pp = ctx->pipes[q]; Run &R = runs[q];
            for (uint32_t b = 0; b < check_every; b++) {
                const uint32_t pass = R.pass;
                while (pp.pass_events.size() < (size_t) (pass + 1) * 3) {
                    hipEvent_t e; HIP_CHECK(ctx, hipEventCreate(&e)); pp.pass_events.push_back(e);
                }
                if (pass_events) HIP_CHECK(ctx, hipEventRecord(pp.pass_events[pass * 3 + 0], pp.stream));
                for (int g = 0; R.work_left && g < (pass == 0 ? 6 : 1); g++) hipLaunchKernelGGL(kge, dim3(R.gen_blocks), dim3(MER_BLOCK), 0, pp.stream, R.P);
                hipLaunchKernelGGL(kev, dim3(R.blocks), dim3(MER_BLOCK), 0, pp.stream, R.P, pass);
                if (connect_stage && ++R.since_connect >= (uint32_t) R.connect_every) {
                    hipLaunchKernelGGL(kco, dim3(R.blocks), dim3(MER_BLOCK), 0, pp.stream, R.P, pass);
                    R.since_connect = 0; R.P.cq_row++;
                }
                if (pass_events) HIP_CHECK(ctx, hipEventRecord(pp.pass_events[pass * 3 + 1], pp.stream));
                hipLaunchKernelGGL(kma, dim3(R.blocks), dim3(MER_BLOCK), 0, pp.stream, R.P, pass);
                if (pass_events) HIP_CHECK(ctx, hipEventRecord(pp.pass_events[pass * 3 + 2], pp.stream));
                R.pass++;
            }
            HIP_CHECK(ctx, hipGetLastError());
            HIP_CHECK(ctx, hipMemcpyAsync(pp.host_live + 4 * rb, pp.live, sizeof(uint32_t), hipMemcpyDeviceToHost, pp.stream));
            HIP_CHECK(ctx, hipMemcpyAsync(pp.host_live + 4 * rb + 2, R.P.work_counter, sizeof(unsigned long long), hipMemcpyDeviceToHost, pp.stream));
            HIP_CHECK(ctx, hipEventRecord(pp.readback[rb], pp.stream));
            return 0;
        };
        for (int q = 0; q < npipes; q++) if (!runs[q].done && enqueue_batch(q, 0)) return 1;
        for (int q = 0; q < npipes; q++) if (!runs[q].done && enqueue_batch(q, 1)) return 1;
        for (;;) {
            bool any = false;
            for (int q = 0; q < npipes; q++) {
                Pipe &pp = ctx->pipes[q]; Run &R = runs[q];
                if (R.done) continue;
                any = true;
                const int rb = R.cur; R.cur ^= 1;
                HIP_CHECK(ctx, hipEventSynchronize(pp.readback[rb]));
                const uint32_t finished_slots = pp.host_live[4 * rb];
                if (finished_slots >= R.nslots) { R.done = true; continue; }
                R.work_left = *(unsigned long long *) (pp.host_live + 4 * rb + 2) < R.P.total_work;
                if (adaptive) {          // tail: few lanes left => longer passes, fewer launches
                    const uint32_t alive = R.nslots - finished_slots;
                    R.P.ksteps = alive < R.nslots / 64 ? ksteps0 * 32 : (alive < R.nslots / 16 ? ksteps0 * 8 : (alive < R.nslots / 4 ? ksteps0 * 2 : ksteps0));
                    R.connect_every = alive < R.nslots / 4 ? 1 : connect_every0;
                }
                if (R.pass > (1u << 24)) return fail(ctx, "mer_render: pass limit exceeded");
                if (enqueue_batch(q, rb)) return 1;
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
        {   // per-kernel device time of this render, from HIP events on the launch streams (summed over the pipelines: with two of
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
        if (getenv("MER_VERBOSE")) { float ms = 0; (void) hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1); fprintf(stderr, "[mer] wavefront: %d pipelines, %u + %u passes, K=%d, nslots=%u each, %.3f ms\n", npipes, runs[0].pass, npipes > 1 ? runs[1].pass : 0u, ksteps0, runs[0].nslots, ms); }
        return 0;
    };
    return scene->boundary == MER_BOUNDARY_SDF ? dispatch_modes_sdf(ctx, scene, body) : dispatch_modes(ctx, scene, body);
}

int mer_render(mer_context *ctx, const mer_scene_desc *scene, const mer_shard *shard, uint64_t seed, float *film_dev) {
    if (!ctx || !scene || !film_dev) return 1;
    return launch_render(ctx, scene, shard, seed, film_dev, nullptr);
}

int mer_render_paths(mer_context *ctx, const mer_scene_desc *scene, int32_t sample_index, uint64_t seed, float *out_rgb) {
    if (!ctx || !scene || !out_rgb) return 1;
    const size_t n = (size_t) scene->width * scene->height * 3;
    DevBuf buf(ctx);
    if (buf.alloc(n * 4)) return 1;
    HIP_CHECK(ctx, hipMemsetAsync(buf.p, 0, n * 4, ctx->stream));
    mer_shard sh = {sample_index, 1, 1, 0, 1};
    if (launch_render(ctx, scene, &sh, seed, buf.as<float>(), buf.as<float>())) return 1;
    return buf.download(out_rgb, n * 4);
}

int mer_synchronize(mer_context *ctx) { HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream)); return 0; }

int mer_last_kernel_ms(mer_context *ctx, float *ms) {
    if (!ctx->timed) return fail(ctx, "no render has been launched");
    HIP_CHECK(ctx, hipEventSynchronize(ctx->ev1));
    HIP_CHECK(ctx, hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return 0;
}

int mer_last_render_stats(mer_context *ctx, int32_t *passes, float *march_ms, float *event_ms) {
    if (!ctx->timed) return fail(ctx, "no render has been launched");
    if (passes) *passes = ctx->last_passes;
    if (march_ms) *march_ms = ctx->last_march_ms;
    if (event_ms) *event_ms = ctx->last_event_ms;
    return 0;
}

int mer_counters_read(mer_context *ctx, uint64_t out[MER_C_COUNT]) {
    HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<uint64_t> all((size_t) MER_C_COUNT * MER_COUNTER_REPLICAS);
    HIP_CHECK(ctx, hipMemcpy(all.data(), ctx->counters, sizeof(uint64_t) * all.size(), hipMemcpyDeviceToHost));
    for (int k = 0; k < MER_C_COUNT; k++) {
        out[k] = 0;
        for (int r = 0; r < MER_COUNTER_REPLICAS; r++) out[k] += all[(size_t) r * MER_C_COUNT + k];
    }
    return 0;
}
int mer_counters_reset(mer_context *ctx) {
    HIP_CHECK(ctx, hipMemsetAsync(ctx->counters, 0, sizeof(uint64_t) * MER_C_COUNT * MER_COUNTER_REPLICAS, ctx->stream));
    return 0;
}

// ---- leaf entry points ------------------------------------------------------------------------------
int mer_lookup_trilinear(mer_context *ctx, mer_volume h, const float *pts, int64_t n, float *out_val, int32_t *out_idx) {
    auto it = ctx->volumes.find(h);
    if (it == ctx->volumes.end()) return fail(ctx, "invalid volume handle");
    if (it->second.desc.channels != 1) return fail(ctx, "lookupFloat(): volume does not support float lookups");
    DGrid g; fill_dgrid(it->second, g);
    DevBuf dp(ctx), dv(ctx), di(ctx);
    if (dp.upload(pts, n * 12) || dv.alloc(n * 4) || di.alloc(n * 16)) return 1;
    hipLaunchKernelGGL(lookup_trilinear_kernel, dim3(nblocks(n)), dim3(256), 0, ctx->stream, g, dp.as<float>(), n, dv.as<float>(),
                       out_idx ? di.as<int32_t>() : (int32_t *) nullptr);
    HIP_CHECK(ctx, hipGetLastError());
    if (dv.download(out_val, n * 4)) return 1;
    if (out_idx && di.download(out_idx, n * 16)) return 1;
    return 0;
}
int mer_lookup_trilinear_rgb(mer_context *ctx, mer_volume h, const float *pts, int64_t n, float *out_rgb) {
    auto it = ctx->volumes.find(h);
    if (it == ctx->volumes.end()) return fail(ctx, "invalid volume handle");
    if (it->second.desc.channels != 3) return fail(ctx, "lookupSpectrum(): volume does not support spectrum lookups");
    Volume tmp = it->second; tmp.cell8 = nullptr;
    DGrid g; fill_dgrid(tmp, g);
    DevBuf dp(ctx), dv(ctx);
    if (dp.upload(pts, n * 12) || dv.alloc(n * 12)) return 1;
    hipLaunchKernelGGL(lookup_rgb_kernel, dim3(nblocks(n)), dim3(256), 0, ctx->stream, g, dp.as<float>(), n, dv.as<float>());
    HIP_CHECK(ctx, hipGetLastError());
    return dv.download(out_rgb, n * 12);
}
int mer_rif_value_grad(mer_context *ctx, mer_volume h, int32_t interp, const float *pts, int64_t n, float *out_val, float *out_grad) {
    auto it = ctx->volumes.find(h);
    if (it == ctx->volumes.end()) return fail(ctx, "invalid volume handle");
    if (it->second.desc.channels != 1 || it->second.desc.dtype != MER_VOL_F32) return fail(ctx, "value(): not implemented for this volume type"); // volume.cpp:57-80
    if (interp != MER_RIF_TRILINEAR && interp != MER_RIF_BSPLINE3) return fail(ctx, "unknown rif_interp");
    if (interp == MER_RIF_BSPLINE3 && !it->second.coeff) return fail(ctx, "volume has no spline coefficients");
    DGrid g; fill_dgrid(it->second, g);
    DevBuf dp(ctx), dv(ctx), dg(ctx);
    if (dp.upload(pts, n * 12) || dv.alloc(n * 4) || dg.alloc(n * 12)) return 1;
    hipLaunchKernelGGL(rif_value_grad_kernel, dim3(nblocks(n)), dim3(256), 0, ctx->stream, g, interp, dp.as<float>(), n, dv.as<float>(), dg.as<float>());
    HIP_CHECK(ctx, hipGetLastError());
    if (dv.download(out_val, n * 4)) return 1;
    return dg.download(out_grad, n * 12);
}

int mer_er_trace(mer_context *ctx, const mer_scene_desc *scene, const float *p0, const float *d0, const float *dist, int64_t n,
                 float *out_p, float *out_v, float *out_dist_surf, float *out_opt, int32_t *out_success) {
    Params P;
    if (make_params(ctx, scene, P)) return 1;
    if (scene->rif_mode == MER_RIF_CONST) return fail(ctx, "mer_er_trace needs a RIF volume");
    DevBuf a(ctx), b(ctx), c(ctx), op(ctx), ov(ctx), od(ctx), oo(ctx), ok(ctx);
    if (a.upload(p0, n * 12) || b.upload(d0, n * 12) || c.upload(dist, n * 4) || op.alloc(n * 12) || ov.alloc(n * 12) ||
        od.alloc(n * 4) || oo.alloc(n * 4) || ok.alloc(n * 4)) return 1;
    int rifk = scene->rif_mode;
    if (scene->rif_mode == MER_RIF_TRILINEAR) {
        if (P.rif.layout == MER_LAYOUT_BRICK27 || P.rif.layout == MER_LAYOUT_BRICK125) return fail(ctx, "this leaf entry point takes the RIF in the dense or cell8 layout");
        if (P.rif.layout == MER_LAYOUT_CELL8) rifk = P.rif.buf_bytes ? RIFK_CELL8_BUF : RIFK_CELL8;
        else rifk = P.rif.buf_bytes ? RIFK_DENSE_BUF : MER_RIF_TRILINEAR;
    }
#define MER_TRACE_CASE(R, S)                                                                                       \
    if (rifk == R && scene->stepper == S)                                                                          \
        hipLaunchKernelGGL((er_trace_kernel<R, S>), dim3(nblocks(n, 64)), dim3(64), 0, ctx->stream, P, a.as<float>(), b.as<float>(), \
                           c.as<float>(), n, op.as<float>(), ov.as<float>(), od.as<float>(), oo.as<float>(), ok.as<int32_t>());
    MER_TRACE_CASE(MER_RIF_TRILINEAR, MER_STEP_VERLET) MER_TRACE_CASE(MER_RIF_TRILINEAR, MER_STEP_RK4)
    MER_TRACE_CASE(RIFK_DENSE_BUF, MER_STEP_VERLET) MER_TRACE_CASE(RIFK_DENSE_BUF, MER_STEP_RK4)
    MER_TRACE_CASE(RIFK_CELL8, MER_STEP_VERLET) MER_TRACE_CASE(RIFK_CELL8, MER_STEP_RK4)
    MER_TRACE_CASE(RIFK_CELL8_BUF, MER_STEP_VERLET) MER_TRACE_CASE(RIFK_CELL8_BUF, MER_STEP_RK4)
    MER_TRACE_CASE(MER_RIF_BSPLINE3, MER_STEP_VERLET) MER_TRACE_CASE(MER_RIF_BSPLINE3, MER_STEP_RK4)
    MER_TRACE_CASE(RIFK_ACOUSTIC, MER_STEP_VERLET) MER_TRACE_CASE(RIFK_ACOUSTIC, MER_STEP_RK4)
#undef MER_TRACE_CASE
    HIP_CHECK(ctx, hipGetLastError());
    if (op.download(out_p, n * 12) || ov.download(out_v, n * 12) || od.download(out_dist_surf, n * 4) || oo.download(out_opt, n * 4) ||
        ok.download(out_success, n * 4)) return 1;
    return 0;
}

int mer_connect(mer_context *ctx, const mer_scene_desc *scene, const float *p1, const float *p2, int64_t n, uint64_t seed, float *out) {
    Params P;
    if (make_params(ctx, scene, P)) return 1;
    if (scene->rif_mode == MER_RIF_CONST) return fail(ctx, "mer_connect needs a RIF volume");
    P.seed = seed;
    DevBuf a(ctx), b(ctx), r(ctx);
    if (a.upload(p1, n * 12) || b.upload(p2, n * 12) || r.alloc(n * 48)) return 1;
    int rifk = scene->rif_mode;
    if (scene->rif_mode == MER_RIF_TRILINEAR) {
        if (P.rif.layout == MER_LAYOUT_BRICK27 || P.rif.layout == MER_LAYOUT_BRICK125) return fail(ctx, "this leaf entry point takes the RIF in the dense or cell8 layout");
        if (P.rif.layout == MER_LAYOUT_CELL8) rifk = P.rif.buf_bytes ? RIFK_CELL8_BUF : RIFK_CELL8;
        else rifk = P.rif.buf_bytes ? RIFK_DENSE_BUF : MER_RIF_TRILINEAR;
    }
#define MER_CONNECT_CASE(R) if (rifk == R) hipLaunchKernelGGL((connect_kernel<R>), dim3(nblocks(n, 64)), dim3(64), 0, ctx->stream, P, a.as<float>(), b.as<float>(), n, r.as<float>());
    MER_CONNECT_CASE(MER_RIF_TRILINEAR) MER_CONNECT_CASE(MER_RIF_BSPLINE3) MER_CONNECT_CASE(RIFK_DENSE_BUF) MER_CONNECT_CASE(RIFK_CELL8) MER_CONNECT_CASE(RIFK_CELL8_BUF)
#undef MER_CONNECT_CASE
    HIP_CHECK(ctx, hipGetLastError());
    return r.download(out, n * 48);
}

int mer_sample_distance(mer_context *ctx, const mer_scene_desc *scene, const float *o, const float *d, const float *maxt, int64_t n,
                        uint64_t seed, float *rec) {
    Params P;
    if (make_params(ctx, scene, P)) return 1;
    P.seed = seed;
    DevBuf a(ctx), b(ctx), c(ctx), r(ctx);
    if (a.upload(o, n * 12) || b.upload(d, n * 12) || c.upload(maxt, n * 4) || r.alloc(n * 80)) return 1;
    int rc = dispatch_modes(ctx, scene, [&](auto curved, auto rif, auto stepper, auto sigma, auto bnd) -> int {
        hipLaunchKernelGGL((sample_distance_kernel<decltype(curved)::value, decltype(rif)::value, decltype(stepper)::value, decltype(sigma)::value>),
                           dim3(nblocks(n, 64)), dim3(64), 0, ctx->stream, P, a.as<float>(), b.as<float>(), c.as<float>(), n, r.as<float>());
        HIP_CHECK(ctx, hipGetLastError());
        return 0;
    });
    if (rc) return rc;
    return r.download(rec, n * 80);
}

int mer_eval_transmittance(mer_context *ctx, const mer_scene_desc *scene, const float *o, const float *d, const float *maxt, int64_t n,
                           uint64_t seed, float *out_tr) {
    Params P;
    if (make_params(ctx, scene, P)) return 1;
    P.seed = seed;
    DevBuf a(ctx), b(ctx), c(ctx), r(ctx);
    if (a.upload(o, n * 12) || b.upload(d, n * 12) || c.upload(maxt, n * 4) || r.alloc(n * 12)) return 1;
    int rc = dispatch_modes(ctx, scene, [&](auto curved, auto rif, auto stepper, auto sigma, auto bnd) -> int {
        hipLaunchKernelGGL((eval_transmittance_kernel<decltype(curved)::value, decltype(rif)::value, decltype(stepper)::value, decltype(sigma)::value>),
                           dim3(nblocks(n, 64)), dim3(64), 0, ctx->stream, P, a.as<float>(), b.as<float>(), c.as<float>(), n, r.as<float>());
        HIP_CHECK(ctx, hipGetLastError());
        return 0;
    });
    if (rc) return rc;
    return r.download(out_tr, n * 12);
}

int mer_phase_sample(mer_context *ctx, int32_t phase, float g, const float *wi, const float *u2, int64_t n, float *wo, float *pdf) {
    if (phase == MER_PHASE_HG && (g >= 1 || g <= -1)) return fail(ctx, "The asymmetry parameter must lie in the interval (-1, 1)!");
    DevBuf a(ctx), b(ctx), c(ctx), e(ctx);
    if (a.upload(wi, n * 12) || b.upload(u2, n * 8) || c.alloc(n * 12) || e.alloc(n * 4)) return 1;
    hipLaunchKernelGGL(phase_sample_kernel, dim3(nblocks(n)), dim3(256), 0, ctx->stream, phase, g, a.as<float>(), b.as<float>(), n, c.as<float>(), e.as<float>());
    HIP_CHECK(ctx, hipGetLastError());
    if (c.download(wo, n * 12)) return 1;
    return e.download(pdf, n * 4);
}
int mer_phase_eval(mer_context *ctx, int32_t phase, float g, const float *wi, const float *wo, int64_t n, float *val) {
    DevBuf a(ctx), b(ctx), c(ctx);
    if (a.upload(wi, n * 12) || b.upload(wo, n * 12) || c.alloc(n * 4)) return 1;
    hipLaunchKernelGGL(phase_eval_kernel, dim3(nblocks(n)), dim3(256), 0, ctx->stream, phase, g, a.as<float>(), b.as<float>(), n, c.as<float>());
    HIP_CHECK(ctx, hipGetLastError());
    return c.download(val, n * 4);
}
int mer_camera_rays(mer_context *ctx, const mer_scene_desc *scene, const float *pos2, int64_t n, float *o, float *d) {
    Params P;
    mer_scene_desc sc = *scene; sc.sigma_mode = MER_SIGMA_HOMOGENEOUS; sc.rif_mode = MER_RIF_CONST; sc.albedo_mode = MER_ALBEDO_CONST;
    if (make_params(ctx, &sc, P)) return 1;
    DevBuf a(ctx), b(ctx), c(ctx);
    if (a.upload(pos2, n * 8) || b.alloc(n * 12) || c.alloc(n * 12)) return 1;
    hipLaunchKernelGGL(camera_rays_kernel, dim3(nblocks(n)), dim3(256), 0, ctx->stream, P, a.as<float>(), n, b.as<float>(), c.as<float>());
    HIP_CHECK(ctx, hipGetLastError());
    if (b.download(o, n * 12)) return 1;
    return c.download(d, n * 12);
}
int mer_correlation(mer_context *ctx, const mer_scene_desc *scene, const float *path_length, int64_t n, float *out) {
    Params P;
    mer_scene_desc sc = *scene; sc.sigma_mode = MER_SIGMA_HOMOGENEOUS; sc.rif_mode = MER_RIF_CONST; sc.albedo_mode = MER_ALBEDO_CONST;
    if (make_params(ctx, &sc, P)) return 1;
    if (sc.modulation == MER_MODULATION_NONE) return fail(ctx, "Cannot call correlation function when the modulation type is not defined");   // pathlengthsampler.cpp:71-73
    DevBuf a(ctx), b(ctx);
    if (a.upload(path_length, n * 4) || b.alloc(n * 4)) return 1;
    hipLaunchKernelGGL(correlation_kernel, dim3(nblocks(n)), dim3(256), 0, ctx->stream, P, a.as<float>(), n, b.as<float>());
    HIP_CHECK(ctx, hipGetLastError());
    return b.download(out, n * 4);
}
int mer_rng_floats(mer_context *ctx, uint64_t seed, uint32_t pixel, uint32_t sample, int32_t n, float *out) {
    DevBuf a(ctx);
    if (a.alloc((size_t) n * 4)) return 1;
    hipLaunchKernelGGL(rng_kernel, dim3(1), dim3(64), 0, ctx->stream, seed, pixel, sample, n, a.as<float>());
    HIP_CHECK(ctx, hipGetLastError());
    return a.download(out, (size_t) n * 4);
}
int mer_synth_field_dev(mer_context *ctx, int32_t kind, int32_t N, float **data_dev) {
    if (kind < 0 || kind > 2 || N < 2) return fail(ctx, "mer_synth_field_dev: bad arguments");
    HIP_CHECK(ctx, hipMalloc((void **) data_dev, (size_t) N * N * N * 4));
    hipLaunchKernelGGL(synth_field_kernel, dim3(8192), dim3(256), 0, ctx->stream, kind, N, *data_dev);
    HIP_CHECK(ctx, hipGetLastError());
    HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

}  // extern "C"
