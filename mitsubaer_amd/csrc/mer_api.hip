// mer_api.hip -- libmer.so: C-ABI (include/mer.h) over the gfx950 kernels in mer_kernels.hpp.
// Host side is plain HIP runtime: device memory, one stream, HIP events.  No CPU compute path exists:
// every entry point that computes launches a kernel, and fails loudly when no device is present.
#include "mer_internal.hpp"
// every entry point runs on its context's device: a process may hold contexts of several GPUs (mer_multi.hip), and the current device is per-thread state
#define MER_USE_DEVICE(ctx) do { if (ctx) (void) hipSetDevice((ctx)->device); } while (0)
#include "mer_kernels.hpp"
#include <algorithm>
#include <functional>

using namespace mer;

namespace { thread_local std::string g_create_error; }

namespace mer {

// worldToGrid = scale((res-1)/extents) * translate(-min) * toWorld^-1 with toWorld = identity
// (GridDataSource::configure, src/volume/gridvolume.cpp:188-195).  Float arithmetic as in the reference.
void fill_dgrid(const mer_context *ctx, const Volume &v, DGrid &g) {
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
        g.buf_bytes = bytes < 0xFFFFFFFFull && ctx->opt.buffer_loads ? (uint32_t) bytes : 0u;
        g.n_record = v.cell8 ? bytes / 4 : 0;
        g.n_dense = (uint64_t) v.desc.res[0] * v.desc.res[1] * v.desc.res[2] * (uint64_t) v.desc.channels;
        g.chk = ctx->chk;
    }
    // worldToVolume: the desc's matrix, all zeros = identity
    float W[12]; bool zero = true;
    for (int i = 0; i < 12; i++) { W[i] = v.desc.world_to_volume[i]; zero = zero && W[i] == 0.0f; }
    if (zero) for (int i = 0; i < 12; i++) W[i] = (i % 5 == 0) ? 1.0f : 0.0f;
    g.affine = 0;
    for (int i = 0; i < 12; i++) { g.w2v[i] = W[i]; if (W[i] != ((i % 5 == 0) ? 1.0f : 0.0f)) g.affine = 1; }
    for (int i = 0; i < 3; i++) {
        g.res[i] = v.desc.res[i];
        g.bmin[i] = v.desc.aabb_min[i]; g.bmax[i] = v.desc.aabb_max[i];
        const float extent = g.bmax[i] - g.bmin[i];
        const float s = (float) (g.res[i] - 1) / extent;
        g.s[i] = s;
        g.t[i] = s * (-g.bmin[i]);
        // (scale * translate) * worldToVolume as Mitsuba's 4x4 product forms it (src/libcore/transform.cpp operator*): row i of the
        // left factor is (s_i e_i, s_i * (-min_i)); the zero terms of the sums add exactly nothing
        for (int j = 0; j < 3; j++) g.m[i * 4 + j] = s * W[i * 4 + j];
        g.m[i * 4 + 3] = s * W[i * 4 + 3] + g.t[i];
        // SplineDataSource interpolatable limits (src/volume/splinevolume.cpp:280-281): stride = 1/xres
        const float stride = (float) (1.0 / s);
        g.lim_min[i] = g.bmin[i] + (2.0f * stride + MER_EPSILON);
        g.lim_max[i] = g.bmax[i] + (-2.0f * stride - MER_EPSILON);
    }
    {   // m_aabb: bounding box of the data box's corners under volumeToWorld (gridvolume.cpp:199-203); volumeToWorld = W^-1 by cofactors
        // in double (the oracle forms it with the same expressions)
        const double a = W[0], b = W[1], c = W[2], d = W[4], e = W[5], f = W[6], gg = W[8], h = W[9], k = W[10];
        const double det = a * (e * k - f * h) - b * (d * k - f * gg) + c * (d * h - e * gg);
        const double inv[9] = {(e * k - f * h) / det, (c * h - b * k) / det, (b * f - c * e) / det,
                               (f * gg - d * k) / det, (a * k - c * gg) / det, (c * d - a * f) / det,
                               (d * h - e * gg) / det, (b * gg - a * h) / det, (a * e - b * d) / det};
        for (int i = 0; i < 3; i++) { g.wmin[i] = std::numeric_limits<float>::infinity(); g.wmax[i] = -std::numeric_limits<float>::infinity(); }
        for (int corner = 0; corner < 8; corner++) {
            const double q[3] = {((corner & 1) ? g.bmax[0] : g.bmin[0]) - (double) W[3], ((corner & 2) ? g.bmax[1] : g.bmin[1]) - (double) W[7],
                                 ((corner & 4) ? g.bmax[2] : g.bmin[2]) - (double) W[11]};
            for (int i = 0; i < 3; i++) {
                const float w = (float) (inv[i * 3] * q[0] + inv[i * 3 + 1] * q[1] + inv[i * 3 + 2] * q[2]);
                g.wmin[i] = std::min(g.wmin[i], w); g.wmax[i] = std::max(g.wmax[i], w);
            }
        }
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
    if (sc->decomposition != MER_DECOMPOSITION_TRANSIENT && sc->decomposition != MER_DECOMPOSITION_BOUNCE)
        return fail(ctx, "The \"decomposition\" parameter must be equal toeither \"none\", \"transient\", or \"bounce\"!");   // film.cpp:66-68
    const float f = std::ceil((sc->max_bound - sc->min_bound) / sc->bin_width);                                               // film.cpp:74
    if (!(f >= 1.0f) || f > 4096.0f) return fail(ctx, "film: a decomposition needs 1 <= ceil((maxBound-minBound)/binWidth) <= 4096 frames");
    frames = (int) f;
    return 0;
}
int make_params(mer_context *ctx, const mer_scene_desc *sc, Params &P, bool allow_sdf) {
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
        fill_dgrid(ctx, it->second, P.density);
        if (!(sc->density_scale > 0)) return fail(ctx, "heterogeneous medium: 'scale' must be positive");
        // m_maxDensity = m_scale * getMaximumFloatValue() (= 1.0 for gridvolume): heterogeneous.cpp:239-242
        P.inv_max_density = 1.0f / (sc->density_scale * 1.0f);
        if (sc->method != MER_METHOD_WOODCOCK && sc->method != MER_METHOD_SIMPSON) return fail(ctx, "Unsupported integration method!");    // heterogeneous.cpp:195-202
        if (sc->method == MER_METHOD_SIMPSON) {
            if (sc->rif_mode != MER_RIF_CONST) return fail(ctx, "method = simpson belongs to the heterogeneous medium (straight rays)");
            auto step_of = [](const mer_grid_desc &g) {                      // gridvolume.cpp:196-198
                float s = std::numeric_limits<float>::infinity();
                for (int i = 0; i < 3; i++) s = std::min(s, 0.5f * (g.aabb_max[i] - g.aabb_min[i]) / (float) (g.res[i] - 1));
                return s;
            };
            float h = sc->het_stepsize;                                      // heterogeneous.cpp:245-257
            if (h == 0) {
                h = step_of(ctx->volumes.find(sc->density)->second.desc);
                if (sc->albedo_mode == MER_ALBEDO_GRID) { auto ia = ctx->volumes.find(sc->albedo_grid); if (ia != ctx->volumes.end()) h = std::min(h, step_of(ia->second.desc)); }
            }
            if (!(h > 0) || !std::isfinite(h))
                return fail(ctx, "Unable to infer a suitable step size for deterministic integration, please specify one manually using the 'stepSize' parameter.");
            P.het_step = h;
            P.sc.tr_estimator = MER_TR_RATIO;        // one walk per transmittance query (the estimator choice is the Woodcock method's)
        }
    }
    if (sc->albedo_mode == MER_ALBEDO_GRID) {
        auto it = ctx->volumes.find(sc->albedo_grid);
        if (it == ctx->volumes.end()) return fail(ctx, "No albedo specified!");                                     // heterogeneous.cpp:231-232
        if (it->second.desc.channels != 3) return fail(ctx, "albedo volume must support spectrum lookups");
        Volume tmp = it->second; tmp.cell8 = nullptr;
        fill_dgrid(ctx, tmp, P.albedo);
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
        fill_dgrid(ctx, it->second, P.rif);
        // the fetch index (z * res_y + y) * res_x + x is formed with 24-bit multiplies (v_mul_u32_u24)
        if ((int64_t) P.rif.res[1] * P.rif.res[2] > ((int64_t) 1 << 24) || P.rif.res[0] > (1 << 24))
            return fail(ctx, "RIF volume: res_y * res_z must not exceed 2^24 (index arithmetic of the trilinear fetch)");
        if ((int64_t) P.rif.res[0] * P.rif.res[1] * P.rif.res[2] >= ((int64_t) 1 << 31))
            return fail(ctx, "RIF volume: more than 2^31 nodes (the cell id of the trilinear fetch is a 32-bit integer)");
        if (P.rif.affine && it->second.cell8)
            return fail(ctx, "RIF volume with a toWorld transform: upload it in the dense layout (the CELL8 / BRICK record layouts carry no transform)");
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
    } else if (sc->strategy == MER_STRATEGY_MAXIMUM) {
        // MaxExpDist's constructor (src/medium/maxexp.h:30-58), in the reference's float arithmetic
        MaxExp &m = P.maxexp;
        for (int i = 0; i < 3; i++) m.sigmaT[i] = sT[i];
        std::sort(m.sigmaT, m.sigmaT + 3, std::greater<float>());
        m.cdf[0] = 0;
        for (int i = 0; i < 3; ++i) {
            if (i > 0 && m.sigmaT[i] == m.sigmaT[i - 1]) return fail(ctx, "Internal error: sigmaT must vary across channels");
            if (!(m.sigmaT[i] > 0)) return fail(ctx, "strategy maximum: sigmaT must be positive in every channel");
            const float lower = (i == 0) ? -1 : -std::pow((m.sigmaT[i] / m.sigmaT[i - 1]), -m.sigmaT[i] / (m.sigmaT[i] - m.sigmaT[i - 1]));
            const float upper = (i == 2) ? 0 : -std::pow((m.sigmaT[i + 1] / m.sigmaT[i]), -m.sigmaT[i] / (m.sigmaT[i + 1] - m.sigmaT[i]));
            m.cdf[i + 1] = m.cdf[i] + (upper - lower);
            m.intervalStart[i] = (i == 0) ? 0 : std::log(m.sigmaT[i] / m.sigmaT[i - 1]) / (m.sigmaT[i] - m.sigmaT[i - 1]);
        }
        m.normalization = m.cdf[3]; m.invNormalization = 1 / m.normalization;
        for (int i = 0; i < 4; ++i) m.cdf[i] *= m.invNormalization;
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
    {   // the table goes to device memory once per (kind, parameter); no kernel of this context is in flight here (renders and leaf calls return synchronised)
        float fv[33];
        filter_table(sc->rfilter, sc->rfilter_param, fv, P.fradius, P.fscale);
        if (!ctx->ftable) HIP_CHECK(ctx, hipMalloc((void **) &ctx->ftable, sizeof(fv)));
        if (ctx->ftable_kind != sc->rfilter || ctx->ftable_param != sc->rfilter_param) {
            HIP_CHECK(ctx, hipMemcpy(ctx->ftable, fv, sizeof(fv), hipMemcpyHostToDevice));
            ctx->ftable_kind = sc->rfilter; ctx->ftable_param = sc->rfilter_param;
        }
        P.ftable = ctx->ftable;
    }
    if (P.fradius > 7.0f) return fail(ctx, "reconstruction filter radius too large");
    if (sc->boundary_bsdf != MER_BSDF_NULL && sc->boundary_bsdf != MER_BSDF_HDIELECTRIC) return fail(ctx, "boundary BSDF must be null or hdielectric");
    P.has_area = (sc->area_radiance[0] != 0 || sc->area_radiance[1] != 0 || sc->area_radiance[2] != 0) ? 1 : 0;
    if (P.has_area) {               // Rectangle::configure (src/shapes/rectangle.cpp:99-110)
        if (sc->rif_mode != MER_RIF_CONST) return fail(ctx, "the area emitter is built for straight rays (rif_mode = CONST)");
        if (sc->boundary_bsdf != MER_BSDF_NULL || sc->boundary == MER_BOUNDARY_SDF) return fail(ctx, "the area emitter needs an index-matched cube / sphere boundary");
        double M[3][4], inv[3][3];
        for (int i = 0; i < 12; i++) { P.rect_o2w[i] = sc->area_to_world[i]; M[i / 4][i % 4] = sc->area_to_world[i]; }
        const double det = M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) + M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
        if (!(std::fabs(det) > 0)) return fail(ctx, "area emitter: 'toWorld' is singular");
        inv[0][0] = (M[1][1] * M[2][2] - M[1][2] * M[2][1]) / det; inv[0][1] = (M[0][2] * M[2][1] - M[0][1] * M[2][2]) / det; inv[0][2] = (M[0][1] * M[1][2] - M[0][2] * M[1][1]) / det;
        inv[1][0] = (M[1][2] * M[2][0] - M[1][0] * M[2][2]) / det; inv[1][1] = (M[0][0] * M[2][2] - M[0][2] * M[2][0]) / det; inv[1][2] = (M[0][2] * M[1][0] - M[0][0] * M[1][2]) / det;
        inv[2][0] = (M[1][0] * M[2][1] - M[1][1] * M[2][0]) / det; inv[2][1] = (M[0][1] * M[2][0] - M[0][0] * M[2][1]) / det; inv[2][2] = (M[0][0] * M[1][1] - M[0][1] * M[1][0]) / det;
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) P.rect_w2o[4 * i + j] = (float) inv[i][j];
            P.rect_w2o[4 * i + 3] = (float) -(inv[i][0] * M[0][3] + inv[i][1] * M[1][3] + inv[i][2] * M[2][3]);
        }
        const double du[3] = {2 * M[0][0], 2 * M[1][0], 2 * M[2][0]}, dv[3] = {2 * M[0][1], 2 * M[1][1], 2 * M[2][1]};
        const double lu = std::sqrt(du[0] * du[0] + du[1] * du[1] + du[2] * du[2]), lv = std::sqrt(dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2]);
        if (std::fabs((du[0] * dv[0] + du[1] * dv[1] + du[2] * dv[2]) / (lu * lv)) > MER_EPSILON) return fail(ctx, "Error: 'toWorld' transformation contains shear!");    // :108-109
        const double nn[3] = {inv[2][0], inv[2][1], inv[2][2]}, ln = std::sqrt(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]);   // o2w(Normal(0,0,1)): inverse transpose
        for (int i = 0; i < 3; i++) P.rect_n[i] = (float) (nn[i] / ln);
        P.rect_inv_area = (float) (1.0 / (lu * lv));
        // the rectangle must lie outside the (convex) medium shape: corners and centre are tested
        for (int k = 0; k < 5; k++) {
            const float lx = k == 4 ? 0.0f : (k & 1 ? 1.0f : -1.0f), ly = k == 4 ? 0.0f : (k & 2 ? 1.0f : -1.0f);
            const float q[3] = {(float) (M[0][0] * lx + M[0][1] * ly + M[0][3]), (float) (M[1][0] * lx + M[1][1] * ly + M[1][3]), (float) (M[2][0] * lx + M[2][1] * ly + M[2][3])};
            bool in = true;
            if (sc->boundary == MER_BOUNDARY_SPHERE) { float d2 = 0; for (int i = 0; i < 3; i++) d2 += (q[i] - sc->sph_center[i]) * (q[i] - sc->sph_center[i]); in = d2 < sc->sph_radius * sc->sph_radius; }
            else for (int i = 0; i < 3; i++) in = in && q[i] >= sc->bmin[i] && q[i] <= sc->bmax[i];
            if (in) return fail(ctx, "the area emitter's rectangle must lie outside the medium shape");
        }
    }
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
        fill_dgrid(ctx, it->second, P.sdf);
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
        (void) has_point;      // curved rays reach a point emitter outside the shape through the boundary (Connector::path_lengths, cross = true)
    }
    P.counters = ctx->counters;
    P.work_counter = ctx->counters + MER_C_COUNT * MER_COUNTER_REPLICAS;
    P.chk = ctx->chk;
    P.dbg_pixel = (int32_t) ctx->opt.debug_pixel;
    return 0;
}

int rif_fetch_kind(mer_context *ctx, const mer_scene_desc *sc) {
    if (sc->rif_mode != MER_RIF_TRILINEAR) return sc->rif_mode;
    const Volume &rv = ctx->volumes.find(sc->rif)->second;
    DGrid tmp; fill_dgrid(ctx, rv, tmp);
    if (tmp.layout == MER_LAYOUT_BRICK27 || tmp.layout == MER_LAYOUT_BRICK125) return tmp.buf_bytes ? RIFK_BRICK27_BUF : RIFK_BRICK27;
    if (tmp.layout == MER_LAYOUT_CELL8) return tmp.buf_bytes ? RIFK_CELL8_BUF : RIFK_CELL8;
    return tmp.buf_bytes ? RIFK_DENSE_BUF : MER_RIF_TRILINEAR;
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
    const int rifk = rif_fetch_kind(ctx, sc);
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
}  // namespace mer

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
#ifdef MER_BOUNDS_CHECK
    if (hipMalloc((void **) &ctx->chk, 4 * sizeof(unsigned long long)) != hipSuccess || hipMemset(ctx->chk, 0, 4 * sizeof(unsigned long long)) != hipSuccess) {
        g_create_error = "mer_context_create: device allocation failed"; delete ctx; return 1;
    }
#endif
    // MER_OPTIONS="name=value,name=value": initial option values for A/B scripts (read once, here; mer_context_set_option afterwards)
    if (const char *e = getenv("MER_OPTIONS")) {
        std::string all(e); size_t pos = 0;
        while (pos < all.size()) {
            size_t end = all.find(',', pos); if (end == std::string::npos) end = all.size();
            const std::string kv = all.substr(pos, end - pos); const size_t eq = kv.find('=');
            if (eq != std::string::npos && mer_context_set_option(ctx, kv.substr(0, eq).c_str(), atoll(kv.c_str() + eq + 1)) != 0) {
                g_create_error = "mer_context_create: MER_OPTIONS: " + ctx->error; mer_context_destroy(ctx); return 1;
            }
            pos = end + 1;
        }
    }
    *out = ctx;
    return 0;
}

static int64_t *option_slot(mer_context *ctx, const char *name) {
    Options &o = ctx->opt;
    const struct { const char *n; int64_t *p; } table[] = {
        {"pipes", &o.pipes}, {"nslots", &o.nslots}, {"ksteps", &o.ksteps}, {"mq_sort", &o.mq_sort}, {"connect_launches", &o.connect_launches},
        {"adaptive_k", &o.adaptive_k}, {"pass_events", &o.pass_events}, {"buffer_loads", &o.buffer_loads}, {"gen_all", &o.gen_all},
        {"prefilter", &o.prefilter}, {"verbose", &o.verbose}, {"debug_pixel", &o.debug_pixel}, {"lds_bricks", &o.lds_bricks}, {"march_lds_kb", &o.march_lds_kb},
        {"tile_deal", &o.tile_deal}, {"small_render_slots", &o.small_render_slots}, {"inline_walks", &o.inline_walks}, {"spawn_walks", &o.spawn_walks},
        {"grid_fit", &o.grid_fit}, {"check_every", &o.check_every}, {"march_sort", &o.march_sort}, {"march_sort_major", &o.march_sort_major}};
    for (const auto &t : table) if (std::strcmp(t.n, name) == 0) return t.p;
    return nullptr;
}
int mer_context_set_option(mer_context *ctx, const char *name, int64_t value) {
    MER_USE_DEVICE(ctx);
    if (!ctx || !name) return 1;
    int64_t *p = option_slot(ctx, name);
    if (!p) return fail(ctx, std::string("unknown option '") + name + "'");
    const std::string n(name);
    if ((n == "pipes" && (value < 1 || value > MER_MAX_PIPES)) || (n == "ksteps" && (value < 1 || value > (1 << 20))) ||
        (n == "connect_launches" && (value < 1 || value > 64)) ||
        // nslots: 0 = default, otherwise at least one block per pipeline (a smaller value would launch empty grids)
        (n == "nslots" && (value < 0 || (value > 0 && value < MER_BLOCK * MER_MAX_PIPES) || value > ((int64_t) 1 << 28))) ||
        (n == "prefilter" && (value < 0 || value > 5)) || (n == "mq_sort" && (value < -1 || value > 1)) ||
        (n == "march_lds_kb" && (value < 0 || value > 64)) ||          // dynamic LDS above 64 KiB would need hipFuncSetAttribute
        (n == "debug_pixel" && (value < -1 || value > ((int64_t) 1 << 31) - 1)) || (n == "tile_deal" && (value < 0 || value > 1)) ||
        (n == "small_render_slots" && (value < 0 || value > 1)) || (n == "inline_walks" && (value < 0 || value > 1)) || (n == "spawn_walks" && (value < 0 || value > 1)) || (n == "adaptive_k" && (value < 0 || value > 2)) ||
        (n == "grid_fit" && (value < 0 || value > 1)) || (n == "check_every" && (value < 1 || value > 64)) || (n == "march_sort" && (value < 0 || value > 4)) || (n == "march_sort_major" && (value < 0 || value > 1)))
        return fail(ctx, std::string("option '") + name + "': value out of range");
    *p = value;
    return 0;
}
int mer_context_get_option(mer_context *ctx, const char *name, int64_t *value) {
    MER_USE_DEVICE(ctx);
    if (!ctx || !name || !value) return 1;
    int64_t *p = option_slot(ctx, name);
    if (!p) return fail(ctx, std::string("unknown option '") + name + "'");
    *value = *p;
    return 0;
}
int mer_debug_bounds(mer_context *ctx, int32_t *enabled, uint64_t out[4]) {
    MER_USE_DEVICE(ctx);
    if (!ctx || !enabled || !out) return 1;
    out[0] = out[1] = out[2] = out[3] = 0;
    *enabled = ctx->chk != nullptr;
    if (ctx->chk) {
        HIP_CHECK(ctx, hipDeviceSynchronize());
        HIP_CHECK(ctx, hipMemcpy(out, ctx->chk, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost));
        HIP_CHECK(ctx, hipMemset(ctx->chk, 0, 4 * sizeof(uint64_t)));
    }
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
    if (ctx->ftable) (void) hipFree(ctx->ftable);
    if (ctx->chk) (void) hipFree(ctx->chk);
    for (Pipe &pp : ctx->pipes) {
        if (pp.slots) (void) hipFree(pp.slots);
        if (pp.live) (void) hipFree(pp.live);
        for (SegQueue *q : {&pp.eq, &pp.mq[0], &pp.mq[1], &pp.sq[0], &pp.sq[1], &pp.cq[0], &pp.cq[1]}) { if (q->items) (void) hipFree(q->items); if (q->counts) (void) hipFree(q->counts); if (q->keys) (void) hipFree(q->keys); }
        if (pp.msort) (void) hipFree(pp.msort);
        if (pp.cstate) (void) hipFree(pp.cstate);
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
    MER_USE_DEVICE(ctx);
    if (name && name_len > 0) { std::strncpy(name, ctx->prop.name, name_len - 1); name[name_len - 1] = 0; }
    if (cu_count) *cu_count = ctx->prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t) ctx->prop.totalGlobalMem;
    return 0;
}

static int volume_finish(mer_context *ctx, Volume &v, int32_t layout, mer_volume *out) {
    if (layout == MER_LAYOUT_AUTO) {
        const int64_t nodes = (int64_t) v.desc.res[0] * v.desc.res[1] * v.desc.res[2];
        bool affine = false, zero = true;          // a toWorld transform: the record layouts carry none
        for (int i = 0; i < 12; i++) { const float w = v.desc.world_to_volume[i]; zero = zero && w == 0; affine = affine || w != ((i % 5 == 0) ? 1.0f : 0.0f); }
        layout = (v.desc.channels != 1 || v.desc.dtype != MER_VOL_F32 || (affine && !zero)) ? MER_LAYOUT_DENSE : (nodes <= ((int64_t) 1 << 28) ? MER_LAYOUT_BRICK27 : MER_LAYOUT_CELL8);
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
    {   // world_to_volume: all zeros (identity) or an invertible affine map
        const float *W = d->world_to_volume; bool zero = true, finite = true;
        for (int i = 0; i < 12; i++) { zero = zero && W[i] == 0.0f; finite = finite && std::isfinite(W[i]); }
        const double det = (double) W[0] * ((double) W[5] * W[10] - (double) W[6] * W[9]) - (double) W[1] * ((double) W[4] * W[10] - (double) W[6] * W[8]) +
                           (double) W[2] * ((double) W[4] * W[9] - (double) W[5] * W[8]);
        if (!zero && (!finite || !(std::fabs(det) > 1e-12))) return fail(ctx, "volume: the toWorld transform is not invertible");
    }
    return 0;
}

int mer_volume_upload(mer_context *ctx, const mer_grid_desc *desc, const void *host_data, int32_t layout, mer_volume *out) {
    MER_USE_DEVICE(ctx);
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
    MER_USE_DEVICE(ctx);
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
    MER_USE_DEVICE(ctx);
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
    const int64_t form = ctx->opt.prefilter;
    const bool seq = form == 1 || std::min(nx, std::min(ny, nz)) < 16;
    if (seq) {                 // one thread per line (reference order of operations; tiny grids)
        hipLaunchKernelGGL(bspline_pass_kernel, dim3(nblocks((int64_t) nx * nz)), dim3(256), 0, ctx->stream,
                           (const float *) v.dense, a, nx, nz, (int64_t) 1, (int64_t) nx * ny, (int64_t) nx, ny);
        hipLaunchKernelGGL(bspline_pass_kernel, dim3(nblocks((int64_t) ny * nz)), dim3(256), 0, ctx->stream,
                           (const float *) a, b, ny, nz, (int64_t) nx, (int64_t) nx * ny, (int64_t) 1, nx);
        hipLaunchKernelGGL(bspline_pass_kernel, dim3(nblocks((int64_t) nx * ny)), dim3(256), 0, ctx->stream,
                           (const float *) b, a, nx, ny, (int64_t) 1, (int64_t) nx, (int64_t) nx * ny, nz);
    } else {
        if (form == 2 || form == 3) HIP_CHECK(ctx, hipMalloc((void **) &t, n * 4));
        auto pass = [&](const float *src, float *dst, int na, int nb, int64_t sa, int64_t sb, int64_t sl, int size) {
            const int64_t threads = (int64_t) na * nb * ((size + MER_PF_SEG - 1) / MER_PF_SEG);
            hipLaunchKernelGGL(bspline_causal_kernel, dim3(nblocks(threads)), dim3(256), 0, ctx->stream, src, t, na, nb, sa, sb, sl, size);
            hipLaunchKernelGGL(bspline_anticausal_kernel, dim3(nblocks(threads)), dim3(256), 0, ctx->stream, (const float *) t, dst, na, nb, sa, sb, sl, size);
        };
        auto win = [&](const float *src, float *dst, int na, int nb, int64_t sb, int64_t sl, int size) {     // fused sweeps, unit stride in a
            const int nseg = (size + MER_PF_SEG - 1) / MER_PF_SEG;
            // interior segments (window inside the line): wave-uniform addressing, no per-sample conditions (bspline_win2_kernel); border segments: generic kernel
            int s_lo = 0, s_hi = nseg;
            if (form != 5 && nb <= 65535 && sl * 4 * (int64_t) MER_PFX_W < ((int64_t) 1 << 31)) {
                while (s_lo < nseg && !(s_lo * MER_PF_SEG - MER_PF_WARM > 0)) s_lo++;
                s_hi = s_lo;
                while (s_hi < nseg && s_hi * MER_PF_SEG - MER_PF_WARM + MER_PFX_W < size) s_hi++;
            } else s_lo = nseg;
            if (s_hi > s_lo)
                hipLaunchKernelGGL(bspline_win2_kernel, dim3((unsigned) ((na + 255) / 256), (unsigned) (s_hi - s_lo), (unsigned) nb), dim3(256), 0, ctx->stream, src, dst, na, sb, sl, size, s_lo);
            else { s_lo = nseg; s_hi = nseg; }
            const int64_t threads = (int64_t) na * nb * (s_lo + (nseg - s_hi));
            if (threads > 0) hipLaunchKernelGGL(bspline_win_kernel, dim3(nblocks(threads)), dim3(256), 0, ctx->stream, src, dst, na, nb, sb, sl, size, s_lo, s_hi);
        };
        const bool two_kernel = form == 2;
        if (two_kernel) pass((const float *) v.dense, a, nx, nz, 1, (int64_t) nx * ny, nx, ny);          // y
        else win((const float *) v.dense, a, nx, nz, (int64_t) nx * ny, nx, ny);
        if (form == 3) pass(a, b, ny, nz, nx, (int64_t) nx * ny, 1, nx);           // x, strided form
        else {                                                                                        // x: lines are contiguous -> LDS tiles
            const int64_t nlines = (int64_t) ny * nz;
            const int ntile = (nx + MER_PFX_COLS - 1) / MER_PFX_COLS;
            int t_lo = ntile, t_hi = ntile;                       // interior segments in registers (n % 4 == 0), border tiles through LDS
            if (nx % 4 == 0 && form != 4) {
                t_lo = 0; while (t_lo < ntile && !(t_lo * MER_PF_SEG - MER_PF_WARM > 0)) t_lo++;
                t_hi = t_lo; while (t_hi < ntile && t_hi * MER_PF_SEG - MER_PF_WARM + MER_PFX_W < nx) t_hi++;
                if (t_hi > t_lo) {
                    const int64_t threads = nlines * (t_hi - t_lo);
                    hipLaunchKernelGGL(bspline_x_reg_kernel, dim3(nblocks(threads)), dim3(256), 0, ctx->stream, (const float *) a, b, nlines, nx, t_lo, t_hi - t_lo);
                } else t_lo = t_hi = ntile;
            }
            if (nx % 4 == 0 && form != 4) {        // border segments: the guarded register form
                const int64_t threads = nlines * (t_lo + (ntile - t_hi));
                if (threads > 0) hipLaunchKernelGGL(bspline_x_reg_border_kernel, dim3(nblocks(threads)), dim3(256), 0, ctx->stream, (const float *) a, b, nlines, nx, t_lo, t_hi);
            } else {                                // any line length: LDS tiles
                const int64_t blocks = ((nlines + MER_PFX_ROWS - 1) / MER_PFX_ROWS) * (t_lo + (ntile - t_hi));
                if (blocks > 0) hipLaunchKernelGGL(bspline_x_kernel, dim3((unsigned) blocks), dim3(256), 0, ctx->stream, (const float *) a, b, nlines, nx, t_lo, t_hi);
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
    MER_USE_DEVICE(ctx);
    auto it = ctx->volumes.find(h);
    if (it == ctx->volumes.end() || !it->second.coeff) return fail(ctx, "volume has no spline coefficients");
    const size_t n = (size_t) it->second.desc.res[0] * it->second.desc.res[1] * it->second.desc.res[2];
    HIP_CHECK(ctx, hipMemcpy(coeff_host, it->second.coeff, n * 4, hipMemcpyDeviceToHost));
    return 0;
}

int mer_volume_destroy(mer_context *ctx, mer_volume h) {
    MER_USE_DEVICE(ctx);
    auto it = ctx->volumes.find(h);
    if (it == ctx->volumes.end()) return fail(ctx, "invalid volume handle");
    if (it->second.dense && it->second.owns_dense) (void) hipFree(it->second.dense);
    if (it->second.cell8) (void) hipFree(it->second.cell8);
    if (it->second.coeff) (void) hipFree(it->second.coeff);
    ctx->volumes.erase(it);
    return 0;
}

int mer_film_channels(mer_context *ctx, const mer_scene_desc *scene, int32_t *channels) {
    MER_USE_DEVICE(ctx);
    int frames;
    if (!scene || !channels) return 1;
    if (film_frames(ctx, scene, frames)) return 1;
    *channels = frames * 3 + 2;
    return 0;
}
int mer_film_alloc_n(mer_context *ctx, int32_t width, int32_t height, int32_t channels, float **film_dev) {
    MER_USE_DEVICE(ctx);
    HIP_CHECK(ctx, hipMalloc((void **) film_dev, (size_t) width * height * channels * sizeof(float)));
    HIP_CHECK(ctx, hipMemsetAsync(*film_dev, 0, (size_t) width * height * channels * sizeof(float), ctx->stream));
    return 0;
}
int mer_film_zero_n(mer_context *ctx, float *film_dev, int32_t width, int32_t height, int32_t channels) {
    MER_USE_DEVICE(ctx);
    HIP_CHECK(ctx, hipMemsetAsync(film_dev, 0, (size_t) width * height * channels * sizeof(float), ctx->stream));
    return 0;
}
int mer_film_download_n(mer_context *ctx, const float *film_dev, int32_t width, int32_t height, int32_t channels, float *film_host) {
    MER_USE_DEVICE(ctx);
    HIP_CHECK(ctx, hipMemcpyAsync(film_host, film_dev, (size_t) width * height * channels * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}
int mer_film_alloc(mer_context *ctx, int32_t width, int32_t height, float **film_dev) { return mer_film_alloc_n(ctx, width, height, 5, film_dev); }
int mer_film_zero(mer_context *ctx, float *film_dev, int32_t width, int32_t height) { return mer_film_zero_n(ctx, film_dev, width, height, 5); }
int mer_film_download(mer_context *ctx, const float *film_dev, int32_t width, int32_t height, float *film_host) {
    MER_USE_DEVICE(ctx);
    return mer_film_download_n(ctx, film_dev, width, height, 5, film_host);
}
int mer_film_free(mer_context *ctx, float *film_dev) { HIP_CHECK(ctx, hipFree(film_dev)); return 0; }
int mer_device_free(mer_context *ctx, void *p) { HIP_CHECK(ctx, hipFree(p)); return 0; }

int mer_render(mer_context *ctx, const mer_scene_desc *scene, const mer_shard *shard, uint64_t seed, float *film_dev) {
    MER_USE_DEVICE(ctx);
    if (!ctx || !scene || !film_dev) return 1;
    int32_t ch = 5;
    if (mer_film_channels(ctx, scene, &ch)) return 1;
    return launch_render(ctx, scene, shard, seed, film_dev, nullptr, (uint64_t) scene->width * scene->height * (uint64_t) ch, 0);
}

int mer_render_paths(mer_context *ctx, const mer_scene_desc *scene, int32_t sample_index, uint64_t seed, float *out_rgb) {
    MER_USE_DEVICE(ctx);
    if (!ctx || !scene || !out_rgb) return 1;
    const size_t n = (size_t) scene->width * scene->height * 3;
    DevBuf buf(ctx);
    if (buf.alloc(n * 4)) return 1;
    HIP_CHECK(ctx, hipMemsetAsync(buf.p, 0, n * 4, ctx->stream));
    mer_shard sh = {sample_index, 1, 1, 0, 1};
    if (launch_render(ctx, scene, &sh, seed, buf.as<float>(), buf.as<float>(), n, n)) return 1;
    return buf.download(out_rgb, n * 4);
}

int mer_synchronize(mer_context *ctx) { HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream)); return 0; }

int mer_last_kernel_ms(mer_context *ctx, float *ms) {
    MER_USE_DEVICE(ctx);
    if (!ctx->timed) return fail(ctx, "no render has been launched");
    HIP_CHECK(ctx, hipEventSynchronize(ctx->ev1));
    HIP_CHECK(ctx, hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return 0;
}

int mer_last_render_stats(mer_context *ctx, int32_t *passes, float *march_ms, float *event_ms) {
    MER_USE_DEVICE(ctx);
    if (!ctx->timed) return fail(ctx, "no render has been launched");
    if (passes) *passes = ctx->last_passes;
    if (march_ms) *march_ms = ctx->last_march_ms;
    if (event_ms) *event_ms = ctx->last_event_ms;
    return 0;
}

int mer_counters_read(mer_context *ctx, uint64_t out[MER_C_COUNT]) {
    MER_USE_DEVICE(ctx);
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
    MER_USE_DEVICE(ctx);
    HIP_CHECK(ctx, hipMemsetAsync(ctx->counters, 0, sizeof(uint64_t) * MER_C_COUNT * MER_COUNTER_REPLICAS, ctx->stream));
    return 0;
}

// ---- leaf entry points ------------------------------------------------------------------------------
int mer_lookup_trilinear(mer_context *ctx, mer_volume h, const float *pts, int64_t n, float *out_val, int32_t *out_idx) {
    MER_USE_DEVICE(ctx);
    auto it = ctx->volumes.find(h);
    if (it == ctx->volumes.end()) return fail(ctx, "invalid volume handle");
    if (it->second.desc.channels != 1) return fail(ctx, "lookupFloat(): volume does not support float lookups");
    DGrid g; fill_dgrid(ctx, it->second, g);
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
    MER_USE_DEVICE(ctx);
    auto it = ctx->volumes.find(h);
    if (it == ctx->volumes.end()) return fail(ctx, "invalid volume handle");
    if (it->second.desc.channels != 3) return fail(ctx, "lookupSpectrum(): volume does not support spectrum lookups");
    Volume tmp = it->second; tmp.cell8 = nullptr;
    DGrid g; fill_dgrid(ctx, tmp, g);
    DevBuf dp(ctx), dv(ctx);
    if (dp.upload(pts, n * 12) || dv.alloc(n * 12)) return 1;
    hipLaunchKernelGGL(lookup_rgb_kernel, dim3(nblocks(n)), dim3(256), 0, ctx->stream, g, dp.as<float>(), n, dv.as<float>());
    HIP_CHECK(ctx, hipGetLastError());
    return dv.download(out_rgb, n * 12);
}
int mer_rif_value_grad(mer_context *ctx, mer_volume h, int32_t interp, const float *pts, int64_t n, float *out_val, float *out_grad) {
    MER_USE_DEVICE(ctx);
    auto it = ctx->volumes.find(h);
    if (it == ctx->volumes.end()) return fail(ctx, "invalid volume handle");
    if (it->second.desc.channels != 1 || it->second.desc.dtype != MER_VOL_F32) return fail(ctx, "value(): not implemented for this volume type"); // volume.cpp:57-80
    if (interp != MER_RIF_TRILINEAR && interp != MER_RIF_BSPLINE3) return fail(ctx, "unknown rif_interp");
    if (interp == MER_RIF_BSPLINE3 && !it->second.coeff) return fail(ctx, "volume has no spline coefficients");
    DGrid g; fill_dgrid(ctx, it->second, g);
    DevBuf dp(ctx), dv(ctx), dg(ctx);
    if (dp.upload(pts, n * 12) || dv.alloc(n * 4) || dg.alloc(n * 12)) return 1;
    hipLaunchKernelGGL(rif_value_grad_kernel, dim3(nblocks(n)), dim3(256), 0, ctx->stream, g, interp, dp.as<float>(), n, dv.as<float>(), dg.as<float>());
    HIP_CHECK(ctx, hipGetLastError());
    if (dv.download(out_val, n * 4)) return 1;
    return dg.download(out_grad, n * 12);
}

int mer_er_trace(mer_context *ctx, const mer_scene_desc *scene, const float *p0, const float *d0, const float *dist, int64_t n,
                 float *out_p, float *out_v, float *out_dist_surf, float *out_opt, int32_t *out_success) {
    MER_USE_DEVICE(ctx);
    Params P;
    if (make_params(ctx, scene, P)) return 1;
    if (scene->rif_mode == MER_RIF_CONST) return fail(ctx, "mer_er_trace needs a RIF volume");
    DevBuf a(ctx), b(ctx), c(ctx), op(ctx), ov(ctx), od(ctx), oo(ctx), ok(ctx);
    if (a.upload(p0, n * 12) || b.upload(d0, n * 12) || c.upload(dist, n * 4) || op.alloc(n * 12) || ov.alloc(n * 12) ||
        od.alloc(n * 4) || oo.alloc(n * 4) || ok.alloc(n * 4)) return 1;
    const int rifk = rif_fetch_kind(ctx, scene);
#define MER_TRACE_CASE(R, S)                                                                                       \
    if (rifk == R && scene->stepper == S)                                                                          \
        hipLaunchKernelGGL((er_trace_kernel<R, S>), dim3(nblocks(n, 64)), dim3(64), 0, ctx->stream, P, a.as<float>(), b.as<float>(), \
                           c.as<float>(), n, op.as<float>(), ov.as<float>(), od.as<float>(), oo.as<float>(), ok.as<int32_t>());
    MER_TRACE_CASE(MER_RIF_TRILINEAR, MER_STEP_VERLET) MER_TRACE_CASE(MER_RIF_TRILINEAR, MER_STEP_RK4)
    MER_TRACE_CASE(RIFK_DENSE_BUF, MER_STEP_VERLET) MER_TRACE_CASE(RIFK_DENSE_BUF, MER_STEP_RK4)
    MER_TRACE_CASE(RIFK_CELL8, MER_STEP_VERLET) MER_TRACE_CASE(RIFK_CELL8, MER_STEP_RK4)
    MER_TRACE_CASE(RIFK_CELL8_BUF, MER_STEP_VERLET) MER_TRACE_CASE(RIFK_CELL8_BUF, MER_STEP_RK4)
    MER_TRACE_CASE(RIFK_BRICK27_BUF, MER_STEP_VERLET) MER_TRACE_CASE(RIFK_BRICK27_BUF, MER_STEP_RK4)      // the bench layout (< 4 GiB: buffer loads)
    MER_TRACE_CASE(RIFK_BRICK27, MER_STEP_VERLET) MER_TRACE_CASE(RIFK_BRICK27, MER_STEP_RK4)
    MER_TRACE_CASE(MER_RIF_BSPLINE3, MER_STEP_VERLET) MER_TRACE_CASE(MER_RIF_BSPLINE3, MER_STEP_RK4)
    MER_TRACE_CASE(RIFK_ACOUSTIC, MER_STEP_VERLET) MER_TRACE_CASE(RIFK_ACOUSTIC, MER_STEP_RK4)
#undef MER_TRACE_CASE
    HIP_CHECK(ctx, hipGetLastError());
    if (op.download(out_p, n * 12) || ov.download(out_v, n * 12) || od.download(out_dist_surf, n * 4) || oo.download(out_opt, n * 4) ||
        ok.download(out_success, n * 4)) return 1;
    return 0;
}

int mer_connect(mer_context *ctx, const mer_scene_desc *scene, const float *p1, const float *p2, int64_t n, uint64_t seed, float *out) {
    MER_USE_DEVICE(ctx);
    Params P;
    if (make_params(ctx, scene, P, true)) return 1;
    if (scene->rif_mode == MER_RIF_CONST) return fail(ctx, "mer_connect needs a RIF volume");
    if (scene->rif_mode == MER_RIF_ACOUSTIC) return fail(ctx, "mer_connect: the analytic acoustic RIF is connected inside mer_render only");
    P.seed = seed;
    DevBuf a(ctx), b(ctx), r(ctx);
    if (a.upload(p1, n * 12) || b.upload(p2, n * 12) || r.alloc(n * 48)) return 1;
    const int rifk = rif_fetch_kind(ctx, scene);
    bool launched = false;
#define MER_CONNECT_CASE(R, B) if (!launched && rifk == R && (scene->boundary == MER_BOUNDARY_SDF) == (B == 1)) { launched = true;   \
        hipLaunchKernelGGL((connect_kernel<R, B>), dim3(nblocks(n, 64)), dim3(64), 0, ctx->stream, P, a.as<float>(), b.as<float>(), n, r.as<float>()); }
    MER_CONNECT_CASE(MER_RIF_TRILINEAR, 0) MER_CONNECT_CASE(MER_RIF_BSPLINE3, 0) MER_CONNECT_CASE(RIFK_DENSE_BUF, 0) MER_CONNECT_CASE(RIFK_CELL8, 0) MER_CONNECT_CASE(RIFK_CELL8_BUF, 0)
    MER_CONNECT_CASE(RIFK_BRICK27_BUF, 0) MER_CONNECT_CASE(RIFK_BRICK27, 0)
    // signed-distance boundary: the fetch kinds mer_render instantiates for it
    MER_CONNECT_CASE(MER_RIF_TRILINEAR, 1) MER_CONNECT_CASE(RIFK_CELL8_BUF, 1) MER_CONNECT_CASE(MER_RIF_BSPLINE3, 1)
#undef MER_CONNECT_CASE
    if (!launched) {
        // a dense RIF below 4 GiB selects buffer loads; the signed-distance kernels read it with global loads
        if (scene->boundary == MER_BOUNDARY_SDF && rifk == RIFK_DENSE_BUF) {
            hipLaunchKernelGGL((connect_kernel<MER_RIF_TRILINEAR, 1>), dim3(nblocks(n, 64)), dim3(64), 0, ctx->stream, P, a.as<float>(), b.as<float>(), n, r.as<float>());
        } else return fail(ctx, "mer_connect: unsupported RIF layout for this boundary");
    }
    HIP_CHECK(ctx, hipGetLastError());
    return r.download(out, n * 48);
}

int mer_sample_distance(mer_context *ctx, const mer_scene_desc *scene, const float *o, const float *d, const float *maxt, int64_t n,
                        uint64_t seed, float *rec) {
    MER_USE_DEVICE(ctx);
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
    MER_USE_DEVICE(ctx);
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
    MER_USE_DEVICE(ctx);
    if (phase == MER_PHASE_HG && (g >= 1 || g <= -1)) return fail(ctx, "The asymmetry parameter must lie in the interval (-1, 1)!");
    DevBuf a(ctx), b(ctx), c(ctx), e(ctx);
    if (a.upload(wi, n * 12) || b.upload(u2, n * 8) || c.alloc(n * 12) || e.alloc(n * 4)) return 1;
    hipLaunchKernelGGL(phase_sample_kernel, dim3(nblocks(n)), dim3(256), 0, ctx->stream, phase, g, a.as<float>(), b.as<float>(), n, c.as<float>(), e.as<float>());
    HIP_CHECK(ctx, hipGetLastError());
    if (c.download(wo, n * 12)) return 1;
    return e.download(pdf, n * 4);
}
int mer_phase_eval(mer_context *ctx, int32_t phase, float g, const float *wi, const float *wo, int64_t n, float *val) {
    MER_USE_DEVICE(ctx);
    DevBuf a(ctx), b(ctx), c(ctx);
    if (a.upload(wi, n * 12) || b.upload(wo, n * 12) || c.alloc(n * 4)) return 1;
    hipLaunchKernelGGL(phase_eval_kernel, dim3(nblocks(n)), dim3(256), 0, ctx->stream, phase, g, a.as<float>(), b.as<float>(), n, c.as<float>());
    HIP_CHECK(ctx, hipGetLastError());
    return c.download(val, n * 4);
}
int mer_camera_rays(mer_context *ctx, const mer_scene_desc *scene, const float *pos2, int64_t n, float *o, float *d) {
    MER_USE_DEVICE(ctx);
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
    MER_USE_DEVICE(ctx);
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
    MER_USE_DEVICE(ctx);
    DevBuf a(ctx);
    if (a.alloc((size_t) n * 4)) return 1;
    hipLaunchKernelGGL(rng_kernel, dim3(1), dim3(64), 0, ctx->stream, seed, pixel, sample, n, a.as<float>());
    HIP_CHECK(ctx, hipGetLastError());
    return a.download(out, (size_t) n * 4);
}
int mer_synth_field_dev(mer_context *ctx, int32_t kind, int32_t N, float **data_dev) {
    MER_USE_DEVICE(ctx);
    if (kind < 0 || kind > 2 || N < 2) return fail(ctx, "mer_synth_field_dev: bad arguments");
    HIP_CHECK(ctx, hipMalloc((void **) data_dev, (size_t) N * N * N * 4));
    hipLaunchKernelGGL(synth_field_kernel, dim3(8192), dim3(256), 0, ctx->stream, kind, N, *data_dev);
    HIP_CHECK(ctx, hipGetLastError());
    HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

}  // extern "C"
