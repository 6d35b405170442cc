/*
 * mer.h -- C-ABI of libmer.so: the MI355X-native refractive volumetric path-tracing hot path.
 *
 * This is the drop-in boundary.  The reference (cmu-ci-lab/MitsubaER) has no
 * C-ABI for this path: its plugins are C++ classes behind `extern "C" CreateInstance(const Properties&)`
 * (include/mitsuba/core/cobject.h:99-107).  Each entry point below names the reference interface it
 * replaces.  Signatures use plain pointers, sizes and POD structs only -- no C++ / torch types.
 *
 * Conventions: every call returns 0 on success, non-zero on failure (reference: Log(EError) throws
 * std::runtime_error, src/libcore/logger.cpp:100-147; the message is kept for mer_last_error()).
 * The caller owns host buffers; the library owns device buffers behind handles.  One context per
 * (process, GPU); a context is single-threaded.  Grids are dense, x fastest:
 * data[((z*yres+y)*xres+x)*channels+c] (VOL v3 payload order, src/volume/gridvolume.cpp:54-89).
 * Pointers named *_dev are DEVICE pointers (e.g. a torch tensor's data_ptr()); all others are host.
 */
#ifndef MER_H
#define MER_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MER_ABI_VERSION 3

typedef struct mer_context mer_context;
typedef int32_t mer_volume;            /* handle, > 0; 0 = none */

/* VOL v3 type codes (src/volume/gridvolume.cpp:54-89) */
enum { MER_VOL_F32 = 1, MER_VOL_U8 = 3 };
enum { MER_SIGMA_HOMOGENEOUS = 0, MER_SIGMA_GRID = 1 };       /* medium `homogeneous` | `heterogeneous` */
enum { MER_RIF_CONST = 0, MER_RIF_TRILINEAR = 1, MER_RIF_BSPLINE3 = 2,   /* none | gridvolume | splinevolume */
       MER_RIF_ACOUSTIC = 8 };   /* acousticrifvolume: analytic, no grid (values 3..7 are internal fetch kinds of the trilinear RIF) */
enum { MER_STEP_VERLET = 0, MER_STEP_RK4 = 1 };
enum { MER_BOUNDARY_AABB = 0, MER_BOUNDARY_SPHERE = 1, MER_BOUNDARY_SDF = 2 };
enum { MER_PHASE_ISOTROPIC = 0, MER_PHASE_HG = 1 };
enum { MER_TR_WOODCOCK2 = 0, MER_TR_RATIO = 1 };
enum { MER_STRATEGY_BALANCE = 0, MER_STRATEGY_SINGLE = 1, MER_STRATEGY_MANUAL = 2, MER_STRATEGY_MAXIMUM = 3 /* MaxExpDist, src/medium/maxexp.h */ };
enum { MER_FILTER_BOX = 0, MER_FILTER_GAUSSIAN = 1 };
enum { MER_ALBEDO_CONST = 0, MER_ALBEDO_GRID = 1 };
/* device memory layout of an uploaded grid (the integer index contract stays (x,y,z)) */
/* device layouts of a 1-channel float32 grid (the index contract stays (x,y,z)): DENSE = the VOL payload; CELL8 = the 8 corners of every
   cell in one 32-byte record; BRICK27 = the 3x3x3 corners of every 2x2x2-cell brick in one 128-byte record (one cache line serves
   eight cells: half the memory requests of CELL8 along a ray).  CELL8 / BRICK27 are for the refractive-index field. */
enum { MER_LAYOUT_DENSE = 0, MER_LAYOUT_CELL8 = 1, MER_LAYOUT_BRICK27 = 2, MER_LAYOUT_BRICK125 = 3 /* 4x4x4-cell bricks: 5x5x5 corners per 512-byte record */,
       MER_LAYOUT_AUTO = 4 /* 1-channel float32 grids: BRICK27 up to 2^28 nodes (the records stay near the caches), CELL8 above (half the
                              bytes per fetch once every fetch goes to HBM; measured 256^3 .. 1024^3); other grids: DENSE */ };

/* replaces GridDataSource::loadFromFile header fields (src/volume/gridvolume.cpp:217-287) */
typedef struct {
    int32_t res[3];
    int32_t channels;               /* 1 or 3 */
    int32_t dtype;                  /* MER_VOL_F32 | MER_VOL_U8 */
    float   aabb_min[3], aabb_max[3];
    /* inverse of the plugin's `toWorld` (GridDataSource: m_worldToVolume = m_volumeToWorld.inverse(), src/volume/gridvolume.cpp:110,
       188-195), row-major 3x4; all zeros = identity.  worldToGrid = scale((res-1)/extents) * translate(-min) * world_to_volume. */
    float   world_to_volume[12];
} mer_grid_desc;

/* Flat scene: what Integrator::render() (include/mitsuba/render/integrator.h:74) sees through
   Scene/Sensor/Film/Medium/PhaseFunction/VolumeDataSource/Emitter objects. */
typedef struct {
    /* sensor `perspective` + film `hdrfilm` (src/sensors/perspective.cpp:130-158,247-269) */
    int32_t width, height;
    float   fov_x_deg, near_clip, far_clip;
    float   cam_to_world[12];       /* row-major 3x4, columns (left,newUp,dir,origin): Transform::lookAt */
    int32_t rfilter; float rfilter_param;   /* box radius | gaussian stddev (src/rfilters) */
    /* integrator `volpath` (src/librender/integrator.cpp:190-225) */
    int32_t max_depth, rr_depth, hide_emitters;
    /* shape with `interior` medium, null BSDF (src/librender/shape.cpp:48-70) */
    int32_t boundary;
    float   bmin[3], bmax[3];
    float   sph_center[3], sph_radius;
    /* medium */
    int32_t sigma_mode;
    float   sigma_a[3], sigma_s[3]; /* homogeneous sigmaA / sigmaS */
    int32_t strategy, channel; float sampling_density, medium_sampling_weight;  /* homogeneous.cpp:156-228 */
    mer_volume density; float density_scale;     /* heterogeneous `density`, `scale` */
    int32_t albedo_mode; float albedo[3]; mer_volume albedo_grid;
    int32_t rif_mode; float rif_const; mer_volume rif;   /* heterogeneousrefractive `rif` */
    int32_t stepper; float stepsize;             /* `stepsize` (heterogeneousrefractive.cpp:208) */
    /* phase `hg` / `isotropic` */
    int32_t phase; float g;
    int32_t tr_estimator;
    /* emitter `constant` radiance; medium emission per unit density (config 5) */
    float   env_radiance[3];
    float   emission[3];
    /* emitter `point` (src/emitters/point.cpp): position + radiant intensity; all-zero intensity = none.  With curved rays the
       emitter is reached by mer_connect's shooting solver; an emitter outside the medium shape through the boundary
       (Snell refraction at the shape, exterior index 1: src/medium/heterogeneousrefractive.cpp:873-919,963-992). */
    float   point_position[3], point_intensity[3];
    /* film decomposition (src/librender/film.cpp:56-84): 0 = none, 1 = transient, 2 = bounce.  Transient: every radiance
       contribution is binned by its optical path length (sum of h*n along curved segments, length*n otherwise:
       src/integrators/bdpt/bdpt_proc.cpp:151-176,449-470) into frames = ceil((max_bound-min_bound)/bin_width) RGB slices.
       The film then is float[height][width][frames*3 + 2]: RGB per frame, alpha, weight (bdpt_proc.cpp:230-245,484-485).
       calibrated_transient != 0 leaves the camera edge out of the path length (bdpt_proc.cpp:163-170).
       Bounce: the same film, binned by the number of path edges instead -- every edge counts 1.0 where transient adds its optical
       length (bdpt_proc.cpp:179-187,335-381); no modulation. */
    int32_t decomposition; float min_bound, max_bound, bin_width; int32_t calibrated_transient;
    /* path-length modulation of a transient film (continuous-wave time of flight; PathLengthSampler,
       src/librender/pathlengthsampler.cpp:12-114): MER_MODULATION_*; lambda, phase in degrees, P, neighbors.  With a
       modulation the film has one frame and every contribution is weighted by correlationFunction(pathLength)
       (src/integrators/bdpt/bdpt_proc.cpp:446-447; src/librender/film.cpp:76-78). */
    int32_t modulation; float mod_lambda, mod_phase_deg; int32_t mod_P, mod_neighbors;
    /* BSDF of the medium's boundary shape: MER_BSDF_NULL (index-matched, src/librender/shape.cpp:48-70) or MER_BSDF_HDIELECTRIC
       (src/bsdfs/hdielectric.cpp: smooth dielectric whose eta is the RIF at the hit point, exterior index 1) */
    int32_t boundary_bsdf;
    /* boundary = MER_BOUNDARY_SDF: the medium shape is the negative region of this signed-distance grid (1-channel float32 volume,
       trilinear; the reference's `sdf` child of heterogeneousrefractive, src/medium/heterogeneousrefractive.cpp:366-375, negative
       inside :481).  Camera rays find it by sphere tracing, the dielectric normal is the normalized gradient (:980-984).
       mer_render only; the leaf entry points know the cube / sphere boundaries. */
    mer_volume sdf;
    /* `aggressivetracing` of heterogeneousrefractive (src/medium/heterogeneousrefractive.cpp:230,473-493,697-704): with the
       signed-distance boundary, a trace of length s first advances, WITHOUT inside tests, by min(depth below the surface -
       sdf_max_error, distance left) as long as that depth is at least Epsilon (1e-4), each such leg being int(d/h) full steps plus
       the remainder step, and walks what is left with the ordinary tested trace.  sdf_max_error = the volume's maxSDFError(). */
    int32_t aggressive_tracing;
    float   sdf_max_error;
    /* rif_mode = MER_RIF_ACOUSTIC: the ultrasound-modulated index of `acousticrifvolume` (src/volume/acousticrifvolume.cpp:101-106,
       224-342), evaluated analytically: n = n_o + n_max J_m(k_r r) cos(m phi), r = sqrt(y^2 + z^2), phi = atan2(y, z),
       k_r = 2 pi freq / speed; gradient and Hessian as written there (r clamped at 1e-8).  No `rif` volume. */
    float   ac_n_o, ac_n_max, ac_k_r;
    int32_t ac_mode;
    /* `method` of the heterogeneous medium (src/medium/heterogeneous.cpp:195-202): MER_METHOD_WOODCOCK (default) or MER_METHOD_SIMPSON --
       composite Simpson quadrature of the density along straight rays for the transmittance (integrateDensity, :301-376) and its
       inversion for the free flight (invertDensityIntegral, :419-544); sigma_mode = GRID, rif_mode = CONST only.  het_stepsize = the
       plugin's `stepSize`; 0 = inferred as the reference does (:245-257): 0.5 x the smallest voxel extent of the density / albedo grids. */
    int32_t method; float het_stepsize;
    /* emitter `area` on a `rectangle` shape (src/emitters/area.cpp:67-187, src/shapes/rectangle.cpp:99-222): the rectangle is the image of
       [-1,1]^2 x {0} under area_to_world (row-major 3x4, no shear: "Error: 'toWorld' transformation contains shear!"); one-sided -- radiance
       area_radiance into the half space of its normal toWorld(0,0,1) -- and all-absorbing otherwise (an emitter shape without a BSDF gets a
       diffuse one of reflectance 0, src/librender/shape.cpp:48-56): it shadows the environment.  Sampled at every real collision
       (Shape::sampleDirect, src/librender/shape.cpp:102-115: area sampling converted to solid angle) with the phase-function sample as its
       MIS partner (volpath.cpp:120-173,370-428).  All-zero radiance = none.  The rectangle lies outside the medium shape.  Straight rays
       (rif_mode = MER_RIF_CONST), index-matched cube / sphere boundary. */
    float   area_to_world[12], area_radiance[3];
} mer_scene_desc;
enum { MER_METHOD_WOODCOCK = 0, MER_METHOD_SIMPSON = 1 };
enum { MER_BSDF_NULL = 0, MER_BSDF_HDIELECTRIC = 1 };
enum { MER_MODULATION_NONE = 0, MER_MODULATION_SINE, MER_MODULATION_SQUARE, MER_MODULATION_HAMILTONIAN, MER_MODULATION_MSEQ,
       MER_MODULATION_DEPTHSELECTIVE };
enum { MER_DECOMPOSITION_NONE = 0, MER_DECOMPOSITION_TRANSIENT = 1, MER_DECOMPOSITION_BOUNCE = 2 };

/* which part of the image-sample space this call renders (multi-GPU sharding):
   sample indices spp_begin + k*spp_stride, k in [0, spp_count); 32x32 image tiles t with
   t % tile_count == tile_rank. */
typedef struct {
    int32_t spp_begin, spp_count, spp_stride;
    int32_t tile_rank, tile_count;
} mer_shard;

enum {
    MER_C_PATHS = 0, MER_C_STEPS, MER_C_RIF_EVALS, MER_C_TENTATIVE, MER_C_REAL,
    MER_C_SEGMENTS, MER_C_NEE, MER_C_LOOP_ITERS, MER_C_ACTIVE_LANES,
    MER_C_CONNECT_UNITS,        /* solver units (traced rays) K_connect ran: a connection costs 3 ... 100+ */
    MER_C_CONNECT_STEPS,        /* sensitivity / Verlet steps inside them */
    MER_C_CONNECT_LANE_SLOTS,   /* 64 x the steps of the longest unit of every K_connect wave: CONNECT_STEPS / this = its active-lane fraction */
    MER_C_SIDE_SPAWNED,         /* luminaire-sample / look-up walks handed to a side-walk slot (option spawn_walks) */
    MER_C_SIDE_INLINE,          /* ... and those that ran in the path's own lane because the side-walk slot was still busy */
    MER_C_COUNT = 16
};

/* ---- context ------------------------------------------------------------------------------------ */
int  mer_abi_version(void);
/* replaces PluginManager::createObject + Scheduler worker setup (src/libcore/plugin.cpp:180-196) */
int  mer_context_create(int32_t device_id, mer_context **out);
void mer_context_destroy(mer_context *ctx);
const char *mer_last_error(mer_context *ctx);       /* ctx may be NULL: error of a failed create */
/* run kernels on this hipStream_t (NULL = default stream).  mer_render cuts its shard into a few independent pipelines: the first runs on
   this stream, the others on internal non-blocking streams that start after the work already queued on this stream (an event) and
   are joined into it before mer_render returns -- the caller sees one stream. */
int  mer_context_set_stream(mer_context *ctx, void *hip_stream);
int  mer_device_info(mer_context *ctx, char *name, int32_t name_len, int32_t *cu_count, int64_t *hbm_bytes);
/* Scheduling / A-B options of a context -- the analogue of the reference's Scheduler / RenderJob settings (block size, worker count:
   src/mitsuba/mitsuba.cpp:80-81,281).  None changes a per-path result (tested); they are NOT read from the process environment by the
   render calls (one hook: MER_OPTIONS="name=value,..." gives initial values when a context is created).  Names:
     pipes          concurrent pipelines a render is cut into (1..4, default 4)
     nslots         path-state slots over all pipelines (0 = 4 x the resident lanes of the chip)
     ksteps         eikonal steps / tentative collisions per lane per K_march launch (default 128)
     mq_sort        march lists sorted by steps-to-boundary class: -1 by field size (default), 0 off, 1 on
     lds_bricks     1 = K_march keeps every lane's current BRICK27 record in LDS (BRICK27 fields below 4 GiB) instead of re-gathering cells
                    through L1 / L2 (default 0: measured slower)
     connect_launches  K_connect launches per pass: each runs one solver unit (one traced ray) per pending curved-ray connection (default 2)
     adaptive_k     pass length in the tail of a render: 0 fixed (default), 1 longer, 2 shorter
     inline_walks   1 = straight rays in a gridded sigma_t: K_event runs the walks itself instead of handing them to K_march (default), 0 = two kernels
     spawn_walks    1 = curved rays, steady-state film: the transmittance walks of luminaire samples and emitter look-ups run in side-walk slots while
                    the path goes on to its next scattering event (default); 0 = every walk in the path's own lane
     check_every    passes per batch of launches; the host reads the finished-slot count back once per batch, two batches in flight (default 4)
     grid_fit       1 = launch grids sized by what the work lists can still hold -- live path slots x records per path + side walks in flight at
                    the last read-back -- instead of by every record (default); 0 = full grids (A/B)
     march_sort     curved rays: the march list of every pass is also counting-sorted by position (cells of a 2^b x 2^b x 2^b grid over the RIF's
                    box in Morton order, b = 1 ... 4) and swept in XCD-contiguous chunks; 0 = off (default: measured -40 % L2 misses, no gain in time)
     march_sort_major  bin order of that sort: 0 = cell-major (default), 1 = exit-time-class-major
     tile_deal      1 = image-tile shards dealt on diagonals of the tile grid (default), 0 = plain row-major round robin (whole tile columns)
     small_render_slots  1 = a render with fewer than ~8 paths per slot runs on a quarter of the slots (default)
     pass_events    per-pass HIP events feeding mer_last_render_stats (default 1)
     buffer_loads   0 = read fields with global loads even below 4 GiB, i.e. run the kernels a >= 4 GiB field selects (default 1)
     gen_all        K_gen hands every camera sample to K_event (A/B, default 0)
     prefilter      K_prefilter form: 0 register windows (default), 1 one thread per line, 2 two kernels per axis, 3 strided x, 4 LDS x
     march_lds_kb   KiB of unused dynamic LDS requested per K_march block: an occupancy cap for A/B runs (33 -> 4 blocks per CU, 41 -> 3; default 0)
     verbose, debug_pixel */
int  mer_context_set_option(mer_context *ctx, const char *name, int64_t value);
int  mer_context_get_option(mer_context *ctx, const char *name, int64_t *value);
/* libmer_check.so (the same sources built with -DMER_BOUNDS_CHECK): every index a kernel forms into a device buffer is compared with
   the buffer's extent; out = {violations since the last call, kind, index, limit of the first}; *enabled = 0 in the product build
   (which compiles the checks away and always reports zeros).  No reference analogue (the reference relies on host sanitizers). */
int  mer_debug_bounds(mer_context *ctx, int32_t *enabled, uint64_t out[4]);

/* ---- volumes: replaces GridDataSource / SplineDataSource construction
        (src/volume/gridvolume.cpp:108-198, src/volume/splinevolume.cpp:204-317) ------------------------- */
int  mer_volume_upload(mer_context *ctx, const mer_grid_desc *desc, const void *host_data,
                       int32_t layout, mer_volume *out);
/* same, from a device-resident dense grid (generated on the GPU) */
int  mer_volume_upload_dev(mer_context *ctx, const mer_grid_desc *desc, const void *data_dev,
                           int32_t layout, mer_volume *out);
/* builds cubic-B-spline coefficients on the GPU (Spline<3>::build3d, include/mitsuba/core/basisspline.h:812-890) */
int  mer_volume_build_spline(mer_context *ctx, mer_volume v);
int  mer_volume_download_spline(mer_context *ctx, mer_volume v, float *coeff_host);
int  mer_volume_destroy(mer_context *ctx, mer_volume v);

/* ---- film (ImageBlock, include/mitsuba/render/imageblock.h:124-205): float[height][width][channels], channels = 5
        (R,G,B,alpha,weight) in steady state; mer_film_channels() for a scene with a transient decomposition; the *_n
        variants take the channel count ------ */
int  mer_film_channels(mer_context *ctx, const mer_scene_desc *scene, int32_t *channels);
int  mer_film_alloc_n(mer_context *ctx, int32_t width, int32_t height, int32_t channels, float **film_dev);
int  mer_film_zero_n(mer_context *ctx, float *film_dev, int32_t width, int32_t height, int32_t channels);
int  mer_film_download_n(mer_context *ctx, const float *film_dev, int32_t width, int32_t height, int32_t channels, float *film_host);
int  mer_film_alloc(mer_context *ctx, int32_t width, int32_t height, float **film_dev);
int  mer_film_zero(mer_context *ctx, float *film_dev, int32_t width, int32_t height);
int  mer_film_download(mer_context *ctx, const float *film_dev, int32_t width, int32_t height, float *film_host);
int  mer_film_free(mer_context *ctx, float *film_dev);

/* ---- the hot path: replaces SamplingIntegrator::render -> renderBlock -> Li -> ImageBlock::put
        (src/librender/integrator.cpp:95-188, src/integrators/path/volpath.cpp:84-343).
        Accumulates (R,G,B,alpha,weight) splats into film_dev.  Launches on the context stream; returns when the
        last wavefront pass has been issued and found no live path (it synchronises the stream internally). ---- */
int  mer_render(mer_context *ctx, const mer_scene_desc *scene, const mer_shard *shard,
                uint64_t seed, float *film_dev);
int  mer_synchronize(mer_context *ctx);
/* HIP-event time of the last mer_render kernel in ms (synchronizes) */
int  mer_last_kernel_ms(mer_context *ctx, float *ms);
/* wavefront passes of the last mer_render and the summed HIP-event device time of its K_march / K_event launches (summed over the
   concurrent pipelines: the sums can exceed the wall time) */
int  mer_last_render_stats(mer_context *ctx, int32_t *passes, float *march_ms, float *event_ms);
int  mer_counters_read(mer_context *ctx, uint64_t out[MER_C_COUNT]);    /* StatsCounter analogue */
int  mer_counters_reset(mer_context *ctx);

/* ---- leaf entry points for parity tests (host pointers, batched SoA) ------------------------------ */
/* GridDataSource::lookupFloat (gridvolume.cpp:337-388); out_idx[4*i..] = x1,y1,z1,linear index or -1 */
int  mer_lookup_trilinear(mer_context *ctx, mer_volume v, const float *pts, int64_t n, float *out_val, int32_t *out_idx);
/* GridDataSource::lookupSpectrum (gridvolume.cpp:390-421) */
int  mer_lookup_trilinear_rgb(mer_context *ctx, mer_volume v, const float *pts, int64_t n, float *out_rgb);
/* VolumeDataSource::valueAndGradient (splinevolume.cpp:354-360) for rif_interp in {TRILINEAR,BSPLINE3} */
int  mer_rif_value_grad(mer_context *ctx, mer_volume v, int32_t rif_interp, const float *pts, int64_t n,
                        float *out_val, float *out_grad);
/* HeterogeneousRefractiveMedium::trace / traceTillBoundary (heterogeneousrefractive.cpp:671-691,742-776);
   dist[i] = +inf selects traceTillBoundary */
int  mer_er_trace(mer_context *ctx, const mer_scene_desc *scene, const float *p0, const float *d0, const float *dist,
                  int64_t n, float *out_p, float *out_v, float *out_dist_surf, float *out_opt, int32_t *out_success);
/* Medium::sampleDistance (heterogeneous.cpp:589-663, homogeneous.cpp:275-352, heterogeneousrefractive.cpp:402-568);
   rec stride 20: success,t,p[3],sigmaS[3],transmittance[3],pdfSuccess,pdfFailure,refRatioSq,d[3],0,0,0;
   RNG stream of item i = (seed, pixel=i, sample=0) */
int  mer_sample_distance(mer_context *ctx, const mer_scene_desc *scene, const float *o, const float *d,
                         const float *maxt, int64_t n, uint64_t seed, float *rec);
/* Medium::evalTransmittance (heterogeneous.cpp:546-587 / ratio tracking) */
int  mer_eval_transmittance(mer_context *ctx, const mer_scene_desc *scene, const float *o, const float *d,
                            const float *maxt, int64_t n, uint64_t seed, float *out_tr);
/* HeterogeneousRefractiveMedium::eval -> makeDirectConnections (heterogeneousrefractive.cpp:571-640,798-1163): connect p1 to p2
   (p1 inside the medium shape -- cube, sphere or signed-distance grid --, p2 inside or outside it) by a curved ray; out stride 12: ok, weight, dirToP2[3] (optical momentum at p1),
   revDirToP1[3], distance, opticalLength, 0, 0; RNG stream of item i = (seed, pixel=i, sample=0) */
int  mer_connect(mer_context *ctx, const mer_scene_desc *scene, const float *p1, const float *p2, int64_t n, uint64_t seed, float *out);
/* PhaseFunction::sample / eval (src/phase/hg.cpp:74-110, src/phase/isotropic.cpp:62-78) */
int  mer_phase_sample(mer_context *ctx, int32_t phase, float g, const float *wi, const float *u2, int64_t n, float *wo, float *pdf);
int  mer_phase_eval(mer_context *ctx, int32_t phase, float g, const float *wi, const float *wo, int64_t n, float *val);
/* PerspectiveCamera::sampleRay (src/sensors/perspective.cpp:247-269) */
int  mer_camera_rays(mer_context *ctx, const mer_scene_desc *scene, const float *pos2, int64_t n, float *o, float *d);
/* PathLengthSampler::correlationFunction for the scene's modulation (src/librender/pathlengthsampler.cpp:68-114) */
int  mer_correlation(mer_context *ctx, const mer_scene_desc *scene, const float *path_length, int64_t n, float *out);
/* per-path radiance Li of sample `sample_index` for every pixel: out[(y*w+x)*3] (no filter) */
int  mer_render_paths(mer_context *ctx, const mer_scene_desc *scene, int32_t sample_index, uint64_t seed, float *out_rgb);
/* sampler stream known answers */
int  mer_rng_floats(mer_context *ctx, uint64_t seed, uint32_t pixel, uint32_t sample, int32_t n, float *out);
/* synthetic fields of BASELINE.json's configs generated on the device:
   kind 0 = sigma_t density, 1 = linear RIF (y), 2 = radial RIF; returns a device pointer the caller frees
   with mer_device_free */
int  mer_synth_field_dev(mer_context *ctx, int32_t kind, int32_t N, float **data_dev);
int  mer_device_free(mer_context *ctx, void *ptr_dev);

/* ---- several GPUs in one process: replaces the reference's N local workers whose image blocks are merged by
        film->put under a mutex (src/librender/renderproc.cpp:142-149; worker count: src/mitsuba/mitsuba.cpp:281).
        A mer_multi owns one context per entry of device_ids and one host thread per context for the duration of a render.  Volumes are
        replicated (every device holds every grid); the sample space of a render is cut into one mer_shard per context
        (MER_SHARD_SAMPLES: sample s goes to context s mod n -- perfect balance, result independent of n up to summation order;
        MER_SHARD_TILES: the 32x32 image tiles, dealt on diagonals of the tile grid -- the reference's block partition); the films are then
        sum-reduced onto the first device: with RCCL (ncclReduce inside ncclGroupStart/End over a communicator made by ncclCommInitAll,
        librccl.so loaded at run time) when all devices are distinct, by peer copy + an add kernel otherwise (a device listed twice,
        e.g. {0, 0}, runs the whole code path on a one-GPU machine; RCCL refuses duplicate devices).  No other exchange: paths are
        independent. ---- */
typedef struct mer_multi mer_multi;
enum { MER_SHARD_SAMPLES = 0, MER_SHARD_TILES = 1 };
enum { MER_REDUCE_NONE = 0 /* one context */, MER_REDUCE_RCCL = 1, MER_REDUCE_PEER_COPY = 2 };
int  mer_multi_create(const int32_t *device_ids, int32_t n, mer_multi **out);
void mer_multi_destroy(mer_multi *m);
const char *mer_multi_last_error(mer_multi *m);          /* m may be NULL: error of a failed create */
int32_t mer_multi_size(mer_multi *m);
mer_context *mer_multi_context(mer_multi *m, int32_t i); /* context i (options, leaf calls); owned by m */
/* mer_context_set_option on every context */
int  mer_multi_set_option(mer_multi *m, const char *name, int64_t value);
/* mer_volume_upload / mer_volume_build_spline / mer_volume_destroy on every context; ONE handle, valid in all of them */
int  mer_multi_volume_upload(mer_multi *m, const mer_grid_desc *desc, const void *host_data, int32_t layout, mer_volume *out);
int  mer_multi_volume_build_spline(mer_multi *m, mer_volume v);
int  mer_multi_volume_destroy(mer_multi *m, mer_volume v);
/* renders sample indices spp_begin .. spp_begin + spp_count - 1 of every pixel, sharded over the contexts, and returns the reduced film
   float[height][width][mer_film_channels] in film_host.  rccl: 1 = use RCCL when the devices allow it and the library initialises (default choice; peer copy + add otherwise --
   mer_multi_last_stats reports which), 0 = always peer copy + add, 2 = RCCL even for a single context (a one-rank communicator: exercises the library binding on one GPU). */
int  mer_multi_render(mer_multi *m, const mer_scene_desc *scene, int32_t shard_mode, int32_t spp_begin, int32_t spp_count, uint64_t seed,
                      int32_t rccl, float *film_host);
/* of the last mer_multi_render: how the films were reduced (MER_REDUCE_*), wall milliseconds of every context's render (n floats, may be
   NULL), of the reduction, and the counters summed over the contexts (may be NULL) */
int  mer_multi_last_stats(mer_multi *m, int32_t *reduce_path, float *render_ms, float *reduce_ms, uint64_t counters[MER_C_COUNT]);

#ifdef __cplusplus
}
#endif
#endif /* MER_H */
