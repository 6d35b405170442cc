/*
 * plugins/volpath_hip.cpp -- the Mitsuba-side plugin a maintainer of cmu-ci-lab/MitsubaER adds to put libmer.so behind
 * `<integrator type="volpath_hip"/>`.
 *
 * NOT compiled in this repository: it needs the reference's own tree (Mitsuba headers, Boost, Xerces-C, SCons), none of which is
 * in this image.  It is written against the reference's headers as they are
 * (include/mitsuba/render/{scene,integrator,sensor,film,medium,phase,shape,emitter,imageblock,renderqueue}.h) and is built inside
 * the reference's tree with one SConscript line in src/integrators/SConscript:
 *
 *     plugins += env.SharedLibrary('volpath_hip', ['path/volpath_hip.cpp'], CPPPATH=env['CPPPATH']+['<this repo>/include'],
 *                                  LIBPATH=env['LIBPATH']+['<this repo>/mitsubaer_amd'], LIBS=env['LIBS']+['mer'])
 *
 * What it does: an Integrator that overrides render() owns its parallelism (include/mitsuba/render/integrator.h:74; bdpt, ptracer
 * and the photon mappers do the same).  render() flattens the scene Mitsuba has already loaded into the C-ABI's POD descs
 * (include/mer.h), uploads the volumes, calls mer_render, and hands the accumulated (R,G,B,alpha,weight) image to the film as ONE
 * ImageBlock (Film::put, src/librender/renderproc.cpp:142-149).  Everything else -- scene loading, plugin manager, film development,
 * GUI preview, network rendering of OTHER integrators -- stays Mitsuba's.
 *
 * What Mitsuba's classes keep private, and how the shim gets it: GridDataSource / SplineDataSource keep their file name and payload
 * in protected members (src/volume/gridvolume.cpp:608-623), HeterogeneousMedium its `density` / `albedo` children
 * (src/medium/heterogeneous.cpp:757-775), HeterogeneousRefractiveMedium its `rif` / `sdf` children and step size
 * (src/medium/heterogeneousrefractive.cpp:1213-1240).  The plugin therefore reads the few things it cannot ask the objects for
 * from ITS OWN properties, with the parameter names of the plugins they belong to:
 *
 *   <integrator type="volpath_hip">
 *     <integer name="maxDepth" value="-1"/> <integer name="rrDepth" value="5"/>      <!-- MonteCarloIntegrator, as volpath -->
 *     <string name="density" value="density.vol"/> <float name="scale" value="4"/>   <!-- heterogeneous: density file, scale -->
 *     <spectrum name="albedo" value="0.9"/>                                          <!-- constvolume albedo -->
 *     <string name="rif" value="rif.vol"/> <string name="rifType" value="gridvolume|splinevolume"/>
 *     <string name="sdf" value="sdf.vol"/> <float name="stepsize" value="1e-3"/>     <!-- heterogeneousrefractive -->
 *     <string name="stepper" value="verlet|rk4"/> <string name="transmittance" value="woodcock|ratio"/>
     <string name="method" value="woodcock|simpson"/> <float name="stepSize" value="0"/>   <!-- heterogeneous -->
 *     <transform name="toWorld"> ... </transform>                                    <!-- the volumes' toWorld (gridvolume) -->
 *     <string name="albedoFile" value="albedo.vol"/>                                 <!-- heterogeneous: gridded (RGB) albedo, heterogeneous.cpp:262-281 -->
 *     <spectrum name="emission" value="0"/>                                          <!-- emission per unit density (configs[4]) -->
 *     <string name="strategy" value="balance|single|manual|maximum"/> <float name="samplingDensity" value=".."/>
 *     <float name="mediumSamplingWeight" value="-1"/>                                <!-- homogeneous.cpp:156-228 -->
 *     <boolean name="aggressivetracing" value="false"/>                              <!-- heterogeneousrefractive.cpp:230 -->
 *     <float name="lambda" .../> <float name="phase" .../> <integer name="P" .../> <integer name="neighbors" .../>   <!-- PathLengthSampler, pathlengthsampler.cpp:12-40 -->
 *     <integer name="device" value="0"/> <string name="devices" value="0,1,2,3"/> <string name="shard" value="samples|tiles"/>
 *     <string name="rifLayout" value="auto|dense|cell8|brick27"/>
 *   </integrator>
 *
 * sigmaA / sigmaS, the phase function, the shape, the sensor, the reconstruction filter, the sampler's sample count, the emitters and
 * the film -- its size, its decomposition (Film::getDecompositionType / MinBound / MaxBound / BinWidth / getFrames / isCalibratedTransient,
 * include/mitsuba/render/film.h:84-93) and the modulation TYPE of its PathLengthSampler (getModulationType; the sampler keeps lambda,
 * phase, P and neighbors private, include/mitsuba/render/pathlengthsampler.h) -- are read from the objects themselves.
 *
 * Several GPUs: `devices` lists the GPUs of this machine to render on (mer_multi_*: one context and one host thread per GPU, volumes
 * replicated, samples or image tiles sharded, films reduced with RCCL) -- what `mitsuba -p N` does with CPU workers
 * (src/mitsuba/mitsuba.cpp:281, src/librender/renderproc.cpp:142-149).
 *
 * UNVERIFIED: this file has never been compiled (no Mitsuba tree in this image).  tests/test_plugin_shim.py keeps it from drifting
 * against include/mer.h (every desc field it assigns and every mer_* function it calls must exist there), nothing more.
 */
#include <mitsuba/render/scene.h>
#include <mitsuba/render/integrator.h>
#include <mitsuba/render/renderqueue.h>
#include <mitsuba/render/renderjob.h>
#include <mitsuba/render/imageblock.h>
#include <mitsuba/render/medium.h>
#include <mitsuba/render/phase.h>
#include <mitsuba/render/emitter.h>
#include <mitsuba/render/film.h>
#include <mitsuba/render/pathlengthsampler.h>
#include <mitsuba/core/bitmap.h>
#include <mitsuba/core/fresolver.h>
#include <mitsuba/core/plugin.h>
#include <fstream>
#include <vector>
#include <cstring>
#include <cstdlib>
#include <cmath>
#include "mer.h"

MTS_NAMESPACE_BEGIN

class HIPVolPathIntegrator : public MonteCarloIntegrator {          /* reads maxDepth / rrDepth / hideEmitters (src/librender/integrator.cpp:190-225) */
public:
    HIPVolPathIntegrator(const Properties &props) : MonteCarloIntegrator(props) {
        m_device = props.getInteger("device", 0);
        m_densityFile = props.getString("density", "");
        m_rifFile = props.getString("rif", "");
        m_sdfFile = props.getString("sdf", "");
        m_rifSpline = props.getString("rifType", "splinevolume") == "splinevolume";     /* the reference's RIF volume type */
        m_scale = props.getFloat("scale", 1.0f);
        m_albedo = props.getSpectrum("albedo", Spectrum(0.9f));
        m_stepsize = props.getFloat("stepsize", 1e-3f);                                 /* heterogeneousrefractive.cpp:208 */
        m_rk4 = props.getString("stepper", "verlet") == "rk4";
        m_ratio = props.getString("transmittance", "woodcock") == "ratio";
        m_simpson = props.getString("method", "woodcock") == "simpson";                 /* heterogeneous.cpp:195-202 */
        m_hetStepSize = props.getFloat("stepSize", 0);
        const std::string l = props.getString("rifLayout", "auto");
        m_layout = l == "auto" ? MER_LAYOUT_AUTO : l == "brick27" ? MER_LAYOUT_BRICK27 : l == "cell8" ? MER_LAYOUT_CELL8 : MER_LAYOUT_DENSE;
        m_seed = (uint64_t) props.getSize("seed", 0);
        m_volumeToWorld = props.getTransform("toWorld", Transform());                   /* gridvolume.cpp:110: the volumes' common toWorld */
        m_albedoFile = props.getString("albedoFile", "");                               /* heterogeneous.cpp:262-281: `albedo` volume child */
        m_emission = props.getSpectrum("emission", Spectrum(0.0f));
        const std::string st = props.getString("strategy", "balance");                  /* homogeneous.cpp:183-228 */
        if (st == "balance") m_strategy = MER_STRATEGY_BALANCE;
        else if (st == "single") m_strategy = MER_STRATEGY_SINGLE;
        else if (st == "maximum") m_strategy = MER_STRATEGY_MAXIMUM;
        else if (st == "manual") m_strategy = MER_STRATEGY_MANUAL;
        else Log(EError, "Specified an unknown sampling strategy");                     /* homogeneous.cpp:224-226 */
        m_samplingDensity = m_strategy == MER_STRATEGY_MANUAL ? props.getFloat("samplingDensity") : 0.0f;
        m_mediumSamplingWeight = props.getFloat("mediumSamplingWeight", -1);
        m_aggressive = props.getBoolean("aggressivetracing", false);
        m_lambda = props.getFloat("lambda", 1.0f); m_phase = props.getFloat("phase", 0.0f);   /* pathlengthsampler.cpp:12-40 */
        m_P = props.getInteger("P", 32); m_neighbors = props.getInteger("neighbors", 3);
        /* GPUs: `devices` = "0,1,..." (a GPU may be listed twice), else the single `device` */
        const std::string dl = props.getString("devices", "");
        for (size_t p = 0; p <= dl.size() && !dl.empty();) {
            size_t e = dl.find(',', p); if (e == std::string::npos) e = dl.size();
            if (e > p) m_devices.push_back(atoi(dl.substr(p, e - p).c_str()));
            p = e + 1;
        }
        if (m_devices.empty()) m_devices.push_back(m_device);
        m_shardTiles = props.getString("shard", "samples") == "tiles";
    }

    /* network rendering / serialisation of the plugin itself (src/librender/integrator.cpp:227-238) */
    HIPVolPathIntegrator(Stream *stream, InstanceManager *manager) : MonteCarloIntegrator(stream, manager) {
        m_device = stream->readInt(); m_layout = stream->readInt();
        m_densityFile = stream->readString(); m_rifFile = stream->readString(); m_sdfFile = stream->readString();
        m_rifSpline = stream->readBool(); m_scale = stream->readFloat(); m_albedo = Spectrum(stream);
        m_stepsize = stream->readFloat(); m_rk4 = stream->readBool(); m_ratio = stream->readBool(); m_seed = stream->readSize();
        m_volumeToWorld = Transform(stream); m_simpson = stream->readBool(); m_hetStepSize = stream->readFloat();
        m_albedoFile = stream->readString(); m_emission = Spectrum(stream); m_strategy = stream->readInt();
        m_samplingDensity = stream->readFloat(); m_mediumSamplingWeight = stream->readFloat(); m_aggressive = stream->readBool();
        m_lambda = stream->readFloat(); m_phase = stream->readFloat(); m_P = stream->readInt(); m_neighbors = stream->readInt();
        m_devices.resize(stream->readSize()); for (size_t i = 0; i < m_devices.size(); ++i) m_devices[i] = stream->readInt();
        m_shardTiles = stream->readBool();
    }
    void serialize(Stream *stream, InstanceManager *manager) const {
        MonteCarloIntegrator::serialize(stream, manager);
        stream->writeInt(m_device); stream->writeInt(m_layout);
        stream->writeString(m_densityFile); stream->writeString(m_rifFile); stream->writeString(m_sdfFile);
        stream->writeBool(m_rifSpline); stream->writeFloat(m_scale); m_albedo.serialize(stream);
        stream->writeFloat(m_stepsize); stream->writeBool(m_rk4); stream->writeBool(m_ratio); stream->writeSize((size_t) m_seed);
        m_volumeToWorld.serialize(stream); stream->writeBool(m_simpson); stream->writeFloat(m_hetStepSize);
        stream->writeString(m_albedoFile); m_emission.serialize(stream); stream->writeInt(m_strategy);
        stream->writeFloat(m_samplingDensity); stream->writeFloat(m_mediumSamplingWeight); stream->writeBool(m_aggressive);
        stream->writeFloat(m_lambda); stream->writeFloat(m_phase); stream->writeInt(m_P); stream->writeInt(m_neighbors);
        stream->writeSize(m_devices.size()); for (size_t i = 0; i < m_devices.size(); ++i) stream->writeInt(m_devices[i]);
        stream->writeBool(m_shardTiles);
    }

    bool render(Scene *scene, RenderQueue *queue, const RenderJob *job, int sceneResID, int sensorResID, int samplerResID) {
        /* one context per listed GPU (mer_multi_*); volumes are replicated, the films reduced (RCCL between distinct devices) */
        mer_multi *ctx = NULL;
        std::vector<int32_t> ids(m_devices.begin(), m_devices.end());
        if (mer_multi_create(&ids[0], (int32_t) ids.size(), &ctx)) Log(EError, "%s", mer_multi_last_error(NULL));
        mer_scene_desc d; memset(&d, 0, sizeof(d));
        std::vector<mer_volume> volumes;

        /* ---- sensor + film (src/sensors/perspective.cpp:130-158, src/librender/film.cpp) */
        const Sensor *sensor = scene->getSensor();
        if (sensor->getClass()->getName() != "PerspectiveCamera") Log(EError, "volpath_hip: the sensor must be 'perspective'");
        const PerspectiveCamera *cam = static_cast<const PerspectiveCamera *>(sensor);
        Film *film = const_cast<Film *>(sensor->getFilm());
        d.width = film->getCropSize().x; d.height = film->getCropSize().y;
        if (film->getCropSize() != film->getSize()) Log(EError, "volpath_hip: crop windows are not supported");
        d.fov_x_deg = cam->getXFov(); d.near_clip = cam->getNearClip(); d.far_clip = cam->getFarClip();
        const Matrix4x4 &tw = cam->getWorldTransform(0).getMatrix();
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) d.cam_to_world[4 * r + c] = (float) tw(r, c);
        const ReconstructionFilter *rf = film->getReconstructionFilter();
        if (rf->getClass()->getName() == "BoxFilter") { d.rfilter = MER_FILTER_BOX; d.rfilter_param = 0.5f; }
        else if (rf->getClass()->getName() == "GaussianFilter") { d.rfilter = MER_FILTER_GAUSSIAN; d.rfilter_param = rf->getRadius() / 4; }   /* radius = 4 stddev, gaussian.cpp:42 */
        else Log(EError, "volpath_hip: reconstruction filter must be 'box' or 'gaussian'");
        d.max_depth = m_maxDepth; d.rr_depth = m_rrDepth; d.hide_emitters = m_hideEmitters ? 1 : 0;
        /* film decomposition (src/librender/film.cpp:56-84; include/mitsuba/render/film.h:84-93): transient / bounce frames, or one frame
           weighted by the PathLengthSampler's correlation function when a modulation is set (film.cpp:76-78) */
        switch (film->getDecompositionType()) {
            case Film::ESteadyState: d.decomposition = MER_DECOMPOSITION_NONE; break;
            case Film::ETransient:   d.decomposition = MER_DECOMPOSITION_TRANSIENT; break;
            case Film::EBounce:      d.decomposition = MER_DECOMPOSITION_BOUNCE; break;
            default: Log(EError, "volpath_hip: film decomposition must be none, transient or bounce");
        }
        d.min_bound = film->getDecompositionMinBound(); d.max_bound = film->getDecompositionMaxBound(); d.bin_width = film->getDecompositionBinWidth();
        d.calibrated_transient = film->isCalibratedTransient() ? 1 : 0;
        if (d.decomposition != MER_DECOMPOSITION_NONE) {
            ref<PathLengthSampler> pls = film->getPathLengthSampler();
            d.modulation = pls ? (int32_t) pls->getModulationType() : MER_MODULATION_NONE;     /* ENone .. EDepthSelective = MER_MODULATION_* (same order) */
            d.mod_lambda = m_lambda; d.mod_phase_deg = m_phase; d.mod_P = m_P; d.mod_neighbors = m_neighbors;
        }

        /* ---- the one shape that carries an interior medium */
        const Shape *shape = findMediumShape(scene);
        const AABB bb = shape->getAABB();
        if (shape->getClass()->getName() == "Sphere") {
            d.boundary = MER_BOUNDARY_SPHERE;
            const Point c = bb.getCenter();
            d.sph_center[0] = c.x; d.sph_center[1] = c.y; d.sph_center[2] = c.z; d.sph_radius = 0.5f * bb.getExtents().x;
        } else d.boundary = MER_BOUNDARY_AABB;               /* cube, or the bounding box of an obj (scenes/volumetric/bounds.obj) */
        for (int i = 0; i < 3; ++i) { d.bmin[i] = bb.min[i]; d.bmax[i] = bb.max[i]; }
        const BSDF *bsdf = shape->getBSDF();
        const std::string bsdfName = bsdf ? bsdf->getClass()->getName() : "Null";
        if (bsdfName == "HDielectric") d.boundary_bsdf = MER_BSDF_HDIELECTRIC;
        else if (bsdfName == "Null") d.boundary_bsdf = MER_BSDF_NULL;               /* no BSDF => `null` (src/librender/shape.cpp:48-70) */
        else Log(EError, "volpath_hip: the medium shape's BSDF must be null or hdielectric");

        /* ---- medium, phase function, volumes */
        fillMedium(ctx, shape->getInteriorMedium(), d, volumes);

        /* ---- emitters: constant environment and / or one point emitter */
        if (const Emitter *env = scene->getEnvironmentEmitter()) {
            if (env->getClass()->getName() != "ConstantBackgroundEmitter") Log(EError, "volpath_hip: the environment emitter must be 'constant'");
            const Spectrum L = env->evalEnvironment(RayDifferential(Point(0.0f), Vector(0, 0, 1), 0));
            Float r, g, b; L.toLinearRGB(r, g, b);
            d.env_radiance[0] = r; d.env_radiance[1] = g; d.env_radiance[2] = b;
        }
        const ref_vector<Emitter> &emitters = scene->getEmitters();
        for (size_t i = 0; i < emitters.size(); ++i) {
            const Emitter *e = emitters[i].get();
            if (e->isEnvironmentEmitter()) continue;
            if (e->getClass()->getName() == "AreaLight") {
                /* `area` emitter on a `rectangle` shape (src/emitters/area.cpp, src/shapes/rectangle.cpp): the rectangle keeps its objectToWorld private,
                   so the transform is recovered from three corner samples (Rectangle::samplePosition, :210-216: p = toWorld(2u - 1, 2v - 1, 0)) */
                const Shape *rs = e->getShape();
                if (!rs || rs->getClass()->getName() != "Rectangle") Log(EError, "volpath_hip: an area emitter must sit on a 'rectangle' shape");
                PositionSamplingRecord p00(0.0f), p10(0.0f), p01(0.0f);
                rs->samplePosition(p00, Point2(0, 0)); rs->samplePosition(p10, Point2(1, 0)); rs->samplePosition(p01, Point2(0, 1));
                const Vector du = (p10.p - p00.p) * 0.5f, dv = (p01.p - p00.p) * 0.5f;
                const Point c = p00.p + du + dv;
                const Normal n = p00.n;                                                          /* the frame normal, toWorld(Normal(0,0,1)) normalized */
                const float cols[3][4] = { { du.x, dv.x, n.x, c.x }, { du.y, dv.y, n.y, c.y }, { du.z, dv.z, n.z, c.z } };
                for (int r = 0; r < 3; ++r) for (int k = 0; k < 4; ++k) d.area_to_world[4 * r + k] = cols[r][k];
                PositionSamplingRecord pr(0.0f); rs->samplePosition(pr, Point2(0.5f));
                const Spectrum Le = e->evalPosition(pr) * INV_PI;                                 /* AreaLight::evalPosition = radiance * pi (area.cpp:98-100) */
                Float r, g, b; Le.toLinearRGB(r, g, b);
                d.area_radiance[0] = r; d.area_radiance[1] = g; d.area_radiance[2] = b;
                continue;
            }
            if (e->getClass()->getName() != "PointEmitter") Log(EError, "volpath_hip: emitters must be 'constant', 'point' or 'area' (on a rectangle)");
            PositionSamplingRecord pRec(0.0f);
            const Spectrum I = e->samplePosition(pRec, Point2(0.5f)) / (4 * M_PI);       /* src/emitters/point.cpp:82-90 */
            Float r, g, b; I.toLinearRGB(r, g, b);
            d.point_intensity[0] = r; d.point_intensity[1] = g; d.point_intensity[2] = b;
            d.point_position[0] = pRec.p.x; d.point_position[1] = pRec.p.y; d.point_position[2] = pRec.p.z;
        }

        /* ---- render on every listed GPU, then hand the reduced image to the film as one block */
        int32_t channels = 5;
        if (mer_film_channels(mer_multi_context(ctx, 0), &d, &channels)) { const std::string msg = mer_last_error(mer_multi_context(ctx, 0)); mer_multi_destroy(ctx); Log(EError, "%s", msg.c_str()); }
        if (d.decomposition != MER_DECOMPOSITION_NONE && d.modulation == MER_MODULATION_NONE && (size_t) channels != film->getFrames() * 3 + 2)
            Log(EError, "volpath_hip: the film holds %i frames, the scene description %i", (int) film->getFrames(), (channels - 2) / 3);
        std::vector<float> host((size_t) d.width * d.height * channels);
        if (mer_multi_render(ctx, &d, m_shardTiles ? MER_SHARD_TILES : MER_SHARD_SAMPLES, 0, (int32_t) scene->getSampler()->getSampleCount(), m_seed, 1, &host[0])) fail(ctx);

        /* steady state: (R,G,B,alpha,weight); decomposed films: frames x RGB + alpha + weight, the reference's multichannel block
           (src/integrators/bdpt/bdpt_wr.cpp:51-55, bdpt_proc.cpp:230-245,484-485) */
        ref<ImageBlock> block = channels == 5
            ? new ImageBlock(Bitmap::ESpectrumAlphaWeight, film->getCropSize(), film->getReconstructionFilter())
            : new ImageBlock(Bitmap::EMultiSpectrumAlphaWeight, film->getCropSize(), film->getReconstructionFilter(), channels);
        block->setOffset(Point2i(0, 0));
        block->clear();
        copyInto(block, host, d.width, d.height, channels);
        film->put(block);                                                              /* src/librender/renderproc.cpp:142-149 */
        queue->signalWorkEnd(job, block, false);

        for (size_t i = 0; i < volumes.size(); ++i) mer_multi_volume_destroy(ctx, volumes[i]);
        mer_multi_destroy(ctx);
        return true;
    }

    /* SamplingIntegrator's per-ray interface is not used: render() is overridden */
    Spectrum Li(const RayDifferential &, RadianceQueryRecord &) const { return Spectrum(0.0f); }

    std::string toString() const {
        std::ostringstream oss;
        oss << "HIPVolPathIntegrator[device=" << m_device << ", maxDepth=" << m_maxDepth << ", rrDepth=" << m_rrDepth << "]";
        return oss.str();
    }

    MTS_DECLARE_CLASS()
private:
    void fail(mer_multi *ctx) const {                                /* Log(EError) throws std::runtime_error (src/libcore/logger.cpp:100-147) */
        const std::string msg = mer_multi_last_error(ctx);
        mer_multi_destroy(ctx);
        Log(EError, "%s", msg.c_str());
    }

    static const Shape *findMediumShape(const Scene *scene) {
        const Shape *found = NULL;
        const ref_vector<Shape> &shapes = scene->getShapes();
        for (size_t i = 0; i < shapes.size(); ++i) {
            if (!shapes[i]->getInteriorMedium()) {
                if (!shapes[i]->isEmitter()) SLog(EError, "volpath_hip: a shape without an interior medium must be a rectangle carrying an area emitter");
                continue;
            }
            if (found && shapes[i]->getInteriorMedium() != found->getInteriorMedium())
                SLog(EError, "volpath_hip: exactly one medium (on one convex shape) is supported");
            if (!found) found = shapes[i].get();
            if (shapes[i]->getExteriorMedium()) SLog(EError, "volpath_hip: exterior media are not supported");
        }
        if (!found) SLog(EError, "volpath_hip: no shape with an 'interior' medium");
        return found;
    }

    /* a VOL v3 file (src/volume/gridvolume.cpp:54-89,217-287) -> mer_volume_upload */
    mer_volume uploadVol(mer_multi *ctx, const std::string &name, int layout, bool spline, std::vector<mer_volume> &keep) const {
        const fs::path path = Thread::getThread()->getFileResolver()->resolve(name);
        std::ifstream f(path.string().c_str(), std::ios::binary);
        if (!f) Log(EError, "\"%s\": file does not exist!", path.string().c_str());
        char hdr[48]; f.read(hdr, 48);
        if (f.gcount() != 48 || hdr[0] != 'V' || hdr[1] != 'O' || hdr[2] != 'L') Log(EError, "Encountered an invalid volume data file (incorrect header identifier)");
        if (hdr[3] != 3) Log(EError, "Encountered an invalid volume data file (incorrect file version)");
        int32_t h[5]; memcpy(h, hdr + 4, 20);
        float bb[6]; memcpy(bb, hdr + 24, 24);
        mer_grid_desc g; memset(&g, 0, sizeof(g));
        g.dtype = h[0]; g.res[0] = h[1]; g.res[1] = h[2]; g.res[2] = h[3]; g.channels = h[4];
        for (int i = 0; i < 3; ++i) { g.aabb_min[i] = bb[i]; g.aabb_max[i] = bb[3 + i]; }
        const Matrix4x4 &w2v = m_volumeToWorld.getInverseMatrix();                      /* m_worldToVolume, gridvolume.cpp:188-195 */
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) g.world_to_volume[4 * r + c] = (float) w2v(r, c);
        {   /* one voxel diagonal of this grid: maxSDFError() of a signed-distance volume (src/volume/splinevolume.cpp:282) */
            double e2 = 0; for (int i = 0; i < 3; ++i) { const double v = (bb[3 + i] - bb[i]) / (g.res[i] - 1); e2 += v * v; }
            m_sdfMaxError = (float) std::sqrt(e2);
        }
        const size_t n = (size_t) g.res[0] * g.res[1] * g.res[2] * g.channels * (g.dtype == MER_VOL_F32 ? 4 : 1);
        std::vector<char> data(n);
        f.read(&data[0], (std::streamsize) n);
        if ((size_t) f.gcount() != n) Log(EError, "Volume data file \"%s\" is truncated", path.string().c_str());
        mer_volume v = 0;
        if (mer_multi_volume_upload(ctx, &g, &data[0], layout, &v)) fail(ctx);             /* every listed GPU receives the grid; one handle */
        if (spline && mer_multi_volume_build_spline(ctx, v)) fail(ctx);
        keep.push_back(v);
        return v;
    }

    void fillMedium(mer_multi *ctx, const Medium *med, mer_scene_desc &d, std::vector<mer_volume> &keep) const {
        const std::string cls = med->getClass()->getName();
        /* phase function (src/phase/hg.cpp, src/phase/isotropic.cpp) */
        const PhaseFunction *phase = med->getPhaseFunction();
        if (phase->getClass()->getName() == "HGPhaseFunction") { d.phase = MER_PHASE_HG; d.g = phase->getMeanCosine(); }
        else if (phase->getClass()->getName() == "IsotropicPhaseFunction") d.phase = MER_PHASE_ISOTROPIC;
        else Log(EError, "volpath_hip: the phase function must be 'hg' or 'isotropic'");
        /* homogeneous coefficients (src/librender/medium.cpp:26-36): public getters */
        Float r, g, b;
        med->getSigmaA().toLinearRGB(r, g, b); d.sigma_a[0] = r; d.sigma_a[1] = g; d.sigma_a[2] = b;
        med->getSigmaS().toLinearRGB(r, g, b); d.sigma_s[0] = r; d.sigma_s[1] = g; d.sigma_s[2] = b;
        /* the media keep these private (homogeneous.cpp:426-427): the integrator's own `strategy` / `samplingDensity` / `mediumSamplingWeight` */
        d.strategy = m_strategy; d.channel = -1; d.sampling_density = m_samplingDensity; d.medium_sampling_weight = m_mediumSamplingWeight;
        d.density_scale = m_scale;
        m_albedo.toLinearRGB(r, g, b); d.albedo[0] = r; d.albedo[1] = g; d.albedo[2] = b;
        d.albedo_mode = MER_ALBEDO_CONST;
        m_emission.toLinearRGB(r, g, b); d.emission[0] = r; d.emission[1] = g; d.emission[2] = b;
        d.tr_estimator = m_ratio ? MER_TR_RATIO : MER_TR_WOODCOCK2;
        d.stepper = m_rk4 ? MER_STEP_RK4 : MER_STEP_VERLET;
        d.stepsize = m_stepsize;
        d.method = m_simpson ? MER_METHOD_SIMPSON : MER_METHOD_WOODCOCK; d.het_stepsize = m_hetStepSize;
        d.rif_const = 1.0f;
        if (cls == "HomogeneousMedium") {
            d.sigma_mode = MER_SIGMA_HOMOGENEOUS; d.rif_mode = MER_RIF_CONST;
        } else if (cls == "HeterogeneousMedium") {
            if (m_densityFile.empty()) Log(EError, "No density specified!");                   /* heterogeneous.cpp:229-230 */
            d.sigma_mode = MER_SIGMA_GRID; d.rif_mode = MER_RIF_CONST;
            d.density = uploadVol(ctx, m_densityFile, MER_LAYOUT_DENSE, false, keep);
            if (!m_albedoFile.empty()) { d.albedo_mode = MER_ALBEDO_GRID; d.albedo_grid = uploadVol(ctx, m_albedoFile, MER_LAYOUT_DENSE, false, keep); }   /* heterogeneous.cpp:262-281 */
        } else if (cls == "HeterogeneousRefractiveMedium") {
            if (m_rifFile.empty()) Log(EError, "No RIF specified!");                           /* heterogeneousrefractive.cpp:368-369 */
            d.rif_mode = m_rifSpline ? MER_RIF_BSPLINE3 : MER_RIF_TRILINEAR;
            d.rif = uploadVol(ctx, m_rifFile, m_rifSpline ? MER_LAYOUT_DENSE : m_layout, m_rifSpline, keep);
            if (!m_densityFile.empty()) { d.sigma_mode = MER_SIGMA_GRID; d.density = uploadVol(ctx, m_densityFile, MER_LAYOUT_CELL8, false, keep); }
            else d.sigma_mode = MER_SIGMA_HOMOGENEOUS;
            if (!m_albedoFile.empty() && d.sigma_mode == MER_SIGMA_GRID) { d.albedo_mode = MER_ALBEDO_GRID; d.albedo_grid = uploadVol(ctx, m_albedoFile, MER_LAYOUT_DENSE, false, keep); }
            if (!m_sdfFile.empty()) {
                d.boundary = MER_BOUNDARY_SDF; d.sdf = uploadVol(ctx, m_sdfFile, MER_LAYOUT_DENSE, false, keep);
                if (m_aggressive) {                                                          /* heterogeneousrefractive.cpp:230,473-493 */
                    d.aggressive_tracing = 1;
                    d.sdf_max_error = m_sdfMaxError;                                         /* maxSDFError(): one voxel diagonal, set by uploadVol (splinevolume.cpp:282) */
                }
            } else if (m_aggressive) Log(EError, "aggressivetracing needs a signed-distance volume ('sdf')");
        } else Log(EError, "volpath_hip: medium \"%s\" is not on this path (homogeneous, heterogeneous, heterogeneousrefractive)", cls.c_str());
    }

    /* film image float[h][w][channels] -> the block's bitmap (which carries a border of getBorderSize() pixels on every side);
       channels = 5 in steady state, frames x 3 + 2 for a decomposed film */
    static void copyInto(ImageBlock *block, const std::vector<float> &host, int w, int h, int channels) {
        Bitmap *bmp = block->getBitmap();
        const int border = block->getBorderSize(), ch = bmp->getChannelCount(), bw = bmp->getWidth();
        SAssert(ch == channels && SPECTRUM_SAMPLES == 3 && bmp->getComponentFormat() == Bitmap::EFloat32);
        float *dst = bmp->getFloat32Data();
        for (int y = 0; y < h; ++y)
            memcpy(dst + ((size_t) (y + border) * bw + border) * ch, &host[(size_t) y * w * channels], (size_t) w * channels * sizeof(float));
    }

    int m_device, m_layout, m_strategy, m_P, m_neighbors;
    std::vector<int> m_devices;
    std::string m_densityFile, m_rifFile, m_sdfFile, m_albedoFile;
    bool m_rifSpline, m_rk4, m_ratio, m_simpson, m_aggressive, m_shardTiles;
    Float m_scale, m_stepsize, m_hetStepSize, m_samplingDensity, m_mediumSamplingWeight, m_lambda, m_phase;
    mutable float m_sdfMaxError;
    Spectrum m_albedo, m_emission;
    Transform m_volumeToWorld;
    uint64_t m_seed;
};

MTS_IMPLEMENT_CLASS_S(HIPVolPathIntegrator, false, MonteCarloIntegrator)
MTS_EXPORT_PLUGIN(HIPVolPathIntegrator, "HIP volumetric path tracer (MI355X, libmer.so)");
MTS_NAMESPACE_END
