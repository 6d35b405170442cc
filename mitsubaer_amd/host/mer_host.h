// mer_host.h -- C++ host side above the C-ABI (include/mer.h): a mirror of the reference's plugin interface for
// the hot path.  Same plugin type names, parameter names, defaults and error texts as cmu-ci-lab/MitsubaER
// (SURVEY.md section 9.1), none of its machinery (no Boost, Xerces, ref-counting, serialization, scheduler).
//
//   Properties            include/mitsuba/core/properties.h:46
//   ConfigurableObject    addChild(name, child) / configure()        src/librender/scenehandler.cpp:712-777
//   VolumeDataSource      gridvolume | splinevolume | constvolume    include/mitsuba/render/volume.h:32-109
//   PhaseFunction         hg | isotropic                             include/mitsuba/render/phase.h:117-241
//   Medium                homogeneous | heterogeneous | heterogeneousrefractive   include/mitsuba/render/medium.h:113-234
//   Shape                 cube | sphere | obj (bounding box), `interior` medium, null BSDF   src/librender/shape.cpp:48-70,166-190
//   Sensor / Film / ReconstructionFilter / Sampler      perspective, hdrfilm, gaussian | box, independent | ldsampler
//   Emitter               constant | point | area (on a rectangle shape)
//   Integrator            volpath -> render() flattens the scene to mer_scene_desc and calls mer_render
//   SceneHandler          scene-XML subset with $param substitution    src/librender/scenehandler.cpp, src/mitsuba/mitsuba.cpp:58,168-173
//
// Errors: like Log(EError) in the reference (src/libcore/logger.cpp:100-147) every failure throws std::runtime_error.
#pragma once
#include <cmath>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/mer.h"

namespace merhost {

struct Spectrum { float c[3]; };
struct Vec3 { float x, y, z; };

[[noreturn]] void Log_EError(const std::string &msg);      // throws std::runtime_error(msg)

// ---------------------------------------------------------------------------------------------------------------
class Properties {
public:
    enum Type { EBoolean, EInteger, EFloat, EString, ESpectrum, EPoint, ETransform };
    explicit Properties(const std::string &pluginName = "") : m_pluginName(pluginName) {}
    const std::string &getPluginName() const { return m_pluginName; }
    const std::string &getID() const { return m_id; }
    void setID(const std::string &id) { m_id = id; }
    bool hasProperty(const std::string &name) const { return m_entries.count(name) != 0; }
    Type getType(const std::string &name) const;
    void setBoolean(const std::string &n, bool v);
    void setInteger(const std::string &n, int v);
    void setFloat(const std::string &n, float v);
    void setString(const std::string &n, const std::string &v);
    void setSpectrum(const std::string &n, const Spectrum &v);
    void setPoint(const std::string &n, const Vec3 &v);
    void setTransform(const std::string &n, const float m[16]);
    bool getBoolean(const std::string &n) const;
    bool getBoolean(const std::string &n, bool def) const;
    int getInteger(const std::string &n) const;
    int getInteger(const std::string &n, int def) const;
    float getFloat(const std::string &n) const;
    float getFloat(const std::string &n, float def) const;
    std::string getString(const std::string &n) const;
    std::string getString(const std::string &n, const std::string &def) const;
    Spectrum getSpectrum(const std::string &n) const;
    Spectrum getSpectrum(const std::string &n, const Spectrum &def) const;
    Vec3 getPoint(const std::string &n) const;
    Vec3 getPoint(const std::string &n, const Vec3 &def) const;
    void getTransform(const std::string &n, float m[16]) const;      // identity when absent
    /// names never queried by the plugin: the reference warns about them (properties.cpp getUnqueried)
    std::vector<std::string> getUnqueried() const;
private:
    struct Entry { Type type; bool b = false; int i = 0; float f = 0; std::string s; Spectrum spec{}; Vec3 p{}; float m[16]; mutable bool queried = false; };
    const Entry &get(const std::string &n, Type t) const;
    std::string m_pluginName, m_id;
    std::map<std::string, Entry> m_entries;
};

// ---------------------------------------------------------------------------------------------------------------
class ConfigurableObject {
public:
    virtual ~ConfigurableObject() {}
    virtual const char *getClassName() const = 0;           // MTS_CLASS analogue: "Medium", "VolumeDataSource", ...
    virtual void addChild(const std::string &name, std::shared_ptr<ConfigurableObject> child);
    virtual void configure() {}
    virtual std::string toString() const { return getClassName(); }
};
typedef std::shared_ptr<ConfigurableObject> ObjRef;

class VolumeDataSource : public ConfigurableObject {
public:
    const char *getClassName() const override { return "VolumeDataSource"; }
    virtual bool supportsFloatLookups() const { return false; }
    virtual bool supportsSpectrumLookups() const { return false; }
    virtual bool isConstant() const { return false; }
    virtual bool isSpline() const { return false; }
    virtual bool isAcoustic() const { return false; }           // acousticrifvolume: analytic, no payload
    float ac_n_o = 1.3333f, ac_n_max = 0.0f, ac_k_r = 0.0f; int ac_mode = 0;
    float aabb_min[3] = {0, 0, 0}, aabb_max[3] = {0, 0, 0};
    float worldToVolume[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // inverse of `toWorld`, row-major 3x4; all zeros = identity (gridvolume.cpp:110,188-189)
    int res[3] = {0, 0, 0}, channels = 0, dtype = MER_VOL_F32;
    std::vector<unsigned char> data;          // dense payload (gridvolume / splinevolume)
    Spectrum constant{};                      // constvolume
    std::string filename;
    float getStepSize() const;                // gridvolume.cpp:196-198
    float getMaximumFloatValue() const { return 1.0f; }     // gridvolume.cpp:583-585
};

class PhaseFunction : public ConfigurableObject {
public:
    const char *getClassName() const override { return "PhaseFunction"; }
    int kind = MER_PHASE_ISOTROPIC; float g = 0.0f;
    float getMeanCosine() const { return kind == MER_PHASE_HG ? g : 0.0f; }
};

class Medium : public ConfigurableObject {
public:
    const char *getClassName() const override { return "Medium"; }
    void addChild(const std::string &name, ObjRef child) override;
    void configure() override;
    virtual bool isHomogeneous() const { return kind == "homogeneous"; }
    bool isheterogeneousrefractive() const { return kind == "heterogeneousrefractive"; }
    std::string kind;                          // plugin name
    Spectrum sigmaA{}, sigmaS{};               // homogeneous coefficients after `scale`
    int strategy = MER_STRATEGY_BALANCE, channel = -1; float samplingDensity = 0, mediumSamplingWeight = -1;
    float scale = 1.0f;                        // heterogeneous `scale`
    float stepsize = 1e-3f;                    // heterogeneousrefractive `stepsize`
    bool aggressiveTracing = false;            // heterogeneousrefractive `aggressivetracing` (needs the `sdf` child)
    int stepper = MER_STEP_VERLET, trEstimator = MER_TR_WOODCOCK2;
    int method = MER_METHOD_WOODCOCK; float hetStepSize = 0;       ///< heterogeneous `method`, `stepSize` (heterogeneous.cpp:183-202)
    Spectrum emission{};
    std::shared_ptr<VolumeDataSource> density, albedo, rif, sdf;
    std::shared_ptr<PhaseFunction> phase;
};

class Shape : public ConfigurableObject {
public:
    const char *getClassName() const override { return "Shape"; }
    void addChild(const std::string &name, ObjRef child) override;
    int boundary = MER_BOUNDARY_AABB;
    float bmin[3] = {-1, -1, -1}, bmax[3] = {1, 1, 1}, center[3] = {0, 0, 0}, radius = 1;
    std::shared_ptr<Medium> interior;
    bool isRectangle = false; float rectToWorld[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};   ///< `rectangle` (src/shapes/rectangle.cpp): carrier of an area emitter
    std::shared_ptr<class Emitter> areaEmitter;
    bool hasBSDF = false;              ///< a bsdf child was given (null | hdielectric)
    int bsdf = MER_BSDF_NULL;          ///< MER_BSDF_*
};
/// BSDF of the medium shape: `null` (index-matched) or `hdielectric` (src/bsdfs/hdielectric.cpp: eta = RIF at the hit point)
class BSDF : public ConfigurableObject {
public:
    const char *getClassName() const override { return "BSDF"; }
    int kind = MER_BSDF_NULL;
    bool isheterogeneousbsdf() const { return kind == MER_BSDF_HDIELECTRIC; }
};

class ReconstructionFilter : public ConfigurableObject {
public:
    const char *getClassName() const override { return "ReconstructionFilter"; }
    int kind = MER_FILTER_GAUSSIAN; float param = 0.5f;
};
class Sampler : public ConfigurableObject {
public:
    const char *getClassName() const override { return "Sampler"; }
    int sampleCount = 4;
};
class Film : public ConfigurableObject {
public:
    const char *getClassName() const override { return "Film"; }
    void addChild(const std::string &name, ObjRef child) override;
    int width = 768, height = 576;
    /// src/librender/film.cpp:56-84
    int decomposition = MER_DECOMPOSITION_NONE; float minBound = 0.0f, maxBound = 0.0f, binWidth = 1.0f; bool calibratedTransient = false;
    /// PathLengthSampler (src/librender/pathlengthsampler.cpp:12-40)
    int modulation = MER_MODULATION_NONE; float lambda = 1.0f, phase = 0.0f; int P = 32, neighbors = 3;
    int frames() const { return (decomposition != MER_DECOMPOSITION_NONE && modulation == MER_MODULATION_NONE) ? (int) std::ceil((maxBound - minBound) / binWidth) : 1; }
    int channels() const { return frames() * 3 + 2; }
    std::shared_ptr<ReconstructionFilter> rfilter;
};
class Sensor : public ConfigurableObject {
public:
    const char *getClassName() const override { return "Sensor"; }
    void addChild(const std::string &name, ObjRef child) override;
    float fov = 50.0f; std::string fovAxis = "x"; float nearClip = 1e-2f, farClip = 1e4f;
    float toWorld[16];
    std::shared_ptr<Film> film; std::shared_ptr<Sampler> sampler;
};
class Emitter : public ConfigurableObject {
public:
    const char *getClassName() const override { return "Emitter"; }
    enum Kind { EConstant, EPoint, EArea } kind = EConstant;
    Spectrum radiance{};                        // constant / area: radiance ; point: intensity
    Vec3 position{0, 0, 0};                     // point (src/emitters/point.cpp:60-68)
};

class Scene;
/// `volpath` executed on the GPU: Integrator::render() (include/mitsuba/render/integrator.h:74) owns its parallelism
class Integrator : public ConfigurableObject {
public:
    const char *getClassName() const override { return "Integrator"; }
    int maxDepth = -1, rrDepth = 5; bool hideEmitters = false, strictNormals = false;
    /// flatten (validates like the reference's configure()) -- no GPU needed
    void flatten(const Scene &scene, mer_scene_desc &desc) const;
    /// upload volumes, render `spp` samples per pixel (0 = the sampler's sampleCount), return the film [h][w][5]
    std::vector<float> render(const Scene &scene, int device, int spp, unsigned long long seed, int layout) const;
    /// the same on several GPUs of this machine (mer_multi_*: replicated volumes, shardMode = MER_SHARD_SAMPLES | MER_SHARD_TILES, films reduced with RCCL)
    std::vector<float> render(const Scene &scene, const std::vector<int> &devices, int shardMode, int spp, unsigned long long seed, int layout) const;
};

class Scene : public ConfigurableObject {
public:
    const char *getClassName() const override { return "Scene"; }
    void addChild(const std::string &name, ObjRef child) override;
    void configure() override;
    std::shared_ptr<Integrator> integrator; std::shared_ptr<Sensor> sensor;
    std::vector<std::shared_ptr<Shape>> shapes; std::vector<std::shared_ptr<Emitter>> emitters;
    std::vector<std::shared_ptr<Medium>> media;
};

/// PluginManager::createObject (src/libcore/plugin.cpp:180-196): plugin found by its short name = XML `type`
ObjRef createObject(const std::string &tag, const Properties &props, const std::string &baseDir);

/// SceneHandler: parse a scene file; `defines` are the -D key=value substitutions for $key
std::shared_ptr<Scene> loadScene(const std::string &path, const std::map<std::string, std::string> &defines);
std::shared_ptr<Scene> loadSceneFromString(const std::string &xml, const std::map<std::string, std::string> &defines, const std::string &baseDir);

/// film [h][w][frames*3+2] -> developed RGB [frames][h][w][3] (HDRFilm::develop: divide by the weight channel)
std::vector<float> develop(const std::vector<float> &film, int w, int h, int frames = 1);
void writeNpy(const std::string &path, const float *data, int h, int w, int c);
void writePfm(const std::string &path, const float *rgb, int h, int w);
/// OpenEXR 2 scan-line file, uncompressed, three float32 channels B, G, R (what HDRFilm::develop writes through OpenEXR,
/// src/films/hdrfilm.cpp:527; no OpenEXR library is needed for this subset)
void writeExr(const std::string &path, const float *rgb, int h, int w);

}  // namespace merhost

extern "C" {
/* C entry points of libmer_host.so for non-C++ callers (tests): return 0 / 1, message via merhost_last_error() */
const char *merhost_last_error(void);
/* parse + validate only: fills the flat scene (volumes = 0 handles) and width/height/spp */
int merhost_flatten_xml(const char *path, const char *defines /* "k=v;k=v" */, mer_scene_desc *out, int32_t *spp);
/* parse, upload, render on `device`; film_host = float[h][w][5] of the scene's film size (query with flatten first) */
int merhost_render_xml(const char *path, const char *defines, int32_t device, int32_t spp, uint64_t seed, int32_t layout, float *film_host);
/* the same on n devices (a device may be listed twice): shard_mode = MER_SHARD_SAMPLES | MER_SHARD_TILES */
int merhost_render_xml_multi(const char *path, const char *defines, const int32_t *devices, int32_t n, int32_t shard_mode, int32_t spp, uint64_t seed, int32_t layout,
                             float *film_host);
}
