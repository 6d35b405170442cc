// mer_host.cpp -- implementation of the C++ host mirror (see mer_host.h).  Plain C++17, links libmer.so.
#include "mer_host.h"
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <functional>
#include <sstream>

namespace merhost {

void Log_EError(const std::string &msg) { throw std::runtime_error(msg); }

static std::string lower(std::string s) { for (auto &c : s) c = (char) std::tolower((unsigned char) c); return s; }

// ------------------------------------------------------------------------------------------------ Properties
#define SETTER(fn, T, field, TYPE) void Properties::fn(const std::string &n, T v) { Entry e; e.type = TYPE; e.field = v; m_entries[n] = e; }
SETTER(setBoolean, bool, b, EBoolean) SETTER(setInteger, int, i, EInteger) SETTER(setFloat, float, f, EFloat)
SETTER(setString, const std::string &, s, EString) SETTER(setSpectrum, const Spectrum &, spec, ESpectrum) SETTER(setPoint, const Vec3 &, p, EPoint)
#undef SETTER
void Properties::setTransform(const std::string &n, const float m[16]) { Entry e; e.type = ETransform; std::memcpy(e.m, m, sizeof(e.m)); m_entries[n] = e; }

const Properties::Entry &Properties::get(const std::string &n, Type t) const {
    auto it = m_entries.find(n);
    if (it == m_entries.end()) Log_EError("Property \"" + n + "\" missing");                                  // properties.cpp
    if (it->second.type != t && !(t == EFloat && it->second.type == EInteger) && !(t == ESpectrum && it->second.type == EFloat))
        Log_EError("The property \"" + n + "\" has the wrong type");
    it->second.queried = true;
    return it->second;
}
Properties::Type Properties::getType(const std::string &n) const {
    auto it = m_entries.find(n);
    if (it == m_entries.end()) Log_EError("Property \"" + n + "\" missing");
    return it->second.type;
}
bool Properties::getBoolean(const std::string &n) const { return get(n, EBoolean).b; }
bool Properties::getBoolean(const std::string &n, bool d) const { return hasProperty(n) ? getBoolean(n) : d; }
int Properties::getInteger(const std::string &n) const { return get(n, EInteger).i; }
int Properties::getInteger(const std::string &n, int d) const { return hasProperty(n) ? getInteger(n) : d; }
float Properties::getFloat(const std::string &n) const { const Entry &e = get(n, EFloat); return e.type == EInteger ? (float) e.i : e.f; }
float Properties::getFloat(const std::string &n, float d) const { return hasProperty(n) ? getFloat(n) : d; }
std::string Properties::getString(const std::string &n) const { return get(n, EString).s; }
std::string Properties::getString(const std::string &n, const std::string &d) const { return hasProperty(n) ? getString(n) : d; }
Spectrum Properties::getSpectrum(const std::string &n) const {
    const Entry &e = get(n, ESpectrum);
    if (e.type == EFloat) return Spectrum{{e.f, e.f, e.f}};
    return e.spec;
}
Spectrum Properties::getSpectrum(const std::string &n, const Spectrum &d) const { return hasProperty(n) ? getSpectrum(n) : d; }
Vec3 Properties::getPoint(const std::string &n) const { return get(n, EPoint).p; }
Vec3 Properties::getPoint(const std::string &n, const Vec3 &d) const { return hasProperty(n) ? getPoint(n) : d; }
void Properties::getTransform(const std::string &n, float m[16]) const {
    if (!hasProperty(n)) { for (int i = 0; i < 16; i++) m[i] = (i % 5 == 0) ? 1.0f : 0.0f; return; }
    std::memcpy(m, get(n, ETransform).m, sizeof(float) * 16);
}
std::vector<std::string> Properties::getUnqueried() const {
    std::vector<std::string> r;
    for (auto &kv : m_entries) if (!kv.second.queried) r.push_back(kv.first);
    return r;
}

// ------------------------------------------------------------------------------------------------ objects
void ConfigurableObject::addChild(const std::string &name, ObjRef child) {
    Log_EError(std::string(getClassName()) + ": Invalid child node! (\"" + child->getClassName() + "\")");   // cobject / medium.cpp:54
}

float VolumeDataSource::getStepSize() const {
    if (isConstant()) return std::numeric_limits<float>::infinity();                                       // constvolume
    float s = std::numeric_limits<float>::infinity();
    for (int i = 0; i < 3; ++i) s = std::min(s, 0.5f * (aabb_max[i] - aabb_min[i]) / (float) (res[i] - 1));
    return s;
}

namespace {
struct ConstVolume : VolumeDataSource {
    bool isSpec = false;
    bool supportsFloatLookups() const override { return !isSpec; }
    bool supportsSpectrumLookups() const override { return isSpec; }
    bool isConstant() const override { return true; }
};
struct AcousticRIFVolume : VolumeDataSource {                  // src/volume/acousticrifvolume.cpp:101-106
    bool supportsFloatLookups() const override { return true; }
    bool isAcoustic() const override { return true; }
};
struct GridVolume : VolumeDataSource {
    bool spline = false;
    bool supportsFloatLookups() const override { return channels == 1; }
    bool supportsSpectrumLookups() const override { return channels == 3; }
    bool isSpline() const override { return spline; }
};

// GridDataSource::loadFromFile (src/volume/gridvolume.cpp:217-287)
void loadVol(GridVolume &v, const std::string &path, bool aabbGiven) {
    std::ifstream f(path, std::ios::binary);
    if (!f) Log_EError("The file \"" + path + "\" does not exist!");
    unsigned char hdr[48];
    f.read((char *) hdr, 48);
    if (f.gcount() != 48 || hdr[0] != 'V' || hdr[1] != 'O' || hdr[2] != 'L')
        Log_EError("Encountered an invalid volume data file (incorrect header identifier)");
    if (hdr[3] != 3) Log_EError("Encountered an invalid volume data file (incorrect file version)");
    int32_t h[5]; std::memcpy(h, hdr + 4, 20);
    float bb[6]; std::memcpy(bb, hdr + 24, 24);
    const int type = h[0];
    v.res[0] = h[1]; v.res[1] = h[2]; v.res[2] = h[3]; v.channels = h[4];
    if (type == 2) Log_EError("Error: float16 volumes are not yet supported!");
    if (type != MER_VOL_F32 && type != MER_VOL_U8) {
        char buf[160]; std::snprintf(buf, sizeof(buf), "Encountered a volume data file of unknown type (type=%i, channels=%i)!", type, v.channels);
        Log_EError(buf);
    }
    if (v.channels != 1 && v.channels != 3) {
        char buf[200];
        std::snprintf(buf, sizeof(buf), "Encountered an unsupported %s volume data file (%i channels, only 1 and 3 are supported)",
                      type == MER_VOL_F32 ? "float32" : "uint8", v.channels);
        Log_EError(buf);
    }
    v.dtype = type;
    if (!aabbGiven) for (int i = 0; i < 3; i++) { v.aabb_min[i] = bb[i]; v.aabb_max[i] = bb[3 + i]; }
    const size_t n = (size_t) v.res[0] * v.res[1] * v.res[2] * v.channels * (type == MER_VOL_F32 ? 4 : 1);
    v.data.resize(n);
    f.read((char *) v.data.data(), (std::streamsize) n);
    if ((size_t) f.gcount() != n) Log_EError("Volume data file \"" + path + "\" is truncated");
    v.filename = path;
}

void requireIdentity(const Properties &props, const char *who) {
    float m[16]; props.getTransform("toWorld", m);
    for (int i = 0; i < 16; i++) if (std::fabs(m[i] - ((i % 5 == 0) ? 1.0f : 0.0f)) > 1e-6f)
        Log_EError(std::string(who) + ": a non-identity 'toWorld' transform is not supported on the GPU path");
}
}  // namespace

void Medium::addChild(const std::string &name, ObjRef child) {
    const std::string cls = child->getClassName();
    if (cls == "VolumeDataSource" && kind != "homogeneous") {
        auto vol = std::static_pointer_cast<VolumeDataSource>(child);
        if (name == "albedo") {                                   // heterogeneous.cpp:262-281
            if (!vol->supportsSpectrumLookups()) Log_EError("Assertion 'volume->supportsSpectrumLookups()' failed");
            albedo = vol;
        } else if (name == "density") {
            if (!vol->supportsFloatLookups()) Log_EError("Assertion 'volume->supportsFloatLookups()' failed");
            density = vol;
        } else if (name == "rif" && kind == "heterogeneousrefractive") {        // heterogeneousrefractive.cpp:1177-1193
            rif = vol;
        } else if (name == "sdf" && kind == "heterogeneousrefractive") {
            sdf = vol;
        } else if (name == "orientation") {
            Log_EError("heterogeneous: anisotropic media ('orientation') are not supported on the GPU path");
        } else Log_EError("Medium: Invalid child node! (\"VolumeDataSource\")");
    } else if (cls == "PhaseFunction") {
        if (phase) Log_EError("Assertion 'm_phaseFunction == NULL' failed");                 // medium.cpp:50
        phase = std::static_pointer_cast<PhaseFunction>(child);
    } else Log_EError("Medium: Invalid child node! (\"" + cls + "\")");                      // medium.cpp:53-55
}

void Medium::configure() {
    if (!phase) phase = std::static_pointer_cast<PhaseFunction>(createObject("phase", Properties("isotropic"), ""));   // medium.cpp:58-64
    if (kind == "heterogeneous") {
        if (!density) Log_EError("No density specified!");          // heterogeneous.cpp:229-232
        if (!albedo) Log_EError("No albedo specified!");
    } else if (kind == "heterogeneousrefractive") {
        if (!rif) Log_EError("No RIF specified!");                  // heterogeneousrefractive.cpp:368-369
        // the reference demands an `sdf` child; here it is optional: without it the boundary is the shape itself (D5), with it the
        // negative region of the grid
        if (rif->channels != 1 || rif->dtype != MER_VOL_F32) Log_EError("The RIF must be a 1-channel float32 volume");
        if (density && !albedo) Log_EError("No albedo specified!");
    }
}

void Shape::addChild(const std::string &name, ObjRef child) {
    const std::string cls = child->getClassName();
    if (cls == "Medium") {
        if (name == "interior") {
            interior = std::static_pointer_cast<Medium>(child);
            if (interior->isheterogeneousrefractive() && hasBSDF && bsdf != MER_BSDF_HDIELECTRIC)        // shape.cpp:172-176
                Log_EError("A shape with heterogeneous refractive index medium should only have a bsdf that is also heterogeneous!");
        } else if (name == "exterior") Log_EError("Shape: an 'exterior' medium is not supported on the GPU path (the sensor must be in vacuum)");
        else Log_EError("Shape: Invalid medium child (must be named 'interior' or 'exterior')!");    // shape.cpp:186-188
    } else if (cls == "BSDF") { /* recorded before the children are attached (see build()) */ }
    else if (cls == "Emitter") {                                         // src/librender/shape.cpp:137-146
        auto e = std::static_pointer_cast<Emitter>(child);
        if (areaEmitter) Log_EError("Tried to attach multiple emitters to a shape!");
        if (e->kind != Emitter::EArea) Log_EError("Tried to attach a non-surface emitter to a shape");
        if (!isRectangle) Log_EError("area emitter: only a 'rectangle' shape can carry one on the GPU path");
        areaEmitter = e;
    }
    else ConfigurableObject::addChild(name, child);
}
void Film::addChild(const std::string &name, ObjRef child) {
    if (std::string(child->getClassName()) == "ReconstructionFilter") rfilter = std::static_pointer_cast<ReconstructionFilter>(child);
    else ConfigurableObject::addChild(name, child);
}
void Sensor::addChild(const std::string &name, ObjRef child) {
    const std::string cls = child->getClassName();
    if (cls == "Film") film = std::static_pointer_cast<Film>(child);
    else if (cls == "Sampler") sampler = std::static_pointer_cast<Sampler>(child);
    else ConfigurableObject::addChild(name, child);
}
void Scene::addChild(const std::string &name, ObjRef child) {
    const std::string cls = child->getClassName();
    if (cls == "Integrator") integrator = std::static_pointer_cast<Integrator>(child);
    else if (cls == "Sensor") sensor = std::static_pointer_cast<Sensor>(child);
    else if (cls == "Shape") shapes.push_back(std::static_pointer_cast<Shape>(child));
    else if (cls == "Emitter") emitters.push_back(std::static_pointer_cast<Emitter>(child));
    else if (cls == "Medium") media.push_back(std::static_pointer_cast<Medium>(child));
    else if (cls == "PhaseFunction" || cls == "VolumeDataSource") { /* referenced objects */ }
    else ConfigurableObject::addChild(name, child);
}
void Scene::configure() {
    if (!integrator) Log_EError("Scene: no integrator was specified");
    if (!sensor) Log_EError("Scene: no sensor was specified");
    if (!sensor->film) sensor->film = std::static_pointer_cast<Film>(createObject("film", Properties("hdrfilm"), ""));
    if (!sensor->film->rfilter) sensor->film->rfilter = std::static_pointer_cast<ReconstructionFilter>(createObject("rfilter", Properties("gaussian"), ""));
    if (!sensor->sampler) sensor->sampler = std::static_pointer_cast<Sampler>(createObject("sampler", Properties("independent"), ""));
}

// ------------------------------------------------------------------------------------------------ factory
ObjRef createObject(const std::string &tag, const Properties &props, const std::string &baseDir) {
    const std::string type = props.getPluginName();
    auto resolve = [&](const std::string &fn) { return (fn.empty() || fn[0] == '/' || baseDir.empty()) ? fn : baseDir + "/" + fn; };
    ObjRef out;
    if (tag == "integrator") {
        if (type != "volpath") Log_EError("integrator \"" + type + "\": only 'volpath' runs on the GPU path");
        auto o = std::make_shared<Integrator>();
        o->rrDepth = props.getInteger("rrDepth", 5); o->maxDepth = props.getInteger("maxDepth", -1);        // integrator.cpp:190-225
        o->strictNormals = props.getBoolean("strictNormals", false); o->hideEmitters = props.getBoolean("hideEmitters", false);
        if (o->rrDepth <= 0) Log_EError("'rrDepth' must be set to a value greater than zero!");
        if (o->maxDepth <= 0 && o->maxDepth != -1) Log_EError("'maxDepth' must be set to -1 (infinite) or a value greater than zero!");
        out = o;
    } else if (tag == "phase") {
        auto o = std::make_shared<PhaseFunction>();
        if (type == "hg") {
            o->kind = MER_PHASE_HG; o->g = props.getFloat("g", 0.8f);                                     // hg.cpp:47-54
            if (o->g >= 1 || o->g <= -1) Log_EError("The asymmetry parameter must lie in the interval (-1, 1)!");
        } else if (type == "isotropic") o->kind = MER_PHASE_ISOTROPIC;
        else Log_EError("phase function \"" + type + "\" is not supported on the GPU path (hg, isotropic)");
        out = o;
    } else if (tag == "volume") {
        if (type == "constvolume") {                                                                       // constvolume.cpp:57-64
            auto o = std::make_shared<ConstVolume>();
            if (!props.hasProperty("value")) Log_EError("Property \"value\" missing");
            const Properties::Type t = props.getType("value");
            if (t != Properties::EFloat && t != Properties::EInteger && t != Properties::ESpectrum)
                Log_EError("The value of a 'constvolume' must have one of the following types: float, vector, spectrum");
            o->constant = props.getSpectrum("value");
            o->isSpec = t == Properties::ESpectrum;
            o->channels = o->isSpec ? 3 : 1;
            out = o;
        } else if (type == "gridvolume" || type == "splinevolume") {                                        // gridvolume.cpp:108-129
            auto o = std::make_shared<GridVolume>();
            o->spline = type == "splinevolume";
            {   // m_worldToVolume = m_volumeToWorld.inverse() (gridvolume.cpp:110,188-189): affine inverse by cofactors, in double
                float m[16]; props.getTransform("toWorld", m);
                bool ident = true; for (int i = 0; i < 16; i++) ident = ident && m[i] == ((i % 5 == 0) ? 1.0f : 0.0f);
                if (!ident) {
                    if (m[12] != 0 || m[13] != 0 || m[14] != 0 || m[15] != 1) Log_EError(type + ": 'toWorld' must be an affine transform");
                    const double a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], h = m[9], k = m[10];
                    const double det = a * (e * k - f * h) - b * (d * k - f * g) + c * (d * h - e * g);
                    if (!(std::fabs(det) > 1e-12)) Log_EError(type + ": 'toWorld' is not invertible");
                    const double inv[9] = {(e * k - f * h) / det, (c * h - b * k) / det, (b * f - c * e) / det, (f * g - d * k) / det, (a * k - c * g) / det,
                                           (c * d - a * f) / det, (d * h - e * g) / det, (b * g - a * h) / det, (a * e - b * d) / det};
                    for (int i = 0; i < 3; i++) {
                        for (int j = 0; j < 3; j++) o->worldToVolume[i * 4 + j] = (float) inv[i * 3 + j];
                        o->worldToVolume[i * 4 + 3] = (float) -(inv[i * 3] * m[3] + inv[i * 3 + 1] * m[7] + inv[i * 3 + 2] * m[11]);
                    }
                }
            }
            bool given = props.hasProperty("min") && props.hasProperty("max");
            if (given) {
                Vec3 a = props.getPoint("min"), b = props.getPoint("max");
                o->aabb_min[0] = a.x; o->aabb_min[1] = a.y; o->aabb_min[2] = a.z; o->aabb_max[0] = b.x; o->aabb_max[1] = b.y; o->aabb_max[2] = b.z;
            }
            (void) props.getBoolean("sendData", false);
            loadVol(*o, resolve(props.getString("filename")), given);
            out = o;
        } else if (type == "acousticrifvolume") {                                                          // acousticrifvolume.cpp:101-106
            auto o = std::make_shared<AcousticRIFVolume>();
            requireIdentity(props, type.c_str());
            if (props.hasProperty("min")) (void) props.getPoint("min");
            if (props.hasProperty("max")) (void) props.getPoint("max");
            const float f_u = props.getFloat("freq", 832000.0f), speed_u = props.getFloat("speed", 1500.0f);
            o->ac_n_o = props.getFloat("n_o", 1.3333f); o->ac_n_max = props.getFloat("n_max", 0.0f); o->ac_mode = props.getInteger("mode", 0);
            const float wavelength_u = speed_u / f_u;
            o->ac_k_r = (float) ((2 * 3.14159265358979323846) / wavelength_u);
            o->channels = 1;
            out = o;
        } else Log_EError("volume \"" + type + "\" is not supported on the GPU path (gridvolume, splinevolume, constvolume, acousticrifvolume)");
    } else if (tag == "medium") {
        auto o = std::make_shared<Medium>();
        o->kind = type;
        if (type == "homogeneous" || type == "heterogeneousrefractive") {
            // lookupMaterial (src/medium/materials.h:61-190) without the preset table
            const bool hasAS = props.hasProperty("sigmaS") || props.hasProperty("sigmaA");
            const bool hasTA = props.hasProperty("sigmaT") || props.hasProperty("albedo");
            if (hasAS && hasTA) Log_EError("You can either specify sigmaS & sigmaA *or* sigmaT & albedo, but no other combinations!");
            if (props.hasProperty("material")) Log_EError("medium: material presets are not available on the GPU path; give sigmaS/sigmaA or sigmaT/albedo");
            Spectrum sS{{0, 0, 0}}, sA{{0, 0, 0}};
            if (hasAS) { sS = props.getSpectrum("sigmaS", sS); sA = props.getSpectrum("sigmaA", sA); }
            else if (hasTA) {
                Spectrum sT = props.getSpectrum("sigmaT"), al = props.getSpectrum("albedo");
                for (int i = 0; i < 3; i++) { sS.c[i] = al.c[i] * sT.c[i]; sA.c[i] = sT.c[i] - sS.c[i]; }
            } else if (type == "homogeneous") Log_EError("medium: specify sigmaS/sigmaA or sigmaT/albedo (material presets are not available on the GPU path)");
            float g = props.getFloat("g", 0.0f);
            if (g <= -1 || g >= 1) Log_EError("The anisotropy parameter 'g' must be in the range (-1, 1)!");
            const float scale = props.getFloat("scale", 1.0f);
            for (int i = 0; i < 3; i++) { sS.c[i] *= scale * (1.0f - g); sA.c[i] *= scale; }               // materials.h:188-190, medium.cpp:33
            o->sigmaS = sS; o->sigmaA = sA;
            const std::string strategy = props.getString("strategy", "balance");                           // homogeneous.cpp:156-228
            if (strategy == "balance") o->strategy = MER_STRATEGY_BALANCE;
            else if (strategy == "single") { o->strategy = MER_STRATEGY_SINGLE; o->channel = props.getInteger("channel", -1); }
            else if (strategy == "manual") { o->strategy = MER_STRATEGY_MANUAL; o->samplingDensity = props.getFloat("samplingDensity"); }
            else if (strategy == "maximum") o->strategy = MER_STRATEGY_MAXIMUM;                                   // homogeneous.cpp:215-220
            else Log_EError("Specified an unknown sampling strategy");
            o->mediumSamplingWeight = props.getFloat("mediumSamplingWeight", -1);
            if (type == "heterogeneousrefractive") {
                o->stepsize = props.getFloat("stepsize", 1e-3f);                                           // heterogeneousrefractive.cpp:208
                o->scale = scale;
                (void) props.getFloat("tol2", 1e-6f); (void) props.getFloat("rrweight", 1e-2f); (void) props.getInteger("boundaryprecision", 3);
                o->aggressiveTracing = props.getBoolean("aggressivetracing", false);                       // :230
            }
        } else if (type == "heterogeneous") {                                                              // heterogeneous.cpp:183-202
            if (props.hasProperty("sigmaS") || props.hasProperty("sigmaA"))
                Log_EError("The 'sigmaS' and 'sigmaA' properties are only supported by homogeneous media. Please use nested volume instances to supply these parameters");
            if (props.hasProperty("densityMultiplier")) Log_EError("The 'densityMultiplier' parameter has been deprecated and is now called 'scale'.");
            o->scale = props.getFloat("scale", 1);
            o->hetStepSize = props.getFloat("stepSize", 0);
            const std::string method = lower(props.getString("method", "woodcock"));
            if (method == "simpson") o->method = MER_METHOD_SIMPSON;
            else if (method != "woodcock") Log_EError("Unsupported integration method \"" + method + "\"!");
        } else Log_EError("medium \"" + type + "\" is not supported on the GPU path");
        // extensions (documented in INTEGRATION.md): stepper, transmittance estimator, emission
        const std::string stepper = lower(props.getString("stepper", "verlet"));
        if (stepper == "rk4") o->stepper = MER_STEP_RK4; else if (stepper == "verlet") o->stepper = MER_STEP_VERLET;
        else Log_EError("Unknown eikonal stepper \"" + stepper + "\" (verlet, rk4)");
        const std::string tr = lower(props.getString("transmittance", "woodcock"));
        if (tr == "ratio") o->trEstimator = MER_TR_RATIO; else if (tr == "woodcock") o->trEstimator = MER_TR_WOODCOCK2;
        else Log_EError("Unknown transmittance estimator \"" + tr + "\" (woodcock, ratio)");
        o->emission = props.getSpectrum("emission", Spectrum{{0, 0, 0}});
        out = o;
    } else if (tag == "shape") {
        auto o = std::make_shared<Shape>();
        float m[16]; props.getTransform("toWorld", m);
        if (type == "rectangle") {                                       // src/shapes/rectangle.cpp:99-110: the carrier of an `area` emitter; any shear-free toWorld
            o->isRectangle = true;
            for (int i = 0; i < 12; i++) o->rectToWorld[i] = m[i];
            const double du[3] = {m[0], m[4], m[8]}, dv[3] = {m[1], m[5], m[9]};
            const double lu = std::sqrt(du[0] * du[0] + du[1] * du[1] + du[2] * du[2]), lv = std::sqrt(dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2]);
            if (!(lu > 0 && lv > 0) || std::fabs((du[0] * dv[0] + du[1] * dv[1] + du[2] * dv[2]) / (lu * lv)) > 1e-4) Log_EError("Error: 'toWorld' transformation contains shear!");   // :108-109
            out = o;
            return out;
        }
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++)
            if (r != c && std::fabs(m[r * 4 + c]) > 1e-6f) Log_EError("shape: only scale + translate 'toWorld' transforms are supported on the GPU path");
        if (type == "cube") {
            for (int i = 0; i < 3; i++) { float s = std::fabs(m[i * 4 + i]); o->bmin[i] = m[i * 4 + 3] - s; o->bmax[i] = m[i * 4 + 3] + s; }
        } else if (type == "sphere") {
            o->boundary = MER_BOUNDARY_SPHERE;
            Vec3 c = props.getPoint("center", Vec3{0, 0, 0}); const float r = props.getFloat("radius", 1.0f);
            if (std::fabs(m[0] - m[5]) > 1e-6f || std::fabs(m[0] - m[10]) > 1e-6f) Log_EError("sphere: non-uniform scales are not supported");
            o->center[0] = c.x * m[0] + m[3]; o->center[1] = c.y * m[5] + m[7]; o->center[2] = c.z * m[10] + m[11]; o->radius = r * std::fabs(m[0]);
        } else if (type == "obj") {
            // bounding box of the vertices (scenes/volumetric/bounds.obj is the cube [-1,1]^3)
            std::ifstream f(resolve(props.getString("filename")));
            if (!f) Log_EError("The file \"" + resolve(props.getString("filename")) + "\" does not exist!");
            std::string line; bool any = false; float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
            while (std::getline(f, line)) {
                if (line.size() > 2 && line[0] == 'v' && std::isspace((unsigned char) line[1])) {
                    float v[3]; if (std::sscanf(line.c_str() + 1, "%f %f %f", v, v + 1, v + 2) == 3) { any = true; for (int i = 0; i < 3; i++) { lo[i] = std::min(lo[i], v[i]); hi[i] = std::max(hi[i], v[i]); } }
                }
            }
            if (!any) Log_EError("obj: no vertices found");
            for (int i = 0; i < 3; i++) { float a = lo[i] * m[i * 4 + i] + m[i * 4 + 3], b = hi[i] * m[i * 4 + i] + m[i * 4 + 3]; o->bmin[i] = std::min(a, b); o->bmax[i] = std::max(a, b); }
        } else Log_EError("shape \"" + type + "\" is not supported on the GPU path (cube, sphere, obj bounding box; rectangle with an area emitter)");
        out = o;
    } else if (tag == "sensor") {
        if (type != "perspective") Log_EError("sensor \"" + type + "\" is not supported on the GPU path (perspective)");
        auto o = std::make_shared<Sensor>();
        if (props.hasProperty("focalLength")) Log_EError("Please specify either a focal length ('focalLength') or a field of view ('fov')!");
        o->fov = props.getFloat("fov", 50.0f); o->fovAxis = lower(props.getString("fovAxis", "x"));
        o->nearClip = props.getFloat("nearClip", 1e-2f); o->farClip = props.getFloat("farClip", 1e4f);
        (void) props.getFloat("focusDistance", 0.0f);
        props.getTransform("toWorld", o->toWorld);
        out = o;
    } else if (tag == "film") {
        auto o = std::make_shared<Film>();
        o->width = props.getInteger("width", 768); o->height = props.getInteger("height", 576);          // film.cpp
        (void) props.getBoolean("banner", true);
        const std::string dec = lower(props.getString("decomposition", "none"));                         // film.cpp:56-84
        if (dec == "none") o->decomposition = MER_DECOMPOSITION_NONE;
        else if (dec == "transient") o->decomposition = MER_DECOMPOSITION_TRANSIENT;
        else if (dec == "bounce") o->decomposition = MER_DECOMPOSITION_BOUNCE;
        else Log_EError("The \"decomposition\" parameter must be equal toeither \"none\", \"transient\", or \"bounce\"!");
        o->minBound = props.getFloat("minBound", 0.0f); o->maxBound = props.getFloat("maxBound", 0.0f);
        o->binWidth = props.getFloat("binWidth", 1.0f);
        o->calibratedTransient = props.getBoolean("calibratedTransient", false);
        {   // PathLengthSampler(props): pathlengthsampler.cpp:12-38
            const std::string mod = lower(props.getString("modulation", "none"));
            static const char *names[] = {"none", "sine", "square", "hamiltonian", "mseq", "depthselective"};
            o->modulation = -1;
            for (int i = 0; i < 6; i++) if (mod == names[i]) o->modulation = i;
            if (o->modulation < 0) Log_EError("The \"modulation\" parameter must be equal toeither \"none\", \"square\", or \"hamiltonian\", or \"mseq\", or \"depthselective\"!");
            o->lambda = props.getFloat("lambda", 1.0f); o->phase = props.getFloat("phase", 0.0f);
            o->P = props.getInteger("P", 32); o->neighbors = props.getInteger("neighbors", 3);
            if (o->modulation != MER_MODULATION_NONE && o->decomposition != MER_DECOMPOSITION_TRANSIENT)
                Log_EError("film: a path-length modulation needs decomposition = transient");
        }
        if (o->decomposition != MER_DECOMPOSITION_NONE && !(o->frames() >= 1 && o->frames() <= 4096))
            Log_EError("film: a decomposition needs 1 <= ceil((maxBound-minBound)/binWidth) <= 4096 frames");
        out = o;
    } else if (tag == "rfilter") {
        auto o = std::make_shared<ReconstructionFilter>();
        if (type == "gaussian") { o->kind = MER_FILTER_GAUSSIAN; o->param = props.getFloat("stddev", 0.5f); }
        else if (type == "box") { o->kind = MER_FILTER_BOX; o->param = props.getFloat("radius", 0.5f); }
        else Log_EError("rfilter \"" + type + "\" is not supported on the GPU path (gaussian, box)");
        out = o;
    } else if (tag == "sampler") {
        auto o = std::make_shared<Sampler>();
        if (type != "independent" && type != "ldsampler") Log_EError("sampler \"" + type + "\" is not supported on the GPU path");
        o->sampleCount = props.getInteger("sampleCount", 4);
        out = o;
    } else if (tag == "bsdf") {
        auto o = std::make_shared<BSDF>();
        if (type == "null") o->kind = MER_BSDF_NULL;
        else if (type == "hdielectric") {                                  // src/bsdfs/hdielectric.cpp:46-60
            o->kind = MER_BSDF_HDIELECTRIC;
            const Spectrum r = props.getSpectrum("specularReflectance", Spectrum{{1, 1, 1}}), t = props.getSpectrum("specularTransmittance", Spectrum{{1, 1, 1}});
            for (int i = 0; i < 3; i++) if (r.c[i] != 1.0f || t.c[i] != 1.0f) Log_EError("hdielectric: specularReflectance / specularTransmittance other than 1 are not supported on the GPU path");
        } else Log_EError("bsdf \"" + type + "\" is not supported on the GPU path (null, hdielectric)");
        out = o;
    } else if (tag == "emitter") {
        auto o = std::make_shared<Emitter>();
        if (type == "constant") o->radiance = props.getSpectrum("radiance", Spectrum{{1, 1, 1}});
        else if (type == "point") {                                      // src/emitters/point.cpp:57-69
            o->kind = Emitter::EPoint;
            if (props.hasProperty("position")) {
                if (props.hasProperty("toWorld")) Log_EError("Only one of the parameters 'position' and 'toWorld' can be used!'");
                o->position = props.getPoint("position");
            } else { float m[16]; props.getTransform("toWorld", m); o->position = Vec3{m[3], m[7], m[11]}; }
            o->radiance = props.getSpectrum("intensity", Spectrum{{1, 1, 1}});
        } else if (type == "area") {                                     // src/emitters/area.cpp:67-80: child of a shape, which gives it its transformation
            o->kind = Emitter::EArea;
            if (props.hasProperty("toWorld")) Log_EError("Found a 'toWorld' transformation -- this is not allowed -- the area light inherits this transformation from its parent shape");
            o->radiance = props.getSpectrum("radiance", Spectrum{{1, 1, 1}});
        } else Log_EError("emitter \"" + type + "\" is not supported on the GPU path (constant, point, area)");
        out = o;
    } else Log_EError("Unsupported scene element <" + tag + ">");
    return out;
}

// ------------------------------------------------------------------------------------------------ XML subset
namespace {
struct Node { std::string tag; std::map<std::string, std::string> attr; std::vector<Node> children; };

struct Parser {
    const std::string &s; size_t i = 0;
    explicit Parser(const std::string &str) : s(str) {}
    void skipWs() { while (i < s.size() && std::isspace((unsigned char) s[i])) i++; }
    bool starts(const char *p) const { return s.compare(i, std::strlen(p), p) == 0; }
    void skipMisc() {
        for (;;) {
            skipWs();
            if (starts("<!--")) { size_t e = s.find("-->", i); if (e == std::string::npos) Log_EError("XML: unterminated comment"); i = e + 3; }
            else if (starts("<?")) { size_t e = s.find("?>", i); if (e == std::string::npos) Log_EError("XML: unterminated declaration"); i = e + 2; }
            else break;
        }
    }
    Node parseElement() {
        skipMisc();
        if (i >= s.size() || s[i] != '<') Log_EError("XML: expected '<'");
        i++;
        Node n;
        while (i < s.size() && (std::isalnum((unsigned char) s[i]) || s[i] == '_')) n.tag += s[i++];
        for (;;) {
            skipWs();
            if (i >= s.size()) Log_EError("XML: unexpected end of file");
            if (s[i] == '/') { if (s.compare(i, 2, "/>") != 0) Log_EError("XML: malformed tag"); i += 2; return n; }
            if (s[i] == '>') { i++; break; }
            std::string k;
            while (i < s.size() && s[i] != '=' && !std::isspace((unsigned char) s[i])) k += s[i++];
            skipWs();
            if (i >= s.size() || s[i] != '=') Log_EError("XML: attribute \"" + k + "\" has no value");
            i++; skipWs();
            const char q = s[i];
            if (q != '"' && q != '\'') Log_EError("XML: attribute value must be quoted");
            size_t e = s.find(q, i + 1);
            if (e == std::string::npos) Log_EError("XML: unterminated attribute value");
            n.attr[k] = s.substr(i + 1, e - i - 1);
            i = e + 1;
        }
        for (;;) {
            skipMisc();
            if (i >= s.size()) Log_EError("XML: unexpected end of file in <" + n.tag + ">");
            if (starts("</")) {
                size_t e = s.find('>', i);
                if (e == std::string::npos || s.substr(i + 2, e - i - 2).find(n.tag) == std::string::npos) Log_EError("XML: mismatched closing tag for <" + n.tag + ">");
                i = e + 1; return n;
            }
            if (s[i] == '<') n.children.push_back(parseElement());
            else i++;       // character data is ignored (Mitsuba scenes carry none)
        }
    }
};

std::vector<float> parseFloats(const std::string &v) {
    std::vector<float> r; std::string t = v;
    for (auto &c : t) if (c == ',') c = ' ';
    std::istringstream is(t); float f;
    while (is >> f) r.push_back(f);
    return r;
}
float toFloat(const std::string &v, const std::string &what) {
    char *end = NULL; const float f = std::strtof(v.c_str(), &end);
    if (end == v.c_str() || (*end && !std::isspace((unsigned char) *end))) Log_EError("Could not parse floating point value \"" + v + "\" (" + what + ")");
    return f;
}
void matMul(const float a[16], const float b[16], float o[16]) {
    float r[16];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { float s = 0; for (int k = 0; k < 4; k++) s += a[i * 4 + k] * b[k * 4 + j]; r[i * 4 + j] = s; }
    std::memcpy(o, r, sizeof(r));
}
Vec3 vec3Attr(const Node &n, const char *name) {
    auto it = n.attr.find(name);
    if (it == n.attr.end()) Log_EError(std::string("<") + n.tag + ">: missing attribute \"" + name + "\"");
    auto f = parseFloats(it->second);
    if (f.size() != 3) Log_EError(std::string("<") + n.tag + ">: \"" + name + "\" must have three components");
    return Vec3{f[0], f[1], f[2]};
}
// Transform::lookAt (src/libcore/transform.cpp:191-214), left-handed
void lookAt(Vec3 p, Vec3 t, Vec3 up, float m[16]) {
    auto norm = [](Vec3 v) { float l = std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z); if (l == 0) Log_EError("lookAt(): 'origin' and 'target' coincide!"); return Vec3{v.x / l, v.y / l, v.z / l}; };
    auto cross = [](Vec3 a, Vec3 b) { return Vec3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; };
    Vec3 dir = norm(Vec3{t.x - p.x, t.y - p.y, t.z - p.z}), left = norm(cross(up, dir)), nu = cross(dir, left);
    const float r[16] = {left.x, nu.x, dir.x, p.x, left.y, nu.y, dir.y, p.y, left.z, nu.z, dir.z, p.z, 0, 0, 0, 1};
    std::memcpy(m, r, sizeof(r));
}

struct Loader {
    std::map<std::string, std::string> defines;
    std::map<std::string, ObjRef> byId;
    std::string baseDir;

    std::string subst(const std::string &v) const {                       // mitsuba.cpp:58,168-173: $name
        std::string out; size_t i = 0;
        while (i < v.size()) {
            if (v[i] == '$') {
                size_t j = i + 1; std::string k;
                while (j < v.size() && (std::isalnum((unsigned char) v[j]) || v[j] == '_')) k += v[j++];
                auto it = defines.find(k);
                if (it == defines.end()) Log_EError("The parameter \"$" + k + "\" was never specified (use -D " + k + "=value)");
                out += it->second; i = j;
            } else out += v[i++];
        }
        return out;
    }
    std::string attr(const Node &n, const char *name) const {
        auto it = n.attr.find(name);
        if (it == n.attr.end()) Log_EError("<" + n.tag + ">: missing attribute \"" + std::string(name) + "\"");
        return subst(it->second);
    }
    void transform(const Node &n, float m[16]) const {
        for (int i = 0; i < 16; i++) m[i] = (i % 5 == 0) ? 1.0f : 0.0f;
        for (const Node &c : n.children) {
            float t[16]; for (int i = 0; i < 16; i++) t[i] = (i % 5 == 0) ? 1.0f : 0.0f;
            auto opt = [&](const char *k, float d) { auto it = c.attr.find(k); return it == c.attr.end() ? d : toFloat(subst(it->second), k); };
            if (c.tag == "lookat" || c.tag == "lookAt") {
                Node cc = c; for (auto &kv : cc.attr) kv.second = subst(kv.second);
                lookAt(vec3Attr(cc, "origin"), vec3Attr(cc, "target"), cc.attr.count("up") ? vec3Attr(cc, "up") : Vec3{0, 1, 0}, t);
            } else if (c.tag == "translate") { t[3] = opt("x", 0); t[7] = opt("y", 0); t[11] = opt("z", 0); }
            else if (c.tag == "scale") {
                if (c.attr.count("value")) { float v = toFloat(subst(c.attr.at("value")), "scale"); t[0] = t[5] = t[10] = v; }
                else { t[0] = opt("x", 1); t[5] = opt("y", 1); t[10] = opt("z", 1); }
            } else if (c.tag == "rotate") {                                  // Transform::rotate(axis, angle in degrees), src/libcore/transform.cpp:65-91
                float ax = opt("x", 0), ay = opt("y", 0), az = opt("z", 0); const float ang = toFloat(attr(c, "angle"), "angle");
                const float len = std::sqrt(ax * ax + ay * ay + az * az);
                if (len == 0) Log_EError("<rotate>: the axis must not be the zero vector");
                ax /= len; ay /= len; az /= len;
                const float rad = ang * 3.14159265358979323846f / 180.0f, sn = std::sin(rad), cs = std::cos(rad);
                t[0] = ax * ax + (1 - ax * ax) * cs; t[1] = ax * ay * (1 - cs) - az * sn; t[2] = ax * az * (1 - cs) + ay * sn;
                t[4] = ax * ay * (1 - cs) + az * sn; t[5] = ay * ay + (1 - ay * ay) * cs; t[6] = ay * az * (1 - cs) - ax * sn;
                t[8] = ax * az * (1 - cs) - ay * sn; t[9] = ay * az * (1 - cs) + ax * sn; t[10] = az * az + (1 - az * az) * cs;
            } else if (c.tag == "matrix") {
                auto f = parseFloats(attr(c, "value"));
                if (f.size() != 16) Log_EError("<matrix>: expected 16 values");
                for (int i = 0; i < 16; i++) t[i] = f[i];
            } else Log_EError("<transform>: unsupported element <" + c.tag + ">");
            matMul(t, m, m);                                               // later operations apply after earlier ones
        }
    }
    static bool isPluginTag(const std::string &t) {
        static const char *tags[] = {"integrator", "medium", "volume", "phase", "shape", "sensor", "sampler", "film", "rfilter", "emitter", "bsdf"};
        for (const char *p : tags) if (t == p) return true;
        return false;
    }
    ObjRef build(const Node &n) {
        Properties props(attr(n, "type"));
        std::vector<std::pair<std::string, ObjRef>> children;
        for (const Node &c : n.children) {
            if (isPluginTag(c.tag)) { children.emplace_back(c.attr.count("name") ? subst(c.attr.at("name")) : "", build(c)); continue; }
            if (c.tag == "ref") {
                auto it = byId.find(attr(c, "id"));
                if (it == byId.end()) Log_EError("Referenced object \"" + attr(c, "id") + "\" not found");
                children.emplace_back(c.attr.count("name") ? subst(c.attr.at("name")) : "", it->second);
                continue;
            }
            const std::string name = attr(c, "name");
            if (c.tag == "float") props.setFloat(name, toFloat(attr(c, "value"), name));
            else if (c.tag == "integer") props.setInteger(name, (int) std::lround(toFloat(attr(c, "value"), name)));
            else if (c.tag == "boolean") { std::string v = lower(attr(c, "value")); if (v != "true" && v != "false") Log_EError("Could not parse boolean value \"" + v + "\""); props.setBoolean(name, v == "true"); }
            else if (c.tag == "string") props.setString(name, attr(c, "value"));
            else if (c.tag == "spectrum" || c.tag == "rgb") {
                auto f = parseFloats(attr(c, "value"));
                if (f.size() == 1) props.setSpectrum(name, Spectrum{{f[0], f[0], f[0]}});
                else if (f.size() == 3) props.setSpectrum(name, Spectrum{{f[0], f[1], f[2]}});
                else Log_EError("<" + c.tag + " name=\"" + name + "\">: expected 1 or 3 values (SPECTRUM_SAMPLES=3)");
            } else if (c.tag == "point" || c.tag == "vector") {
                Vec3 v{0, 0, 0};
                auto g = [&](const char *k) { auto it = c.attr.find(k); return it == c.attr.end() ? 0.0f : toFloat(subst(it->second), k); };
                v.x = g("x"); v.y = g("y"); v.z = g("z");
                props.setPoint(name, v);
            } else if (c.tag == "transform") { float m[16]; transform(c, m); props.setTransform(name, m); }
            else Log_EError("Unsupported property tag <" + c.tag + ">");
        }
        ObjRef obj = createObject(n.tag, props, baseDir);
        if (n.tag == "shape") for (auto &ch : children) if (std::string(ch.second->getClassName()) == "BSDF") {
            std::static_pointer_cast<Shape>(obj)->hasBSDF = true; std::static_pointer_cast<Shape>(obj)->bsdf = std::static_pointer_cast<BSDF>(ch.second)->kind; }
        for (auto &ch : children) obj->addChild(ch.first, ch.second);
        obj->configure();
        if (n.attr.count("id")) byId[subst(n.attr.at("id"))] = obj;
        return obj;
    }
};
}  // namespace

std::shared_ptr<Scene> loadSceneFromString(const std::string &xml, const std::map<std::string, std::string> &defines, const std::string &baseDir) {
    Parser p(xml);
    Node root = p.parseElement();
    if (root.tag != "scene") Log_EError("XML: the root element must be <scene>");
    if (!root.attr.count("version")) Log_EError("The <scene> element is missing a 'version' attribute");
    Loader L; L.defines = defines; L.baseDir = baseDir;
    auto scene = std::make_shared<Scene>();
    for (const Node &c : root.children) {
        if (!Loader::isPluginTag(c.tag)) Log_EError("Unsupported scene element <" + c.tag + ">");
        scene->addChild(c.attr.count("name") ? c.attr.at("name") : "", L.build(c));
    }
    scene->configure();
    return scene;
}
std::shared_ptr<Scene> loadScene(const std::string &path, const std::map<std::string, std::string> &defines) {
    std::ifstream f(path);
    if (!f) Log_EError("Unable to open scene file \"" + path + "\"");
    std::stringstream ss; ss << f.rdbuf();
    size_t slash = path.find_last_of('/');
    return loadSceneFromString(ss.str(), defines, slash == std::string::npos ? "." : path.substr(0, slash));
}

// ------------------------------------------------------------------------------------------------ integrator
void Integrator::flatten(const Scene &scene, mer_scene_desc &d) const {
    std::memset(&d, 0, sizeof(d));
    const Sensor &se = *scene.sensor; const Film &fi = *se.film;
    d.width = fi.width; d.height = fi.height;
    d.decomposition = fi.decomposition; d.min_bound = fi.minBound; d.max_bound = fi.maxBound; d.bin_width = fi.binWidth;
    d.calibrated_transient = fi.calibratedTransient ? 1 : 0;
    d.modulation = fi.modulation; d.mod_lambda = fi.lambda; d.mod_phase_deg = fi.phase; d.mod_P = fi.P; d.mod_neighbors = fi.neighbors;
    const float aspect = (float) fi.width / (float) fi.height;
    std::string axis = se.fovAxis;                                        // sensor.cpp:246-260
    if (axis == "smaller") axis = aspect > 1 ? "y" : "x"; else if (axis == "larger") axis = aspect > 1 ? "x" : "y";
    if (axis == "x") d.fov_x_deg = se.fov;
    else if (axis == "y") d.fov_x_deg = (float) (2.0 * std::atan(std::tan(0.5 * se.fov * M_PI / 180.0) * aspect) * 180.0 / M_PI);
    else Log_EError("The 'fovAxis' parameter must be set to one of 'smaller', 'larger', 'diagonal', 'x', or 'y'!");
    d.near_clip = se.nearClip; d.far_clip = se.farClip;
    for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) d.cam_to_world[r * 4 + c] = se.toWorld[r * 4 + c];
    d.rfilter = fi.rfilter->kind; d.rfilter_param = fi.rfilter->param;
    d.max_depth = maxDepth; d.rr_depth = rrDepth; d.hide_emitters = hideEmitters;
    const Shape *shape = NULL;
    for (auto &s : scene.shapes) if (s->interior) { if (shape) Log_EError("Only one shape with an interior medium is supported on the GPU path"); shape = s.get(); }
    if (!shape) Log_EError("No shape with an 'interior' medium was found");
    d.boundary = shape->boundary; d.boundary_bsdf = shape->bsdf; d.sdf = 0;
    for (int i = 0; i < 3; i++) { d.bmin[i] = shape->bmin[i]; d.bmax[i] = shape->bmax[i]; d.sph_center[i] = shape->center[i]; }
    d.sph_radius = shape->radius;
    const Medium &m = *shape->interior;
    if (m.density && m.density->isConstant()) Log_EError("heterogeneous: a constant 'density' volume is a homogeneous medium; use the 'homogeneous' plugin");
    // an `sdf` child makes the medium shape the negative region of that grid (heterogeneousrefractive.cpp:366-375); the mesh then only
    // has to enclose it
    if (m.sdf) {
        if (m.sdf->isConstant() || m.sdf->channels != 1 || m.sdf->dtype != MER_VOL_F32) Log_EError("heterogeneousrefractive: the sdf must be a 1-channel float32 grid volume");
        d.boundary = MER_BOUNDARY_SDF;
    }
    const bool grid = m.density != NULL;
    d.sigma_mode = grid ? MER_SIGMA_GRID : MER_SIGMA_HOMOGENEOUS;
    for (int i = 0; i < 3; i++) { d.sigma_a[i] = m.sigmaA.c[i]; d.sigma_s[i] = m.sigmaS.c[i]; }
    d.strategy = m.strategy; d.channel = m.channel; d.sampling_density = m.samplingDensity; d.medium_sampling_weight = m.mediumSamplingWeight;
    d.density_scale = m.scale;
    d.albedo_mode = MER_ALBEDO_CONST; d.albedo[0] = d.albedo[1] = d.albedo[2] = 0;
    if (m.albedo) {
        if (m.albedo->isConstant()) for (int i = 0; i < 3; i++) d.albedo[i] = m.albedo->constant.c[i];
        else d.albedo_mode = MER_ALBEDO_GRID;
    }
    d.rif_mode = MER_RIF_CONST; d.rif_const = 1.0f;
    if (m.rif) d.rif_mode = m.rif->isAcoustic() ? MER_RIF_ACOUSTIC : (m.rif->isSpline() ? MER_RIF_BSPLINE3 : MER_RIF_TRILINEAR);
    d.ac_n_o = 1.0f; d.ac_n_max = 0.0f; d.ac_k_r = 1.0f; d.ac_mode = 0;
    if (m.rif && m.rif->isAcoustic()) { d.ac_n_o = m.rif->ac_n_o; d.ac_n_max = m.rif->ac_n_max; d.ac_k_r = m.rif->ac_k_r; d.ac_mode = m.rif->ac_mode; }
    d.stepper = m.stepper; d.stepsize = m.stepsize;
    d.aggressive_tracing = 0; d.sdf_max_error = 0.0f;
    if (m.aggressiveTracing) {
        if (!m.sdf) Log_EError("aggressivetracing needs the medium's `sdf` volume");
        d.aggressive_tracing = 1;
        double e2 = 0;                                                       // maxSDFError(): one voxel diagonal (splinevolume.cpp:282)
        for (int i = 0; i < 3; i++) { const double st = ((double) m.sdf->aabb_max[i] - m.sdf->aabb_min[i]) / (m.sdf->res[i] - 1); e2 += st * st; }
        d.sdf_max_error = (float) std::sqrt(e2);
    }
    d.phase = m.phase->kind; d.g = m.phase->g;
    d.tr_estimator = m.trEstimator; d.method = m.method; d.het_stepsize = m.hetStepSize;
    int nconst = 0, npoint = 0, narea = 0;
    for (int i = 0; i < 3; i++) { d.env_radiance[i] = 0; d.point_intensity[i] = 0; d.point_position[i] = 0; d.emission[i] = m.emission.c[i]; d.area_radiance[i] = 0; }
    for (int i = 0; i < 12; i++) d.area_to_world[i] = (i % 5 == 0) ? 1.0f : 0.0f;
    for (auto &sh : scene.shapes) {
        if (!sh->isRectangle) continue;
        if (!sh->areaEmitter) Log_EError("shape \"rectangle\" is supported on the GPU path as the carrier of an area emitter only");
        if (++narea > 1) Log_EError("Only one area emitter is supported on the GPU path");
        for (int i = 0; i < 12; i++) d.area_to_world[i] = sh->rectToWorld[i];
        for (int i = 0; i < 3; i++) d.area_radiance[i] = sh->areaEmitter->radiance.c[i];
    }
    for (auto &e : scene.emitters) {
        if (e->kind == Emitter::EArea) Log_EError("An area light must be child of a shape instance");            // area.cpp:200-201
        if (e->kind == Emitter::EPoint) {
            if (++npoint > 1) Log_EError("Only one point emitter is supported on the GPU path");
            d.point_position[0] = e->position.x; d.point_position[1] = e->position.y; d.point_position[2] = e->position.z;
            for (int i = 0; i < 3; i++) d.point_intensity[i] = e->radiance.c[i];
        } else {
            if (++nconst > 1) Log_EError("Only one constant emitter is supported on the GPU path");
            for (int i = 0; i < 3; i++) d.env_radiance[i] = e->radiance.c[i];
        }
    }
}

std::vector<float> Integrator::render(const Scene &scene, int device, int spp, unsigned long long seed, int layout) const {
    return render(scene, std::vector<int>(1, device), MER_SHARD_SAMPLES, spp, seed, layout);
}

// One or several GPUs: mer_multi owns a context per device, replicates the volumes and reduces the films (RCCL between distinct devices).
// The reference's counterpart: `mitsuba -p <workers>` local workers merged by film->put (src/mitsuba/mitsuba.cpp:281,
// src/librender/renderproc.cpp:142-149).
std::vector<float> Integrator::render(const Scene &scene, const std::vector<int> &devices, int shardMode, int spp, unsigned long long seed, int layout) const {
    mer_scene_desc d; flatten(scene, d);
    if (spp <= 0) spp = scene.sensor->sampler->sampleCount;
    if (devices.empty()) Log_EError("Integrator::render: no device given");
    mer_multi *mm = NULL;
    std::vector<int32_t> ids(devices.begin(), devices.end());
    if (mer_multi_create(ids.data(), (int32_t) ids.size(), &mm)) Log_EError(mer_multi_last_error(NULL));
    int32_t channels = 5;
    auto fail = [&](void) { std::string msg = mer_multi_last_error(mm); mer_multi_destroy(mm); Log_EError(msg); };
    const Medium &m = *([&]() -> const Shape * { for (auto &s : scene.shapes) if (s->interior) return s.get(); return (const Shape *) NULL; }())->interior;
    auto upload = [&](const VolumeDataSource &v, int lay) -> mer_volume {
        mer_grid_desc g; std::memset(&g, 0, sizeof(g));
        for (int i = 0; i < 3; i++) { g.res[i] = v.res[i]; g.aabb_min[i] = v.aabb_min[i]; g.aabb_max[i] = v.aabb_max[i]; }
        for (int i = 0; i < 12; i++) g.world_to_volume[i] = v.worldToVolume[i];
        g.channels = v.channels; g.dtype = v.dtype;
        mer_volume h = 0;
        if (mer_multi_volume_upload(mm, &g, v.data.data(), lay, &h)) fail();
        return h;
    };
    if (m.density) d.density = upload(*m.density, MER_LAYOUT_DENSE);
    if (m.albedo && !m.albedo->isConstant()) d.albedo_grid = upload(*m.albedo, MER_LAYOUT_DENSE);
    if (m.rif && !m.rif->isAcoustic()) {
        d.rif = upload(*m.rif, m.rif->isSpline() ? MER_LAYOUT_DENSE : layout);
        if (m.rif->isSpline() && mer_multi_volume_build_spline(mm, d.rif)) fail();
    }
    if (m.sdf) d.sdf = upload(*m.sdf, MER_LAYOUT_DENSE);
    if (mer_film_channels(mer_multi_context(mm, 0), &d, &channels)) { std::string msg = mer_last_error(mer_multi_context(mm, 0)); mer_multi_destroy(mm); Log_EError(msg); }
    std::vector<float> film((size_t) d.width * d.height * channels, 0.0f);
    if (mer_multi_render(mm, &d, shardMode, 0, spp, seed, 1, film.data())) fail();
    mer_multi_destroy(mm);
    return film;
}

std::vector<float> develop(const std::vector<float> &film, int w, int h, int frames) {
    const int ch = frames * 3 + 2;
    std::vector<float> rgb((size_t) frames * w * h * 3);
    for (size_t p = 0; p < (size_t) w * h; p++) {
        const float wgt = film[p * ch + ch - 1], inv = wgt > 0 ? 1.0f / wgt : 0.0f;     // HDRFilm::develop
        for (int f = 0; f < frames; f++)
            for (int c = 0; c < 3; c++) rgb[((size_t) f * w * h + p) * 3 + c] = film[p * ch + f * 3 + c] * inv;
    }
    return rgb;
}
void writeNpy(const std::string &path, const float *data, int h, int w, int c) {
    std::ofstream f(path, std::ios::binary);
    if (!f) Log_EError("Unable to write \"" + path + "\"");
    char dict[128];
    std::snprintf(dict, sizeof(dict), "{'descr': '<f4', 'fortran_order': False, 'shape': (%d, %d, %d), }", h, w, c);
    std::string hdr = dict;
    while ((10 + hdr.size() + 1) % 64 != 0) hdr += ' ';
    hdr += '\n';
    const unsigned short len = (unsigned short) hdr.size();
    f.write("\x93NUMPY\x01\x00", 8); f.write((const char *) &len, 2); f.write(hdr.data(), (std::streamsize) hdr.size());
    f.write((const char *) data, (std::streamsize) ((size_t) h * w * c * 4));
}
void writePfm(const std::string &path, const float *rgb, int h, int w) {
    std::ofstream f(path, std::ios::binary);
    if (!f) Log_EError("Unable to write \"" + path + "\"");
    f << "PF\n" << w << " " << h << "\n-1.0\n";
    for (int y = h - 1; y >= 0; y--) f.write((const char *) (rgb + (size_t) y * w * 3), (std::streamsize) ((size_t) w * 12));
}

void writeExr(const std::string &path, const float *rgb, int h, int w) {
    std::ofstream f(path, std::ios::binary);
    if (!f) Log_EError("Unable to write \"" + path + "\"");
    auto put32 = [&](uint32_t v) { f.write((const char *) &v, 4); };
    auto puts0 = [&](const char *s) { f.write(s, (std::streamsize) std::strlen(s) + 1); };
    auto attr = [&](const char *name, const char *type, const void *data, uint32_t size) { puts0(name); puts0(type); put32(size); f.write((const char *) data, size); };
    put32(20000630u); put32(2u);                                            // magic, version 2 (scan lines, no flags)
    {   // chlist: name\0, int32 pixel type (2 = FLOAT), uint8 pLinear + 3 reserved, int32 xSampling, ySampling; terminated by \0
        std::string ch;
        for (const char *c : {"B", "G", "R"}) { ch += c; ch += '\0'; int32_t t = 2, one = 1; char lin[4] = {0, 0, 0, 0};
            ch.append((const char *) &t, 4); ch.append(lin, 4); ch.append((const char *) &one, 4); ch.append((const char *) &one, 4); }
        ch += '\0';
        attr("channels", "chlist", ch.data(), (uint32_t) ch.size());
    }
    { unsigned char c = 0; attr("compression", "compression", &c, 1); }    // NO_COMPRESSION
    { int32_t b[4] = {0, 0, w - 1, h - 1}; attr("dataWindow", "box2i", b, 16); attr("displayWindow", "box2i", b, 16); }
    { unsigned char c = 0; attr("lineOrder", "lineOrder", &c, 1); }        // INCREASING_Y
    { float a = 1.0f; attr("pixelAspectRatio", "float", &a, 4); }
    { float c[2] = {0, 0}; attr("screenWindowCenter", "v2f", c, 8); }
    { float a = 1.0f; attr("screenWindowWidth", "float", &a, 4); }
    f.put('\0');                                                           // end of header
    const uint64_t line_bytes = (uint64_t) w * 3 * 4, table = (uint64_t) f.tellp() + (uint64_t) h * 8;
    for (int y = 0; y < h; y++) { uint64_t off = table + (uint64_t) y * (8 + line_bytes); f.write((const char *) &off, 8); }
    std::vector<float> row((size_t) w);
    for (int y = 0; y < h; y++) {
        put32((uint32_t) y); put32((uint32_t) line_bytes);
        for (int c = 2; c >= 0; c--) {                                      // channels in alphabetical order: B, G, R
            for (int x = 0; x < w; x++) row[(size_t) x] = rgb[((size_t) y * w + x) * 3 + c];
            f.write((const char *) row.data(), (std::streamsize) w * 4);
        }
    }
}

}  // namespace merhost

// ------------------------------------------------------------------------------------------------ C exports
static thread_local std::string g_host_error;
static std::map<std::string, std::string> parseDefines(const char *defs) {
    std::map<std::string, std::string> m;
    if (!defs) return m;
    std::string s = defs; size_t i = 0;
    while (i < s.size()) {
        size_t e = s.find(';', i); if (e == std::string::npos) e = s.size();
        std::string kv = s.substr(i, e - i); size_t q = kv.find('=');
        if (q != std::string::npos) m[kv.substr(0, q)] = kv.substr(q + 1);
        i = e + 1;
    }
    return m;
}
extern "C" {
int merhost_write_exr(const char *path, const float *rgb, int32_t h, int32_t w) {
    try { merhost::writeExr(path, rgb, h, w); return 0; } catch (const std::exception &e) { g_host_error = e.what(); return 1; }
}
const char *merhost_last_error(void) { return g_host_error.c_str(); }
int merhost_flatten_xml(const char *path, const char *defines, mer_scene_desc *out, int32_t *spp) {
    try {
        auto scene = merhost::loadScene(path, parseDefines(defines));
        scene->integrator->flatten(*scene, *out);
        if (spp) *spp = scene->sensor->sampler->sampleCount;
        return 0;
    } catch (const std::exception &e) { g_host_error = e.what(); return 1; }
}
int merhost_render_xml_multi(const char *path, const char *defines, const int32_t *devices, int32_t n, int32_t shard_mode, int32_t spp, uint64_t seed, int32_t layout,
                             float *film_host) {
    try {
        auto scene = merhost::loadScene(path, parseDefines(defines));
        std::vector<float> film = scene->integrator->render(*scene, std::vector<int>(devices, devices + (n > 0 ? n : 0)), shard_mode, spp, seed, layout);
        std::memcpy(film_host, film.data(), film.size() * sizeof(float));
        return 0;
    } catch (const std::exception &e) { g_host_error = e.what(); return 1; }
}
int merhost_render_xml(const char *path, const char *defines, int32_t device, int32_t spp, uint64_t seed, int32_t layout, float *film_host) {
    try {
        auto scene = merhost::loadScene(path, parseDefines(defines));
        std::vector<float> film = scene->integrator->render(*scene, device, spp, seed, layout);
        std::memcpy(film_host, film.data(), film.size() * sizeof(float));
        return 0;
    } catch (const std::exception &e) { g_host_error = e.what(); return 1; }
}
}
