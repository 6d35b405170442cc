/*
 * mer_oracle.cpp -- CPU ORACLE: restatement of the reference's refractive volumetric
 * path-tracing hot path (cmu-ci-lab/MitsubaER, mounted at /root/reference at authoring time).
 *
 * TEST INFRASTRUCTURE ONLY (see mer_oracle.h).  Never linked into or called by the product.
 *
 * Every function cites the reference file:line it follows.  Type conventions of the
 * reference build (-DSINGLE_PRECISION -DSPECTRUM_SAMPLES=3): Float = float, Spectrum = 3 floats,
 * FLOAT = float (config_custom_release) or double (-DFLOATDEBUG) for the RIF path.
 * Compile with -ffp-contract=off: fused operations appear only where written as fmaf().
 *
 * Deliberate, documented departures (no reference analogue exists):
 *   - Sampler: counter-based PCG32 stream per (pixel, sample) instead of SFMT (src/samplers are OOS);
 *     float conversion follows src/libcore/random.cpp:630-639.
 *   - Composed estimator "delta tracking along a curved ray" (SURVEY D3, section 9.2).
 *   - Trilinear RIF value + analytic gradient, RK4 stepper.
 *   - Ratio-tracking transmittance (north star) next to the reference's 2-sample Woodcock.
 *   - Medium boundary taken as data (AABB / sphere) instead of the hard-coded sphere.
 */
#include "mer_oracle.h"
#include <cmath>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <algorithm>
#include <vector>
#include <thread>
#include <atomic>
#include <string>

namespace {

typedef float Float;
static const Float Epsilon = 1e-4f;            /* include/mitsuba/core/constants.h:25-31 */
static const Float M_PI_F  = 3.14159265358979323846f;
static const Float INV_FOURPI_F = 0.07957747154594766788f;
static const Float INV_PI_F = 0.31830988618379067154f;      /* constants.h INV_PI (single precision build) */

thread_local std::string g_err;

/* ------------------------------------------------------------------ small vector types */
template <typename T> struct V3 {
    T x, y, z;
    V3() {}
    V3(T a, T b, T c) : x(a), y(b), z(c) {}
    template <typename U> explicit V3(const V3<U> &o) : x((T) o.x), y((T) o.y), z((T) o.z) {}
    V3 operator+(const V3 &o) const { return V3(x + o.x, y + o.y, z + o.z); }
    V3 operator-(const V3 &o) const { return V3(x - o.x, y - o.y, z - o.z); }
    V3 operator*(T s) const { return V3(x * s, y * s, z * s); }
    V3 operator/(T s) const { T recip = (T) 1 / s; return V3(x * recip, y * recip, z * recip); }   /* vector.h:548-557 */
    V3 operator-() const { return V3(-x, -y, -z); }
    V3 &operator+=(const V3 &o) { x += o.x; y += o.y; z += o.z; return *this; }
    V3 &operator*=(T s) { x *= s; y *= s; z *= s; return *this; }
};
template <typename T> inline V3<T> operator*(T s, const V3<T> &v) { return V3<T>(s * v.x, s * v.y, s * v.z); }
template <typename T> inline T dot(const V3<T> &a, const V3<T> &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename T> inline V3<T> cross(const V3<T> &a, const V3<T> &b) {
    return V3<T>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
template <typename T> inline V3<T> normalize(const V3<T> &a) { return a / std::sqrt(dot(a, a)); }
typedef V3<Float> Vec;

struct Spec {
    Float c[3];
    Spec() {}
    explicit Spec(Float v) { c[0] = c[1] = c[2] = v; }
    Spec(Float r, Float g, Float b) { c[0] = r; c[1] = g; c[2] = b; }
    Float &operator[](int i) { return c[i]; }
    Float operator[](int i) const { return c[i]; }
    Spec operator*(const Spec &o) const { return Spec(c[0] * o.c[0], c[1] * o.c[1], c[2] * o.c[2]); }
    Spec operator*(Float s) const { return Spec(c[0] * s, c[1] * s, c[2] * s); }
    Spec operator/(Float s) const { Float recip = 1.0f / s; return Spec(c[0] * recip, c[1] * recip, c[2] * recip); }  /* spectrum.h:415-425 */
    Spec operator+(const Spec &o) const { return Spec(c[0] + o.c[0], c[1] + o.c[1], c[2] + o.c[2]); }
    Spec operator-(const Spec &o) const { return Spec(c[0] - o.c[0], c[1] - o.c[1], c[2] - o.c[2]); }
    Spec &operator*=(const Spec &o) { c[0] *= o.c[0]; c[1] *= o.c[1]; c[2] *= o.c[2]; return *this; }
    Spec &operator*=(Float s) { c[0] *= s; c[1] *= s; c[2] *= s; return *this; }
    Spec &operator/=(Float s) { Float recip = 1.0f / s; c[0] *= recip; c[1] *= recip; c[2] *= recip; return *this; }
    Spec &operator+=(const Spec &o) { c[0] += o.c[0]; c[1] += o.c[1]; c[2] += o.c[2]; return *this; }
    bool isZero() const { return c[0] == 0 && c[1] == 0 && c[2] == 0; }
    Float max() const { return std::max(c[0], std::max(c[1], c[2])); }
};

/* ------------------------------------------------------------------ sampler (departure: PCG32) */
struct Pcg32 {
    uint64_t state, inc;
    static uint64_t splitmix64(uint64_t x) {
        x += 0x9E3779B97F4A7C15ULL;
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
        x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
        return x ^ (x >> 31);
    }
    void seed(uint64_t seedv, uint32_t pixel, uint32_t sample) {
        uint64_t initseq = ((uint64_t) sample << 32) | (uint64_t) pixel;
        state = 0; inc = (initseq << 1) | 1ULL;
        next();
        state += splitmix64(seedv);
        next();
    }
    /* The stream of a SIDE WALK (the transmittance walk of a luminaire sample, kind 1, or of an emitter look-up, kind 2): a child of the path's
       stream at the point where the walk starts.  The path's own stream does not advance while the walk runs, so the walk is an independent piece of
       work -- the HIP path runs it in a lane of its own while the path goes on (csrc/mer_wavefront.hpp: spawned walks). */
    Pcg32 fork(uint64_t kind) const { Pcg32 c; c.state = splitmix64(state ^ (kind * 0xD1B54A32D192ED03ULL)); c.inc = inc; return c; }
    uint32_t next() {
        uint64_t old = state;
        state = old * 6364136223846793005ULL + inc;
        uint32_t xorshifted = (uint32_t) (((old >> 18u) ^ old) >> 27u);
        uint32_t rot = (uint32_t) (old >> 59u);
        return (xorshifted >> rot) | (xorshifted << ((32u - rot) & 31u));
    }
    /* src/libcore/random.cpp:630-639: 23 mantissa bits in [1,2) minus 1 */
    Float next1D() {
        union { uint32_t u; float f; } x;
        x.u = (next() >> 9) | 0x3f800000u;
        return x.f - 1.0f;
    }
};

/* ------------------------------------------------------------------ A1/A2 grid volume */
struct Grid {
    int res[3], channels, dtype;
    const void *data;
    Float bmin[3], bmax[3];
    Float m[3][4];          /* worldToGrid, gridvolume.cpp:188-195 */
    Float w2v[3][4];        /* worldToVolume = toWorld^-1 */
    Float s[3], t[3];       /* volumeToGrid: scale((res-1)/extents) * translate(-min) */
    Float wmin[3], wmax[3]; /* m_aabb: world-space box of the transformed data box, gridvolume.cpp:199-203 */
    bool affine;
    Float stepSize;         /* gridvolume.cpp:196-198 */
    Float densityMap[256];  /* gridvolume.cpp:204-214 */
    bool valid;

    Grid() : data(NULL), valid(false) {}
    void configure(const orc_grid &g) {
        valid = g.data != NULL;
        for (int i = 0; i < 3; ++i) { res[i] = g.res[i]; bmin[i] = g.aabb_min[i]; bmax[i] = g.aabb_max[i]; }
        channels = g.channels; dtype = g.dtype; data = g.data;
        std::memset(m, 0, sizeof(m));
        bool zero = true; affine = false;
        for (int i = 0; i < 12; i++) zero = zero && g.world_to_volume[i] == 0.0f;
        for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) {
            w2v[i][j] = zero ? (i == j ? 1.0f : 0.0f) : g.world_to_volume[i * 4 + j];
            if (w2v[i][j] != (i == j ? 1.0f : 0.0f)) affine = true;
        }
        stepSize = std::numeric_limits<Float>::infinity();
        for (int i = 0; i < 3; ++i) {
            Float extent = bmax[i] - bmin[i];
            Float s = (Float) (res[i] - 1) / extent;       /* Transform::scale((res-1)/extents) */
            /* (scale * translate(-min)) * worldToVolume: the 4x4 products of src/libcore/transform.cpp; zero terms add nothing */
            this->s[i] = s; t[i] = s * (-bmin[i]);
            for (int j = 0; j < 3; j++) m[i][j] = s * w2v[i][j];
            m[i][3] = s * w2v[i][3] + t[i];
            stepSize = std::min(stepSize, 0.5f * extent / (Float) (res[i] - 1));
        }
        {   /* m_aabb (gridvolume.cpp:199-203): corners of the data box under volumeToWorld = W^-1 (cofactors, double) */
            const double a = w2v[0][0], b = w2v[0][1], c = w2v[0][2], d = w2v[1][0], e = w2v[1][1], f = w2v[1][2], gg = w2v[2][0], h = w2v[2][1], k = w2v[2][2];
            const double det = a * (e * k - f * h) - b * (d * k - f * gg) + c * (d * h - e * gg);
            const double inv[9] = {(e * k - f * h) / det, (c * h - b * k) / det, (b * f - c * e) / det,
                                   (f * gg - d * k) / det, (a * k - c * gg) / det, (c * d - a * f) / det,
                                   (d * h - e * gg) / det, (b * gg - a * h) / det, (a * e - b * d) / det};
            for (int i = 0; i < 3; i++) { wmin[i] = std::numeric_limits<Float>::infinity(); wmax[i] = -std::numeric_limits<Float>::infinity(); }
            for (int corner = 0; corner < 8; corner++) {
                const double q[3] = {((corner & 1) ? bmax[0] : bmin[0]) - (double) w2v[0][3], ((corner & 2) ? bmax[1] : bmin[1]) - (double) w2v[1][3],
                                     ((corner & 4) ? bmax[2] : bmin[2]) - (double) w2v[2][3]};
                for (int i = 0; i < 3; i++) {
                    const Float w = (Float) (inv[i * 3] * q[0] + inv[i * 3 + 1] * q[1] + inv[i * 3 + 2] * q[2]);
                    wmin[i] = std::min(wmin[i], w); wmax[i] = std::max(wmax[i], w);
                }
            }
        }
        for (int i = 0; i < 255; ++i) densityMap[i] = i / 255.0f;
        densityMap[255] = 1.0f;
    }
    /* include/mitsuba/core/transform.h:147-155 transformAffine, operation order kept */
    inline Vec toGrid(const Vec &p) const {
        Float x = m[0][0] * p.x + m[0][1] * p.y + m[0][2] * p.z + m[0][3];
        Float y = m[1][0] * p.x + m[1][1] * p.y + m[1][2] * p.z + m[1][3];
        Float z = m[2][0] * p.x + m[2][1] * p.y + m[2][2] * p.z + m[2][3];
        return Vec(x, y, z);
    }
    inline Float fetch(int idx) const {
        return dtype == ORC_VOL_F32 ? ((const float *) data)[idx] : densityMap[((const uint8_t *) data)[idx]];
    }
    /* gridvolume.cpp:337-388 lookupFloat.  idx4 (optional) = x1,y1,z1,linear index / -1 */
    inline Float lookupFloat(const Vec &_p, int *idx4 = NULL) const {
        const Vec p = toGrid(_p);
        const int x1 = (int) std::floor(p.x), y1 = (int) std::floor(p.y), z1 = (int) std::floor(p.z),
                  x2 = x1 + 1, y2 = y1 + 1, z2 = z1 + 1;
        if (idx4) { idx4[0] = x1; idx4[1] = y1; idx4[2] = z1; idx4[3] = -1; }
        if (x1 < 0 || y1 < 0 || z1 < 0 || x2 >= res[0] || y2 >= res[1] || z2 >= res[2])
            return 0;
        const Float fx = p.x - x1, fy = p.y - y1, fz = p.z - z1,
                    _fx = 1.0f - fx, _fy = 1.0f - fy, _fz = 1.0f - fz;
        if (idx4) idx4[3] = (z1 * res[1] + y1) * res[0] + x1;
        const Float d000 = fetch((z1 * res[1] + y1) * res[0] + x1), d001 = fetch((z1 * res[1] + y1) * res[0] + x2),
                    d010 = fetch((z1 * res[1] + y2) * res[0] + x1), d011 = fetch((z1 * res[1] + y2) * res[0] + x2),
                    d100 = fetch((z2 * res[1] + y1) * res[0] + x1), d101 = fetch((z2 * res[1] + y1) * res[0] + x2),
                    d110 = fetch((z2 * res[1] + y2) * res[0] + x1), d111 = fetch((z2 * res[1] + y2) * res[0] + x2);
        return ((d000 * _fx + d001 * fx) * _fy + (d010 * _fx + d011 * fx) * fy) * _fz +
               ((d100 * _fx + d101 * fx) * _fy + (d110 * _fx + d111 * fx) * fy) * fz;
    }
    /* gridvolume.cpp:390-421 lookupSpectrum (float32, 3 channels) */
    inline Spec lookupSpectrum(const Vec &_p) const {
        const Vec p = toGrid(_p);
        const int x1 = (int) std::floor(p.x), y1 = (int) std::floor(p.y), z1 = (int) std::floor(p.z),
                  x2 = x1 + 1, y2 = y1 + 1, z2 = z1 + 1;
        if (x1 < 0 || y1 < 0 || z1 < 0 || x2 >= res[0] || y2 >= res[1] || z2 >= res[2])
            return Spec(0.0f);
        const Float fx = p.x - x1, fy = p.y - y1, fz = p.z - z1,
                    _fx = 1.0f - fx, _fy = 1.0f - fy, _fz = 1.0f - fz;
        Spec out;
        for (int c = 0; c < 3; ++c) {
            auto F = [&](int z, int y, int x) -> Float {
                int idx = ((z * res[1] + y) * res[0] + x) * 3 + c;
                return dtype == ORC_VOL_F32 ? ((const float *) data)[idx] : densityMap[((const uint8_t *) data)[idx]];
            };
            out[c] = ((F(z1, y1, x1) * _fx + F(z1, y1, x2) * fx) * _fy + (F(z1, y2, x1) * _fx + F(z1, y2, x2) * fx) * fy) * _fz +
                     ((F(z2, y1, x1) * _fx + F(z2, y1, x2) * fx) * _fy + (F(z2, y2, x1) * _fx + F(z2, y2, x2) * fx) * fy) * fz;
        }
        return out;
    }
    /* include/mitsuba/core/aabb.h:308-339 */
    inline bool rayIntersect(const Vec &o, const Vec &d, Float &nearT, Float &farT) const {
        return aabbIntersect(wmin, wmax, o, d, nearT, farT);          /* m_aabb: world space */
    }
    /* splinevolume.cpp:320-376: p = m_worldToVolume(pc); gradients come back through m_worldToVolume_RotT */
    template <typename FLOAT> inline V3<FLOAT> toVolume(const V3<FLOAT> &p) const {
        return V3<FLOAT>((FLOAT) w2v[0][0] * p.x + (FLOAT) w2v[0][1] * p.y + (FLOAT) w2v[0][2] * p.z + (FLOAT) w2v[0][3],
                         (FLOAT) w2v[1][0] * p.x + (FLOAT) w2v[1][1] * p.y + (FLOAT) w2v[1][2] * p.z + (FLOAT) w2v[1][3],
                         (FLOAT) w2v[2][0] * p.x + (FLOAT) w2v[2][1] * p.y + (FLOAT) w2v[2][2] * p.z + (FLOAT) w2v[2][3]);
    }
    template <typename FLOAT> inline V3<FLOAT> rotT(const V3<FLOAT> &v) const {
        return V3<FLOAT>((FLOAT) w2v[0][0] * v.x + (FLOAT) w2v[1][0] * v.y + (FLOAT) w2v[2][0] * v.z, (FLOAT) w2v[0][1] * v.x + (FLOAT) w2v[1][1] * v.y + (FLOAT) w2v[2][1] * v.z,
                         (FLOAT) w2v[0][2] * v.x + (FLOAT) w2v[1][2] * v.y + (FLOAT) w2v[2][2] * v.z);
    }
    static inline bool aabbIntersect(const Float mn[3], const Float mx[3], const Vec &o, const Vec &d, Float &nearT, Float &farT) {
        nearT = -std::numeric_limits<Float>::infinity();
        farT = std::numeric_limits<Float>::infinity();
        const Float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
        for (int i = 0; i < 3; i++) {
            const Float origin = oo[i], minVal = mn[i], maxVal = mx[i];
            if (dd[i] == 0) {
                if (origin < minVal || origin > maxVal) return false;
            } else {
                const Float dRcp = 1.0f / dd[i];          /* ray.dRcp, include/mitsuba/core/ray.h */
                Float t1 = (minVal - origin) * dRcp, t2 = (maxVal - origin) * dRcp;
                if (t1 > t2) std::swap(t1, t2);
                nearT = std::max(t1, nearT);
                farT = std::min(t2, farT);
                if (!(nearT <= farT)) return false;
            }
        }
        return true;
    }
};

/* trilinear value + analytic gradient of the interpolant, cell clamped to the grid (new: SURVEY D2 --
   gridvolume has no value()/gradient()).  Being new functionality its arithmetic is a definition, not a
   restatement: the cell's interpolant in monomial form about its base corner,
   f = a0 + ax x + ay y + az z + axy xy + axz xz + ayz yz + axyz xyz, coefficients from the corners in the operation order below,
   evaluated by fused Horner steps -- the same expression tree the HIP kernel uses (CellCache::set, trilinear_value_grad). */
template <typename FLOAT> struct TriCoeff {
    FLOAT a0, ax, ay, az, axy, axz, ayz, axyz;
    TriCoeff(const float *D, int base, int sy, int sz) {
        const FLOAT d000 = D[base], d001 = D[base + 1], d010 = D[base + sy], d011 = D[base + sy + 1],
                    d100 = D[base + sz], d101 = D[base + sz + 1], d110 = D[base + sz + sy], d111 = D[base + sz + sy + 1];
        a0 = d000; ax = d001 - d000; ay = d010 - d000; az = d100 - d000;
        const FLOAT x1 = d011 - d010, x2 = d101 - d100, x3 = d111 - d110;
        axy = x1 - ax; axz = x2 - ax; ayz = (d110 - d100) - ay;
        axyz = (x3 - x2) - axy;
    }
};
template <typename FLOAT>
inline void trilinearValueGrad(const Grid &g, const V3<FLOAT> &pw, FLOAT &val, V3<FLOAT> &grad) {
    /* pw: the point in VOLUME space (Rif applies worldToVolume first when the grid has a toWorld) */
    const FLOAT px = std::fma((FLOAT) g.s[0], pw.x, (FLOAT) g.t[0]);
    const FLOAT py = std::fma((FLOAT) g.s[1], pw.y, (FLOAT) g.t[1]);
    const FLOAT pz = std::fma((FLOAT) g.s[2], pw.z, (FLOAT) g.t[2]);
    int x1 = (int) std::floor(px), y1 = (int) std::floor(py), z1 = (int) std::floor(pz);
    x1 = std::min(std::max(x1, 0), g.res[0] - 2);
    y1 = std::min(std::max(y1, 0), g.res[1] - 2);
    z1 = std::min(std::max(z1, 0), g.res[2] - 2);
    const FLOAT fx = px - x1, fy = py - y1, fz = pz - z1;
    const float *D = (const float *) g.data;
    const int base = (z1 * g.res[1] + y1) * g.res[0] + x1, sy = g.res[0], sz = g.res[0] * g.res[1];
    const TriCoeff<FLOAT> c(D, base, sy, sz);
    const FLOAT A = std::fma(c.axyz, fz, c.axy), B = std::fma(c.axz, fz, c.ax), C = std::fma(c.ayz, fz, c.ay), Dz = std::fma(c.az, fz, c.a0);
    const FLOAT gx = std::fma(A, fy, B);
    val = std::fma(gx, fx, std::fma(C, fy, Dz));
    const FLOAT gy = std::fma(A, fx, C);
    const FLOAT gz = std::fma(std::fma(c.axyz, fx, c.ayz), fy, std::fma(c.axz, fx, c.az));
    grad = V3<FLOAT>(gx * (FLOAT) g.s[0], gy * (FLOAT) g.s[1], gz * (FLOAT) g.s[2]);
}

/* ------------------------------------------------------------------ A5 cubic B-spline (basisspline.h) */
template <typename FLOAT> struct SplineConst;
template <> struct SplineConst<float> {   /* constants.h:93-101 non-FLOATDEBUG: *_FLT literals */
    static float half() { return 0.5f; } static float sixth() { return 0.16666666667f; } static float twothird() { return 0.66666666667f; }
};
template <> struct SplineConst<double> {  /* FLOATDEBUG: 1.0/2.0, 1.0/6.0, 2.0/3.0 */
    static double half() { return 1.0 / 2.0; } static double sixth() { return 1.0 / 6.0; } static double twothird() { return 2.0 / 3.0; }
};
template <typename T> inline int sgn(T val) { return (T(0) < val) - (val < T(0)); }

/* basisspline.h:40-46 */
template <typename FLOAT> inline FLOAT kernel0(FLOAT x) {
    x = std::abs(x);
    if (x > 2) return (FLOAT) 0.0;
    if (x > 1) return (SplineConst<FLOAT>::sixth() * (2 - x) * (2 - x) * (2 - x));
    return (SplineConst<FLOAT>::twothird() - x * x + SplineConst<FLOAT>::half() * x * x * x);
}
/* basisspline.h:65-72 (note the double literal 1.5, as in the reference) */
template <typename FLOAT> inline FLOAT kernel1(FLOAT x) {
    int s = sgn(x);
    x = std::abs(x);
    if (x > 2) return (FLOAT) 0.0;
    if (x > 1) return s * (-SplineConst<FLOAT>::half() * (2 - x) * (2 - x));
    return (FLOAT) (s * ((1.5 * x - 2) * x));
}
/* basisspline.h:91-97 */
template <typename FLOAT> inline FLOAT kernel2(FLOAT x) {
    x = std::abs(x);
    if (x > 2) return (FLOAT) 0.0;
    if (x > 1) return 2 - x;
    return 3 * x - 2;
}

template <typename FLOAT> struct Spline3 {
    FLOAT xmin[3], xmax[3], xres[3], dxres[3], dxres2[3];
    std::vector<FLOAT> own;
    const FLOAT *coeff;
    int N[3];
    FLOAT z1;
    Spline3() : coeff(NULL) {}
    /* basisspline.h:124-138 */
    void initialize(const FLOAT mn[3], const FLOAT mx[3], const int n[3]) {
        for (int i = 0; i < 3; i++) {
            xmin[i] = mn[i]; xmax[i] = mx[i]; N[i] = n[i];
            xres[i] = (N[i] - 1) / (xmax[i] - xmin[i]);
            dxres[i] = xres[i];
            dxres2[i] = dxres[i] * dxres[i];
        }
        z1 = -2 + std::sqrt((FLOAT) 3);
    }
    FLOAT getStride(int d) const { return (FLOAT) (1.0 / xres[d]); }   /* basisspline.h:622-624 */
    /* basisspline.h:812-840 */
    void build1d(const FLOAT *data, int offset, int stride, int size, FLOAT *out) const {
        std::vector<FLOAT> cp(size), cn(size);
        cp[0] = 0;
        for (int i = 0; i < size; i++) cp[0] += data[offset + i * stride] * std::pow(z1, i);
        for (int i = size - 2; i > 0; i--) cp[0] += data[offset + i * stride] * std::pow(z1, 2 * size - 2 - i);
        cp[0] /= (1 - std::pow(z1, 2 * size - 2));
        for (int i = 1; i < size; i++) cp[i] = data[offset + i * stride] + z1 * cp[i - 1];
        cn[size - 1] = z1 / (z1 * z1 - 1) * (cp[size - 1] + z1 * cp[size - 2]);
        for (int i = size - 2; i >= 0; i--) cn[i] = z1 * (cn[i + 1] - cp[i]);
        for (int i = 0; i < size; i++) out[i] = 6 * cn[i];
    }
    /* basisspline.h:865-890: along y, then x, then z */
    void build3d(const FLOAT *data, FLOAT *c) const {
        std::vector<FLOAT> temp(std::max(std::max(N[0], N[1]), N[2]));
        for (int k = 0; k < N[2]; k++)
            for (int i = 0; i < N[0]; i++) {
                build1d(data, k * N[0] * N[1] + i, N[0], N[1], temp.data());
                for (int t = 0; t < N[1]; t++) c[i + t * N[0] + k * N[0] * N[1]] = temp[t];
            }
        for (int k = 0; k < N[2]; k++)
            for (int j = 0; j < N[1]; j++) {
                build1d(c, k * N[0] * N[1] + j * N[0], 1, N[0], temp.data());
                for (int t = 0; t < N[0]; t++) c[t + j * N[0] + k * N[0] * N[1]] = temp[t];
            }
        for (int i = 0; i < N[0]; i++)
            for (int j = 0; j < N[1]; j++) {
                build1d(c, j * N[0] + i, N[0] * N[1], N[2], temp.data());
                for (int t = 0; t < N[2]; t++) c[i + j * N[0] + t * N[0] * N[1]] = temp[t];
            }
    }
    void buildFromFloat(const float *data) {   /* splinevolume.cpp:285-289: f32 file data -> FLOAT */
        size_t n = (size_t) N[0] * N[1] * N[2];
        std::vector<FLOAT> tmp(n);
        for (size_t i = 0; i < n; i++) tmp[i] = (FLOAT) data[i];
        own.resize(n);
        build3d(tmp.data(), own.data());
        coeff = own.data();
    }
    inline void convertToX(FLOAT x[3]) const { for (int i = 0; i < 3; i++) x[i] = (x[i] - xmin[i]) * xres[i]; } /* :655-658 */
    /* basisspline.h:302-315 */
    inline FLOAT value(const FLOAT x1[3]) const {
        FLOAT x[3] = {x1[0], x1[1], x1[2]};
        convertToX(x);
        FLOAT v = 0;
        for (int i1 = (int) std::ceil(x[0] - 2); i1 <= std::floor(x[0] + 2); i1++)
            for (int i2 = (int) std::ceil(x[1] - 2); i2 <= std::floor(x[1] + 2); i2++)
                for (int i3 = (int) std::ceil(x[2] - 2); i3 <= std::floor(x[2] + 2); i3++)
                    v += coeff[i1 + i2 * N[0] + i3 * N[0] * N[1]] * kernel0<FLOAT>(x[0] - i1) * kernel0<FLOAT>(x[1] - i2) * kernel0<FLOAT>(x[2] - i3);
        return v;
    }
    /* basisspline.h:318-364 */
    inline V3<FLOAT> gradient(const FLOAT x1[3]) const {
        FLOAT x[3] = {x1[0], x1[1], x1[2]};
        convertToX(x);
        V3<FLOAT> v(0, 0, 0);
        for (int i1 = (int) std::ceil(x[0] - 2); i1 <= std::floor(x[0] + 2); i1++)
            for (int i2 = (int) std::ceil(x[1] - 2); i2 <= std::floor(x[1] + 2); i2++)
                for (int i3 = (int) std::ceil(x[2] - 2); i3 <= std::floor(x[2] + 2); i3++) {
                    FLOAT c = coeff[i1 + i2 * N[0] + i3 * N[0] * N[1]];
                    FLOAT k0x = kernel0<FLOAT>(x[0] - i1), k0y = kernel0<FLOAT>(x[1] - i2), k0z = kernel0<FLOAT>(x[2] - i3);
                    v.x += c * kernel1<FLOAT>(x[0] - i1) * k0y * k0z;
                    v.y += c * k0x * kernel1<FLOAT>(x[1] - i2) * k0z;
                    v.z += c * k0x * k0y * kernel1<FLOAT>(x[2] - i3);
                }
        v.x *= dxres[0]; v.y *= dxres[1]; v.z *= dxres[2];
        return v;
    }
    /* basisspline.h:438-471 */
    inline void valueAndGradient(const FLOAT x1[3], FLOAT &f, V3<FLOAT> &v) const {
        FLOAT x[3] = {x1[0], x1[1], x1[2]};
        v.x = v.y = v.z = 0; f = 0;
        convertToX(x);
        for (int i1 = (int) std::ceil(x[0] - 2); i1 <= std::floor(x[0] + 2); i1++)
            for (int i2 = (int) std::ceil(x[1] - 2); i2 <= std::floor(x[1] + 2); i2++)
                for (int i3 = (int) std::ceil(x[2] - 2); i3 <= std::floor(x[2] + 2); i3++) {
                    FLOAT c = coeff[i1 + i2 * N[0] + i3 * N[0] * N[1]];
                    FLOAT k0x = kernel0<FLOAT>(x[0] - i1), k0y = kernel0<FLOAT>(x[1] - i2), k0z = kernel0<FLOAT>(x[2] - i3);
                    f += c * k0x * k0y * k0z;
                    v.x += c * kernel1<FLOAT>(x[0] - i1) * k0y * k0z;
                    v.y += c * k0x * kernel1<FLOAT>(x[1] - i2) * k0z;
                    v.z += c * k0x * k0y * kernel1<FLOAT>(x[2] - i3);
                }
        v.x *= dxres[0]; v.y *= dxres[1]; v.z *= dxres[2];
    }
    /* basisspline.h:539-606 valueGradientAndHessian; H row-major (Hxx,Hxy,Hzx; Hxy,Hyy,Hyz; Hzx,Hyz,Hzz) */
    inline void valueGradientAndHessian(const FLOAT x1[3], FLOAT &f, V3<FLOAT> &v, FLOAT H[9]) const {
        FLOAT x[3] = {x1[0], x1[1], x1[2]};
        v.x = v.y = v.z = 0; f = 0;
        FLOAT Hxx = 0, Hyy = 0, Hzz = 0, Hxy = 0, Hyz = 0, Hzx = 0;
        convertToX(x);
        for (int i1 = (int) std::ceil(x[0] - 2); i1 <= std::floor(x[0] + 2); i1++)
            for (int i2 = (int) std::ceil(x[1] - 2); i2 <= std::floor(x[1] + 2); i2++)
                for (int i3 = (int) std::ceil(x[2] - 2); i3 <= std::floor(x[2] + 2); i3++) {
                    FLOAT c = coeff[i1 + i2 * N[0] + i3 * N[0] * N[1]];
                    FLOAT k0x = kernel0<FLOAT>(x[0] - i1), k0y = kernel0<FLOAT>(x[1] - i2), k0z = kernel0<FLOAT>(x[2] - i3);
                    FLOAT k1x = kernel1<FLOAT>(x[0] - i1), k1y = kernel1<FLOAT>(x[1] - i2), k1z = kernel1<FLOAT>(x[2] - i3);
                    f += c * k0x * k0y * k0z;
                    v.x += c * k1x * k0y * k0z;
                    v.y += c * k0x * k1y * k0z;
                    v.z += c * k0x * k0y * k1z;
                    Hxx += c * kernel2<FLOAT>(x[0] - i1) * k0y * k0z;
                    Hyy += c * k0x * kernel2<FLOAT>(x[1] - i2) * k0z;
                    Hzz += c * k0x * k0y * kernel2<FLOAT>(x[2] - i3);
                    Hxy += c * k1x * k1y * k0z;
                    Hyz += c * k0x * k1y * k1z;
                    Hzx += c * k1x * k0y * k1z;
                }
        v.x *= dxres[0]; v.y *= dxres[1]; v.z *= dxres[2];
        Hxx *= dxres2[0]; Hyy *= dxres2[1]; Hzz *= dxres2[2];
        Hxy *= dxres[0] * dxres[1]; Hyz *= dxres[1] * dxres[2]; Hzx *= dxres[2] * dxres[0];
        H[0] = Hxx; H[1] = Hxy; H[2] = Hzx; H[3] = Hxy; H[4] = Hyy; H[5] = Hyz; H[6] = Hzx; H[7] = Hyz; H[8] = Hzz;
    }
};

/* ------------------------------------------------------------------ counters */
struct Counters {
    uint64_t c[ORC_C_COUNT];
    Counters() { std::memset(c, 0, sizeof(c)); }
};

/* ------------------------------------------------------------------ scene */
struct MediumRec {   /* include/mitsuba/render/medium.h:40-98 (fields used on this path) */
    Float t; Vec p; Vec d; Spec sigmaA, sigmaS, transmittance;
    Float pdfSuccess, pdfFailure, refRatioSq, opticalLength;
};

/* AcousticRIFVolume (src/volume/acousticrifvolume.cpp:15,224-342): the ultrasound-modulated index in the (y, z) plane, evaluated in
   FLOAT with the C library's Bessel functions (jnf for float, as the GPU's device library; jn for double) */
template <typename FLOAT> struct AcousticRif {
    FLOAT n_o, n_max, k_r; int mode;
    static inline float bessel(int n, float x) { return ::jnf(n, x); }
    static inline double bessel(int n, double x) { return ::jn(n, x); }
    inline void polar(const V3<FLOAT> &pc, FLOAT &py, FLOAT &pz, FLOAT &r, FLOAT &phi) const {
        const FLOAT eps = (FLOAT) 1e-8f;                                     /* EpsilonRIF (:15) */
        py = pc.y; pz = pc.z;
        r = std::sqrt(py * py + pz * pz);
        phi = std::atan2(py, pz);
        if (r < eps) { py = eps; pz = eps; r = eps; }                        /* :235-239 */
    }
    /* valueAndGradient (:312-342) */
    inline void valueAndGradient(const V3<FLOAT> &pc, FLOAT &f, V3<FLOAT> &v) const {
        FLOAT py, pz, r, phi; polar(pc, py, pz, r, phi);
        const FLOAT krr = k_r * r, m = (FLOAT) mode;
        const FLOAT bj = bessel(mode, krr), dbj = m / krr * bj - bessel(mode + 1, krr);
        const FLOAT invr = (FLOAT) 1 / r, invr2 = invr * invr;
        const FLOAT cosmp = std::cos(m * phi), sinmp = std::sin(m * phi);
        f = n_o + n_max * bj * cosmp;
        v = V3<FLOAT>(0, n_max * (dbj * k_r * py * invr * cosmp - bj * m * sinmp * pz * invr2),
                      n_max * (dbj * k_r * pz * invr * cosmp + bj * m * sinmp * py * invr2));
    }
    /* hessian (:254-308); the reference's "x" is the z axis here (cos phi = z / r), its "y" the y axis; H row-major over (x, y, z) */
    inline void hessian(const V3<FLOAT> &pc, FLOAT H[9]) const {
        FLOAT py, pz, r, phi; polar(pc, py, pz, r, phi);
        const FLOAT krr = k_r * r, m = (FLOAT) mode;
        const FLOAT j0 = bessel(mode, krr), j1 = bessel(mode + 1, krr), j2 = bessel(mode + 2, krr);
        const FLOAT d0 = m / krr * j0 - j1, d1 = (m + 1) / krr * j1 - j2;
        const FLOAT invr = (FLOAT) 1 / r, invr2 = invr * invr;
        const FLOAT cosp = std::cos(phi), sinp = std::sin(phi), cosmp = std::cos(m * phi), sinmp = std::sin(m * phi);
        const FLOAT cosm1p = std::cos((m - 1) * phi), sinm1p = std::sin((m - 1) * phi), cosm2p = std::cos((m - 2) * phi), sinm2p = std::sin((m - 2) * phi);
        const FLOAT Hxx = n_max * (j0 * m * invr2 * (-cosm2p + m * sinm1p * sinp) + d0 * m * invr * k_r * cosm1p * cosp
                                   - j1 * k_r * py * invr2 * (sinp * cosmp + m * cosp * sinmp) - d1 * k_r * k_r * cosp * cosp * cosmp);
        const FLOAT Hxy = n_max * (-j0 * m * invr2 * (-sinm2p + m * sinm1p * cosp) + d0 * m * invr * k_r * cosm1p * sinp
                                   + j1 * k_r * pz * invr2 * (sinp * cosmp + m * cosp * sinmp) - d1 * k_r * k_r * cosp * sinp * cosmp);
        const FLOAT Hyy = -n_max * (j0 * m * invr2 * (-cosm2p + m * cosm1p * cosp) + d0 * m * invr * k_r * sinm1p * sinp
                                    + j1 * k_r * pz * invr2 * (cosp * cosmp - m * sinp * sinmp) + d1 * k_r * k_r * sinp * sinp * cosmp);
        for (int i = 0; i < 9; i++) H[i] = 0;
        H[4] = Hyy; H[5] = H[7] = Hxy; H[8] = Hxx;
    }
};

template <typename FLOAT> struct Rif {
    int mode; FLOAT cst;
    AcousticRif<FLOAT> ac;
    Grid grid;
    Spline3<FLOAT> spline;
    FLOAT limMin[3], limMax[3];   /* splinevolume.cpp:280-281 interpolatable limits */
    void configure(const orc_scene &s) {
        mode = s.rif_mode; cst = (FLOAT) s.rif_const;
        if (mode == ORC_RIF_ACOUSTIC) { ac.n_o = (FLOAT) s.ac_n_o; ac.n_max = (FLOAT) s.ac_n_max; ac.k_r = (FLOAT) s.ac_k_r; ac.mode = s.ac_mode; }
        else if (mode != ORC_RIF_CONST) {
            grid.configure(s.rif);
            if (mode == ORC_RIF_BSPLINE3) {
                FLOAT mn[3], mx[3];
                for (int i = 0; i < 3; i++) { mn[i] = s.rif.aabb_min[i]; mx[i] = s.rif.aabb_max[i]; }
                spline.initialize(mn, mx, s.rif.res);
                spline.buildFromFloat((const float *) s.rif.data);
                for (int i = 0; i < 3; i++) {
                    limMin[i] = mn[i] + ((FLOAT) 2.0 * spline.getStride(i) + (FLOAT) Epsilon);
                    limMax[i] = mx[i] + ((FLOAT) -2.0 * spline.getStride(i) - (FLOAT) Epsilon);
                }
            }
        }
    }
    /* splinevolume.cpp:319-324 */
    inline bool insideVolumeLimits(const V3<FLOAT> &pw) const {
        if (mode != ORC_RIF_BSPLINE3) return true;
        const V3<FLOAT> p = grid.affine ? grid.toVolume(pw) : pw;
        return p.x > limMin[0] && p.x < limMax[0] && p.y > limMin[1] && p.y < limMax[1] && p.z > limMin[2] && p.z < limMax[2];
    }
    inline FLOAT value(const V3<FLOAT> &pw, Counters &C) const {
        C.c[ORC_C_RIF_EVALS]++;
        if (mode == ORC_RIF_CONST) return cst;
        if (mode == ORC_RIF_ACOUSTIC) { FLOAT v; V3<FLOAT> g; ac.valueAndGradient(pw, v, g); return v; }
        const V3<FLOAT> p = grid.affine ? grid.toVolume(pw) : pw;
        if (mode == ORC_RIF_TRILINEAR) { FLOAT v; V3<FLOAT> g; trilinearValueGrad<FLOAT>(grid, p, v, g); return v; }
        FLOAT x[3] = {p.x, p.y, p.z};
        return spline.value(x);              /* splinevolume.cpp:330-337 */
    }
    inline void valueAndGradient(const V3<FLOAT> &pw, FLOAT &n, V3<FLOAT> &g, Counters &C) const {
        C.c[ORC_C_RIF_EVALS]++;
        if (mode == ORC_RIF_CONST) { n = cst; g = V3<FLOAT>(0, 0, 0); return; }
        if (mode == ORC_RIF_ACOUSTIC) { ac.valueAndGradient(pw, n, g); return; }
        const V3<FLOAT> p = grid.affine ? grid.toVolume(pw) : pw;
        if (mode == ORC_RIF_TRILINEAR) trilinearValueGrad<FLOAT>(grid, p, n, g);
        else { FLOAT x[3] = {p.x, p.y, p.z}; spline.valueAndGradient(x, n, g); }    /* splinevolume.cpp:354-360 */
        if (grid.affine) g = grid.rotT(g);                                            /* v = m_worldToVolume_RotT * v (:359) */
    }
    inline V3<FLOAT> gradient(const V3<FLOAT> &pw, Counters &C) const {
        C.c[ORC_C_RIF_EVALS]++;
        if (mode == ORC_RIF_CONST) return V3<FLOAT>(0, 0, 0);
        if (mode == ORC_RIF_ACOUSTIC) { FLOAT v; V3<FLOAT> g; ac.valueAndGradient(pw, v, g); return g; }
        const V3<FLOAT> p = grid.affine ? grid.toVolume(pw) : pw;
        V3<FLOAT> g;
        if (mode == ORC_RIF_TRILINEAR) { FLOAT v; trilinearValueGrad<FLOAT>(grid, p, v, g); }
        else { FLOAT x[3] = {p.x, p.y, p.z}; g = spline.gradient(x); }               /* splinevolume.cpp:338-344 */
        return grid.affine ? grid.rotT(g) : g;
    }
    /* splinevolume.cpp:370-376 valueGradientAndHessian; H row-major.  Trilinear: Hessian of the interpolant (only the
       mixed terms are non-zero inside a cell) -- new, SURVEY D2. */
    inline void valueGradientAndHessian(const V3<FLOAT> &pw, FLOAT &n, V3<FLOAT> &g, FLOAT H[9], Counters &C) const {
        if (mode == ORC_RIF_CONST || mode == ORC_RIF_ACOUSTIC || !grid.affine) { valueGradientAndHessianVol(pw, n, g, H, C); return; }
        valueGradientAndHessianVol(grid.toVolume(pw), n, g, H, C);
        g = grid.rotT(g);
        /* M = m_worldToVolume_RotT * M * m_worldToVolume_Rot (splinevolume.cpp:367,375) */
        FLOAT Rm[3][3], T1[3][3];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rm[i][j] = (FLOAT) grid.w2v[i][j];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { FLOAT a = 0; for (int k = 0; k < 3; k++) a += Rm[k][i] * H[k * 3 + j]; T1[i][j] = a; }
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { FLOAT a = 0; for (int k = 0; k < 3; k++) a += T1[i][k] * Rm[k][j]; H[i * 3 + j] = a; }
    }
    /* in volume space */
    inline void valueGradientAndHessianVol(const V3<FLOAT> &p, FLOAT &n, V3<FLOAT> &g, FLOAT H[9], Counters &C) const {
        C.c[ORC_C_RIF_EVALS]++;
        for (int i = 0; i < 9; i++) H[i] = 0;
        if (mode == ORC_RIF_CONST) { n = cst; g = V3<FLOAT>(0, 0, 0); return; }
        if (mode == ORC_RIF_ACOUSTIC) { ac.valueAndGradient(p, n, g); ac.hessian(p, H); return; }
        if (mode == ORC_RIF_BSPLINE3) { FLOAT x[3] = {p.x, p.y, p.z}; spline.valueGradientAndHessian(x, n, g, H); return; }
        trilinearValueGrad<FLOAT>(grid, p, n, g);
        const Grid &G = grid;
        const FLOAT px = std::fma((FLOAT) G.s[0], p.x, (FLOAT) G.t[0]), py = std::fma((FLOAT) G.s[1], p.y, (FLOAT) G.t[1]),
                    pz = std::fma((FLOAT) G.s[2], p.z, (FLOAT) G.t[2]);
        int x1 = std::min(std::max((int) std::floor(px), 0), G.res[0] - 2), y1 = std::min(std::max((int) std::floor(py), 0), G.res[1] - 2),
            z1 = std::min(std::max((int) std::floor(pz), 0), G.res[2] - 2);
        const FLOAT fx = px - x1, fy = py - y1, fz = pz - z1;
        const float *D = (const float *) G.data;
        const int base = (z1 * G.res[1] + y1) * G.res[0] + x1, sy = G.res[0], sz = G.res[0] * G.res[1];
        const TriCoeff<FLOAT> c(D, base, sy, sz);
        const FLOAT sx = G.s[0], syy = G.s[1], szz = G.s[2];
        const FLOAT hxy = std::fma(c.axyz, fz, c.axy) * sx * syy;          /* the mixed second derivatives of the monomial form */
        const FLOAT hyz = std::fma(c.axyz, fx, c.ayz) * syy * szz;
        const FLOAT hzx = std::fma(c.axyz, fy, c.axz) * szz * sx;
        H[1] = H[3] = hxy; H[5] = H[7] = hyz; H[2] = H[6] = hzx;
    }
};

/* MaxExpDist (src/medium/maxexp.h:28-98): the distribution proportional to max_i sigma_i exp(-sigma_i t) over the channels,
   restated for SPECTRUM_SAMPLES = 3 */
struct MaxExpDist {
    Float sigmaT[3], cdfv[4], intervalStart[3], normalization, invNormalization;
    bool configure(const Float *s) {                                   /* :30-58 */
        for (int i = 0; i < 3; i++) sigmaT[i] = s[i];
        std::sort(sigmaT, sigmaT + 3, std::greater<Float>());
        cdfv[0] = 0;
        for (int i = 0; i < 3; ++i) {
            if (i > 0 && sigmaT[i] == sigmaT[i - 1]) return false;     /* "Internal error: sigmaT must vary across channels" */
            Float lower = (i == 0) ? -1 : -std::pow((sigmaT[i] / sigmaT[i - 1]), -sigmaT[i] / (sigmaT[i] - sigmaT[i - 1]));
            Float upper = (i == 2) ? 0 : -std::pow((sigmaT[i + 1] / sigmaT[i]), -sigmaT[i] / (sigmaT[i + 1] - sigmaT[i]));
            cdfv[i + 1] = cdfv[i] + (upper - lower);
            intervalStart[i] = (i == 0) ? 0 : std::log(sigmaT[i] / sigmaT[i - 1]) / (sigmaT[i] - sigmaT[i - 1]);
        }
        normalization = cdfv[3]; invNormalization = 1 / normalization;
        for (int i = 0; i < 4; ++i) cdfv[i] *= invNormalization;
        return true;
    }
    static int lowerBoundIndex(const Float *a, int n, Float v) {       /* std::lower_bound(a, a + n, v) - a */
        int k = 0; while (k < n && a[k] < v) k++; return k;
    }
    Float sample(Float u, Float &pdf) const {                           /* :60-75 */
        const int index = std::max(0, lowerBoundIndex(cdfv, 4, u) - 1);
        const Float t = -std::log(std::exp(-intervalStart[index] * sigmaT[index]) - normalization * (u - cdfv[index])) / sigmaT[index];
        pdf = sigmaT[index] * std::exp(-sigmaT[index] * t) * invNormalization;
        return t;
    }
    Float pdf(Float t) const {                                          /* :77-83 */
        const int index = std::max(0, lowerBoundIndex(intervalStart, 3, t) - 1);
        return sigmaT[index] * std::exp(-sigmaT[index] * t) * invNormalization;
    }
    Float cdf(Float t) const {                                          /* :85-96 */
        const int index = std::max(0, lowerBoundIndex(intervalStart, 3, t) - 1);
        Float lower = (index == 0) ? -1 : -std::pow((sigmaT[index] / sigmaT[index - 1]), -sigmaT[index] / (sigmaT[index] - sigmaT[index - 1]));
        Float upper = -std::exp(-sigmaT[index] * t);
        return cdfv[index] + (upper - lower) * invNormalization;
    }
};

struct Scene {
    orc_scene s;
    Grid density, albedoGrid;
    Rif<float> rifF; Rif<double> rifD;
    /* camera */
    Float camM[3][4], aspect, cotHalfFov, invResX, invResY;
    /* filter table, src/libcore/rfilter.cpp:40-55 */
    Float fvalues[33], fradius, fscale;
    /* medium derived */
    Spec sigmaA, sigmaS, sigmaT;
    Float mediumSamplingWeight, samplingDensity;
    MaxExpDist maxExp;
    Float maxDensity, invMaxDensity;
    Float hetStepSize = 0;          /* heterogeneous `stepSize` (method = simpson), heterogeneous.cpp:183,245-257 */
    bool curved;
    Grid sdfGrid; Float sdfEps = 0;
    int frames = 1;                /* film.cpp:71-78 */
    Float modPhase = 0;            /* radians */
    /* `area` emitter on a `rectangle` (rectangle.cpp:99-110): objectToWorld, its inverse, the frame normal, 1 / area */
    bool hasArea = false; Float rectO2W[12], rectW2O[12], rectInvArea = 0; Vec rectN; Spec rectRadiance;
    bool configureArea() {
        hasArea = s.area_radiance[0] != 0 || s.area_radiance[1] != 0 || s.area_radiance[2] != 0;
        if (!hasArea) return true;
        if (s.rif_mode != ORC_RIF_CONST) { g_err = "the area emitter is built for straight rays (rif_mode = CONST)"; return false; }
        if (s.boundary_bsdf != ORC_BSDF_NULL || s.boundary == 2) { g_err = "the area emitter needs an index-matched cube / sphere boundary"; return false; }
        double M[3][4], inv[3][3];
        for (int i = 0; i < 12; i++) { rectO2W[i] = s.area_to_world[i]; M[i / 4][i % 4] = s.area_to_world[i]; }
        const double det = M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) + M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
        if (!(std::fabs(det) > 0)) { g_err = "area emitter: 'toWorld' is singular"; return false; }
        inv[0][0] = (M[1][1] * M[2][2] - M[1][2] * M[2][1]) / det; inv[0][1] = (M[0][2] * M[2][1] - M[0][1] * M[2][2]) / det; inv[0][2] = (M[0][1] * M[1][2] - M[0][2] * M[1][1]) / det;
        inv[1][0] = (M[1][2] * M[2][0] - M[1][0] * M[2][2]) / det; inv[1][1] = (M[0][0] * M[2][2] - M[0][2] * M[2][0]) / det; inv[1][2] = (M[0][2] * M[1][0] - M[0][0] * M[1][2]) / det;
        inv[2][0] = (M[1][0] * M[2][1] - M[1][1] * M[2][0]) / det; inv[2][1] = (M[0][1] * M[2][0] - M[0][0] * M[2][1]) / det; inv[2][2] = (M[0][0] * M[1][1] - M[0][1] * M[1][0]) / det;
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) rectW2O[4 * i + j] = (Float) inv[i][j];
            rectW2O[4 * i + 3] = (Float) -(inv[i][0] * M[0][3] + inv[i][1] * M[1][3] + inv[i][2] * M[2][3]);
        }
        /* rectangle.cpp:102-110: dpdu = o2w(2,0,0), dpdv = o2w(0,2,0), normal = normalize(o2w(Normal(0,0,1))) = the inverse transpose's third column */
        const double du[3] = {2 * M[0][0], 2 * M[1][0], 2 * M[2][0]}, dv[3] = {2 * M[0][1], 2 * M[1][1], 2 * M[2][1]};
        const double lu = std::sqrt(du[0] * du[0] + du[1] * du[1] + du[2] * du[2]), lv = std::sqrt(dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2]);
        if (std::fabs((du[0] * dv[0] + du[1] * dv[1] + du[2] * dv[2]) / (lu * lv)) > Epsilon) { g_err = "Error: 'toWorld' transformation contains shear!"; return false; }   /* :108-109 */
        const double nn[3] = {inv[2][0], inv[2][1], inv[2][2]}, ln = std::sqrt(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]);
        rectN = Vec((Float) (nn[0] / ln), (Float) (nn[1] / ln), (Float) (nn[2] / ln));
        rectInvArea = (Float) (1.0 / (lu * lv));                                   /* :107,121-123 */
        rectRadiance = Spec(s.area_radiance[0], s.area_radiance[1], s.area_radiance[2]);
        return true;
    }
    /* Rectangle::rayIntersect (rectangle.cpp:125-148): t or -1 */
    inline Float rectIntersect(const Vec &o, const Vec &d, Float mint, Float maxt) const {
        const Float *W = rectW2O;
        const Float oz = W[8] * o.x + W[9] * o.y + W[10] * o.z + W[11], dz = W[8] * d.x + W[9] * d.y + W[10] * d.z;
        const Float hit = -oz / dz;
        if (!(hit >= mint && hit <= maxt)) return -1;
        const Float lx = (W[0] * o.x + W[1] * o.y + W[2] * o.z + W[3]) + hit * (W[0] * d.x + W[1] * d.y + W[2] * d.z),
                    ly = (W[4] * o.x + W[5] * o.y + W[6] * o.z + W[7]) + hit * (W[4] * d.x + W[5] * d.y + W[6] * d.z);
        return (std::abs(lx) <= 1 && std::abs(ly) <= 1) ? hit : -1;
    }
    /* AreaLight::eval (area.cpp:102-107): radiance leaving the rectangle along -d for a ray travelling along d */
    inline Spec rectLe(const Vec &d) const { return dot(rectN, -d) <= 0 ? Spec(0.0f) : rectRadiance; }
    /* Shape::sampleDirect + AreaLight::sampleDirect (shape.cpp:102-115, area.cpp:162-177) for a reference point in a medium (refN = 0);
       returns radiance / pdf (0 = back side), direction, distance and the solid-angle pdf */
    inline Spec rectSampleDirect(const Vec &ref, Float sx, Float sy, Vec &d, Float &dist, Float &pdf) const {
        const Float *M = rectO2W; const Float lx = sx * 2 - 1, ly = sy * 2 - 1;
        const Vec p(M[0] * lx + M[1] * ly + M[3], M[4] * lx + M[5] * ly + M[7], M[8] * lx + M[9] * ly + M[11]);
        d = p - ref;
        const Float distSquared = dot(d, d);
        dist = std::sqrt(distSquared);
        d = d / dist;                                                              /* TVector3::operator/=: reciprocal, then multiply */
        const Float dp = std::abs(dot(d, rectN));
        pdf = rectInvArea * (dp != 0 ? (distSquared / dp) : 0.0f);
        if (dot(d, rectN) < 0 && pdf != 0) return rectRadiance / pdf;
        pdf = 0.0f;
        return Spec(0.0f);
    }
    /* AreaLight::pdfDirect (area.cpp:179-187) for a hit at distance dist along d */
    inline Float rectPdfDirect(const Vec &d, Float dist) const { return dot(d, rectN) < 0 ? rectInvArea * (dist * dist) / std::abs(dot(d, rectN)) : 0.0f; }
    /* include/mitsuba/render/pathlengthsampler.h:32-42 */
    inline Float mSeq(Float t, Float phase) const {
        const Float lambda = s.mod_lambda; const int mP = s.mod_P;
        Float pathLength = t;
        pathLength = pathLength + phase * lambda * INV_PI_F / 2;
        pathLength = std::fmod(pathLength, lambda);
        if (pathLength < lambda / mP) return 1 - pathLength * (mP - 1) / lambda;
        else if (pathLength > (1 - 1.0 / mP) * lambda) return 1 - (lambda - pathLength) * (mP - 1) / lambda;
        else return (Float) (1.0 / mP);
    }
    /* PathLengthSampler::correlationFunction, src/librender/pathlengthsampler.cpp:68-114 (mixed float / double as written there) */
    inline Float correlationFunction(Float t) const {
        const Float lambda = s.mod_lambda;
        Float pathLength = t;
        switch (s.modulation) {
        case 1: pathLength = pathLength + modPhase * lambda * INV_PI_F / 2; return (Float) std::cos(pathLength * 2 * M_PI / lambda);
        case 2: pathLength = pathLength + modPhase * lambda * INV_PI_F / 2;
                return 4 / lambda * (std::fabs(std::fmod(pathLength, lambda) - lambda / 2) - lambda / 4);
        case 3: pathLength = pathLength + modPhase * lambda * INV_PI_F / 2;
                pathLength = std::fmod(pathLength, lambda);
                if (pathLength < lambda / 6) return 6 * pathLength / lambda;
                else if (pathLength < lambda / 2 && pathLength >= lambda / 6) return 1.0f;
                else if (pathLength < 2 * lambda / 3 && pathLength >= lambda / 2) return 1 - (pathLength - lambda / 2) * 6 / lambda;
                else return 0;
        case 4: return mSeq(pathLength, modPhase);
        case 5: { Float value = 0;
                  for (int i = 0; i < s.mod_neighbors; i++) value += mSeq(pathLength, (Float) (modPhase - i * (2 * M_PI) / s.mod_P));
                  value -= (float) (s.mod_neighbors - 1) / s.mod_P;
                  return value; }
        }
        return 1.0f;
    }

    bool configure(const orc_scene &in) {
        s = in;
        if (s.sigma_mode == ORC_SIGMA_GRID) {
            if (!s.density.data) { g_err = "No density specified!"; return false; }   /* heterogeneous.cpp:229-230 */
            density.configure(s.density);
            /* heterogeneous.cpp:239-242 with gridvolume.cpp:583-585 (maximum hard-coded to 1) */
            maxDensity = s.density_scale * 1.0f;
            invMaxDensity = 1.0f / maxDensity;
            if (s.method != 0 && s.method != 1) { g_err = "Unsupported integration method!"; return false; }           /* :195-202 */
            if (s.method == 1) {
                if (s.rif_mode != ORC_RIF_CONST) { g_err = "method = simpson belongs to the heterogeneous medium (straight rays)"; return false; }
                hetStepSize = s.het_stepsize;                                      /* :245-257: 0 => min over the volumes' getStepSize() */
                if (hetStepSize == 0) {
                    hetStepSize = density.stepSize;
                    if (s.albedo_mode == ORC_ALBEDO_GRID) { Grid a; a.configure(s.albedo_grid); hetStepSize = std::min(hetStepSize, a.stepSize); }
                }
                if (!(hetStepSize > 0) || !std::isfinite(hetStepSize)) { g_err = "Unable to infer a suitable step size for deterministic integration, please specify one manually using the 'stepSize' parameter."; return false; }
            }
        }
        if (s.albedo_mode == ORC_ALBEDO_GRID) albedoGrid.configure(s.albedo_grid);
        curved = s.rif_mode != ORC_RIF_CONST;
        if (s.boundary == ORC_BOUNDARY_SDF) {
            if (!s.sdf.data || s.sdf.channels != 1 || s.sdf.dtype != ORC_VOL_F32) { g_err = "heterogeneousrefractive: the sdf must be a 1-channel float32 grid"; return false; }
            sdfGrid.configure(s.sdf);
            Float d2 = 0; for (int i = 0; i < 3; ++i) d2 += (s.sdf.aabb_max[i] - s.sdf.aabb_min[i]) * (s.sdf.aabb_max[i] - s.sdf.aabb_min[i]);
            sdfEps = 1e-4f * std::sqrt(d2);
        }
        frames = 1;
        modPhase = (Float) (s.mod_phase_deg * M_PI / 180);                            /* pathlengthsampler.cpp:15 */
        if (s.modulation < 0 || s.modulation > 5) { g_err = "The \"modulation\" parameter must be equal toeither \"none\", \"square\", or \"hamiltonian\", or \"mseq\", or \"depthselective\"!"; return false; }
        if (s.modulation != 0 && s.decomposition != 1) { g_err = "film: a path-length modulation needs decomposition = transient"; return false; }
        if (s.decomposition == 1 && s.modulation != 0) frames = 1;                    /* film.cpp:76-78 */
        else if (s.decomposition == 1 || s.decomposition == 2) {
            frames = (int) std::ceil((s.max_bound - s.min_bound) / s.bin_width);      /* film.cpp:71-74 */
            if (!(frames >= 1) || frames > 4096) { g_err = "film: a decomposition needs 1 <= ceil((maxBound-minBound)/binWidth) <= 4096 frames"; return false; }
        } else if (s.decomposition != 0) { g_err = "The \"decomposition\" parameter must be equal toeither \"none\", \"transient\", or \"bounce\"!"; return false; }
        if (s.rif_double) rifD.configure(s); else rifF.configure(s);
        if (!s.rif_double && s.rif_mode == ORC_RIF_CONST) rifF.cst = s.rif_const;
        /* homogeneous coefficients: src/librender/medium.cpp:26-36 */
        for (int i = 0; i < 3; i++) { sigmaA[i] = s.sigma_a[i]; sigmaS[i] = s.sigma_s[i]; sigmaT[i] = s.sigma_a[i] + s.sigma_s[i]; }
        /* homogeneous.cpp:172-190 == heterogeneousrefractive.cpp:239-255 */
        mediumSamplingWeight = s.medium_sampling_weight;
        if (mediumSamplingWeight == -1) {
            for (int i = 0; i < 3; ++i) {
                Float albedo = sigmaS[i] / sigmaT[i];
                if (albedo > mediumSamplingWeight && sigmaT[i] != 0) mediumSamplingWeight = albedo;
            }
            if (mediumSamplingWeight > 0) mediumSamplingWeight = std::max(mediumSamplingWeight, (Float) 0.5f);
        }
        samplingDensity = 0;
        if (s.strategy == ORC_STRATEGY_SINGLE) {       /* heterogeneousrefractive.cpp:259-275 */
            int channel = 0; Float smallest = std::numeric_limits<Float>::infinity();
            for (int i = 0; i < 3; ++i) if (sigmaT[i] < smallest) { smallest = sigmaT[i]; channel = i; }
            if (s.channel >= 0) channel = s.channel;
            samplingDensity = sigmaT[channel];
        } else if (s.strategy == ORC_STRATEGY_MANUAL) {
            samplingDensity = s.sampling_density;
        } else if (s.strategy == ORC_STRATEGY_MAXIMUM) {   /* homogeneous.cpp:215-220 == heterogeneousrefractive.cpp:286-291 */
            Float c[3] = {sigmaT[0], sigmaT[1], sigmaT[2]};
            if (!maxExp.configure(c)) { g_err = "Internal error: sigmaT must vary across channels"; return false; }
        }
        /* camera: src/sensors/perspective.cpp:130-158, analytic inverse of cameraToSample at z'=0 */
        for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) camM[i][j] = s.cam_to_world[i * 4 + j];
        aspect = (Float) s.width / (Float) s.height;
        cotHalfFov = 1.0f / std::tan((s.fov_x_deg / 2.0f) * (M_PI_F / 180.0f));  /* transform.cpp:99-123 */
        invResX = 1.0f / s.width; invResY = 1.0f / s.height;
        filterTable(s.rfilter, s.rfilter_param, fvalues, fradius, fscale);
        return configureArea();
    }
    static Float filterEval(int kind, Float param, Float radius, Float x) {
        if (kind == ORC_FILTER_BOX) return std::abs(x) <= radius ? 1.0f : 0.0f;     /* src/rfilters/box.cpp */
        Float alpha = -1.0f / (2.0f * param * param);                                /* src/rfilters/gaussian.cpp:50-55 */
        return std::max((Float) 0.0f, std::exp(alpha * x * x) - std::exp(alpha * radius * radius));
    }
    static void filterTable(int kind, Float param, Float *values, Float &radius, Float &scale) {
        const int RES = 31;                                  /* MTS_FILTER_RESOLUTION, rfilter.h:28 */
        radius = kind == ORC_FILTER_BOX ? param + 1e-5f : 4 * param;
        Float sum = 0.0f;
        for (int i = 0; i < RES; ++i) { Float v = filterEval(kind, param, radius, (radius * i) / RES); values[i] = v; sum += v; }
        values[RES] = 0.0f; values[RES + 1] = 0.0f;
        scale = RES / radius;
        sum *= 2 * radius / RES;
        Float normalization = 1.0f / sum;
        for (int i = 0; i < RES; ++i) values[i] *= normalization;
    }
    inline Float evalDiscretized(Float x) const {            /* rfilter.h:76-77 */
        return fvalues[std::min((int) std::abs(x * fscale), 31)];
    }
    /* perspective.cpp:247-269 */
    inline void sampleRay(Float px, Float py, Vec &o, Vec &d, Float &mint, Float &maxt) const {
        Float sx = px * invResX, sy = py * invResY;
        Vec nearP((1.0f - 2.0f * sx) * s.near_clip / cotHalfFov,
                  (1.0f - 2.0f * sy) / aspect * s.near_clip / cotHalfFov, s.near_clip);
        Vec dl = normalize(nearP);
        Float invZ = 1.0f / dl.z;
        mint = s.near_clip * invZ; maxt = s.far_clip * invZ;
        o = Vec(camM[0][3], camM[1][3], camM[2][3]);
        d = Vec(camM[0][0] * dl.x + camM[0][1] * dl.y + camM[0][2] * dl.z,
                camM[1][0] * dl.x + camM[1][1] * dl.y + camM[1][2] * dl.z,
                camM[2][0] * dl.x + camM[2][1] * dl.y + camM[2][2] * dl.z);
    }
    /* signed distance at p: trilinear lookup (gridvolume lookupFloat); outside the grid (where lookupFloat returns 0) far outside */
    inline Float sdfValue(const Vec &p) const {
        int idx4[4];
        const Float v = sdfGrid.lookupFloat(p, idx4);
        return idx4[3] >= 0 ? v : (Float) 1e30f;
    }
    /* medium boundary shape: heterogeneousrefractive.cpp:707-726 generalised to data */
    template <typename FLOAT> inline bool insideShape(const V3<FLOAT> &p) const {
        if (s.boundary == ORC_BOUNDARY_SDF)                     /* negative inside (:481); lookupFloat is 0 outside the grid */
            return sdfValue(Vec((Float) p.x, (Float) p.y, (Float) p.z)) < 0;
        if (s.boundary == ORC_BOUNDARY_SPHERE) {
            V3<FLOAT> q(p.x - (FLOAT) s.sph_center[0], p.y - (FLOAT) s.sph_center[1], p.z - (FLOAT) s.sph_center[2]);
            return dot(q, q) < (FLOAT) s.sph_radius * (FLOAT) s.sph_radius;
        }
        return p.x >= s.bmin[0] && p.x <= s.bmax[0] && p.y >= s.bmin[1] && p.y <= s.bmax[1] && p.z >= s.bmin[2] && p.z <= s.bmax[2];
    }
    /* ray / boundary-shape intersection in [mint, maxt]; returns t or -1 */
    inline Float intersectShape(const Vec &o, const Vec &d, Float mint, Float maxt) const {
        Float nearT, farT;
        if (s.boundary == ORC_BOUNDARY_SDF) {
            /* sphere tracing on the signed-distance grid (new: the reference intersects the mesh; the SDF is its inside test).  A start
               point near the surface counts as inside (eps = 1e-4 x grid diagonal): the walk then looks for the exit */
            if (!Grid::aabbIntersect(sdfGrid.wmin, sdfGrid.wmax, o, d, nearT, farT)) return -1;
            const Float t0 = std::max(nearT, mint), t1 = std::min(farT, maxt);
            if (!(t0 <= t1)) return -1;
            Float t = t0;
            Float v = sdfValue(o + d * t);
            /* Entry points are returned with sdf < eps/2 and exit points with sdf in [eps, ~2 eps): a start within 4 eps of the surface
               is a start from the inside side; it first has to get below 0 ("armed") before an exit counts */
            const bool insideStart = v < 4 * sdfEps;
            bool armed = false;
            for (int it = 0; it < 1024; ++it) {
                if (insideStart) {
                    if (v < 0) armed = true;
                    else if (armed && v >= sdfEps) return t;
                    else if (!armed && it >= 16) return t;          /* grazing / outward start: it never went in */
                } else if (v < 0.5f * sdfEps) return t;
                t += v > 1e29f ? sdfEps : std::max(std::fabs(v), sdfEps);       /* a sample on the grid's face may round to outside it */
                if (t > t1) return insideStart ? (farT <= maxt ? farT : (Float) -1) : (Float) -1;
                v = sdfValue(o + d * t);
            }
            return -1;
        }
        if (s.boundary == ORC_BOUNDARY_SPHERE) {
            /* src/shapes/sphere.cpp rayIntersect: double-precision quadratic */
            double ox = (double) o.x - s.sph_center[0], oy = (double) o.y - s.sph_center[1], oz = (double) o.z - s.sph_center[2];
            double dx = d.x, dy = d.y, dz = d.z;
            double A = dx * dx + dy * dy + dz * dz, B = 2 * (dx * ox + dy * oy + dz * oz),
                   C = ox * ox + oy * oy + oz * oz - (double) s.sph_radius * s.sph_radius;
            double disc = B * B - 4 * A * C;
            if (disc < 0) return -1;
            double root = std::sqrt(disc);
            double q = B < 0 ? -0.5 * (B - root) : -0.5 * (B + root);
            double t0 = q / A, t1 = C / q;
            if (t0 > t1) std::swap(t0, t1);
            nearT = (Float) t0; farT = (Float) t1;
        } else {
            if (!Grid::aabbIntersect(s.bmin, s.bmax, o, d, nearT, farT)) return -1;
        }
        if (!(nearT <= maxt && farT >= mint)) return -1;
        if (nearT >= mint) return nearT;
        if (farT <= maxt) return farT;
        return -1;
    }
    /* outward geometric normal of the boundary shape at a surface point (cube: the face whose plane the point is closest to) */
    inline Vec shapeNormal(const Vec &x) const {
        if (s.boundary == ORC_BOUNDARY_SDF) {                   /* normalized SDF gradient (heterogeneousrefractive.cpp:980-984) */
            Float v; V3<float> g;
            trilinearValueGrad<float>(sdfGrid, V3<float>(x), v, g);
            return normalize(Vec(g.x, g.y, g.z));
        }
        if (s.boundary == ORC_BOUNDARY_SPHERE) {
            Vec n(x.x - s.sph_center[0], x.y - s.sph_center[1], x.z - s.sph_center[2]);
            return normalize(n);
        }
        Float best = -1; int axis = 0; Float sign = 1;
        const Float xx[3] = {x.x, x.y, x.z};
        for (int i = 0; i < 3; ++i) {
            const Float c = 0.5f * (s.bmin[i] + s.bmax[i]), hsz = 0.5f * (s.bmax[i] - s.bmin[i]);
            const Float r = std::fabs(xx[i] - c) / hsz;
            if (r > best) { best = r; axis = i; sign = xx[i] >= c ? 1.0f : -1.0f; }
        }
        return Vec(axis == 0 ? sign : 0.0f, axis == 1 ? sign : 0.0f, axis == 2 ? sign : 0.0f);
    }
    /* eta of the hdielectric boundary: the RIF at the hit point (hdielectric.cpp:115-118), queried just inside the grid */
    inline Float boundaryEta(const Vec &x) const {
        if (s.rif_mode == ORC_RIF_CONST) return s.rif_const;
        Counters dummy; Vec q = x;
        const Float *mn = s.rif.aabb_min, *mx = s.rif.aabb_max;
        bool w = false; for (int i = 0; i < 12; i++) w = w || (s.rif.world_to_volume[i] != 0.0f && s.rif.world_to_volume[i] != ((i % 5 == 0) ? 1.0f : 0.0f));
        if (s.rif_mode != ORC_RIF_ACOUSTIC && !w) {              /* the analytic field has no grid to stay inside of; a transformed one clamps its cell */
            q.x = std::min(std::max(q.x, mn[0]), mx[0]); q.y = std::min(std::max(q.y, mn[1]), mx[1]); q.z = std::min(std::max(q.z, mn[2]), mx[2]);
        }
        return s.rif_double ? (Float) rifD.value(V3<double>(q), dummy) : rifF.value(V3<float>(q), dummy);
    }
    inline Spec albedoAt(const Vec &p) const {
        if (s.albedo_mode == ORC_ALBEDO_GRID) return albedoGrid.lookupSpectrum(p);
        return Spec(s.albedo[0], s.albedo[1], s.albedo[2]);       /* constvolume.cpp:57-64 */
    }
    template <typename FLOAT> const Rif<FLOAT> &rif() const;
};
template <> const Rif<float> &Scene::rif<float>() const { return rifF; }
template <> const Rif<double> &Scene::rif<double>() const { return rifD; }

/* ------------------------------------------------------------------ A6 er_step, A7 trace */
template <typename FLOAT> struct Tracer {
    const Scene &S; const Rif<FLOAT> &R; Counters &C;
    Tracer(const Scene &s, Counters &c) : S(s), R(s.rif<FLOAT>()), C(c) {}

    /* heterogeneousrefractive.cpp:653-661 (velocity-Verlet), or classic RK4 on (p,v,opt) (new) */
    inline void er_step(V3<FLOAT> &p, V3<FLOAT> &v, FLOAT h, FLOAT &opt) const {
        C.c[ORC_C_STEPS]++;
        if (S.s.stepper == ORC_STEP_VERLET) {
            FLOAT n; V3<FLOAT> G;
            R.valueAndGradient(p, n, G, C);
            v += SplineConst<FLOAT>::half() * h * G;
            p += h * v / n;
            v += SplineConst<FLOAT>::half() * h * R.gradient(p, C);
            opt += h * n;
        } else {
            /* classic RK4 (new), defined with fused multiply-adds; same expression tree as the kernel */
            auto fma3 = [](FLOAT sc, const V3<FLOAT> &a, const V3<FLOAT> &b) {
                return V3<FLOAT>(std::fma(sc, a.x, b.x), std::fma(sc, a.y, b.y), std::fma(sc, a.z, b.z));
            };
            FLOAT n; V3<FLOAT> gr;
            const FLOAT hh = (FLOAT) 0.5 * h;
            R.valueAndGradient(p, n, gr, C);
            V3<FLOAT> kp = v * ((FLOAT) 1 / n);
            V3<FLOAT> ps = kp, vs = gr; FLOAT ns = n;
            V3<FLOAT> vv = fma3(hh, gr, v);
            R.valueAndGradient(fma3(hh, kp, p), n, gr, C);
            kp = vv * ((FLOAT) 1 / n);
            ps = fma3((FLOAT) 2, kp, ps); vs = fma3((FLOAT) 2, gr, vs); ns = std::fma((FLOAT) 2, n, ns);
            vv = fma3(hh, gr, v);
            R.valueAndGradient(fma3(hh, kp, p), n, gr, C);
            kp = vv * ((FLOAT) 1 / n);
            ps = fma3((FLOAT) 2, kp, ps); vs = fma3((FLOAT) 2, gr, vs); ns = std::fma((FLOAT) 2, n, ns);
            vv = fma3(h, gr, v);
            R.valueAndGradient(fma3(h, kp, p), n, gr, C);
            kp = vv * ((FLOAT) 1 / n);
            ps = ps + kp; vs = vs + gr; ns = ns + n;
            const FLOAT h6 = h * ((FLOAT) 1 / (FLOAT) 6);
            p = fma3(h6, ps, p);
            v = fma3(h6, vs, v);
            opt = std::fma(h6, ns, opt);
        }
    }
    /* heterogeneousrefractive.cpp:662-669 (the connection code always uses the reference's own Verlet step) */
    inline void er_step_verlet(V3<FLOAT> &p, V3<FLOAT> &v, FLOAT h, FLOAT &opt) const {
        C.c[ORC_C_STEPS]++;
        FLOAT n; V3<FLOAT> G;
        R.valueAndGradient(p, n, G, C);
        v += SplineConst<FLOAT>::half() * h * G;
        p += h * v / n;
        v += SplineConst<FLOAT>::half() * h * R.gradient(p, C);
        opt += h * n;
    }
    /* heterogeneousrefractive.cpp:671-691 */
    /* aggressive_trace (:697-704): same as trace but no inside / outside tests */
    inline void aggressiveTrace(V3<FLOAT> &p, V3<FLOAT> &v, FLOAT sampledDistance, FLOAT &opt) const {
        const FLOAT h = (FLOAT) S.s.stepsize;
        FLOAT distance = sampledDistance;
        int steps = (int) (distance / h);
        distance = distance - steps * h;
        for (int i = 0; i < steps; i++) er_step(p, v, h, opt);
        er_step(p, v, distance, opt);
    }
    /* sampleDistance's `aggressivetracing` branch (:476-493), applied to every finite trace of a free flight or of a
       transmittance walk (the connection transmittance keeps the plain trace): legs of min(depth below the surface, distance left)
       without tests while that depth is >= Epsilon, then the tested trace of the rest */
    inline bool traceMaybeAggressive(V3<FLOAT> &p, V3<FLOAT> &v, FLOAT sampledDistance, FLOAT &distSurf, FLOAT &opt) const {
        if (!(S.s.aggressive_tracing && S.s.boundary == ORC_BOUNDARY_SDF)) return trace(p, v, sampledDistance, distSurf, opt);
        FLOAT dist_left = sampledDistance, dist_traced = 0;
        while (dist_left > (FLOAT) Epsilon) {
            FLOAT sdf = -(FLOAT) S.sdfValue(Vec(p));
            sdf -= (FLOAT) S.s.sdf_max_error;
            if (sdf < (FLOAT) Epsilon) break;
            const FLOAT traceDist = std::min(sdf, dist_left);
            aggressiveTrace(p, v, traceDist, opt);
            dist_left -= traceDist;
            dist_traced += traceDist;
        }
        const bool success = trace(p, v, dist_left, distSurf, opt);
        distSurf += dist_traced;
        return success;
    }
    inline bool trace(V3<FLOAT> &p, V3<FLOAT> &v, FLOAT sampledDistance, FLOAT &distSurf, FLOAT &opt) const {
        const FLOAT h = (FLOAT) S.s.stepsize;
        FLOAT distance = sampledDistance;
        distSurf = 0;
        int steps = (int) (distance / h);
        distance = distance - steps * h;
        for (int i = 0; i < steps; i++) {
            er_step(p, v, h, opt);
            if (!S.insideShape(p)) { er_step(p, v, -h, opt); return false; }
            distSurf += h;
        }
        er_step(p, v, distance, opt);
        if (!S.insideShape(p)) { er_step(p, v, -distance, opt); return false; }
        distSurf += distance;
        return true;
    }
    /* heterogeneousrefractive.cpp:742-776 */
    inline void traceTillBoundary(V3<FLOAT> &p, V3<FLOAT> &v, FLOAT &distSurf, FLOAT &opt) const {
        distSurf = 0;
        const long maxsteps = 100000;
        const FLOAT h = (FLOAT) S.s.stepsize;
        for (long i = 0; i < maxsteps; i++) {
            er_step(p, v, h, opt);
            if (S.insideShape(p)) distSurf += h;
            else { er_step(p, v, -h, opt); distSurf -= h; return; }
        }
    }
};


/* ------------------------------------------------------------------ A9 phase functions */
/* src/libcore/util.cpp:606-615 */
inline void coordinateSystem(const Vec &a, Vec &b, Vec &c) {
    if (std::abs(a.x) > std::abs(a.y)) {
        Float invLen = 1.0f / std::sqrt(a.x * a.x + a.z * a.z);
        c = Vec(a.z * invLen, 0.0f, -a.x * invLen);
    } else {
        Float invLen = 1.0f / std::sqrt(a.y * a.y + a.z * a.z);
        c = Vec(0.0f, a.z * invLen, -a.y * invLen);
    }
    b = cross(c, a);
}
inline Float safe_sqrt(Float v) { return std::sqrt(std::max((Float) 0.0f, v)); }   /* math.h:260-267 */
/* src/libcore/warp.cpp:25-31 */
inline Vec squareToUniformSphere(Float sx, Float sy) {
    Float z = 1.0f - 2.0f * sy;
    Float r = safe_sqrt(1.0f - z * z);
    Float phi = 2.0f * M_PI_F * sx;
    return Vec(r * std::cos(phi), r * std::sin(phi), z);
}
/* src/phase/hg.cpp:107-110, src/phase/isotropic.cpp:76-78.  wi points away from the vertex (phase.h:40-48) */
inline Float phaseEval(int kind, Float g, const Vec &wi, const Vec &wo) {
    if (kind == ORC_PHASE_ISOTROPIC) return INV_FOURPI_F;
    Float temp = 1.0f + g * g + 2.0f * g * dot(wi, wo);
    return INV_FOURPI_F * (1 - g * g) / (temp * std::sqrt(temp));
}
/* src/phase/hg.cpp:74-103, src/phase/isotropic.cpp:62-74 */
inline Float phaseSample(int kind, Float g, const Vec &wi, Float sx, Float sy, Vec &wo, Float &pdf) {
    if (kind == ORC_PHASE_ISOTROPIC) { wo = squareToUniformSphere(sx, sy); pdf = INV_FOURPI_F; return 1.0f; }
    Float cosTheta;
    if (std::abs(g) < Epsilon) cosTheta = 1 - 2 * sx;
    else {
        Float sqrTerm = (1 - g * g) / (1 - g + 2 * g * sx);
        cosTheta = (1 + g * g - sqrTerm * sqrTerm) / (2 * g);
    }
    Float sinTheta = safe_sqrt(1.0f - cosTheta * cosTheta);
    Float phi = 2 * M_PI_F * sy, sinPhi = std::sin(phi), cosPhi = std::cos(phi);
    Vec n = -wi, s, t;
    coordinateSystem(n, s, t);            /* Frame(n): include/mitsuba/core/frame.h:55-57 */
    Vec l(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta);
    wo = s * l.x + t * l.y + n * l.z;     /* frame.h:83-85 toWorld */
    pdf = phaseEval(kind, g, wi, wo);
    return 1.0f;
}

/* ------------------------------------------------------------------ A12 curved-ray connection (config 5)
   heterogeneousrefractive.cpp:798-1163 restated for two points INSIDE the medium shape (the boundary branch with Snell
   refraction and its Jacobian, :873-919,:980-1001,:1036-1074, belongs to the `hdielectric` boundary = "next" row N2 and is
   treated as a failed connection here).  The reference minimises 0.5|r|^2 with Ceres LINE_SEARCH/BFGS (<= 20
   iterations, function tolerance tol2); Ceres is not available (un-vendored, unpinned) => a Levenberg-damped
   Gauss-Newton with the same analytic Jacobian stands in: PARITY UNPINNED for the iterates, checked on the converged
   direction only. */
template <typename FLOAT> struct M33 {
    FLOAT m[3][3];
    M33() {}
    explicit M33(FLOAT d) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) m[i][j] = i == j ? d : 0; }
    static M33 outer(const V3<FLOAT> &a, const V3<FLOAT> &b) {          /* Matrix3x3(v1, v2), matrix.h:716-720 */
        M33 r; const FLOAT A[3] = {a.x, a.y, a.z}, B[3] = {b.x, b.y, b.z};
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = A[i] * B[j];
        return r;
    }
    M33 operator*(const M33 &o) const { M33 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { FLOAT s = 0; for (int k = 0; k < 3; k++) s += m[i][k] * o.m[k][j]; r.m[i][j] = s; } return r; }
    M33 operator*(FLOAT s) const { M33 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = m[i][j] * s; return r; }
    M33 operator+(const M33 &o) const { M33 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = m[i][j] + o.m[i][j]; return r; }
    M33 operator-(const M33 &o) const { M33 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = m[i][j] - o.m[i][j]; return r; }
    V3<FLOAT> preMult(const V3<FLOAT> &v) const {                       /* matrix.h:640-644: v^T M */
        return V3<FLOAT>(v.x * m[0][0] + v.y * m[1][0] + v.z * m[2][0], v.x * m[0][1] + v.y * m[1][1] + v.z * m[2][1],
                         v.x * m[0][2] + v.y * m[1][2] + v.z * m[2][2]);
    }
};

inline Float fresnelDielectricExt(Float cosThetaI_, Float &cosThetaT_, Float eta);     /* below */
template <typename FLOAT> inline bool finite3(const V3<FLOAT> &a) { return std::isfinite(a.x) && std::isfinite(a.y) && std::isfinite(a.z); }

template <typename FLOAT> struct Connector {
    const Scene &S; const Rif<FLOAT> &R; Counters &C; Pcg32 &rng;
    FLOAT tol, rrweight; int precision, maxIter;
    Connector(const Scene &s, Counters &c, Pcg32 &r) : S(s), R(s.rif<FLOAT>()), C(c), rng(r) {
        tol = (FLOAT) 1e-6; rrweight = (FLOAT) 1e-2; precision = 3; maxIter = 20;     /* :209-213, :217 */
    }
    static M33<FLOAT> fromH(const FLOAT H[9]) { M33<FLOAT> r; for (int i = 0; i < 9; i++) r.m[i / 3][i % 3] = H[i]; return r; }
    int maxSteps() const {
        /* the reference allows 1e5 steps; bound by what a ray can travel inside the shape */
        FLOAT diag = 0; for (int i = 0; i < 3; i++) diag += (S.s.bmax[i] - S.s.bmin[i]) * (S.s.bmax[i] - S.s.bmin[i]);
        if (S.s.boundary == ORC_BOUNDARY_SPHERE) diag = 4 * S.s.sph_radius * S.s.sph_radius;
        return std::min(100000, (int) (4 * std::sqrt(diag) / S.s.stepsize) + 16);
    }
    /* :798-814.  The solver's iterates are not pinned to the reference (Ceres), so the arithmetic of this step is a DEFINITION shared with the
       HIP kernel (Connector::dstep, csrc/mer_connect.hpp): (v (x) G) dp formed as v (x) (G^T dp), every accumulation a fused multiply-add. */
    static void addHessDp(M33<FLOAT> &dv, const FLOAT H[9], const M33<FLOAT> &dp, FLOAT t) {
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                const FLOAT sum = std::fma(H[3 * i + 2], dp.m[2][j], std::fma(H[3 * i + 1], dp.m[1][j], H[3 * i] * dp.m[0][j]));
                dv.m[i][j] = std::fma(t, sum, dv.m[i][j]);
            }
    }
    void er_derivativestep(V3<FLOAT> &p, V3<FLOAT> &v, M33<FLOAT> &dpdv0, M33<FLOAT> &dvdv0, FLOAT h) const {
        FLOAT n, H[9]; V3<FLOAT> G;
        const FLOAT t = SplineConst<FLOAT>::half() * h;
        R.valueGradientAndHessian(p, n, G, H, C);
        v = V3<FLOAT>(std::fma(t, G.x, v.x), std::fma(t, G.y, v.y), std::fma(t, G.z, v.z));
        addHessDp(dvdv0, H, dpdv0, t);
        p += h * v / n;
        R.valueGradientAndHessian(p, n, G, H, C);
        const FLOAT invn = 1 / n, c = -(invn * invn);
        const FLOAT Vc[3] = {c * v.x, c * v.y, c * v.z};
        FLOAT w[3];
        for (int j = 0; j < 3; j++) w[j] = std::fma(G.z, dpdv0.m[2][j], std::fma(G.y, dpdv0.m[1][j], G.x * dpdv0.m[0][j]));
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) dpdv0.m[i][j] = std::fma(h, std::fma(Vc[i], w[j], invn * dvdv0.m[i][j]), dpdv0.m[i][j]);
        v = V3<FLOAT>(std::fma(t, G.x, v.x), std::fma(t, G.y, v.y), std::fma(t, G.z, v.z));
        addHessDp(dvdv0, H, dpdv0, t);
        C.c[ORC_C_STEPS]++;
    }
    /* boundaryVelocity (:1040-1055): Snell's law for the optical momentum at a surface (normal N, index ni on the ray's side, ne beyond) */
    void boundaryVelocity(V3<FLOAT> &v, const V3<FLOAT> &N, FLOAT ni, FLOAT ne) const {
        const FLOAT dotp = dot(v, N);
        FLOAT r = ne / ni; r = r * r - 1;
        const FLOAT n2 = dot(v, v);
        FLOAT sq = r * n2 + dotp * dotp;
        if (sq < (FLOAT) Epsilon) { v = (FLOAT) 2 * dotp * N - v; return; }
        sq = std::sqrt(sq);
        v = v - dotp * N + (FLOAT) sgn(dotp) * sq * N;
    }
    /* boundaryVelocityDerivative (:1061-1074) */
    void boundaryVelocityDerivative(V3<FLOAT> &v, M33<FLOAT> &dvdv0, const V3<FLOAT> &dtbdv0, const V3<FLOAT> &dnb, const V3<FLOAT> &N, FLOAT ni, FLOAT ne) const {
        const FLOAT dotp = dot(v, N);
        FLOAT r = ne / ni; r = r * r - 1;
        const FLOAT n2 = dot(v, v);
        FLOAT sq = r * n2 + dotp * dotp;
        const M33<FLOAT> inner = dvdv0 + M33<FLOAT>::outer(dnb, dtbdv0), NN = M33<FLOAT>::outer(N, N), I((FLOAT) 1);
        if (sq < (FLOAT) Epsilon) {
            v = (FLOAT) 2 * dotp * N - v;
            dvdv0 = (NN * (FLOAT) 2 - I) * inner;
            return;
        }
        sq = std::sqrt(sq);
        dvdv0 = (I - NN + M33<FLOAT>::outer(N, (r * v + dotp * N) / sq) * (FLOAT) sgn(dotp)) * inner;
        v = v - dotp * N + (FLOAT) sgn(dotp) * sq * N;
    }
    /* :816-939.  J[r][c] = d error_r / d v0_c.  Returns false where the reference sets error = p1 - p2 with a zero derivative.
       cross = the reference's isSensorSample (here: p2 lies outside the medium shape). */
    bool computefdf(const V3<FLOAT> &v_i, const V3<FLOAT> &p1, const V3<FLOAT> &p2, bool cross, V3<FLOAT> &error, M33<FLOAT> &J) const {
        M33<FLOAT> dpdv0((FLOAT) 0), dvdv0((FLOAT) 1);
        error = p1 - p2; J = M33<FLOAT>((FLOAT) 0);
        /* a shooting direction that is not a finite non-zero vector has no ray (the stand-in solver's step can overflow: guard, not reference) */
        if (!finite3(v_i) || !(dot(v_i, v_i) > 0) || !std::isfinite(dot(v_i, v_i))) return false;
        if (!R.insideVolumeLimits(p1)) return false;
        FLOAT h = (FLOAT) S.s.stepsize;
        long nBisect = (long) std::ceil(precision / std::log10(2.0));
        V3<FLOAT> p = p1, oldp, v = v_i, oldv;
        M33<FLOAT> olddp, olddv;
        bool signOld = std::signbit(dot(p - p2, v)), signNew;
        FLOAT r = R.value(p, C);
        const FLOAT n1 = std::sqrt(dot(v_i, v_i)), n2 = n1 * n1, n3 = n2 * n1;
        dvdv0 = ((M33<FLOAT>(n2) - M33<FLOAT>::outer(v, v)) * (r / n3)) * dvdv0;
        v = v / n1 * r;
        const int ms = maxSteps();
        int found = 0;
        for (int i = 0; i < ms; i++) {
            oldp = p; oldv = v; olddp = dpdv0; olddv = dvdv0;
            er_derivativestep(p, v, dpdv0, dvdv0, h);
            signNew = std::signbit(dot(p - p2, v));
            if (signNew != signOld) {
                while (nBisect > 0) {
                    nBisect--;
                    p = oldp; v = oldv; dpdv0 = olddp; dvdv0 = olddv;
                    h = h / 2;
                    er_derivativestep(p, v, dpdv0, dvdv0, h);
                    signNew = std::signbit(dot(p - p2, v));
                    if (signNew == signOld) { oldp = p; oldv = v; olddp = dpdv0; olddv = dvdv0; }
                }
                found = 1;
                break;
            } else if (!S.insideShape(p)) {                       /* :873-919 */
                /* towards a point inside the shape a ray that leaves it is a failed trial here (mer_connect.hpp: fewer traced rays for
                   the same connections); the reference refracts it as well */
                if (!cross) return false;
                while (nBisect > 0) {
                    nBisect--;
                    p = oldp; v = oldv; dpdv0 = olddp; dvdv0 = olddv;
                    h = h / 2;
                    er_derivativestep(p, v, dpdv0, dvdv0, h);
                    if (S.insideShape(p)) { oldp = p; oldv = v; olddp = dpdv0; olddv = dvdv0; }
                }
                found = 2;
                break;
            }
        }
        if (!found) return false;
        V3<FLOAT> dpdt, dtstar;
        if (found == 1) {
            V3<FLOAT> dvdt; FLOAT rr;
            R.valueAndGradient(p, rr, dvdt, C);
            dpdt = v / rr;
            dtstar = -(dpdv0.preMult(v) + dvdv0.preMult(p - p2)) / (dot(v, dpdt) + dot(p - p2, dvdt));
        } else {
            if (dot(p - p1, p - p1) < (FLOAT) Epsilon) return false;            /* no progress made (:890-894) */
            FLOAT nb; V3<FLOAT> dnb;
            R.valueAndGradient(p, nb, dnb, C);
            const V3<FLOAT> dpdtb = v / nb;
            const Vec Nf = S.shapeNormal(Vec((Float) p.x, (Float) p.y, (Float) p.z));          /* normalize(m_SDF->gradient(p)) (:899-900) */
            const V3<FLOAT> N((FLOAT) Nf.x, (FLOAT) Nf.y, (FLOAT) Nf.z);
            const V3<FLOAT> dtb = -dpdv0.preMult(N) / dot(N, dpdtb);
            boundaryVelocityDerivative(v, dvdv0, dtb, dnb, N, nb, (FLOAT) 1);
            const FLOAT extra_t = -dot(v, p - p2) / dot(v, v);
            if (cross && extra_t < 0) return false;                               /* :907-912 */
            dpdv0 = dpdv0 + M33<FLOAT>::outer(dpdtb - v, dtb) + dvdv0 * extra_t;
            p += extra_t * v;
            dpdt = v;
            dtstar = -(dpdv0.preMult(v) + dvdv0.preMult(p - p2)) / dot(v, dpdt);
        }
        J = dpdv0 + M33<FLOAT>::outer(dpdt, dtstar);
        error = p - p2;
        return true;
    }
    /* stand-in for ceres::Solve (LINE_SEARCH/BFGS, <= 20 iterations, function_tolerance tol2): damped Gauss-Newton */
    FLOAT solve(V3<FLOAT> &x, const V3<FLOAT> &p1, const V3<FLOAT> &p2) const {
        V3<FLOAT> e; M33<FLOAT> J;
        const bool cross = !S.insideShape(p2);
        bool ok = computefdf(x, p1, p2, cross, e, J);
        FLOAT cost = (FLOAT) 0.5 * dot(e, e), lambda = (FLOAT) 1e-4;
        /* computefdf rescales its argument to |v0| = n(p1): the residual does not depend on |x|, J^T J is singular along x and only the
           damping makes the step finite.  The unknown lives on the sphere |x| = n(p1): every trial iterate is put back on it, and a
           step whose determinant is below float resolution of the product of the pivots, or that is not finite, is retried with
           more damping (same rule on the GPU: mer_connect.hpp). */
        const FLOAT radius = std::sqrt(dot(x, x));
        for (int it = 0; it < maxIter && ok && cost >= tol * (FLOAT) 1e-3; ++it) {
            /* (J^T J + lambda I) d = -J^T e */
            FLOAT A[3][3], b[3];
            const FLOAT E[3] = {e.x, e.y, e.z};
            for (int i = 0; i < 3; i++) { b[i] = 0; for (int k = 0; k < 3; k++) b[i] -= J.m[k][i] * E[k];
                for (int j = 0; j < 3; j++) { A[i][j] = 0; for (int k = 0; k < 3; k++) A[i][j] += J.m[k][i] * J.m[k][j]; } }
            bool improved = false;
            for (int tries = 0; tries < 6 && !improved; ++tries) {
                FLOAT M[3][3]; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M[i][j] = A[i][j] + (i == j ? lambda * (A[i][i] + (FLOAT) 1e-12) : 0);
                const FLOAT det = M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
                                  M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
                if (!(std::fabs(det) > (FLOAT) 1e-6f * std::fabs(M[0][0] * M[1][1] * M[2][2])) || !std::isfinite(det)) { lambda *= 10; continue; }
                FLOAT d[3];
                d[0] = (b[0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (b[1] * M[2][2] - M[1][2] * b[2]) + M[0][2] * (b[1] * M[2][1] - M[1][1] * b[2])) / det;
                d[1] = (M[0][0] * (b[1] * M[2][2] - M[1][2] * b[2]) - b[0] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) + M[0][2] * (M[1][0] * b[2] - b[1] * M[2][0])) / det;
                d[2] = (M[0][0] * (M[1][1] * b[2] - b[1] * M[2][1]) - M[0][1] * (M[1][0] * b[2] - b[1] * M[2][0]) + b[0] * (M[1][0] * M[2][1] - M[1][1] * M[2][0])) / det;
                V3<FLOAT> xn(x.x + d[0], x.y + d[1], x.z + d[2]), en; M33<FLOAT> Jn;
                const FLOAT ln = std::sqrt(dot(xn, xn));
                if (!(ln > 0) || !std::isfinite(ln)) { lambda *= 10; continue; }
                xn = xn * (radius / ln);
                const bool okn = computefdf(xn, p1, p2, cross, en, Jn);
                const FLOAT cn = (FLOAT) 0.5 * dot(en, en);
                if (okn && cn < cost) { x = xn; e = en; J = Jn; cost = cn; lambda = std::max(lambda * (FLOAT) 0.1, (FLOAT) 1e-9); improved = true; }
                else lambda *= 10;
            }
            if (!improved) break;
        }
        return ok ? cost : std::numeric_limits<FLOAT>::infinity();
    }
    /* :941-1030.  cross = false: a ray that leaves the shape is no connection (emitter samples, :960-962); cross = true: it is
       taken to the boundary, refracted (Snell, exterior index 1) and followed straight to its closest approach (sensor samples,
       :963-992).  dist = arc length inside the shape; bweight = weight of the boundary BSDF for the refracted ray. */
    bool computePathLengths(const V3<FLOAT> &p1, const V3<FLOAT> &p2, const V3<FLOAT> &dirToP2, bool cross, V3<FLOAT> &revDir, FLOAT &optDist, FLOAT &dist, FLOAT &bweight) const {
        dist = 0; optDist = 0; bweight = 1;
        FLOAT h = (FLOAT) S.s.stepsize;
        long nBisect = (long) std::ceil(precision / std::log10(2.0));
        V3<FLOAT> p = p1, oldp, v = dirToP2, oldv;
        const bool signOld = std::signbit(dot(p - p2, v)); bool signNew;
        Tracer<FLOAT> T(S, C);
        FLOAT dummy = 0;
        const int ms = maxSteps();
        for (int i = 0; i < ms; i++) {
            oldp = p; oldv = v;
            T.er_step_verlet(p, v, h, dummy);
            signNew = std::signbit(dot(p - p2, v));
            if (!S.insideShape(p)) {
                if (!cross) return false;
                while (nBisect > 0) {
                    nBisect--;
                    p = oldp; v = oldv; h = h / 2;
                    T.er_step_verlet(p, v, h, dummy);
                    if (S.insideShape(p)) { dist += h; optDist += h * R.value(SplineConst<FLOAT>::half() * (p + oldp), C); oldp = p; oldv = v; }
                }
                const Vec Nf = S.shapeNormal(Vec((Float) p.x, (Float) p.y, (Float) p.z));
                const V3<FLOAT> N((FLOAT) Nf.x, (FLOAT) Nf.y, (FLOAT) Nf.z);
                const FLOAT nb = R.value(p, C);
                if (S.s.boundary_bsdf == 1) {                  /* HDielectric's refracted component seen from inside: (1 - F) eta^2 (hdielectric.cpp:183-242) */
                    const Float cosI = (Float) dot(normalize(v), N);
                    Float cosT; const Float F = fresnelDielectricExt(-cosI, cosT, (Float) nb);
                    bweight = (FLOAT) ((1.0f - F) * ((Float) nb * (Float) nb));
                }
                boundaryVelocity(v, N, nb, (FLOAT) 1);           /* :982-984 */
                const FLOAT extra_t = -dot(v, p - p2) / dot(v, v);
                if (extra_t < 0) return false;
                p += extra_t * v;
                optDist += extra_t;
                break;
            }
            if (signNew != signOld) {
                while (nBisect > 0) {
                    nBisect--;
                    p = oldp; v = oldv; h = h / 2;
                    T.er_step_verlet(p, v, h, dummy);
                    signNew = std::signbit(dot(p - p2, v));
                    if (signNew == signOld) { dist += h; optDist += h * R.value(SplineConst<FLOAT>::half() * (p + oldp), C); oldp = p; oldv = v; }
                }
                break;
            } else { dist += h; optDist += h * R.value(SplineConst<FLOAT>::half() * (p + oldp), C); }
        }
        if (dot(p - p2, p - p2) > tol) return false;
        revDir = -normalize(v);
        return true;
    }
    /* :1078-1084 */
    V3<FLOAT> uniformSample(const V3<FLOAT> &in) {
        Vec a((Float) in.x, (Float) in.y, (Float) in.z), ax, ay;
        coordinateSystem(a, ax, ay);
        const Float u1 = rng.next1D(), u2 = rng.next1D();
        const Float z = u1, tmp = safe_sqrt(1.0f - z * z), phi = 2.0f * M_PI_F * u2;          /* warp.cpp:33-41 */
        const Vec t(std::cos(phi) * tmp, std::sin(phi) * tmp, z);
        return V3<FLOAT>((FLOAT) t.x * V3<FLOAT>(ax) + (FLOAT) t.y * V3<FLOAT>(ay) + (FLOAT) t.z * in);
    }
    /* :1087-1163 */
    bool makeDirectConnections(const V3<FLOAT> &p1, const V3<FLOAT> &p2, const V3<FLOAT> &d, FLOAT &weight, V3<FLOAT> &dirToP2,
                               V3<FLOAT> &revDir, FLOAT &optDist, FLOAT &dist) {
        V3<FLOAT> tempSol(0, 0, 0);
        int iterations = 1;
        if (!R.insideVolumeLimits(p1)) return false;
        const FLOAT RIFp = R.value(p1, C);
        while (true) {
            V3<FLOAT> x = uniformSample(d) * RIFp;
            const FLOAT cost = solve(x, p1, p2);
            if (cost < tol) {
                if (iterations == 1) { iterations++; tempSol = normalize(x); }
                else iterations++;
                dirToP2 = normalize(x);
                if (dot(tempSol - dirToP2, tempSol - dirToP2) < 2 * tol) break;
            }
            if (rng.next1D() < (Float) rrweight) weight = weight / rrweight;
            else { dirToP2 = normalize(x); return false; }
            if (iterations > 64) return false;       /* safety bound (the reference has none) */
        }
        dirToP2 = dirToP2 * RIFp;
        weight *= (iterations - 1);
        FLOAT bw = 1;
        const bool okp = computePathLengths(p1, p2, dirToP2, !S.insideShape(p2), revDir, optDist, dist, bw);
        weight *= bw;
        return okp;
    }
};


/* src/libcore/util.cpp:665-695 */
inline Float fresnelDielectricExt(Float cosThetaI_, Float &cosThetaT_, Float eta) {
    if (eta == 1) { cosThetaT_ = -cosThetaI_; return 0.0f; }
    Float scale = (cosThetaI_ > 0) ? 1 / eta : eta, cosThetaTSqr = 1 - (1 - cosThetaI_ * cosThetaI_) * (scale * scale);
    if (cosThetaTSqr <= 0.0f) { cosThetaT_ = 0.0f; return 1.0f; }
    Float cosThetaI = std::abs(cosThetaI_);
    Float cosThetaT = std::sqrt(cosThetaTSqr);
    Float Rs = (cosThetaI - eta * cosThetaT) / (cosThetaI + eta * cosThetaT);
    Float Rp = (eta * cosThetaI - cosThetaT) / (eta * cosThetaI + cosThetaT);
    cosThetaT_ = (cosThetaI_ > 0) ? -cosThetaT : cosThetaT;
    return 0.5f * (Rs * Rs + Rp * Rp);
}

/* ------------------------------------------------------------------ media */
struct Walker {
    const Scene &S; Pcg32 &rng; Counters &C;
    Walker(const Scene &s, Pcg32 &r, Counters &c) : S(s), rng(r), C(c) {}
    /* transient film (N1): optical length of the last transmittance walk that reached the boundary, and the per-sample
       decomposition values (frames x RGB) the contributions are binned into (bdpt_proc.cpp:449-470 restated for volpath) */
    Float lastTrOpt = 0;
    Float *decomp = nullptr;
    /* what one path edge adds to the binned quantity: its optical length (transient) or 1 (bounce, bdpt_proc.cpp:179-187) */
    inline Float el(Float opticalLength) const { return S.s.decomposition == 2 ? (Float) 1.0f : opticalLength; }
    Spec modL = Spec(0.0f);                      /* sum of contributions x correlationFunction(pathLength) (bdpt_proc.cpp:446-447) */
    inline void contribute(const Spec &value, Float pathLength) {
        if (S.s.modulation != 0) { if (!value.isZero()) modL += value * S.correlationFunction(pathLength); return; }
        if (!decomp || value.isZero()) return;
        const Float b = std::floor((pathLength - S.s.min_bound) / S.s.bin_width);
        if (!(b >= 0) || !(b < (Float) S.frames)) return;      /* size_t binIndex >= 0 && binIndex < m_frames */
        const int bin = (int) b;
        for (int k = 0; k < 3; ++k) decomp[bin * 3 + k] += value[k];
    }

    inline Float sigmaTAt(const Vec &p) const {     /* heterogeneous.cpp:707-717 (isotropic medium) * m_scale */
        C.c[ORC_C_TENTATIVE]++;
        return S.density.lookupFloat(p) * S.s.density_scale;
    }

    /* heterogeneous.cpp:301-376: composite Simpson quadrature of the density over the ray segment [0, rayMaxt] clipped to the
       density box; HETVOL_EARLY_EXIT is defined in the reference (:31).  ray.mint = 0 as everywhere on this path. */
    Float integrateDensity(const Vec &o, const Vec &d, Float rayMaxt) {
        Float mint, maxt;
        if (!S.density.rayIntersect(o, d, mint, maxt)) return 0.0f;
        mint = std::max(mint, (Float) 0.0f);
        maxt = std::min(maxt, rayMaxt);
        Float length = maxt - mint, maxComp = 0;
        Vec p = o + d * mint, pLast = o + d * maxt;
        { const Float pc[3] = {p.x, p.y, p.z}, lc[3] = {pLast.x, pLast.y, pLast.z};
          for (int i = 0; i < 3; ++i) maxComp = std::max(std::max(maxComp, std::abs(pc[i])), std::abs(lc[i])); }
        if (length < 1e-6f * maxComp) return 0.0f;
        uint32_t nSteps = (uint32_t) std::ceil(length / S.hetStepSize);
        nSteps += nSteps % 2;
        const Float stepSize = length / nSteps;
        const Vec increment = d * stepSize;
        Float integratedDensity = lookupDensity(p) + lookupDensity(pLast);
        const Float stopAfterDensity = -std::log((Float) 1e-4f);                        /* Epsilon, single precision */
        const Float stopValue = stopAfterDensity * 3.0f / (stepSize * S.s.density_scale);
        p = p + increment;
        Float m = 4;
        for (uint32_t i = 1; i < nSteps; ++i) {
            integratedDensity += m * lookupDensity(p);
            m = 6 - m;
            if (integratedDensity > stopValue) return std::numeric_limits<Float>::infinity();
            Vec next = p + increment;
            if (p.x == next.x && p.y == next.y && p.z == next.z) break;
            p = next;
        }
        return integratedDensity * S.s.density_scale * stepSize * (1.0f / 3.0f);
    }
    inline Float lookupDensity(const Vec &p) { C.c[ORC_C_TENTATIVE]++; return S.density.lookupFloat(p); }   /* :707-717, isotropic */
    /* heterogeneous.cpp:419-544 */
    bool invertDensityIntegral(const Vec &o, const Vec &d, Float rayMaxt, Float desiredDensity, Float &integratedDensity, Float &t, Float &densityAtT) {
        integratedDensity = densityAtT = 0.0f; t = 0.0f;
        Float mint, maxt;
        if (!S.density.rayIntersect(o, d, mint, maxt)) return false;
        mint = std::max(mint, (Float) 0.0f);
        maxt = std::min(maxt, rayMaxt);
        Float length = maxt - mint, maxComp = 0;
        Vec p = o + d * mint, pLast = o + d * maxt;
        { const Float pc[3] = {p.x, p.y, p.z}, lc[3] = {pLast.x, pLast.y, pLast.z};
          for (int i = 0; i < 3; ++i) maxComp = std::max(std::max(maxComp, std::abs(pc[i])), std::abs(lc[i])); }
        if (length < 1e-6f * maxComp) return false;
        uint32_t nSteps = (uint32_t) std::ceil(length / (2 * S.hetStepSize));
        Float stepSize = length / nSteps, multiplier = (1.0f / 6.0f) * stepSize * S.s.density_scale;
        Vec fullStep = d * stepSize, halfStep = fullStep * .5f;
        Float node1 = lookupDensity(p);
        for (uint32_t i = 0; i < nSteps; ++i) {
            Float node2 = lookupDensity(p + halfStep), node3 = lookupDensity(p + fullStep),
                  newDensity = integratedDensity + multiplier * (node1 + node2 * 4 + node3);
            if (newDensity >= desiredDensity) {
                Float a = 0, b = stepSize, x = a, fx = integratedDensity - desiredDensity, stepSizeSqr = stepSize * stepSize,
                      temp = S.s.density_scale / stepSizeSqr;
                int it = 1;
                while (true) {
                    Float dfx = temp * (node1 * stepSizeSqr - (3 * node1 - 4 * node2 + node3) * stepSize * x + 2 * (node1 - 2 * node2 + node3) * x * x);
                    x -= fx / dfx;
                    if (x <= a || x >= b || dfx == 0) x = 0.5f * (b + a);
                    Float intval = integratedDensity + temp * (1.0f / 6.0f) * (x * (6 * node1 * stepSizeSqr - 3 * (3 * node1 - 4 * node2 + node3) * stepSize * x
                                   + 4 * (node1 - 2 * node2 + node3) * x * x));
                    fx = intval - desiredDensity;
                    if (std::abs(fx) < 1e-6f) {
                        t = mint + stepSize * i + x;
                        integratedDensity = intval;
                        densityAtT = temp * (node1 * stepSizeSqr - (3 * node1 - 4 * node2 + node3) * stepSize * x + 2 * (node1 - 2 * node2 + node3) * x * x);
                        return true;
                    } else if (++it > 30) return false;
                    if (fx > 0) b = x; else a = x;
                }
            }
            Vec next = p + fullStep;
            if (p.x == next.x && p.y == next.y && p.z == next.z) break;
            integratedDensity = newDensity;
            node1 = node3;
            p = next;
        }
        return false;
    }
    /* heterogeneous.cpp:589-612: sampleDistance, Simpson branch */
    bool sampleDistanceSimpson(const Vec &o, const Vec &d, Float rayMaxt, MediumRec &mRec) {
        mRec.refRatioSq = 1.0f;
        Float integratedDensity, densityAtT;
        bool success = false;
        Float desiredDensity = -std::log(1 - rng.next1D());
        if (invertDensityIntegral(o, d, rayMaxt, desiredDensity, integratedDensity, mRec.t, densityAtT)) {
            mRec.p = o + d * mRec.t;
            success = true;
            Spec albedo = S.albedoAt(mRec.p);
            mRec.sigmaS = albedo * densityAtT;
            mRec.sigmaA = Spec(densityAtT) - mRec.sigmaS;
        }
        Float expVal = std::exp(-integratedDensity);
        mRec.pdfFailure = expVal;
        mRec.pdfSuccess = expVal * densityAtT;
        mRec.transmittance = Spec(expVal);
        success = success && mRec.pdfSuccess > 0;
        if (success) C.c[ORC_C_REAL]++;
        return success;
    }
    /* (a) heterogeneous.cpp:589-663, Woodcock branch */
    bool sampleDistanceWoodcock(const Vec &o, const Vec &d, Float rayMaxt, MediumRec &mRec) {
        if (S.s.method == 1) return sampleDistanceSimpson(o, d, rayMaxt, mRec);
        mRec.pdfFailure = 1.0f; mRec.pdfSuccess = 1.0f; mRec.transmittance = Spec(1.0f); mRec.refRatioSq = 1.0f;
        Float mint, maxt;
        if (!S.density.rayIntersect(o, d, mint, maxt)) return false;
        mint = std::max(mint, (Float) 0.0f);
        maxt = std::min(maxt, rayMaxt);
        Float t = mint, densityAtT = 0;
        bool success = false;
        while (true) {
            t -= std::log(1 - rng.next1D()) * S.invMaxDensity;
            if (t >= maxt) break;
            Vec p = o + d * t;
            densityAtT = sigmaTAt(p);
            if (densityAtT * S.invMaxDensity > rng.next1D()) {
                mRec.t = t; mRec.p = p;
                Spec albedo = S.albedoAt(p);
                mRec.sigmaS = albedo * densityAtT;
                mRec.sigmaA = Spec(densityAtT) - mRec.sigmaS;
                mRec.transmittance = Spec(densityAtT != 0.0f ? 1.0f / densityAtT : 0);
                if (!std::isfinite(mRec.transmittance[0])) mRec.transmittance = Spec(0.0f);
                success = true;
                C.c[ORC_C_REAL]++;
                break;
            }
        }
        return success && mRec.pdfSuccess > 0;
    }
    /* heterogeneous.cpp:546-587 (nSamples = 2 binary estimator) + ratio tracking (departure) */
    Spec evalTransmittanceHet(const Vec &o, const Vec &d, Float rayMaxt) {
        if (S.s.method == 1) return Spec(std::exp(-integrateDensity(o, d, rayMaxt)));       /* :547-548 */
        Float mint, maxt;
        if (!S.density.rayIntersect(o, d, mint, maxt)) return Spec(1.0f);
        mint = std::max(mint, (Float) 0.0f);
        maxt = std::min(maxt, rayMaxt);
        if (S.s.tr_estimator == ORC_TR_RATIO) {
            Float T = 1.0f, t = mint;
            while (true) {
                t -= std::log(1 - rng.next1D()) * S.invMaxDensity;
                if (t >= maxt) break;
                Float density = sigmaTAt(o + d * t);
                T *= 1.0f - density * S.invMaxDensity;
                if (T == 0.0f) break;
            }
            return Spec(T);
        }
        int nSamples = 2; Float result = 0;
        for (int i = 0; i < nSamples; ++i) {
            Float t = mint;
            while (true) {
                t -= std::log(1 - rng.next1D()) * S.invMaxDensity;
                if (t >= maxt) { result += 1; break; }
                Float density = sigmaTAt(o + d * t);
                if (density * S.invMaxDensity > rng.next1D()) break;
            }
        }
        return Spec(result / nSamples);
    }

    /* strategy pdfs: homogeneous.cpp:317-343 == heterogeneousrefractive.cpp:533-558 */
    inline void strategyPdfs(Float sampledDistance, Float samplingDensity, MediumRec &mRec) const {
        switch (S.s.strategy) {
        case ORC_STRATEGY_BALANCE:
            mRec.pdfFailure = 0; mRec.pdfSuccess = 0;
            for (int i = 0; i < 3; ++i) {
                Float tmp = std::exp(-S.sigmaT[i] * sampledDistance);
                mRec.pdfFailure += tmp; mRec.pdfSuccess += S.sigmaT[i] * tmp;
            }
            mRec.pdfFailure /= 3; mRec.pdfSuccess /= 3;
            break;
        case ORC_STRATEGY_MAXIMUM:                          /* :318-320; pdfSuccess was set by MaxExpDist::sample (carried in samplingDensity) */
            mRec.pdfFailure = 1 - S.maxExp.cdf(sampledDistance);
            mRec.pdfSuccess = samplingDensity;
            break;
        default:
            mRec.pdfFailure = std::exp(-samplingDensity * sampledDistance);
            mRec.pdfSuccess = samplingDensity * mRec.pdfFailure;
        }
        for (int i = 0; i < 3; ++i) mRec.transmittance[i] = std::exp(S.sigmaT[i] * (-sampledDistance));
        mRec.pdfSuccess = mRec.pdfSuccess * S.mediumSamplingWeight;
        mRec.pdfFailure = S.mediumSamplingWeight * mRec.pdfFailure + (1 - S.mediumSamplingWeight);
        if (mRec.transmittance.max() < 1e-20f) mRec.transmittance = Spec(0.0f);
    }
    inline Float sampleExpDistance(Float &samplingDensity) {   /* homogeneous.cpp:277-296 */
        Float rand = rng.next1D(), sampledDistance;
        samplingDensity = S.samplingDensity;
        if (rand < S.mediumSamplingWeight) {
            rand /= S.mediumSamplingWeight;
            if (S.s.strategy == ORC_STRATEGY_MAXIMUM) return S.maxExp.sample(1 - rand, samplingDensity);   /* :291: the pdf rides in samplingDensity */
            if (S.s.strategy == ORC_STRATEGY_BALANCE) {
                int channel = std::min((int) (rng.next1D() * 3), 2);
                samplingDensity = S.sigmaT[channel];
            }
            sampledDistance = -std::log(1 - rand) / samplingDensity;
        } else sampledDistance = std::numeric_limits<Float>::infinity();
        return sampledDistance;
    }
    /* (b) homogeneous.cpp:275-352 */
    bool sampleDistanceHomogeneous(const Vec &o, const Vec &d, Float rayMaxt, MediumRec &mRec) {
        Float samplingDensity;
        Float sampledDistance = sampleExpDistance(samplingDensity);
        Float distSurf = rayMaxt - 0.0f;
        bool success = true;
        mRec.refRatioSq = 1.0f;
        if (sampledDistance < distSurf) {
            mRec.t = sampledDistance; mRec.p = o + d * mRec.t;
            mRec.sigmaA = S.sigmaA; mRec.sigmaS = S.sigmaS;
            if (mRec.p.x == o.x && mRec.p.y == o.y && mRec.p.z == o.z) success = false;
            else C.c[ORC_C_REAL]++;
        } else { sampledDistance = distSurf; success = false; }
        strategyPdfs(sampledDistance, samplingDensity, mRec);
        return success;
    }
    /* (c) heterogeneousrefractive.cpp:402-568 (homogeneous sigma along a curved ray) */
    template <typename FLOAT> bool sampleDistanceRefractive(const Vec &o, const Vec &d, MediumRec &mRec) {
        Float samplingDensity;
        FLOAT sampledDistance = sampleExpDistance(samplingDensity);
        Tracer<FLOAT> T(S, C);
        bool success = true;
        FLOAT distSurf = 0, opticalDistance = 0;
        V3<FLOAT> tempP(o), tempV(d);
        if (!T.R.insideVolumeLimits(tempP)) {               /* :461-466 */
            mRec.transmittance = Spec(0.0f); mRec.pdfSuccess = 1.0f; mRec.pdfFailure = 1.0f; mRec.refRatioSq = 1.0f;
            mRec.p = o; mRec.d = d;
            return false;
        }
        FLOAT refStart = T.R.value(tempP, C);
        FLOAT refRatioSq = (FLOAT) 1.0 / (refStart * refStart);
        tempV *= refStart;
        if (std::isfinite(sampledDistance)) success = T.traceMaybeAggressive(tempP, tempV, sampledDistance, distSurf, opticalDistance);
        else { T.traceTillBoundary(tempP, tempV, distSurf, opticalDistance); success = false; }
        Float refEnd = (Float) T.R.value(tempP, C);
        refRatioSq *= refEnd * refEnd;
        mRec.opticalLength = (Float) opticalDistance; mRec.p = Vec(tempP); mRec.d = Vec(tempV); mRec.refRatioSq = (Float) refRatioSq;
        if (success) {
            mRec.t = (Float) sampledDistance; mRec.sigmaA = S.sigmaA; mRec.sigmaS = S.sigmaS;
            if (mRec.p.x == o.x && mRec.p.y == o.y && mRec.p.z == o.z) success = false;
            else C.c[ORC_C_REAL]++;
        } else { sampledDistance = distSurf; mRec.t = (Float) sampledDistance; }
        strategyPdfs((Float) sampledDistance, samplingDensity, mRec);
        return success;
    }
    /* (d) composed estimator, SURVEY section 9.2: delta tracking (heterogeneous.cpp:613-659) whose free
       paths are walked with trace() (heterogeneousrefractive.cpp:671-691) */
    template <typename FLOAT> bool sampleDistanceComposed(const Vec &o, const Vec &d, MediumRec &mRec) {
        mRec.pdfFailure = 1.0f; mRec.pdfSuccess = 1.0f; mRec.transmittance = Spec(1.0f); mRec.refRatioSq = 1.0f;
        Tracer<FLOAT> T(S, C);
        V3<FLOAT> tempP(o), tempV(d);
        mRec.p = o; mRec.d = d;
        if (!T.R.insideVolumeLimits(tempP)) { mRec.transmittance = Spec(0.0f); return false; }
        FLOAT refStart = T.R.value(tempP, C);
        tempV *= refStart;
        FLOAT total = 0, opt = 0, distSurf = 0;
        bool success = false;
        while (true) {
            FLOAT s = (FLOAT) (-std::log(1 - rng.next1D()) * S.invMaxDensity);
            bool inside = T.traceMaybeAggressive(tempP, tempV, s, distSurf, opt);
            total += distSurf;
            if (!inside) break;
            Vec p(tempP);
            Float densityAtT = sigmaTAt(p);
            if (densityAtT * S.invMaxDensity > rng.next1D()) {
                Spec albedo = S.albedoAt(p);
                mRec.sigmaS = albedo * densityAtT;
                mRec.sigmaA = Spec(densityAtT) - mRec.sigmaS;
                mRec.transmittance = Spec(densityAtT != 0.0f ? 1.0f / densityAtT : 0);
                if (!std::isfinite(mRec.transmittance[0])) mRec.transmittance = Spec(0.0f);
                success = true;
                C.c[ORC_C_REAL]++;
                break;
            }
        }
        Float refEnd = (Float) T.R.value(tempP, C);
        mRec.refRatioSq = (Float) ((FLOAT) 1.0 / (refStart * refStart)) * refEnd * refEnd;
        mRec.t = (Float) total; mRec.opticalLength = (Float) opt; mRec.p = Vec(tempP); mRec.d = Vec(tempV);
        return success;
    }
    /* transmittance from (o,d) along the curved ray to the boundary (no reference analogue for volpath) */
    template <typename FLOAT> Spec evalTransmittanceCurved(const Vec &o, const Vec &d) {
        Tracer<FLOAT> T(S, C);
        if (!T.R.insideVolumeLimits(V3<FLOAT>(o))) return Spec(0.0f);
        if (S.s.sigma_mode == ORC_SIGMA_HOMOGENEOUS) {
            V3<FLOAT> p(o), v(d);
            v *= T.R.value(p, C);
            FLOAT distSurf = 0, opt = 0;
            T.traceTillBoundary(p, v, distSurf, opt);
            lastTrOpt = (Float) opt;
            Spec tr;                                           /* heterogeneousrefractive.cpp:393-400 */
            for (int i = 0; i < 3; ++i) tr[i] = S.sigmaT[i] != 0 ? std::exp(S.sigmaT[i] * (Float) (-distSurf)) : 1.0f;
            return tr;
        }
        const int nWalks = S.s.tr_estimator == ORC_TR_RATIO ? 1 : 2;
        Float result = 0;
        for (int w = 0; w < nWalks; ++w) {
            V3<FLOAT> p(o), v(d);
            v *= T.R.value(p, C);
            FLOAT distSurf = 0, opt = 0;
            Float Tr = 1.0f;
            while (true) {
                FLOAT s = (FLOAT) (-std::log(1 - rng.next1D()) * S.invMaxDensity);
                if (!T.traceMaybeAggressive(p, v, s, distSurf, opt)) { lastTrOpt = (Float) opt; break; }
                Float density = sigmaTAt(Vec(p));
                if (S.s.tr_estimator == ORC_TR_RATIO) { Tr *= 1.0f - density * S.invMaxDensity; if (Tr == 0.0f) break; }
                else if (density * S.invMaxDensity > rng.next1D()) { Tr = 0.0f; break; }
            }
            result += Tr;
        }
        return Spec(result / nWalks);
    }

    /* transmittance along the connecting ray of arc length `dist` starting at o with optical momentum v0
       (Medium::eval's transmittance, heterogeneousrefractive.cpp:617 for homogeneous sigma; delta tracking otherwise) */
    template <typename FLOAT> Spec connectionTransmittance(const Vec &o, const V3<FLOAT> &v0, FLOAT dist) {
        if (S.s.sigma_mode == ORC_SIGMA_HOMOGENEOUS) {
            Spec tr; for (int i = 0; i < 3; ++i) tr[i] = std::exp(S.sigmaT[i] * (Float) (-dist));
            return tr;
        }
        Tracer<FLOAT> T(S, C);
        const int nWalks = S.s.tr_estimator == ORC_TR_RATIO ? 1 : 2;
        Float result = 0;
        for (int w = 0; w < nWalks; ++w) {
            V3<FLOAT> p(o), v = v0; FLOAT left = dist, ds = 0, opt = 0; Float Tr = 1.0f;
            while (true) {
                FLOAT s = (FLOAT) (-std::log(1 - rng.next1D()) * S.invMaxDensity);
                if (s >= left) break;
                if (!T.trace(p, v, s, ds, opt)) break;
                left -= s;
                Float density = sigmaTAt(Vec(p));
                if (S.s.tr_estimator == ORC_TR_RATIO) { Tr *= 1.0f - density * S.invMaxDensity; if (Tr == 0.0f) break; }
                else if (density * S.invMaxDensity > rng.next1D()) { Tr = 0.0f; break; }
            }
            result += Tr;
        }
        return Spec(result / nWalks);
    }

    bool sampleDistance(const Vec &o, const Vec &d, Float maxt, MediumRec &mRec) {
        C.c[ORC_C_SEGMENTS]++;
        if (!S.curved) {
            mRec.d = d;
            return S.s.sigma_mode == ORC_SIGMA_GRID ? sampleDistanceWoodcock(o, d, maxt, mRec)
                                                    : sampleDistanceHomogeneous(o, d, maxt, mRec);
        }
        if (S.s.sigma_mode == ORC_SIGMA_GRID)
            return S.s.rif_double ? sampleDistanceComposed<double>(o, d, mRec) : sampleDistanceComposed<float>(o, d, mRec);
        return S.s.rif_double ? sampleDistanceRefractive<double>(o, d, mRec) : sampleDistanceRefractive<float>(o, d, mRec);
    }
    /* Medium::evalTransmittance over [0,maxt] (straight) or to the boundary (curved) */
    Spec evalTransmittance(const Vec &o, const Vec &d, Float maxt) {
        if (!S.curved) {
            lastTrOpt = maxt * S.s.rif_const;
            if (S.s.sigma_mode == ORC_SIGMA_GRID) return evalTransmittanceHet(o, d, maxt);
            Spec tr;                                           /* homogeneous.cpp:264-273 */
            Float negLength = 0.0f - maxt;
            for (int i = 0; i < 3; ++i) tr[i] = S.sigmaT[i] != 0 ? std::exp(S.sigmaT[i] * negLength) : 1.0f;
            return tr;
        }
        return S.s.rif_double ? evalTransmittanceCurved<double>(o, d) : evalTransmittanceCurved<float>(o, d);
    }

    /* evalTransmittance on the forked stream of a side walk (Pcg32::fork) */
    Spec sideTransmittance(uint64_t kind, const Vec &o, const Vec &d, Float maxt) {
        const Pcg32 saved = rng;
        rng = saved.fork(kind);
        const Spec tr = evalTransmittance(o, d, maxt);
        rng = saved;
        return tr;
    }
    static inline Float miWeight(Float pdfA, Float pdfB) { pdfA *= pdfA; pdfB *= pdfB; return pdfA / (pdfA + pdfB); } /* volpath.cpp:430-433 */

    /* A10: VolumetricPathTracer::Li (src/integrators/path/volpath.cpp:84-343) restricted to the scene
       {one index-matched (null BSDF) convex shape with an interior medium, constant environment emitter};
       refractive deltas from src/libbidir/edge.cpp:45-60,91-93 and src/libbidir/vertex.cpp:251-255. */
    bool dbg = false;
    Spec Li(Vec ro, Vec rd, Float rmint, Float rmaxt) {
        const orc_scene &P = S.s;
        const Spec env(P.env_radiance[0], P.env_radiance[1], P.env_radiance[2]);
        const bool dielectric = P.boundary_bsdf == ORC_BSDF_HDIELECTRIC;
        /* a dielectric boundary blocks emitter sampling and emitter look-ups from inside (Scene::evalTransmittance stops at a non-null
           surface): the environment is reached only by paths that refract out */
        const bool hasEnvAny = !env.isZero();
        const bool hasEnv = hasEnvAny && !dielectric;
        const bool hasEmission = P.emission[0] != 0 || P.emission[1] != 0 || P.emission[2] != 0;
        const Spec pointI(P.point_intensity[0], P.point_intensity[1], P.point_intensity[2]);
        const bool hasPoint = !pointI.isZero();
        const Vec pointP(P.point_position[0], P.point_position[1], P.point_position[2]);
        const bool hasArea = S.hasArea;
        const Float INF = std::numeric_limits<Float>::infinity();
        /* what a ray sees that has left the convex medium shape for good (or never meets it): the rectangle emitter if it is hit -- its front
           side emits, its back side is black, and it hides the environment either way (all-absorbing BSDF) -- else the environment */
        auto escape = [&](const Vec &o, const Vec &d, Float mint, Float &extra) -> Spec {
            extra = 0;
            if (hasArea) { const Float t = S.rectIntersect(o, d, mint, INF); if (t >= 0) { extra = t * P.rif_const; return S.rectLe(d); } }
            return env;
        };
        Spec Li(0.0f), throughput(1.0f);
        Float plen = 0;                                            /* optical path length sensor -> current vertex (transient film) */
        Float eta = 1.0f;
        bool scattered = false, medium = false, emitted = true;   /* rRec.type & EEmittedRadiance */
        int depth = 1;
        MediumRec mRec;
        Float itsT = S.intersectShape(ro, rd, rmint, rmaxt);       /* rRec.rayIntersect(ray) */
        bool itsValid = itsT >= 0;
        const int maxDepth = P.max_depth;
        if (hasArea) {
            /* the camera ray meets the rectangle before the medium shape (or instead of it): its.isEmitter() => Li += Le (volpath.cpp:203-206), then
               the all-absorbing BSDF ends the path */
            const Float tr = S.rectIntersect(ro, rd, rmint, rmaxt);
            if (tr >= 0 && (!itsValid || tr < itsT)) {
                if (!P.hide_emitters) { const Spec Le = S.rectLe(rd); Li += Le; contribute(Le, (P.calibrated_transient && P.decomposition == 1) ? 0.0f : el(tr * P.rif_const)); }
                return S.s.modulation != 0 ? modL : Li;
            }
        }

        while (depth <= maxDepth || maxDepth < 0) {
            if (medium && sampleDistance(ro, rd, itsT, mRec)) {
                plen += el(S.curved ? mRec.opticalLength : mRec.t * P.rif_const);           /* bdpt_proc.cpp:158-176 */
                if (depth >= maxDepth && maxDepth != -1) break;
                if (hasEmission && P.sigma_mode == ORC_SIGMA_GRID) {
                    /* collision estimator for volumetric emission (new, config 5): eps(p)/sigma_t(p),
                       eps = emission * density(p) * scale ; sigma_t = density(p)*scale  => ratio = emission */
                    Li += throughput * Spec(P.emission[0], P.emission[1], P.emission[2]) * mRec.refRatioSq;
                    contribute(throughput * Spec(P.emission[0], P.emission[1], P.emission[2]) * mRec.refRatioSq, plen);
                }
                throughput *= mRec.sigmaS * mRec.transmittance / mRec.pdfSuccess;       /* volpath.cpp:113 */
                if (S.curved) throughput *= mRec.refRatioSq;                              /* edge.cpp:91-93 */
                const Vec wi = S.curved ? normalize(-mRec.d) : -rd;                       /* vertex.cpp:251-255 */

                /* ---- luminaire sampling: scene.cpp:854-874 + constant.cpp:179-214 */
                if (hasEnv) {
                    C.c[ORC_C_NEE]++;
                    int interactions = maxDepth - depth - 1;
                    Float s2x = rng.next1D(), s2y = rng.next1D();
                    Vec dd = squareToUniformSphere(s2x, s2y);
                    Float dpdf = INV_FOURPI_F;
                    Spec value = env / dpdf;
                    Spec tr(0.0f);
                    lastTrOpt = 0;
                    if (interactions != 0) {                 /* scene.cpp:619-678: one null crossing is needed */
                        Float tExit = 0;
                        if (!S.curved) {
                            tExit = S.intersectShape(mRec.p, dd, 0.0f, std::numeric_limits<Float>::infinity());
                            tr = tExit >= 0 ? sideTransmittance(1, mRec.p, dd, tExit) : Spec(1.0f);
                        } else tr = sideTransmittance(1, mRec.p, dd, 0);
                    }
                    if (hasArea && S.rectIntersect(mRec.p, dd, 0.0f, INF) >= 0) tr = Spec(0.0f);     /* the rectangle shadows the environment (Scene::evalTransmittance stops at a non-null surface) */
                    value *= tr;
                    if (dbg) printf("orc NEE depth=%d T=%g L=%g tr=%g rng=%llu\n", depth, throughput[0], Li[0], tr[0], (unsigned long long) rng.state);
                    if (!value.isZero()) {
                        Float phaseVal = phaseEval(P.phase, P.g, wi, dd);
                        if (phaseVal != 0) {
                            Float phasePdf = phaseVal;       /* env emitter isOnSurface: constant.cpp:47 */
                            Float weight = miWeight(dpdf, phasePdf);
                            Li += throughput * value * phaseVal * weight;
                            contribute(throughput * value * phaseVal * weight, plen + el(lastTrOpt));
                        }
                    }
                }
                /* ---- luminaire sampling of a point emitter: src/emitters/point.cpp sampleDirect (pdf 1, EDiscrete => no MIS) */
                if (hasPoint) {
                    C.c[ORC_C_NEE]++;
                    const int interactions = maxDepth - depth - 1;
                    if (!S.curved) {
                        Vec dvec = pointP - mRec.p;
                        const Float dist = std::sqrt(dot(dvec, dvec)), invDist = 1.0f / dist;
                        dvec *= invDist;
                        Spec value = pointI * (invDist * invDist);
                        const Float tExit = S.intersectShape(mRec.p, dvec, 0.0f, std::numeric_limits<Float>::infinity());
                        const bool crosses = tExit >= 0 && tExit < dist;          /* emitter outside the shape: one null crossing */
                        Spec tr(1.0f);
                        if (crosses && interactions == 0) tr = Spec(0.0f);
                        else tr = evalTransmittance(mRec.p, dvec, crosses ? tExit : dist);
                        value *= tr;
                        if (!value.isZero()) {
                            Li += throughput * value * phaseEval(P.phase, P.g, wi, dvec);
                            contribute(throughput * value * phaseEval(P.phase, P.g, wi, dvec), plen + el(dist * P.rif_const));
                        }
                    } else {
                        /* connection through the RIF: Medium::eval (heterogeneousrefractive.cpp:571-640) */
                        bool ok; Float w = 1, dist = 0, optD = 0; Vec dir(0, 0, 1); Spec tr(0.0f);
                        if (S.s.rif_double) {
                            Connector<double> K(S, C, rng); V3<double> d2, rev; double ww = 1, od = 0, di = 0;
                            V3<double> a(mRec.p), b(pointP);
                            ok = K.makeDirectConnections(a, b, normalize(b - a), ww, d2, rev, od, di);
                            if (ok) { w = (Float) ww; dist = (Float) di; optD = (Float) od; dir = Vec(d2); tr = connectionTransmittance<double>(mRec.p, d2, di); }
                        } else {
                            Connector<float> K(S, C, rng); V3<float> d2, rev; float ww = 1, od = 0, di = 0;
                            V3<float> a(mRec.p), b(pointP);
                            ok = K.makeDirectConnections(a, b, normalize(b - a), ww, d2, rev, od, di);
                            if (ok) { w = ww; dist = di; optD = od; dir = d2; tr = connectionTransmittance<float>(mRec.p, d2, di); }
                        }
                        if (ok && !tr.isZero()) {
                            /* PointEmitter::sampleDirect (src/emitters/point.cpp): I / |p - ref|^2 with the straight-line distance -- the
                               reference keeps it for curved connections (bdpt_proc.cpp sampleDirect + pathConnectAndCollapse) */
                            const Vec dv = pointP - mRec.p;
                            const Float invDist = 1.0f / std::sqrt(dot(dv, dv));
                            Spec value = pointI * (invDist * invDist) * tr * w;
                            Li += throughput * value * phaseEval(P.phase, P.g, wi, normalize(dir));
                            contribute(throughput * value * phaseEval(P.phase, P.g, wi, normalize(dir)), plen + el(optD));
                        }
                    }
                }
                /* ---- luminaire sampling of the area emitter: scene.cpp:854-874 + area.cpp:162-177 + shape.cpp:102-115; MIS partner = phase sampling */
                if (hasArea) {
                    C.c[ORC_C_NEE]++;
                    const int interactions = maxDepth - depth - 1;
                    const Float sx = rng.next1D(), sy = rng.next1D();
                    Vec dvec; Float dist, dpdf;
                    Spec value = S.rectSampleDirect(mRec.p, sx, sy, dvec, dist, dpdf);
                    if (!value.isZero()) {
                        /* the segment crosses the (null) boundary of the convex medium shape once on its way out (scene.cpp:619-678) */
                        const Float tExit = S.intersectShape(mRec.p, dvec, 0.0f, INF);
                        const bool crosses = tExit >= 0 && tExit < dist;
                        Spec tr(1.0f);
                        if (crosses && interactions == 0) tr = Spec(0.0f);
                        else tr = evalTransmittance(mRec.p, dvec, crosses ? tExit : dist);
                        value *= tr;
                        if (!value.isZero()) {
                            const Float phaseVal = phaseEval(P.phase, P.g, wi, dvec);
                            if (phaseVal != 0) {
                                const Float weight = miWeight(dpdf, phaseVal);       /* emitter->isOnSurface(), solid-angle measure: phasePdf = phase->pdf = its value */
                                Li += throughput * value * phaseVal * weight;
                                contribute(throughput * value * phaseVal * weight, plen + el(dist * P.rif_const));
                            }
                        }
                    }
                }
                /* ---- phase function sampling: volpath.cpp:149-158 */
                Float phasePdf; Vec wo;
                Float p2x = rng.next1D(), p2y = rng.next1D();
                Float phaseVal = phaseSample(P.phase, P.g, wi, p2x, p2y, wo, phasePdf);
                if (phaseVal == 0) break;
                throughput *= phaseVal;
                ro = mRec.p; rd = wo;
                /* rayIntersectAndLookForEmitter: volpath.cpp:370-428 */
                if (S.curved) { itsT = 0; itsValid = true; }
                else { itsT = S.intersectShape(ro, rd, 0.0f, std::numeric_limits<Float>::infinity()); itsValid = itsT >= 0; }
                if (hasEnv || hasArea) {
                    Spec tr(1.0f);
                    lastTrOpt = 0;
                    int maxInteractions = maxDepth - depth - 1;
                    bool blocked = false;
                    if (!S.curved) {
                        if (itsValid) { tr = sideTransmittance(2, ro, rd, itsT); if (maxInteractions == 0) blocked = true; }
                    } else {
                        tr = sideTransmittance(2, ro, rd, 0);
                        if (maxInteractions == 0) blocked = true;
                    }
                    if (dbg) printf("orc LOOKUP depth=%d T=%g L=%g tr=%g rng=%llu\n", depth, throughput[0], Li[0], tr[0], (unsigned long long) rng.state);
                    if (!blocked && !tr.isZero()) {
                        Spec value = tr * env;
                        Float emitterPdf = INV_FOURPI_F, extra = 0;
                        if (hasArea) {
                            /* rayIntersectAndLookForEmitter (volpath.cpp:370-428): beyond the null boundary the ray meets the rectangle or the environment */
                            const Float tRect = S.rectIntersect(ro, rd, 0.0f, INF);
                            if (tRect >= 0) { value = tr * S.rectLe(rd); emitterPdf = S.rectPdfDirect(rd, tRect); extra = (tRect - (itsValid ? itsT : 0)) * P.rif_const; }
                            else if (!hasEnv) value = Spec(0.0f);
                        }
                        if (!value.isZero()) {
                            Li += throughput * value * miWeight(phasePdf, emitterPdf);
                            contribute(throughput * value * miWeight(phasePdf, emitterPdf), plen + el(lastTrOpt + extra));
                        }
                    }
                }
                emitted = false;                              /* ERadianceNoEmission */
            } else {
                if (medium) {
                    plen += el(S.curved ? mRec.opticalLength : itsT * P.rif_const);
                    throughput *= mRec.transmittance / mRec.pdfFailure;                  /* volpath.cpp:188-189 */
                    if (S.curved) {
                        throughput *= mRec.refRatioSq;
                        /* edge.cpp:45-60: boundary re-hit at mRec.p along mRec.d */
                        ro = mRec.p; rd = normalize(mRec.d); itsValid = true; itsT = 0;
                    }
                }
                if (!itsValid) {
                    if (emitted && (!P.hide_emitters || scattered)) {                                    /* volpath.cpp:194-201 (environment), :203-206 (its.isEmitter()) */
                        Float extra; const Spec Le = escape(ro, rd, 0.0f, extra);
                        Li += throughput * Le; contribute(throughput * Le, plen + (P.decomposition == 2 ? 0.0f : extra));     /* the free-space leg to the rectangle: optical length, no further bounce */
                    }
                    break;
                }
                if (depth >= maxDepth && maxDepth != -1) break;
                if (dielectric) {
                    /* hdielectric boundary (N2): BSDF sampling of volpath.cpp:259-316 with HDielectric::sample (hdielectric.cpp:183-242);
                       not smooth => no luminaire sampling; only sample.x is used (:196) */
                    const Float u1 = rng.next1D(); (void) rng.next1D();
                    if (S.curved && medium) {                  /* edge.cpp:45-60: the surface point is re-found from mRec.p along mRec.d */
                        const Float tHit = S.intersectShape(ro, rd, 0.0f, std::numeric_limits<Float>::infinity());
                        itsT = tHit >= 0 ? tHit : 0;
                    }
                    if (!medium && !(P.calibrated_transient && P.decomposition == 1)) plen += el(itsT);
                    const Vec x = ro + rd * itsT;
                    const Vec n = S.shapeNormal(x);
                    const Float cosI = dot(-rd, n);            /* Frame::cosTheta(bRec.wi), wi = -ray.d */
                    const Float etaB = S.boundaryEta(x), invEtaB = 1 / etaB;          /* hdielectric.cpp:115-118 */
                    Float cosT; const Float F = fresnelDielectricExt(cosI, cosT, etaB);
                    if (dbg) printf("orc DIEL depth=%d medium=%d itsT=%g x=(%g %g %g) n=(%g %g %g) cosI=%g eta=%g F=%g u1=%g T=%g\n", depth, (int) medium, itsT, x.x, x.y, x.z, n.x, n.y, n.z, cosI, etaB, F, u1, throughput[0]);
                    Vec wo; bool inside;
                    if (u1 <= F) {                             /* reflect(wi) = (-x,-y,z) locally: 2 (wi.n) n - wi */
                        wo = rd + n * (2 * cosI);
                        inside = medium;
                    } else {                                   /* refract (:121-126): (scale wi.x, scale wi.y, cosThetaT) locally */
                        const Float scale = -(cosT < 0 ? invEtaB : etaB);
                        const Vec wi = -rd;
                        wo = (wi - n * cosI) * scale + n * cosT;
                        const Float factor = cosT < 0 ? invEtaB : etaB;               /* ERadiance: solid-angle compression (:213-216) */
                        throughput *= factor * factor;
                        eta *= (cosT < 0 ? etaB : invEtaB);                            /* bRec.eta (:208) */
                        inside = cosT < 0;
                    }
                    ro = x; rd = wo; medium = inside;
                    if (!medium) {
                        /* rayIntersectAndLookForEmitter: the shape is convex, the ray escapes to the environment; delta BSDF => weight 1 */
                        if (hasEnvAny) { Li += throughput * env; contribute(throughput * env, plen); }
                        itsValid = false; itsT = -1;
                    } else {
                        itsT = S.curved ? 0 : S.intersectShape(ro, rd, Epsilon, std::numeric_limits<Float>::infinity());
                        itsValid = S.curved ? true : itsT >= 0;
                        if (!itsValid) medium = false;
                    }
                    emitted = false;                           /* ERadianceNoEmission (volpath.cpp:322) */
                } else {
                /* null BSDF (shape.cpp:48-70): no NEE (not smooth), pass-through sample */
                (void) rng.next1D(); (void) rng.next1D();     /* bsdf->sample(bRec, pdf, rRec.nextSample2D()) */
                if (!medium && !(P.calibrated_transient && P.decomposition == 1)) plen += el(itsT);  /* the camera edge (bdpt_proc.cpp:163-176: startIndex 2 | 3) */
                ro = ro + rd * itsT;
                medium = !medium;
                emitted = !scattered;                         /* volpath.cpp:293-301 */
                if (medium) {
                    itsT = S.curved ? 0 : S.intersectShape(ro, rd, Epsilon, std::numeric_limits<Float>::infinity());
                    itsValid = S.curved ? true : itsT >= 0;
                    if (!itsValid) { medium = false; }        /* grazing hit: no exit found */
                } else { itsValid = false; itsT = -1; }
                depth++;
                continue;
                }
            }
            if (depth++ >= P.rr_depth) {                      /* volpath.cpp:326-336 */
                Float q = std::min(throughput.max() * eta * eta, (Float) 0.95f);
                if (rng.next1D() >= q) break;
                throughput /= q;
            }
            scattered = true;
        }
        return S.s.modulation != 0 ? modL : Li;
    }
};

/* ------------------------------------------------------------------ A11 pixel loop + film */
/* include/mitsuba/render/imageblock.h:124-205 with block = whole image (borders cropped) */
inline bool filmPut(const Scene &S, float *film, Float px, Float py, const Float *temp) {
    const int ch = S.frames * 3 + 2;                  /* RGB per frame, alpha, weight (bdpt_proc.cpp:230-245, :484-485) */
    for (int i = 0; i < ch; ++i) if (!std::isfinite(temp[i])) return false;
    const int W = S.s.width, H = S.s.height;
    const Float posx = px - 0.5f, posy = py - 0.5f, r = S.fradius;
    const int minx = std::max((int) std::ceil(posx - r), 0), miny = std::max((int) std::ceil(posy - r), 0),
              maxx = std::min((int) std::floor(posx + r), W - 1), maxy = std::min((int) std::floor(posy + r), H - 1);
    Float wx[16], wy[16];
    for (int x = minx, idx = 0; x <= maxx; ++x) wx[idx++] = S.evalDiscretized(x - posx);
    for (int y = miny, idx = 0; y <= maxy; ++y) wy[idx++] = S.evalDiscretized(y - posy);
    for (int y = miny, yr = 0; y <= maxy; ++y, ++yr) {
        const Float weightY = wy[yr];
        float *dest = film + ((size_t) y * W + minx) * ch;
        for (int x = minx, xr = 0; x <= maxx; ++x, ++xr) {
            const Float weight = wx[xr] * weightY;
            for (int k = 0; k < ch; ++k) *dest++ += weight * temp[k];
        }
    }
    return true;
}

struct SceneHolder { Scene S; bool ok; SceneHolder(const orc_scene *s) { ok = S.configure(*s); } };

}  // namespace

template <typename FLOAT> static void bsplineBuild(const float *data, const int32_t N[3], FLOAT *coeff) {
    Spline3<FLOAT> sp; FLOAT mn[3] = {0, 0, 0}, mx[3] = {1, 1, 1}; int n[3] = {N[0], N[1], N[2]};
    sp.initialize(mn, mx, n);
    size_t tot = (size_t) N[0] * N[1] * N[2];
    std::vector<FLOAT> tmp(tot);
    for (size_t i = 0; i < tot; i++) tmp[i] = (FLOAT) data[i];
    sp.build3d(tmp.data(), coeff);
}
template <typename FLOAT> static void bsplineEval(const FLOAT *coeff, const int32_t N[3], const float xmin[3], const float xmax[3],
                                                   const FLOAT *pts, int64_t n, FLOAT *val, FLOAT *grad, FLOAT *hess) {
    Spline3<FLOAT> sp; FLOAT mn[3], mx[3]; int nn[3];
    for (int i = 0; i < 3; i++) { mn[i] = xmin[i]; mx[i] = xmax[i]; nn[i] = N[i]; }
    sp.initialize(mn, mx, nn); sp.coeff = coeff;
    for (int64_t i = 0; i < n; i++) {
        FLOAT x[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]}, f; V3<FLOAT> g;
        if (hess) sp.valueGradientAndHessian(x, f, g, hess + 9 * i); else sp.valueAndGradient(x, f, g);
        val[i] = f; grad[3 * i] = g.x; grad[3 * i + 1] = g.y; grad[3 * i + 2] = g.z;
    }
}
/* =========================================================================== C exports */
extern "C" {

void orc_correlation(const orc_scene *s, const float *path_length, int64_t n, float *out) {
    SceneHolder H(s);
    for (int64_t i = 0; i < n; i++) out[i] = H.ok ? H.S.correlationFunction(path_length[i]) : 0.0f;
}
int32_t orc_film_channels(const orc_scene *s) { SceneHolder H(s); return H.ok ? H.S.frames * 3 + 2 : -1; }
const char *orc_last_error(void) { return g_err.c_str(); }

void orc_lookup_trilinear(const orc_grid *g, const float *pts, int64_t n, float *out_val, int32_t *out_idx) {
    Grid G; G.configure(*g);
    for (int64_t i = 0; i < n; i++) {
        int idx[4];
        out_val[i] = G.lookupFloat(Vec(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]), idx);
        if (out_idx) for (int k = 0; k < 4; k++) out_idx[4 * i + k] = idx[k];
    }
}
void orc_lookup_trilinear_rgb(const orc_grid *g, const float *pts, int64_t n, float *out_rgb) {
    Grid G; G.configure(*g);
    for (int64_t i = 0; i < n; i++) {
        Spec s = G.lookupSpectrum(Vec(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]));
        for (int k = 0; k < 3; k++) out_rgb[3 * i + k] = s[k];
    }
}
void orc_trilinear_value_grad(const orc_grid *g, const float *pts, int64_t n, float *out_val, float *out_grad) {
    Grid G; G.configure(*g);
    for (int64_t i = 0; i < n; i++) {
        float v; V3<float> gr;
        trilinearValueGrad<float>(G, V3<float>(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]), v, gr);
        out_val[i] = v; out_grad[3 * i] = gr.x; out_grad[3 * i + 1] = gr.y; out_grad[3 * i + 2] = gr.z;
    }
}
void orc_bspline_build_f32(const float *data, const int32_t N[3], float *coeff) { bsplineBuild<float>(data, N, coeff); }
void orc_bspline_build_f64(const float *data, const int32_t N[3], double *coeff) { bsplineBuild<double>(data, N, coeff); }
void orc_bspline_eval_f32(const float *coeff, const int32_t N[3], const float xmin[3], const float xmax[3],
                          const float *pts, int64_t n, float *val, float *grad, float *hess) {
    bsplineEval<float>(coeff, N, xmin, xmax, pts, n, val, grad, hess);
}
void orc_bspline_eval_f64(const double *coeff, const int32_t N[3], const float xmin[3], const float xmax[3],
                          const double *pts, int64_t n, double *val, double *grad, double *hess) {
    bsplineEval<double>(coeff, N, xmin, xmax, pts, n, val, grad, hess);
}

/* the scene's RIF (any rif_mode) at n points: value, gradient[3], Hessian[9] row-major; fp64 when rif_double */
void orc_rif_eval(const orc_scene *s, const float *pts, int64_t n, double *val, double *grad, double *hess) {
    SceneHolder H(s); Counters C;
    for (int64_t i = 0; i < n; i++) {
        if (s->rif_double) {
            V3<double> p(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]), g; double f, M[9];
            H.S.rifD.valueGradientAndHessian(p, f, g, M, C);
            val[i] = f; grad[3 * i] = g.x; grad[3 * i + 1] = g.y; grad[3 * i + 2] = g.z;
            for (int k = 0; k < 9; k++) hess[9 * i + k] = M[k];
        } else {
            V3<float> p(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]), g; float f, M[9];
            H.S.rifF.valueGradientAndHessian(p, f, g, M, C);
            val[i] = f; grad[3 * i] = g.x; grad[3 * i + 1] = g.y; grad[3 * i + 2] = g.z;
            for (int k = 0; k < 9; k++) hess[9 * i + k] = M[k];
        }
    }
}

void orc_er_trace(const orc_scene *s, const float *p0, const float *d0, const float *dist, int64_t n,
                  float *out_p, float *out_v, float *out_dist_surf, float *out_opt, int32_t *out_success) {
    SceneHolder H(s); Counters C;
    for (int64_t i = 0; i < n; i++) {
        Vec o(p0[3 * i], p0[3 * i + 1], p0[3 * i + 2]), d(d0[3 * i], d0[3 * i + 1], d0[3 * i + 2]);
        bool ok; Vec P, V; Float ds, op;
        if (s->rif_double) {
            Tracer<double> T(H.S, C); V3<double> p(o), v(d); v *= T.R.value(p, C); double a = 0, b = 0;
            if (std::isfinite(dist[i])) ok = T.trace(p, v, dist[i], a, b); else { T.traceTillBoundary(p, v, a, b); ok = false; }
            P = Vec(p); V = Vec(v); ds = (Float) a; op = (Float) b;
        } else {
            Tracer<float> T(H.S, C); V3<float> p(o), v(d); v *= T.R.value(p, C); float a = 0, b = 0;
            if (std::isfinite(dist[i])) ok = T.trace(p, v, dist[i], a, b); else { T.traceTillBoundary(p, v, a, b); ok = false; }
            P = p; V = v; ds = a; op = b;
        }
        out_p[3 * i] = P.x; out_p[3 * i + 1] = P.y; out_p[3 * i + 2] = P.z;
        out_v[3 * i] = V.x; out_v[3 * i + 1] = V.y; out_v[3 * i + 2] = V.z;
        out_dist_surf[i] = ds; out_opt[i] = op; out_success[i] = ok ? 1 : 0;
    }
}

void orc_sample_distance(const orc_scene *s, const float *o, const float *d, const float *maxt, int64_t n,
                         uint64_t seed, float *rec) {
    SceneHolder H(s); Counters C;
    for (int64_t i = 0; i < n; i++) {
        Pcg32 rng; rng.seed(seed, (uint32_t) i, 0);
        Walker W(H.S, rng, C);
        MediumRec m; std::memset(&m, 0, sizeof(m));
        bool ok = W.sampleDistance(Vec(o[3 * i], o[3 * i + 1], o[3 * i + 2]), Vec(d[3 * i], d[3 * i + 1], d[3 * i + 2]), maxt[i], m);
        float *r = rec + 20 * i;
        r[0] = ok ? 1.0f : 0.0f; r[1] = m.t; r[2] = m.p.x; r[3] = m.p.y; r[4] = m.p.z;
        for (int k = 0; k < 3; k++) { r[5 + k] = ok ? m.sigmaS[k] : 0.0f; r[8 + k] = m.transmittance[k]; }
        r[11] = m.pdfSuccess; r[12] = m.pdfFailure; r[13] = m.refRatioSq; r[14] = m.d.x; r[15] = m.d.y; r[16] = m.d.z;
        r[17] = r[18] = r[19] = 0;
    }
}

void orc_eval_transmittance(const orc_scene *s, const float *o, const float *d, const float *maxt, int64_t n,
                            uint64_t seed, float *out_tr) {
    SceneHolder H(s); Counters C;
    for (int64_t i = 0; i < n; i++) {
        Pcg32 rng; rng.seed(seed, (uint32_t) i, 0);
        Walker W(H.S, rng, C);
        Spec t = W.evalTransmittance(Vec(o[3 * i], o[3 * i + 1], o[3 * i + 2]), Vec(d[3 * i], d[3 * i + 1], d[3 * i + 2]), maxt[i]);
        for (int k = 0; k < 3; k++) out_tr[3 * i + k] = t[k];
    }
}

void orc_phase_sample(int32_t phase, float g, const float *wi, const float *u2, int64_t n, float *wo, float *pdf) {
    for (int64_t i = 0; i < n; i++) {
        Vec o; Float p;
        phaseSample(phase, g, Vec(wi[3 * i], wi[3 * i + 1], wi[3 * i + 2]), u2[2 * i], u2[2 * i + 1], o, p);
        wo[3 * i] = o.x; wo[3 * i + 1] = o.y; wo[3 * i + 2] = o.z; pdf[i] = p;
    }
}
void orc_phase_eval(int32_t phase, float g, const float *wi, const float *wo, int64_t n, float *val) {
    for (int64_t i = 0; i < n; i++)
        val[i] = phaseEval(phase, g, Vec(wi[3 * i], wi[3 * i + 1], wi[3 * i + 2]), Vec(wo[3 * i], wo[3 * i + 1], wo[3 * i + 2]));
}
void orc_camera_rays(const orc_scene *s, const float *pos2, int64_t n, float *o, float *d) {
    SceneHolder H(s);
    for (int64_t i = 0; i < n; i++) {
        Vec oo, dd; Float a, b;
        H.S.sampleRay(pos2[2 * i], pos2[2 * i + 1], oo, dd, a, b);
        o[3 * i] = oo.x; o[3 * i + 1] = oo.y; o[3 * i + 2] = oo.z; d[3 * i] = dd.x; d[3 * i + 1] = dd.y; d[3 * i + 2] = dd.z;
    }
}
void orc_filter_table(int32_t rfilter, float param, float *values33, float *radius, float *scale) {
    Scene::filterTable(rfilter, param, values33, *radius, *scale);
}
/* A12 leaf: connect p1 -> p2 through the RIF.  out stride 12: ok, weight, dirToP2(3, momentum at p1), revDirToP1(3), dist, opticalDist, 0, 0 */
void orc_connect(const orc_scene *s, const float *p1, const float *p2, int64_t n, uint64_t seed, float *out) {
    SceneHolder H(s); Counters C;
    for (int64_t i = 0; i < n; i++) {
        Pcg32 rng; rng.seed(seed, (uint32_t) i, 0);
        float *o = out + 12 * i;
        for (int k = 0; k < 12; k++) o[k] = 0;
        const uint64_t steps0 = C.c[ORC_C_STEPS];
        const Vec a(p1[3 * i], p1[3 * i + 1], p1[3 * i + 2]), b(p2[3 * i], p2[3 * i + 1], p2[3 * i + 2]);
        if (s->rif_double) {
            Connector<double> K(H.S, C, rng);
            V3<double> A(a), B(b), dir, rev; double w = 1, od = 0, di = 0;
            const bool ok = K.makeDirectConnections(A, B, normalize(B - A), w, dir, rev, od, di);
            o[0] = ok; o[1] = (float) w; o[2] = (float) dir.x; o[3] = (float) dir.y; o[4] = (float) dir.z;
            if (ok) { o[5] = (float) rev.x; o[6] = (float) rev.y; o[7] = (float) rev.z; o[8] = (float) di; o[9] = (float) od; }
        } else {
            Connector<float> K(H.S, C, rng);
            V3<float> A(a), B(b), dir, rev; float w = 1, od = 0, di = 0;
            const bool ok = K.makeDirectConnections(A, B, normalize(B - A), w, dir, rev, od, di);
            o[0] = ok; o[1] = w; o[2] = dir.x; o[3] = dir.y; o[4] = dir.z;
            if (ok) { o[5] = rev.x; o[6] = rev.y; o[7] = rev.z; o[8] = di; o[9] = od; }
        }
        o[10] = (float) (C.c[ORC_C_STEPS] - steps0);           /* eikonal / sensitivity steps this connection cost */
    }
}

/* MaxExpDist known answers: out[4*i..] = sample(u[i]) -> t, its pdf, pdf(t), cdf(t) */
int orc_maxexp(const float sigma_t[3], const float *u, int64_t n, float *out) {
    MaxExpDist m; Float c[3] = {sigma_t[0], sigma_t[1], sigma_t[2]};
    if (!m.configure(c)) { g_err = "Internal error: sigmaT must vary across channels"; return 1; }
    for (int64_t i = 0; i < n; i++) { Float pdf; const Float t = m.sample(u[i], pdf); out[4 * i] = t; out[4 * i + 1] = pdf; out[4 * i + 2] = m.pdf(t); out[4 * i + 3] = m.cdf(t); }
    return 0;
}

void orc_rng_floats(uint64_t seed, uint32_t pixel, uint32_t sample, int32_t n, float *out) {
    Pcg32 r; r.seed(seed, pixel, sample);
    for (int i = 0; i < n; i++) out[i] = r.next1D();
}

static void renderRows(const Scene &S, int spp_begin, int spp_count, uint64_t seed, std::atomic<int> &nextRow, int y0, int y1,
                       float *film, Counters &C, float *pathOut) {
    const int W = S.s.width;
    const int ch = S.frames * 3 + 2;
    std::vector<Float> temp((size_t) ch, 0.0f);
    while (true) {
        int y = nextRow.fetch_add(1);
        if (y >= y1) break;
        if (y < y0) continue;
        for (int x = 0; x < W; x++) {
            for (int sidx = spp_begin; sidx < spp_begin + spp_count; sidx++) {
                /* integrator.cpp:162-187 */
                Pcg32 rng; rng.seed(seed, (uint32_t) (y * W + x), (uint32_t) sidx);
                Walker Wk(S, rng, C);
                { const char *e = getenv("ORC_DEBUG_PIXEL"); Wk.dbg = e && atoi(e) == y * W + x; }
                Float sx = rng.next1D(), sy = rng.next1D();
                Float px = (Float) x + sx, py = (Float) y + sy;
                Vec o, d; Float mint, maxt;
                S.sampleRay(px, py, o, d, mint, maxt);
                std::fill(temp.begin(), temp.end(), 0.0f);
                if (S.s.decomposition != 0 && S.s.modulation == 0) Wk.decomp = temp.data();
                Spec L = Wk.Li(o, d, mint, maxt);
                C.c[ORC_C_PATHS]++;
                if (pathOut) { float *q = pathOut + ((size_t) y * W + x) * 3; q[0] = L[0]; q[1] = L[1]; q[2] = L[2]; }
                else {
                    if (S.s.decomposition == 0 || S.s.modulation != 0) { temp[0] = L[0]; temp[1] = L[1]; temp[2] = L[2]; }
                    temp[ch - 2] = 1.0f; temp[ch - 1] = 1.0f;
                    filmPut(S, film, px, py, temp.data());
                }
            }
        }
    }
}

static int renderImpl(const orc_scene *s, int spp_begin, int spp_count, uint64_t seed, int y0, int y1, int nthreads,
                      float *film, uint64_t *counters, float *pathOut) {
    SceneHolder H(s);
    if (!H.ok) return 1;
    const int W = s->width, Hh = s->height;
    if (y1 > Hh) y1 = Hh;
    if (nthreads < 1) nthreads = 1;
    std::atomic<int> nextRow(y0);
    std::vector<Counters> Cs(nthreads);
    std::vector<std::vector<float>> films(nthreads);
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; t++) {
        if (!pathOut) films[t].assign((size_t) W * Hh * (H.S.frames * 3 + 2), 0.0f);
        th.emplace_back([&, t]() { renderRows(H.S, spp_begin, spp_count, seed, nextRow, y0, y1, pathOut ? NULL : films[t].data(), Cs[t], pathOut); });
    }
    for (auto &t : th) t.join();
    if (!pathOut)
        for (int t = 0; t < nthreads; t++)
            for (size_t i = 0; i < (size_t) W * Hh * (H.S.frames * 3 + 2); i++) film[i] += films[t][i];
    if (counters) {
        for (int k = 0; k < ORC_C_COUNT; k++) { counters[k] = 0; for (int t = 0; t < nthreads; t++) counters[k] += Cs[t].c[k]; }
    }
    return 0;
}

int orc_render(const orc_scene *s, int32_t spp_begin, int32_t spp_count, uint64_t seed, int32_t y0, int32_t y1,
               int32_t nthreads, float *film, uint64_t *counters) {
    return renderImpl(s, spp_begin, spp_count, seed, y0, y1, nthreads, film, counters, NULL);
}
int orc_render_paths(const orc_scene *s, int32_t sample_index, uint64_t seed, int32_t nthreads, float *out_rgb) {
    return renderImpl(s, sample_index, 1, seed, 0, s->height, nthreads, NULL, NULL, out_rgb);
}

}  // extern "C"
