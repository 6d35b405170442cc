// mer_connect.hpp -- K_connect: curved-ray connection p1 -> p2 through the refractive-index field.
//
// Reference: HeterogeneousRefractiveMedium::eval -> makeDirectConnections -> computefdfBDPT / er_derivativestep /
// computePathLengthsTillClosestP2 (src/medium/heterogeneousrefractive.cpp:571-640, 798-1030, 1087-1163).  The shooting
// problem "find the initial optical momentum v0 (|v0| = n(p1)) whose eikonal ray passes through p2" is solved per lane:
// residual r(v0) = p(t*) - p2 at the closest approach t* (sign change of (p-p2).v, bisected), analytic Jacobian from the
// 3x3 sensitivities dp/dv0, dv/dv0 integrated with the Hessian of n, multi-restart with Russian roulette and the
// (iterations-1) solution-count weight exactly as the reference.  The reference minimises with Ceres LINE_SEARCH/BFGS
// (un-vendored, unpinned); here a Levenberg-damped Gauss-Newton on the same residual/Jacobian -- parity unpinned for the
// iterates, pinned on the converged ray.  A ray that leaves the medium shape is refracted at the boundary (Snell's law and its
// Jacobian, :873-919, :1036-1074) and continued straight in the exterior: connections to a point outside the shape cross it.
#pragma once
#include "mer_walk.hpp"

namespace mer {

struct m33 {
    float m[3][3];
    __device__ __forceinline__ m33() {}
    __device__ __forceinline__ explicit m33(float d) {
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) m[i][j] = i == j ? d : 0.0f;
    }
};
__device__ __forceinline__ m33 outer(f3 a, f3 b) {            // Matrix3x3(v1, v2): include/mitsuba/core/matrix.h:716-720
    m33 r; const float A[3] = {a.x, a.y, a.z}, B[3] = {b.x, b.y, b.z};
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) r.m[i][j] = A[i] * B[j];
    return r;
}
__device__ __forceinline__ m33 mul(const m33 &a, const m33 &b) {
    m33 r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) { float s = 0; for (int k = 0; k < 3; k++) s += a.m[i][k] * b.m[k][j]; r.m[i][j] = s; }
    return r;
}
__device__ __forceinline__ m33 scale(const m33 &a, float s) { m33 r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) r.m[i][j] = a.m[i][j] * s;
    return r; }
__device__ __forceinline__ m33 add(const m33 &a, const m33 &b) { m33 r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) r.m[i][j] = a.m[i][j] + b.m[i][j];
    return r; }
__device__ __forceinline__ f3 premult(const m33 &M, f3 v) {   // matrix.h:640-644: v^T M
    return f3(v.x * M.m[0][0] + v.y * M.m[1][0] + v.z * M.m[2][0], v.x * M.m[0][1] + v.y * M.m[1][1] + v.z * M.m[2][1],
              v.x * M.m[0][2] + v.y * M.m[1][2] + v.z * M.m[2][2]);
}

// Second-derivative B-spline weights (include/mitsuba/core/basisspline.h:91-97: |x|<=1: 3|x|-2 ; 1<|x|<=2: 2-|x|)
__device__ __forceinline__ void bspline_weights2(float t, float d2[4]) {
    d2[0] = 2.0f - (t + 1.0f); d2[1] = 3.0f * t - 2.0f; d2[2] = 3.0f * (1.0f - t) - 2.0f; d2[3] = 2.0f - (2.0f - t);
}

// Spline<3>::valueGradientAndHessian (basisspline.h:539-606) / Hessian of the trilinear interpolant (mixed terms only)
template <int RIF> __device__ __forceinline__ void rif_value_grad_hess_vol(const DGrid &g, CellCache &cc, f3 p, float &val, f3 &grad, m33 &H);
// value, gradient and Hessian in world space: H = Rot^T H_vol Rot for a grid with a `toWorld` (splinevolume.cpp:367,375)
template <int RIF>
__device__ __forceinline__ void rif_value_grad_hess(const DGrid &g, CellCache &cc, f3 p, float &val, f3 &grad, m33 &H) {
    if (!rif_affine_capable<RIF>() || !g.affine) { rif_value_grad_hess_vol<RIF>(g, cc, p, val, grad, H); return; }
    rif_value_grad_hess_vol<RIF>(g, cc, to_volume(g, p), val, grad, H);
    grad = rot_t(g, grad);
    m33 Rm, Rt;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { Rm.m[i][j] = g.w2v[i * 4 + j]; Rt.m[j][i] = g.w2v[i * 4 + j]; }
    H = mul(mul(Rt, H), Rm);
}
template <int RIF>
__device__ __forceinline__ void rif_value_grad_hess_vol(const DGrid &g, CellCache &cc, f3 p, float &val, f3 &grad, m33 &H) {
    if (RIF == RIFK_ACOUSTIC) {
        // AcousticRIFVolume::hessian (src/volume/acousticrifvolume.cpp:254-308); its "x" is our z (cos phi = z / r), its "y" our y
        acoustic_value_grad(g, p, val, grad);
        float py = p.y, pz = p.z;
        float r = sqrtf(py * py + pz * pz);
        const float phi = atan2f(py, pz);
        if (r < MER_EPSILON_RIF) { py = MER_EPSILON_RIF; pz = MER_EPSILON_RIF; r = MER_EPSILON_RIF; }
        const float kr = g.ac_k_r, krr = kr * r, m = (float) g.ac_mode, nmax = g.ac_n_max;
        const float j0 = jnf(g.ac_mode, krr), j1 = jnf(g.ac_mode + 1, krr), j2 = jnf(g.ac_mode + 2, krr);
        const float d0 = m / krr * j0 - j1, d1 = (m + 1.0f) / krr * j1 - j2;
        const float invr = 1.0f / r, invr2 = invr * invr;
        const float cosp = cosf(phi), sinp = sinf(phi), cosmp = cosf(m * phi), sinmp = sinf(m * phi);
        const float cosm1p = cosf((m - 1.0f) * phi), sinm1p = sinf((m - 1.0f) * phi), cosm2p = cosf((m - 2.0f) * phi), sinm2p = sinf((m - 2.0f) * phi);
        const float Hxx = nmax * (j0 * m * invr2 * (-cosm2p + m * sinm1p * sinp) + d0 * m * invr * kr * cosm1p * cosp
                                  - j1 * kr * py * invr2 * (sinp * cosmp + m * cosp * sinmp) - d1 * kr * kr * cosp * cosp * cosmp);
        const float Hxy = nmax * (-j0 * m * invr2 * (-sinm2p + m * sinm1p * cosp) + d0 * m * invr * kr * cosm1p * sinp
                                  + j1 * kr * pz * invr2 * (sinp * cosmp + m * cosp * sinmp) - d1 * kr * kr * cosp * sinp * cosmp);
        const float Hyy = -nmax * (j0 * m * invr2 * (-cosm2p + m * cosm1p * cosp) + d0 * m * invr * kr * sinm1p * sinp
                                   + j1 * kr * pz * invr2 * (cosp * cosmp - m * sinp * sinmp) + d1 * kr * kr * sinp * sinp * cosmp);
        H = m33(0.0f);
        H.m[1][1] = Hyy; H.m[1][2] = H.m[2][1] = Hxy; H.m[2][2] = Hxx;
    } else if (RIF == MER_RIF_BSPLINE3) {
        const float px = (p.x - g.bmin[0]) * g.s[0], py = (p.y - g.bmin[1]) * g.s[1], pz = (p.z - g.bmin[2]) * g.s[2];
        const float flx = floorf(px), fly = floorf(py), flz = floorf(pz);
        float wx[4], dx[4], ex[4], wy[4], dy[4], ey[4], wz[4], dz[4], ez[4];
        bspline_weights(px - flx, wx, dx); bspline_weights2(px - flx, ex);
        bspline_weights(py - fly, wy, dy); bspline_weights2(py - fly, ey);
        bspline_weights(pz - flz, wz, dz); bspline_weights2(pz - flz, ez);
        int ix = min(max((int) flx - 1, 0), g.res[0] - 4), iy = min(max((int) fly - 1, 0), g.res[1] - 4), iz = min(max((int) flz - 1, 0), g.res[2] - 4);
        const float *C = g.coeff + MER_CHK(g.chk, CHK_GRID_COEFF, ((size_t) iz * g.res[1] + iy) * g.res[0] + ix, g.n_dense - 3u * (uint32_t) (g.res[0] * g.res[1] + g.res[0] + 1));
        const int sy = g.res[0], sz = g.res[0] * g.res[1];
        float f = 0, gx = 0, gy = 0, gz = 0, hxx = 0, hyy = 0, hzz = 0, hxy = 0, hyz = 0, hzx = 0;
        for (int k = 0; k < 4; k++)
            for (int j = 0; j < 4; j++) {
                const float *row = C + k * sz + j * sy;
                const float c0 = row[0], c1 = row[1], c2 = row[2], c3 = row[3];
                const float r0 = c0 * wx[0] + c1 * wx[1] + c2 * wx[2] + c3 * wx[3];
                const float r1 = c0 * dx[0] + c1 * dx[1] + c2 * dx[2] + c3 * dx[3];
                const float r2 = c0 * ex[0] + c1 * ex[1] + c2 * ex[2] + c3 * ex[3];
                f += r0 * wy[j] * wz[k];
                gx += r1 * wy[j] * wz[k]; gy += r0 * dy[j] * wz[k]; gz += r0 * wy[j] * dz[k];
                hxx += r2 * wy[j] * wz[k]; hyy += r0 * ey[j] * wz[k]; hzz += r0 * wy[j] * ez[k];
                hxy += r1 * dy[j] * wz[k]; hyz += r0 * dy[j] * dz[k]; hzx += r1 * wy[j] * dz[k];
            }
        val = f; grad = f3(gx * g.s[0], gy * g.s[1], gz * g.s[2]);
        H.m[0][0] = hxx * g.s[0] * g.s[0]; H.m[1][1] = hyy * g.s[1] * g.s[1]; H.m[2][2] = hzz * g.s[2] * g.s[2];
        H.m[0][1] = H.m[1][0] = hxy * g.s[0] * g.s[1]; H.m[1][2] = H.m[2][1] = hyz * g.s[1] * g.s[2]; H.m[0][2] = H.m[2][0] = hzx * g.s[2] * g.s[0];
    } else {
        trilinear_value_grad<RIF>(g, cc, p, val, grad);
        const float fx = __builtin_fmaf(g.s[0], p.x, g.t[0]) - cc.cx, fy = __builtin_fmaf(g.s[1], p.y, g.t[1]) - cc.cy,
                    fz = __builtin_fmaf(g.s[2], p.z, g.t[2]) - cc.cz;
        // mixed second derivatives of the monomial form (the pure ones vanish): d2f/dxdy = axy + axyz z, ...
        const float hxy = __builtin_fmaf(cc.axyz, fz, cc.axy) * g.s[0] * g.s[1];
        const float hyz = __builtin_fmaf(cc.axyz, fx, cc.ayz) * g.s[1] * g.s[2];
        const float hzx = __builtin_fmaf(cc.axyz, fy, cc.axz) * g.s[2] * g.s[0];
        H = m33(0.0f);
        H.m[0][1] = H.m[1][0] = hxy; H.m[1][2] = H.m[2][1] = hyz; H.m[0][2] = H.m[2][0] = hzx;
    }
}

__device__ __forceinline__ bool finite3(f3 a) { return isfinite(a.x) && isfinite(a.y) && isfinite(a.z); }

// State of one curved-ray connection between two units of work (Connector::unit): 32 words when parked (K_connect's cstate record)
enum { CP_NEW = 0, CP_EVAL0, CP_TRIAL, CP_PATHLEN, CP_OK, CP_FAIL };
struct ConnState {
    f3 x, e, xn, tempSol, dir;        // iterate, its residual, trial iterate, first converged direction, direction found (x n(p1) once accepted)
    m33 J;                            // Jacobian of the residual at x
    float cost, lambda, radius, weight, RIFp, optDist, dist;
    int it, tries, iterations, phase, ok;
    __device__ __forceinline__ void load(const uint32_t *r) {
        const uint4 *q = (const uint4 *) r; const uint4 a = q[0], b = q[1], c = q[2], d = q[3], g = q[4], h = q[5], k = q[6], l = q[7];
#define F(u) __uint_as_float(u)
        x = f3(F(a.x), F(a.y), F(a.z)); e = f3(F(a.w), F(b.x), F(b.y)); xn = f3(F(b.z), F(b.w), F(c.x)); tempSol = f3(F(c.y), F(c.z), F(c.w));
        dir = f3(F(d.x), F(d.y), F(d.z));
        J.m[0][0] = F(d.w); J.m[0][1] = F(g.x); J.m[0][2] = F(g.y); J.m[1][0] = F(g.z); J.m[1][1] = F(g.w); J.m[1][2] = F(h.x);
        J.m[2][0] = F(h.y); J.m[2][1] = F(h.z); J.m[2][2] = F(h.w);
        cost = F(k.x); lambda = F(k.y); radius = F(k.z); weight = F(k.w); RIFp = F(l.x); optDist = F(l.y); dist = F(l.z);
#undef F
        it = (int) (l.w & 31u); tries = (int) ((l.w >> 5) & 7u); iterations = (int) ((l.w >> 8) & 127u); phase = (int) ((l.w >> 15) & 7u); ok = (int) ((l.w >> 18) & 1u);
    }
    __device__ __forceinline__ void store(uint32_t *r) const {
        uint4 *q = (uint4 *) r;
#define U(f) __float_as_uint(f)
        q[0] = make_uint4(U(x.x), U(x.y), U(x.z), U(e.x)); q[1] = make_uint4(U(e.y), U(e.z), U(xn.x), U(xn.y));
        q[2] = make_uint4(U(xn.z), U(tempSol.x), U(tempSol.y), U(tempSol.z)); q[3] = make_uint4(U(dir.x), U(dir.y), U(dir.z), U(J.m[0][0]));
        q[4] = make_uint4(U(J.m[0][1]), U(J.m[0][2]), U(J.m[1][0]), U(J.m[1][1])); q[5] = make_uint4(U(J.m[1][2]), U(J.m[2][0]), U(J.m[2][1]), U(J.m[2][2]));
        q[6] = make_uint4(U(cost), U(lambda), U(radius), U(weight));
        q[7] = make_uint4(U(RIFp), U(optDist), U(dist), (uint32_t) it | ((uint32_t) tries << 5) | ((uint32_t) iterations << 8) | ((uint32_t) phase << 15) | ((uint32_t) (ok != 0) << 18));
#undef U
    }
};
#define MER_CSTATE_WORDS 32

// XC = false: the kernel is built for connections that stay inside the shape (the boundary code is compiled out: 16 % faster on
// configs[4], where the emitter sits inside); XC = true: a connection may cross (decided per pair: its far end lies outside)
template <int RIF, int BND = 0, bool XC = true> struct Connector {
    const Params &P;
    float tol, rrweight; int precision, maxIter, maxSteps;
    mutable CellCache cc;                       // the 8 corners of the cell the ray is in (trilinear RIF): reused across evaluations
    mutable uint32_t nsteps = 0;                // sensitivity / Verlet steps taken (MER_C_CONNECT_STEPS)
#ifdef MER_CONNECT_DEBUG
    mutable float dbg0 = -1, dbg1 = -1, dbg2 = 0, dbg3 = 0, dbg4 = 0;
#endif
    __device__ Connector(const Params &p) : P(p) {
        cc.reset();
        tol = 1e-6f; rrweight = 1e-2f; precision = 3; maxIter = 20;                 // :209-213, :217
        float diag = 0;
        for (int i = 0; i < 3; i++) diag += (p.sc.bmax[i] - p.sc.bmin[i]) * (p.sc.bmax[i] - p.sc.bmin[i]);
        if (p.sc.boundary == MER_BOUNDARY_SPHERE) diag = 4 * p.sc.sph_radius * p.sc.sph_radius;
        maxSteps = min(100000, (int) (4 * sqrtf(diag) / p.sc.stepsize) + 16);        // the reference allows 1e5 (:829)
    }
    __device__ float rif_value(f3 p) const { float n; f3 g; rif_value_grad<RIF>(P.rif, cc, p, n, g); return n; }

    // er_derivativestep (:798-814): one velocity-Verlet step of the ray with its sensitivities dp/dv0, dv/dv0.  The solver's iterates are not
    // pinned to the reference (its minimiser is Ceres), so the step's ARITHMETIC is defined here, in fused form, and restated identically in
    // the oracle: (v (x) G) dp is formed as v (x) (G^T dp) -- 9 + 9 multiply-adds instead of a 3x3 product of an outer product -- and every
    // accumulation is a fused multiply-add.  ~200 VALU instructions per step against ~350 for the literal matrix expression.
    //   H dp: s = H[i][0] dp[0][j]; s = fma(H[i][1], dp[1][j], s); s = fma(H[i][2], dp[2][j], s).  The Hessian of a TRILINEAR interpolant has a
    //   zero diagonal (the interpolant is linear along every axis): its term is skipped, which leaves every result bit for bit (fma(0, b, s) = s).
    __device__ __forceinline__ void add_hess_dp(m33 &dv, const m33 &H, const m33 &dp, float t) const {
        constexpr bool trilinear = RIF != MER_RIF_BSPLINE3 && RIF != RIFK_ACOUSTIC;
        const bool zero_diag = trilinear && !(rif_affine_capable<RIF>() && P.rif.affine);       // a grid with a `toWorld` is rotated: full product
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) {
                float sum;
                if (zero_diag) {
                    const int k0 = i == 0 ? 1 : 0, k1 = i == 2 ? 1 : 2;
                    sum = __builtin_fmaf(H.m[i][k1], dp.m[k1][j], H.m[i][k0] * dp.m[k0][j]);
                } else sum = __builtin_fmaf(H.m[i][2], dp.m[2][j], __builtin_fmaf(H.m[i][1], dp.m[1][j], H.m[i][0] * dp.m[0][j]));
                dv.m[i][j] = __builtin_fmaf(t, sum, dv.m[i][j]);
            }
    }
    __device__ void dstep(f3 &p, f3 &v, m33 &dp, m33 &dv, float h) const {
        float n; f3 G; m33 H;
        const float t = 0.5f * h;
        rif_value_grad_hess<RIF>(P.rif, cc, p, n, G, H);
        v = fma3(t, G, v);
        add_hess_dp(dv, H, dp, t);
        p = p + h * v / n;
        rif_value_grad_hess<RIF>(P.rif, cc, p, n, G, H);
        const float invn = 1.0f / n, c = -(invn * invn);
        const float V[3] = {c * v.x, c * v.y, c * v.z};
        float w[3];
#pragma unroll
        for (int j = 0; j < 3; j++) w[j] = __builtin_fmaf(G.z, dp.m[2][j], __builtin_fmaf(G.y, dp.m[1][j], G.x * dp.m[0][j]));       // G^T dp
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) dp.m[i][j] = __builtin_fmaf(h, __builtin_fmaf(V[i], w[j], invn * dv.m[i][j]), dp.m[i][j]);
        v = fma3(t, G, v);
        add_hess_dp(dv, H, dp, t);
    }
    // boundaryVelocity (:1040-1055): Snell's law for the optical momentum v at a surface with unit normal N, index ni on the ray's side and
    // ne beyond (total internal reflection when the root is imaginary)
    __device__ void boundary_velocity(f3 &v, f3 N, float ni, float ne) const {
        const float dotp = dot(v, N);
        float r = ne / ni; r = r * r - 1;
        const float n2 = dot(v, v);
        float sq = r * n2 + dotp * dotp;
        if (sq < MER_EPSILON) { v = 2 * dotp * N - v; return; }
        sq = sqrtf(sq);
        v = v - dotp * N + (dotp > 0 ? sq : (dotp < 0 ? -sq : 0.0f)) * N;
    }
    // boundaryVelocityDerivative (:1061-1074): the same, with the sensitivity dv/dv0 taken through the refraction; dtb = d(arrival
    // parameter)/dv0, dnb = grad n at the boundary point
    __device__ void boundary_velocity_derivative(f3 &v, m33 &dv, f3 dtb, f3 dnb, f3 N, float ni, float ne) const {
        const float dotp = dot(v, N);
        float r = ne / ni; r = r * r - 1;
        const float n2 = dot(v, v);
        float sq = r * n2 + dotp * dotp;
        const m33 inner = add(dv, outer(dnb, dtb));
        const m33 NN = outer(N, N);
        if (sq < MER_EPSILON) {
            m33 a = scale(NN, 2.0f); for (int i = 0; i < 3; i++) a.m[i][i] -= 1.0f;
            v = 2 * dotp * N - v;
            dv = mul(a, inner);
            return;
        }
        sq = sqrtf(sq);
        const float sg = dotp > 0 ? 1.0f : (dotp < 0 ? -1.0f : 0.0f);
        m33 a = add(scale(NN, -1.0f), scale(outer(N, (r * v + dotp * N) / sq), sg)); for (int i = 0; i < 3; i++) a.m[i][i] += 1.0f;
        dv = mul(a, inner);
        v = v - dotp * N + sg * sq * N;
    }
    // computefdfBDPT (:816-939); J[r][c] = d error_r / d v0_c.  A ray that leaves the shape before its closest approach to p2 is
    // taken to the boundary (bisection), refracted into the exterior (index 1) and continued straight to its closest approach
    // (:873-919).  `cross`: the connection is meant to cross (the reference's isSensorSample; here: p2 lies outside the shape) -- a
    // refracted ray that moves away from p2 then has no derivative (:907-912).  false = "error = p1 - p2, derivative 0" of the reference.
    __device__ bool computefdf(f3 v_i, f3 p1, f3 p2, bool cross, f3 &error, m33 &J) const {
        m33 dp(0.0f), dv(1.0f);
        error = p1 - p2; J = m33(0.0f);
        // a shooting direction that is not a finite non-zero vector has no ray (the solver's step can overflow: guard, not reference).
        // Past this point a non-finite p or v cannot reach memory: every fetch index is clamped or bounds-tested in a NaN-safe form
        // (mer_device.hpp), a NaN position fails inside_shape (all its comparisons are false) and a NaN residual never beats the
        // current cost -- the march ends at the next test, as for a ray that leaves the shape.
        if (!finite3(v_i) || !(dot(v_i, v_i) > 0.0f) || !isfinite(dot(v_i, v_i))) return false;
        if (RIF == MER_RIF_BSPLINE3 && !inside_volume_limits(P.rif, p1)) return false;
        float h = P.sc.stepsize;
        int nBisect = (int) ceilf((float) precision / 0.30102999566f);
        f3 p = p1, oldp = p1, v = v_i, oldv = v_i; m33 olddp(0.0f), olddv(1.0f);
        const bool signOld = dot(p - p2, v) < 0.0f; bool signNew = signOld;
        const float r = rif_value(p);
        const float n1 = sqrtf(dot(v_i, v_i)), n2 = n1 * n1, n3 = n2 * n1;
        { m33 a(n2); const m33 o = outer(v, v);
          for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) a.m[i][j] -= o.m[i][j];
          dv = mul(scale(a, r / n3), dv); }
        v = v / n1 * r;
        int found = 0;                                    // 1: closest approach inside the shape, 2: beyond the boundary
        for (int i = 0; i < maxSteps; i++) {
            oldp = p; oldv = v; olddp = dp; olddv = dv;
            dstep(p, v, dp, dv, h); nsteps++;
            signNew = dot(p - p2, v) < 0.0f;
            if (signNew != signOld) {
                while (nBisect > 0) {
                    nBisect--;
                    p = oldp; v = oldv; dp = olddp; dv = olddv;
                    h = h / 2;
                    dstep(p, v, dp, dv, h); nsteps++;
                    signNew = dot(p - p2, v) < 0.0f;
                    if (signNew == signOld) { oldp = p; oldv = v; olddp = dp; olddv = dv; }
                }
                found = 1;
                break;
            } else if (!inside_shape_b<BND>(P, p)) {
                // a ray that leaves the shape on its way to a point INSIDE it is a failed trial (the damping grows): the reference refracts
                // it too and lets the exterior leg steer the minimiser, which costs 16 % more traced rays per connection here for the
                // same connections found (measured on configs[4]); the boundary is crossed only by connections that must cross it
                if (!XC || !cross) return false;
                while (nBisect > 0) {                                                   // to the boundary (:874-889)
                    nBisect--;
                    p = oldp; v = oldv; dp = olddp; dv = olddv;
                    h = h / 2;
                    dstep(p, v, dp, dv, h); nsteps++;
                    if (inside_shape_b<BND>(P, p)) { oldp = p; oldv = v; olddp = dp; olddv = dv; }
                }
                found = 2;
                break;
            }
        }
        if (!found) return false;
        f3 dpdt(0, 0, 0), dtstar(0, 0, 0);
        if (found == 1) {
            float rr; f3 dvdt;
            rif_value_grad<RIF>(P.rif, cc, p, rr, dvdt);
            dpdt = v / rr;
            dtstar = -(premult(dp, v) + premult(dv, p - p2)) / (dot(v, dpdt) + dot(p - p2, dvdt));
        } else if (XC) {
            if (dot(p - p1, p - p1) < MER_EPSILON) return false;                       // no progress made (:890-894)
            float nb; f3 dnb;
            rif_value_grad<RIF>(P.rif, cc, p, nb, dnb);
            const f3 dpdtb = v / nb;
            const f3 N = shape_normal_b<BND>(P, p);                                     // normalize(m_SDF->gradient(p)) (:899-900)
            const f3 dtb = -premult(dp, N) / dot(N, dpdtb);
            boundary_velocity_derivative(v, dv, dtb, dnb, N, nb, 1.0f);
            const float extra_t = -dot(v, p - p2) / dot(v, v);
            if (cross && extra_t < 0) return false;                                     // (:907-912)
            dp = add(add(dp, outer(dpdtb - v, dtb)), scale(dv, extra_t));
            p = p + extra_t * v;
            dpdt = v;
            dtstar = -(premult(dp, v) + premult(dv, p - p2)) / dot(v, dpdt);
        }
        J = add(dp, outer(dpdt, dtstar));
        error = p - p2;
        return true;
    }
    // ---- the solver as a resumable state machine ---------------------------------------------------------------------------------
    // makeDirectConnections (:1087-1163) around the stand-in for ceres::Solve (damped Gauss-Newton, <= 20 iterations of <= 6 damping
    // trials).  One connection costs 1 ... 100+ traced rays (`computefdf`), and which lane needs how many is not predictable: run to
    // completion per lane, a wave waits for its slowest solve (round 1: K_connect at a quarter of K_march's step rate).  So the loop
    // nest is flattened: `unit()` performs ONE expensive operation -- one traced ray -- plus the algebra up to the next one, and the
    // state in between (ConnState, 32 words) can be parked in HBM: K_connect runs one unit per pending connection per launch and
    // re-compacts the unfinished ones (mer_wavefront.hpp); the leaf entry point just loops.  Per lane the sequence of arithmetic
    // operations and sampler draws is the loop nest's (the oracle keeps the loops).
    //
    // computefdf rescales its argument to |v0| = n(p1): the residual does not depend on |x|, J^T J is singular along x and only the
    // damping makes the step finite -- with little damping the step along x is rounding noise of any size.  The unknown lives on the
    // sphere |x| = n(p1): every trial iterate is put back on it, and a step whose determinant is below float resolution of the
    // product of the pivots, or that is not finite, is retried with more damping.
    // The next damped Gauss-Newton candidate S.xn from (S.x, S.e, S.J); false: the solve has ended (converged, stuck or out of iterations)
    __device__ bool next_candidate(ConnState &S) const {
        if (S.tries == 0 && !(S.it < maxIter && S.ok && S.cost >= tol * 1e-3f)) return false;
        float A[3][3], b[3]; const float E[3] = {S.e.x, S.e.y, S.e.z};
        for (int i = 0; i < 3; i++) { b[i] = 0; for (int k = 0; k < 3; k++) b[i] -= S.J.m[k][i] * E[k];
            for (int j = 0; j < 3; j++) { A[i][j] = 0; for (int k = 0; k < 3; k++) A[i][j] += S.J.m[k][i] * S.J.m[k][j]; } }
        while (S.tries < 6) {
            S.tries++;
            float M[3][3];
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M[i][j] = A[i][j] + (i == j ? S.lambda * (A[i][i] + 1e-12f) : 0.0f);
            const float det = M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
                              M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
            if (!(fabsf(det) > 1e-6f * fabsf(M[0][0] * M[1][1] * M[2][2])) || !isfinite(det)) { S.lambda *= 10; continue; }
            float d[3];
            d[0] = (b[0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (b[1] * M[2][2] - M[1][2] * b[2]) + M[0][2] * (b[1] * M[2][1] - M[1][1] * b[2])) / det;
            d[1] = (M[0][0] * (b[1] * M[2][2] - M[1][2] * b[2]) - b[0] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) + M[0][2] * (M[1][0] * b[2] - b[1] * M[2][0])) / det;
            d[2] = (M[0][0] * (M[1][1] * b[2] - b[1] * M[2][1]) - M[0][1] * (M[1][0] * b[2] - b[1] * M[2][0]) + b[0] * (M[1][0] * M[2][1] - M[1][1] * M[2][0])) / det;
            const f3 xn(S.x.x + d[0], S.x.y + d[1], S.x.z + d[2]);
            const float ln = sqrtf(dot(xn, xn));
            if (!(ln > 0.0f) || !isfinite(ln)) { S.lambda *= 10; continue; }
            S.xn = xn * (S.radius / ln);
            return true;
        }
        return false;                                  // six trials without an improvement
    }
    // a solve has ended: accept / restart under Russian roulette (makeDirectConnections' loop body after ceres::Solve)
    __device__ void end_solve(ConnState &S, f3 p1, f3 p2, Rng &rng) const {
        const float cost = S.ok ? S.cost : MER_INF;
        if (cost < tol) {
            if (S.iterations == 1) { S.iterations++; S.tempSol = normalize(S.x); }
            else S.iterations++;
            S.dir = normalize(S.x);
            if (dot(S.tempSol - S.dir, S.tempSol - S.dir) < 2 * tol) {
                S.dir = S.dir * S.RIFp;
                S.weight *= (float) (S.iterations - 1);
                S.phase = CP_PATHLEN;
                return;
            }
        }
        if (rng.next1D() < rrweight) S.weight = S.weight / rrweight;
        else { S.dir = normalize(S.x); S.phase = CP_FAIL; return; }
        if (S.iterations > 64) { S.phase = CP_FAIL; return; }
        S.x = uniform_sample(normalize(p2 - p1), rng) * S.RIFp;
        S.phase = CP_EVAL0;
    }
    // one traced ray of the connection p1 -> p2, and the algebra up to the next one.  S.phase = CP_NEW and S.weight set by the caller.
    __device__ __forceinline__ void unit(ConnState &S, f3 p1, f3 p2, Rng &rng, f3 &revDir) const {
        if (S.phase == CP_NEW) {
            S.iterations = 1; S.tempSol = f3(0, 0, 0); S.dir = f3(0, 0, 1); S.optDist = 0; S.dist = 0; S.it = 0; S.tries = 0; S.ok = 0;
            S.cost = 0; S.lambda = 0; S.radius = 0; S.e = f3(0, 0, 0); S.xn = f3(0, 0, 0); S.J = m33(0.0f);
            if (RIF == MER_RIF_BSPLINE3 && !inside_volume_limits(P.rif, p1)) { S.phase = CP_FAIL; return; }
            S.RIFp = rif_value(p1);
            S.x = uniform_sample(normalize(p2 - p1), rng) * S.RIFp;
            S.phase = CP_EVAL0;
        }
        if (S.phase == CP_EVAL0 || S.phase == CP_TRIAL) {
            const bool first = S.phase == CP_EVAL0;
            f3 en; m33 Jn;
            const bool okn = computefdf(first ? S.x : S.xn, p1, p2, XC && !inside_shape_b<BND>(P, p2), en, Jn);            // the one call site
            const float cn = 0.5f * dot(en, en);
            if (first) { S.ok = okn; S.e = en; S.J = Jn; S.cost = cn; S.lambda = 1e-4f; S.radius = sqrtf(dot(S.x, S.x)); S.it = 0; S.tries = 0; }
            else if (okn && cn < S.cost) { S.x = S.xn; S.e = en; S.J = Jn; S.cost = cn; S.lambda = fmaxf(S.lambda * 0.1f, 1e-9f); S.it++; S.tries = 0; }
            else S.lambda *= 10;
            if (next_candidate(S)) S.phase = CP_TRIAL; else end_solve(S, p1, p2, rng);
            return;
        }
        if (S.phase == CP_PATHLEN) {
            float bw = 1.0f;
            S.phase = path_lengths(p1, p2, S.dir, XC && !inside_shape_b<BND>(P, p2), revDir, S.optDist, S.dist, bw) ? CP_OK : CP_FAIL;
            S.weight *= bw;
        }
    }
    // the reference's own Verlet step (:662-669)
    __device__ void verlet(f3 &p, f3 &v, float h) const {
        float n, n2; f3 G, G2;
        rif_value_grad<RIF>(P.rif, cc, p, n, G);
        v = v + 0.5f * h * G;
        p = p + h * v / n;
        rif_value_grad<RIF>(P.rif, cc, p, n2, G2);
        v = v + 0.5f * h * G2;
    }
    // computePathLengthsTillClosestP2 (:941-1030).  cross = false: both ends inside the shape, a ray that leaves it is no connection (the
    // reference's emitter samples, :960-962); cross = true: p2 lies outside -- the ray is taken to the boundary, refracted by Snell's law
    // into the exterior (index 1) and followed straight to its closest approach to p2 (the reference's sensor samples, :963-992).
    // dist = arc length INSIDE the shape (what the medium attenuates); optDist includes the exterior leg.  bweight: the weight the
    // boundary's BSDF gives the refracted ray (1 for the index-matched null boundary; (1 - F) eta^2 for hdielectric, hdielectric.cpp:183-242).
    __device__ bool path_lengths(f3 p1, f3 p2, f3 dirToP2, bool cross, f3 &revDir, float &optDist, float &dist, float &bweight) const {
        dist = 0; optDist = 0; bweight = 1.0f;
        float h = P.sc.stepsize;
        int nBisect = (int) ceilf((float) precision / 0.30102999566f);
        f3 p = p1, oldp = p1, v = dirToP2, oldv = dirToP2;
        const bool signOld = dot(p - p2, v) < 0.0f; bool signNew = signOld;
        for (int i = 0; i < maxSteps; i++) {
            oldp = p; oldv = v;
            verlet(p, v, h); nsteps++;
            signNew = dot(p - p2, v) < 0.0f;
            if (!inside_shape_b<BND>(P, p)) {
                if (!XC || !cross) return false;
                while (nBisect > 0) {                                          // close to the boundary (:965-979)
                    nBisect--;
                    p = oldp; v = oldv; h = h / 2;
                    verlet(p, v, h); nsteps++;
                    if (inside_shape_b<BND>(P, p)) { dist += h; optDist += h * rif_value(0.5f * (p + oldp)); oldp = p; oldv = v; }
                }
                const f3 N = shape_normal_b<BND>(P, p);
                const float nb = rif_value(p);
                if (P.sc.boundary_bsdf == MER_BSDF_HDIELECTRIC) {              // HDielectric, refracted component seen from inside: (1 - F) x eta^2
                    const float cosI = dot(normalize(v), N);                    // the ray leaves: cos > 0 with the outward normal
                    float cosT; const float F = fresnel_dielectric_ext(-cosI, cosT, nb);
                    bweight = (1.0f - F) * (nb * nb);
                }
                boundary_velocity(v, N, nb, 1.0f);                              // Snell's law (:982-984)
                const float extra_t = -dot(v, p - p2) / dot(v, v);
                if (extra_t < 0) return false;
                p = p + extra_t * v;
                optDist += extra_t;
                break;
            }
            if (signNew != signOld) {
                while (nBisect > 0) {
                    nBisect--;
                    p = oldp; v = oldv; h = h / 2;
                    verlet(p, v, h); nsteps++;
                    signNew = dot(p - p2, v) < 0.0f;
                    if (signNew == signOld) { dist += h; optDist += h * rif_value(0.5f * (p + oldp)); oldp = p; oldv = v; }
                }
                break;
            } else { dist += h; optDist += h * rif_value(0.5f * (p + oldp)); }
        }
        if (dot(p - p2, p - p2) > tol) return false;
        revDir = -normalize(v);
        return true;
    }
    // uniformSample (:1078-1084) with squareToUniformHemisphere (src/libcore/warp.cpp:33-41)
    __device__ f3 uniform_sample(f3 in, Rng &rng) const {
        f3 ax, ay;
        coordinate_system(in, ax, ay);
        const float u1 = rng.next1D(), u2 = rng.next1D();
        const float z = u1, tmp = safe_sqrt(1.0f - z * z), phi = 2.0f * MER_PI * u2;
        return (cosf(phi) * tmp) * ax + (sinf(phi) * tmp) * ay + z * in;
    }
    // makeDirectConnections (:1087-1163), run to completion (leaf entry point mer_connect)
    __device__ bool connect(f3 p1, f3 p2, Rng &rng, float &weight, f3 &dirToP2, f3 &revDir, float &optDist, float &dist) const {
        ConnState S; S.phase = CP_NEW; S.weight = weight;
        while (S.phase != CP_OK && S.phase != CP_FAIL) unit(S, p1, p2, rng, revDir);
        weight = S.weight; dirToP2 = S.dir; optDist = S.optDist; dist = S.dist;
        return S.phase == CP_OK;
    }
};

// Medium::evalTransmittance over [0, L] of the straight ray ps + t dvec (homogeneous.cpp:264-273; heterogeneous.cpp:546-587: the reference's 2-walk
// Woodcock estimator or ratio tracking; :301-376 for method = simpson), run synchronously inside K_event (luminaire samples of the point and
// area emitters).  The sampler draws are the oracle's evalTransmittance's.
template <int SIGMA>
__device__ __forceinline__ f3 straight_transmittance(const Params &P, Rng &rng, LaneCounters &C, f3 ps, f3 dvec, float L) {
    const mer_scene_desc &S = P.sc;
    if (SIGMA == MER_SIGMA_HOMOGENEOUS) return homogeneous_transmittance(P, 0.0f - L);
    if (S.method == MER_METHOD_SIMPSON) { const float tv = expf(-simpson_integrate(P, C, ps, dvec, L)); return f3(tv, tv, tv); }    // heterogeneous.cpp:547-548
    const int nwalks = S.tr_estimator == MER_TR_WOODCOCK2 ? 2 : 1;
    float mint, maxt;                                            // heterogeneous.cpp:546-587
    if (!aabb_intersect(P.density.wmin, P.density.wmax, ps, dvec, mint, maxt)) return f3(1, 1, 1);
    mint = fmaxf(mint, 0.0f); maxt = fminf(maxt, L);
    float result = 0.0f;
    for (int w = 0; w < nwalks; ++w) {
        float Tr = 1.0f, t = mint;
        for (;;) {
            t -= logf(1 - rng.next1D()) * P.inv_max_density;
            if (t >= maxt) break;
            const float sigma = lookup_float(P.density, ps + dvec * t) * S.density_scale; C.tentative++;
            if (S.tr_estimator == MER_TR_RATIO) { Tr *= 1.0f - sigma * P.inv_max_density; if (Tr == 0.0f) break; }
            else if (sigma * P.inv_max_density > rng.next1D()) { Tr = 0.0f; break; }
        }
        result += Tr;
    }
    const float tv = result / (float) nwalks;
    return f3(tv, tv, tv);
}

// Luminaire sampling of a point emitter at a medium interaction: PointEmitter::sampleDirect (src/emitters/point.cpp:
// pdf 1, EDiscrete => no MIS partner) + Scene::evalTransmittance (straight rays, src/librender/scene.cpp:619-678) or
// Medium::eval through the RIF (curved rays, heterogeneousrefractive.cpp:571-640).  Returns value * phase (to be
// multiplied by the path throughput).  Synchronous: it runs inside K_event.
template <bool CURVED, int RIF, int STEPPER, int SIGMA, int BND = 0>
__device__ __forceinline__ f3 point_nee(const Params &P, Rng &rng, LaneCounters &C, f3 ps, f3 wi, int depth, float &optLen) {   // inlined: an out-of-line callee taking Params by reference forces a scratch copy of the kernel arguments
    optLen = 0.0f;
    const mer_scene_desc &S = P.sc;
    const f3 I(S.point_intensity[0], S.point_intensity[1], S.point_intensity[2]);
    const f3 pp(S.point_position[0], S.point_position[1], S.point_position[2]);
    const int interactions = S.max_depth - depth - 1;
    C.nee++;
    static_assert(!CURVED, "curved-ray connections run in K_connect");
    {
        f3 dvec = pp - ps;
        const float dist = sqrtf(dot(dvec, dvec)), invDist = 1.0f / dist;
        dvec = dvec * invDist;
        f3 value = I * (invDist * invDist);
        optLen = dist * S.rif_const;
        const float tExit = intersect_shape_b<BND>(P, ps, dvec, 0.0f, MER_INF);
        const bool crosses = tExit >= 0 && tExit < dist;
        const float L = crosses ? tExit : dist;
        f3 tr(1, 1, 1);
        if (crosses && interactions == 0) tr = f3(0, 0, 0);
        else tr = straight_transmittance<SIGMA>(P, rng, C, ps, dvec, L);
        value = value * tr;
        if (is_zero(value)) return f3(0, 0, 0);
        return value * phase_eval(S.phase, S.g, wi, dvec);
    }
    return f3(0, 0, 0);       // curved rays: K_connect (Connector::unit per launch, then connection_value)
}

// The radiance a finished curved-ray connection carries from the point emitter to the scattering point ps (to be multiplied by the
// path throughput): I / |pp - ps|^2 (the straight-line distance of PointEmitter::sampleDirect, which the reference keeps for curved
// connections) x transmittance along the connecting ray (arc length dist, launched along dir) x solver weight x phase function.
template <int RIF, int STEPPER, int SIGMA, int BND = 0>
__device__ __forceinline__ f3 connection_value(const Params &P, Rng &rng, LaneCounters &C, f3 ps, f3 wi, f3 dir, float dist, float w) {
    const mer_scene_desc &S = P.sc;
    const f3 I(S.point_intensity[0], S.point_intensity[1], S.point_intensity[2]);
    const f3 pp(S.point_position[0], S.point_position[1], S.point_position[2]);
    const int nwalks = (SIGMA == MER_SIGMA_GRID && S.tr_estimator == MER_TR_WOODCOCK2) ? 2 : 1;
    f3 tr;
    if (SIGMA == MER_SIGMA_HOMOGENEOUS) tr = f3(expf(P.sigT.x * (-dist)), expf(P.sigT.y * (-dist)), expf(P.sigT.z * (-dist)));
    else {
        float result = 0.0f;
        for (int wk = 0; wk < nwalks; ++wk) {
            f3 p = ps, v = dir; float left = dist, Tr = 1.0f, opt = 0; CellCache cc; cc.reset();
            for (;;) {
                const float s = -logf(1 - rng.next1D()) * P.inv_max_density;
                if (s >= left) break;
                // trace(p, v, s): int(s/h) full steps + remainder, insideShape after each, one step back on exit (:671-691)
                const float h = S.stepsize; int steps = (int) (s / h); const float rem = s - steps * h; bool inside = true;
                for (int q = 0; q <= steps && inside; ++q) {
                    const float hq = q < steps ? h : rem;
                    er_step<RIF, STEPPER>(P.rif, cc, p, v, hq, opt); C.steps++;
                    if (!inside_shape_b<BND>(P, p)) { er_step<RIF, STEPPER>(P.rif, cc, p, v, -hq, opt); C.steps++; inside = false; }
                }
                if (!inside) break;
                left -= s;
                const float sigma = lookup_float(P.density, p) * S.density_scale; C.tentative++;
                if (S.tr_estimator == MER_TR_RATIO) { Tr *= 1.0f - sigma * P.inv_max_density; if (Tr == 0.0f) break; }
                else if (sigma * P.inv_max_density > rng.next1D()) { Tr = 0.0f; break; }
            }
            result += Tr;
        }
        const float tv = result / (float) nwalks; tr = f3(tv, tv, tv);
    }
    if (is_zero(tr)) return f3(0, 0, 0);
    const f3 dv = pp - ps;
    const float invDist = 1.0f / sqrtf(dot(dv, dv));
    const f3 value = I * (invDist * invDist) * tr * w;
    return value * phase_eval(S.phase, S.g, wi, normalize(dir));
}

// leaf kernel: out stride 12: ok, weight, dirToP2[3], revDirToP1[3], dist, opticalDist, 0, 0; RNG stream (seed, i, 0)
template <int RIF, int BND = 0>
__global__ void __launch_bounds__(64) connect_kernel(const Params P, const float *p1, const float *p2, int64_t n, float *out) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Rng rng; rng.seed(P.seed, (uint32_t) i, 0);
    Connector<RIF, BND> K(P);
    const f3 a(p1[3 * i], p1[3 * i + 1], p1[3 * i + 2]), b(p2[3 * i], p2[3 * i + 1], p2[3 * i + 2]);
    float w = 1.0f, od = 0, di = 0; f3 dir(0, 0, 0), rev(0, 0, 0);
    const bool ok = K.connect(a, b, rng, w, dir, rev, od, di);
    float *o = out + 12 * i;
    o[0] = ok ? 1.0f : 0.0f; o[1] = w; o[2] = dir.x; o[3] = dir.y; o[4] = dir.z;
    o[5] = ok ? rev.x : 0.0f; o[6] = ok ? rev.y : 0.0f; o[7] = ok ? rev.z : 0.0f; o[8] = ok ? di : 0.0f; o[9] = ok ? od : 0.0f; o[10] = o[11] = 0.0f;
#ifdef MER_CONNECT_DEBUG
    o[10] = K.dbg0; o[11] = K.dbg1; o[5] = K.dbg2; o[6] = K.dbg3; o[7] = K.dbg4;
#endif
}

}  // namespace mer
