// Reproducer of the K_connect miscompile: Connector<RIF,BND>::connect on 64 different pairs, standalone (no libmer).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-math-errno -fno-slp-vectorize -DMER_SDF_BRANCHING -o pl_O3_branching pl.hip
//   ... -O1 -DMER_SDF_BRANCHING -o pl_O1_branching      ... -O3 -o pl_O3
// MER_SDF_BRANCHING selects round 1's sdf_value (lookupFloat with its early `return 0` for a point off the grid).  Measured on MI355X,
// ROCm 7.2.0: "connect BND=1" finds 1 of 64 connections at -O3 (and -O2) with the branching form, 61 of 64 at -O1, and 61 of 64 at -O3
// with the branch-free form that ships; BND=0 (no grid look-up in the inside test) finds 61 of 64 in every build.  With 64 identical
// pairs (no lane divergence) the -O3 branching build is correct too.
#include "frozen/mer_connect.hpp"      // the headers as they were when the miscompile was found (the library has moved on)
#include <cstdio>
#include <vector>
#include <cmath>
#include <cstring>
using namespace mer;

template <int RIF, int BND>
__global__ void pl_kernel(const Params P, f3 p1, f3 p2, f3 dir, float *out) {
    Connector<RIF, BND> K(P);
    f3 rev(0, 0, 0); float od = 0, di = 0;
    const bool ok = K.path_lengths(p1, p2, dir, rev, od, di);
    out[threadIdx.x * 4 + 0] = ok ? 1.f : 0.f; out[threadIdx.x * 4 + 1] = di; out[threadIdx.x * 4 + 2] = od; out[threadIdx.x * 4 + 3] = rev.x;
}

static void fill(DGrid &g, void *data, int N, float lo, float hi) {
    std::memset((void *) &g, 0, sizeof(g));
    g.data = data; g.layout = MER_LAYOUT_DENSE; g.channels = 1; g.dtype = MER_VOL_F32;
    for (int i = 0; i < 3; i++) { g.res[i] = N; g.bmin[i] = lo; g.bmax[i] = hi; g.s[i] = (N - 1) / (hi - lo); g.t[i] = g.s[i] * -lo; }
    g.n_dense = (uint64_t) N * N * N; g.buf_bytes = (uint32_t) (g.n_dense * 4);
}

template <int RIF, int BND>
__global__ void __launch_bounds__(64) cn_kernel(const Params P, f3 p1_, f3 p2_, float *out) {
    Rng rng; rng.seed(1, (uint32_t) threadIdx.x, 0);
    // 64 different pairs: the lanes of the wave diverge in every loop of the solver
    Rng g; g.seed(77, threadIdx.x, 3);
    f3 p1(0.9f * (g.next1D() - 0.5f), 0.9f * (g.next1D() - 0.5f), 0.9f * (g.next1D() - 0.5f));
    f3 p2(0.9f * (g.next1D() - 0.5f), 0.9f * (g.next1D() - 0.5f), 0.9f * (g.next1D() - 0.5f));
    if (threadIdx.x == 0) { p1 = p1_; p2 = p2_; }
    Connector<RIF, BND> K(P);
    float w = 1.0f, od = 0, di = 0; f3 dir(0, 0, 0), rev(0, 0, 0);
    const bool ok = K.connect(p1, p2, normalize(p2 - p1), rng, w, dir, rev, od, di);
    out[threadIdx.x * 4 + 0] = ok ? 1.f : 0.f; out[threadIdx.x * 4 + 1] = di; out[threadIdx.x * 4 + 2] = od; out[threadIdx.x * 4 + 3] = dir.x;
}

int main() {
    const int N = 24, M = 64;
    std::vector<float> rif(N * N * N), sdf(M * M * M);
    for (int k = 0; k < N; k++) for (int j = 0; j < N; j++) for (int i = 0; i < N; i++) {
        const double x = -1 + 2.0 * i / (N - 1), y = -1 + 2.0 * j / (N - 1), z = -1 + 2.0 * k / (N - 1);
        rif[(k * N + j) * N + i] = (float) (2.0 - (x * x + y * y + z * z) / 3.0);
    }
    for (int k = 0; k < M; k++) for (int j = 0; j < M; j++) for (int i = 0; i < M; i++) {
        const double x = -1.2 + 2.4 * i / (M - 1), y = -1.2 + 2.4 * j / (M - 1), z = -1.2 + 2.4 * k / (M - 1);
        sdf[(k * M + j) * M + i] = (float) (std::sqrt(x * x + y * y + z * z) - 0.9);
    }
    float *drif, *dsdf, *dout;
    hipMalloc(&drif, rif.size() * 4); hipMalloc(&dsdf, sdf.size() * 4); hipMalloc(&dout, 64 * 16);
    hipMemcpy(drif, rif.data(), rif.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dsdf, sdf.data(), sdf.size() * 4, hipMemcpyHostToDevice);
    Params P; std::memset((void *) &P, 0, sizeof(P));
    fill(P.rif, drif, N, -1, 1); fill(P.sdf, dsdf, M, -1.2f, 1.2f);
    P.sc.boundary = MER_BOUNDARY_SDF; P.sc.stepsize = 0.5f * 2.0f / (N - 1);
    for (int i = 0; i < 3; i++) { P.sc.bmin[i] = -1; P.sc.bmax[i] = 1; }
    const f3 p1(0.04393215f, 0.19367044f, 0.09248704f), p2(-0.2824263f, 0.39993516f, 0.21559572f), dir(-1.5805206f, 1.0293975f, 0.61161023f);
    float h[256];
    hipLaunchKernelGGL((pl_kernel<MER_RIF_TRILINEAR, 1>), dim3(1), dim3(64), 0, 0, P, p1, p2, dir, dout);
    hipMemcpy(h, dout, 64 * 16, hipMemcpyDeviceToHost);
    printf("BND=1 global : ok %g dist %g opt %g rev.x %g   (expected ok 1 dist 0.40574 opt 0.79205 rev.x 0.81385)\n", h[0], h[1], h[2], h[3]);
    hipLaunchKernelGGL((cn_kernel<MER_RIF_TRILINEAR, 1>), dim3(1), dim3(64), 0, 0, P, p1, p2, dout);
    hipMemcpy(h, dout, 64 * 16, hipMemcpyDeviceToHost);
    { int nok = 0; double sd = 0; for (int t = 0; t < 64; t++) { nok += h[4 * t] == 1.f; sd += h[4 * t + 1]; } printf("connect BND=1 global : %d / 64 connected, sum of lengths %.6f\n", nok, sd); }
    P.sc.boundary = MER_BOUNDARY_AABB;
    hipLaunchKernelGGL((cn_kernel<RIFK_DENSE_BUF, 0>), dim3(1), dim3(64), 0, 0, P, p1, p2, dout);
    hipMemcpy(h, dout, 64 * 16, hipMemcpyDeviceToHost);
    { int nok = 0; double sd = 0; for (int t = 0; t < 64; t++) { nok += h[4 * t] == 1.f; sd += h[4 * t + 1]; } printf("connect BND=0 buffer : %d / 64 connected, sum of lengths %.6f\n", nok, sd); }
    hipLaunchKernelGGL((pl_kernel<RIFK_DENSE_BUF, 0>), dim3(1), dim3(64), 0, 0, P, p1, p2, dir, dout);
    hipMemcpy(h, dout, 64 * 16, hipMemcpyDeviceToHost);
    printf("BND=0 buffer : ok %g dist %g opt %g rev.x %g\n", h[0], h[1], h[2], h[3]);
    return 0;
}
