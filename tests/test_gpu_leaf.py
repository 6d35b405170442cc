"""GPU parity of the leaf entry points (through the C-ABI) against the CPU oracle."""
import numpy as np
import pytest
from mitsubaer_amd import params as P, synth, capi
from tests import scenes

pytestmark = pytest.mark.gpu


def test_rng_stream_bit_exact(ctx, orc):
    for seed, pix, smp in [(0, 0, 0), (7, 12345, 3), (2 ** 40 + 5, 2 ** 20 - 1, 1023)]:
        a = ctx.rng_floats(seed, pix, smp, 64)
        b = orc.rng_floats(seed, pix, smp, 64)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        assert a.min() >= 0.0 and a.max() < 1.0


@pytest.mark.parametrize("layout", [capi.LAYOUT_DENSE, capi.LAYOUT_CELL8])
@pytest.mark.parametrize("shape", [(24, 24, 24), (23, 30, 20)])
def test_trilinear_lookup_indices_and_values_bit_exact(ctx, orc, layout, shape):
    rng = np.random.RandomState(0)
    data = rng.rand(*shape).astype(np.float32)
    mn, mx = [-1, -2, 0.5], [1, 2.5, 3]
    pts = np.stack([rng.uniform(mn[i] - 0.2, mx[i] + 0.2, 200000) for i in range(3)], 1).astype(np.float32)
    # exact nodes / faces too
    pts[:8] = [[-1, -2, 0.5], [1, 2.5, 3], [0, 0, 1], [1, 0, 1], [-1, 0, 1], [0, 2.5, 1], [0, -2, 3], [0, 0, 0.5]]
    vol = ctx.upload_volume(data, mn, mx, layout)
    v, idx = ctx.lookup_trilinear(vol, pts)
    vo, idxo = orc.lookup_trilinear(data, mn, mx, pts)
    assert np.array_equal(idx, idxo)                      # integer grid-index arithmetic: bit-exact
    assert np.array_equal(v.view(np.uint32), vo.view(np.uint32))   # same blend order, no contraction
    vol.destroy()


def test_trilinear_lookup_u8_and_rgb(ctx, orc):
    rng = np.random.RandomState(1)
    d8 = rng.randint(0, 256, size=(17, 19, 21)).astype(np.uint8)
    pts = scenes.rand_points(50000, -1.1, 1.1)
    vol = ctx.upload_volume(d8, [-1] * 3, [1] * 3)
    v, idx = ctx.lookup_trilinear(vol, pts)
    vo, idxo = orc.lookup_trilinear(d8, [-1] * 3, [1] * 3, pts)
    assert np.array_equal(idx, idxo) and np.array_equal(v.view(np.uint32), vo.view(np.uint32))
    rgb = rng.rand(12, 13, 14, 3).astype(np.float32)
    vol3 = ctx.upload_volume(rgb, [-1] * 3, [1] * 3)
    a = ctx.lookup_trilinear_rgb(vol3, pts)
    b = orc.lookup_trilinear_rgb(rgb, [-1] * 3, [1] * 3, pts)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    with pytest.raises(capi.MerError):
        ctx.lookup_trilinear(vol3, pts)      # lookupFloat on a 3-channel volume (gridvolume.cpp:570)


@pytest.mark.parametrize("layout", [capi.LAYOUT_DENSE, capi.LAYOUT_CELL8])
def test_trilinear_value_grad_bit_exact(ctx, orc, layout):
    data = synth.radial_rif(20)
    pts = scenes.rand_points(100000, -1.02, 1.02)
    vol = ctx.upload_volume(data, [-1] * 3, [1] * 3, layout)
    v, g = ctx.rif_value_grad(vol, P.RIF_TRILINEAR, pts)
    vo, go = orc.trilinear_value_grad(data, [-1] * 3, [1] * 3, pts)
    assert np.array_equal(v.view(np.uint32), vo.view(np.uint32))
    assert np.array_equal(g.view(np.uint32), go.view(np.uint32))


@pytest.mark.parametrize("shape", [(40, 33, 70), (16, 130, 17), (97, 16, 64), (20, 18, 16), (18, 24, 128), (17, 19, 36)])
def test_bspline_parallel_prefilter_matches_sequential_and_oracle(ctx, orc, shape, monkeypatch):
    """K_prefilter's segmented form (warm-up 16, segments of 64; interior segments without guards, border segments guarded) against the one-thread-per-line form and the oracle's build3d
    (include/mitsuba/core/basisspline.h:812-890): lines shorter than, equal to and longer than a segment + warm-up, ragged tails."""
    rng = np.random.RandomState(7)
    data = (1.3 + 0.3 * rng.rand(*shape)).astype(np.float32)
    par = ctx.upload_volume(data, [-1] * 3, [1] * 3).build_spline().download_spline()
    with ctx.options(prefilter=1):
        seq = ctx.upload_volume(data, [-1] * 3, [1] * 3).build_spline().download_spline()
    assert np.abs(par - seq).max() < 2e-6               # same recursion per sample; only the truncated warm-up differs
    assert np.abs(par - orc.bspline_build(data)).max() < 5e-6


def test_bspline_prefilter_and_eval(ctx, orc):
    rng = np.random.RandomState(3)
    shape = (17, 20, 33)
    data = (1.3 + 0.3 * rng.rand(*shape)).astype(np.float32)
    mn, mx = [-1, -2, 0], [1, 2, 3]
    vol = ctx.upload_volume(data, mn, mx).build_spline()
    coeff = vol.download_spline()
    co = orc.bspline_build(data)
    # tolerance: fp32 recursion, reference build flags are not bit-reproducible
    assert np.abs(coeff - co).max() < 5e-6
    stride = np.array([(mx[i] - mn[i]) / (shape[2 - i] - 1) for i in range(3)])
    lo = np.array(mn) + 2.01 * stride; hi = np.array(mx) - 2.01 * stride
    pts = np.stack([rng.uniform(lo[i], hi[i], 50000) for i in range(3)], 1).astype(np.float32)
    v, g = ctx.rif_value_grad(vol, P.RIF_BSPLINE3, pts)
    vo, go = orc.bspline_eval(co, mn, mx, pts)
    assert np.abs(v - vo).max() < 2e-5
    assert np.abs(g - go).max() < 4e-4        # gradient of O(1) random data times dxres ~ 16
    c64 = orc.bspline_build(data, double=True)
    v64, g64 = orc.bspline_eval(c64, mn, mx, pts.astype(np.float64))
    assert np.abs(v - v64).max() < 2e-5


@pytest.mark.parametrize("g", [0.9, -0.3, 0.8, 0.0])
def test_hg_sample_eval(ctx, orc, g):
    kind = P.PHASE_HG
    wi = scenes.rand_dirs(100000)
    u2 = np.random.RandomState(5).rand(100000, 2).astype(np.float32)
    wo, pdf = ctx.phase_sample(kind, g, wi, u2)
    woo, pdfo = orc.phase_sample(kind, g, wi, u2)
    assert np.abs(wo - woo).max() < 2e-5 and np.abs(pdf - pdfo).max() < 2e-4 * max(1.0, pdfo.max())
    ev = ctx.phase_eval(kind, g, wi, wo)
    assert np.allclose(ev, pdf, rtol=1e-5, atol=1e-7)
    # mean cosine = g (HGPhaseFunction::getMeanCosine); wi points away, so the scattered direction is -wi-relative
    cos = -(wi * wo).sum(1)
    assert abs(cos.mean() - g) < 5e-3
    wo_i, pdf_i = ctx.phase_sample(P.PHASE_ISOTROPIC, 0.0, wi, u2)
    assert np.allclose(pdf_i, 1 / (4 * np.pi))
    with pytest.raises(capi.MerError):
        ctx.phase_sample(kind, 1.0, wi[:4], u2[:4])       # hg.cpp:52-53


@pytest.mark.parametrize("kind,g", [(P.PHASE_ISOTROPIC, 0.0), (P.PHASE_HG, 0.9), (P.PHASE_HG, -0.3)])
def test_phase_chisquare_reference_fixture_on_the_hip_path(ctx, kind, g):
    """The one fixture the reference holds for a hot-path row (A9), applied to the HIP kernels themselves: src/tests/test_chisquare.cpp:508-573
    on the isotropic / hg g=0.9 / hg g=-0.3 plugins of data/tests/test_phase.xml -- wiSamples = 20 incident directions, 10 x 20 (theta, phi)
    bins, 200 000 samples each drawn by mer_phase_sample, expected frequencies integrated from mer_phase_eval, significance 0.0025 with
    Sidak correction over the 20 tests, cells pooled below 5.  No oracle involved: this is the reference's own acceptance test."""
    from tests.test_oracle_kat import _chi2_phase
    rng = np.random.RandomState(42)
    wis = scenes.rand_dirs(20, seed=7)
    for wi in wis:
        pval, alpha = _chi2_phase(lambda a, b: ctx.phase_sample(kind, g, a, b), lambda a, b: ctx.phase_eval(kind, g, a, b), wi, rng)
        assert pval >= alpha, (kind, g, wi, pval, alpha)


def test_camera_rays(ctx, orc):
    p = scenes.straight_scene()
    sc, vols = ctx.upload_scene(p)
    pos = np.random.RandomState(4).uniform(0, 48, size=(4096, 2)).astype(np.float32)
    o, d = ctx.camera_rays(sc, pos)
    oo, do = orc.camera_rays(p, pos)
    assert np.array_equal(o, oo)
    assert np.abs(d - do).max() < 1e-6
    # pixel centre of the image maps to the optical axis (+x)
    o, d = ctx.camera_rays(sc, np.array([[24.0, 20.0]], np.float32))
    assert np.allclose(d[0], [1, 0, 0], atol=1e-6)


@pytest.mark.parametrize("buffer_loads", [1, 0])
@pytest.mark.parametrize("layout", ["cell8", "brick27"])
def test_er_trace_and_connect_in_the_record_layouts(ctx, orc, layout, buffer_loads):
    """the leaf entry points on the record layouts the renders use (BRICK27 = the bench layout; buffer loads below 4 GiB, global loads as a
    >= 4 GiB field selects): mer_er_trace against the oracle AND bit for bit against the dense layout; mer_connect equal to the dense layout's"""
    lay = {"cell8": capi.LAYOUT_CELL8, "brick27": capi.LAYOUT_BRICK27}[layout]
    p = scenes.curved_scene(N=24, rif="radial", stepper=P.STEP_RK4)
    n = 4096
    p0 = scenes.rand_points(n, -0.9, 0.9); d0 = scenes.rand_dirs(n)
    dist = np.random.RandomState(9).uniform(0.0, 1.5, n).astype(np.float32); dist[::7] = np.inf
    sd, vd = ctx.upload_scene(p, layout=capi.LAYOUT_DENSE)
    dense = ctx.er_trace(sd, p0, d0, dist)
    a = scenes.rand_points(256, -0.6, 0.6, seed=3); b = scenes.rand_points(256, -0.6, 0.6, seed=4)
    cd = ctx.connect(sd, a, b, 7)
    with ctx.options(buffer_loads=buffer_loads):
        sc, vols = ctx.upload_scene(p, layout=lay)
        got = ctx.er_trace(sc, p0, d0, dist)
        cg = ctx.connect(sc, a, b, 7)
    for x, y in zip(got, dense):
        assert np.array_equal(x, y)                      # the record layouts return the same corner values: not a bit differs
    assert np.array_equal(cg, cd)
    rp, rv, rds, roo, rok = orc.er_trace(p, p0, d0, dist)
    same = got[4] == rok
    assert same.mean() > 0.999 and np.abs(got[0] - rp)[same].max() < 2e-5 and np.abs(got[3] - roo)[same].max() < 1e-3
    assert cd[:, 0].mean() > 0.8                         # most pairs connect
    for v in vols + vd:
        v.destroy()


@pytest.mark.parametrize("stepper", [P.STEP_VERLET, P.STEP_RK4])
@pytest.mark.parametrize("rifkind", ["trilinear", "bspline"])
def test_er_trace(ctx, orc, stepper, rifkind):
    p = scenes.curved_scene(N=24, stepper=stepper) if rifkind == "trilinear" else scenes.bspline_scene(N=24, stepper=stepper)
    sc, vols = ctx.upload_scene(p)
    n = 4096
    p0 = scenes.rand_points(n, -0.9, 0.9)
    d0 = scenes.rand_dirs(n)
    dist = np.random.RandomState(9).uniform(0.0, 1.5, n).astype(np.float32)
    dist[::7] = np.inf                                   # traceTillBoundary
    op, ov, ds, oo, ok = ctx.er_trace(sc, p0, d0, dist)
    rp, rv, rds, roo, rok = orc.er_trace(p, p0, d0, dist)
    same = ok == rok
    assert same.mean() > 0.999
    tol = 1e-6 if rifkind == "trilinear" else 2e-4
    assert np.abs(op - rp)[same].max() < max(tol, 2e-5)
    assert np.abs(ov - rv)[same].max() < max(tol * 10, 2e-4)
    assert np.abs(ds - rds)[same].max() < 1e-4
    assert np.abs(oo - roo)[same].max() < 1e-3
    # eikonal invariant: |v| = n(p) along the ray
    nv, _ = ctx.rif_value_grad(vols[-1], P.RIF_TRILINEAR if rifkind == "trilinear" else P.RIF_BSPLINE3, op)
    assert np.abs(np.linalg.norm(ov, axis=1) - nv).max() < 5e-3


@pytest.mark.parametrize("stepper", [P.STEP_VERLET, P.STEP_RK4])
@pytest.mark.parametrize("rifkind", ["trilinear", "bspline"])
def test_er_trace_fp32_against_the_fp64_oracle(ctx, orc, stepper, rifkind):
    """D6: the reference's refractive path is `FLOAT` = double in one of its two build configurations (config_release.py:7) and float in the other;
    the GPU is fp32.  Stated tolerance against the oracle built in double, at the reference's default step (h = 1e-3, ~1000 steps over a distance
    of 1): position 6e-5, momentum 2e-4, optical length 5e-5 (observed on the CPU, fp32 vs fp64 oracle: 1.3e-5, 3.5e-5, 6e-6) -- the fp32 round-off of
    a thousand accumulated steps, not a modelling difference."""
    mk = scenes.curved_scene if rifkind == "trilinear" else scenes.bspline_scene
    p = mk(N=24, stepper=stepper, stepsize=1e-3)
    sc, vols = ctx.upload_scene(p)
    n = 2048
    p0 = scenes.rand_points(n, -0.45, 0.45); d0 = scenes.rand_dirs(n)
    dist = np.random.RandomState(2).uniform(0.2, 1.0, n).astype(np.float32)
    op, ov, ds, oo, ok = ctx.er_trace(sc, p0, d0, dist)
    rp, rv, rds, roo, rok = orc.er_trace(p.copy(rif_double=1), p0, d0, dist)
    same = (ok == rok) & (ok == 1)
    assert same.mean() > 0.6 and (ok == rok).mean() > 0.999
    assert np.abs(op - rp)[same].max() < 6e-5, np.abs(op - rp)[same].max()
    assert np.abs(ov - rv)[same].max() < 2e-4
    assert np.abs(oo - roo)[same].max() < 5e-5


@pytest.mark.parametrize("mode", ["woodcock", "simpson", "simpson_stepsize", "homogeneous", "homogeneous_maximum", "refractive_maximum", "refractive_homog", "composed_rk4", "composed_verlet"])
def test_sample_distance(ctx, orc, mode):
    if mode == "woodcock":
        p = scenes.straight_scene(N=24)
    elif mode == "simpson":                              # method = simpson: invertDensityIntegral (heterogeneous.cpp:419-544)
        p = scenes.straight_scene(N=24, method=P.METHOD_SIMPSON)
    elif mode == "simpson_stepsize":
        p = scenes.straight_scene(N=24, method=P.METHOD_SIMPSON, het_stepsize=0.013, albedo_mode=P.ALBEDO_GRID, albedo_grid=scenes.rgb_albedo(24))
    elif mode == "homogeneous":
        p = scenes.homogeneous_scene()
    elif mode == "homogeneous_maximum":                  # strategy = maximum: MaxExpDist (src/medium/maxexp.h)
        p = scenes.homogeneous_scene(strategy=P.STRATEGY_MAXIMUM)
    elif mode == "refractive_maximum":
        p = scenes.curved_scene(N=24, sigma_mode=P.SIGMA_HOMOGENEOUS, stepper=P.STEP_VERLET, strategy=P.STRATEGY_MAXIMUM)
    elif mode == "refractive_homog":
        p = scenes.curved_scene(N=24, sigma_mode=P.SIGMA_HOMOGENEOUS, stepper=P.STEP_VERLET)
    elif mode == "composed_rk4":
        p = scenes.curved_scene(N=24, stepper=P.STEP_RK4)
    else:
        p = scenes.curved_scene(N=24, stepper=P.STEP_VERLET)
    sc, vols = ctx.upload_scene(p)
    n = 8192
    o = scenes.rand_points(n, -0.95, 0.95)
    d = scenes.rand_dirs(n)
    maxt = np.random.RandomState(3).uniform(0.2, 2.0, n).astype(np.float32)
    a = ctx.sample_distance(sc, o, d, maxt, 11)
    b = orc.sample_distance(p, o, d, maxt, 11)
    same = a[:, 0] == b[:, 0]
    assert same.mean() > 0.995          # same RNG stream => same accept/reject decisions (up to libm ulps)
    ok = same & (a[:, 0] == 1)
    assert np.abs(a - b)[ok].max() < 2e-3                        # full record on a medium interaction
    assert np.median(np.abs(a - b)[ok].max(1)) < 1e-5
    fl = same & (a[:, 0] == 0)                                    # failure: only transmittance / pdfs / refRatioSq are defined
    if mode == "simpson":                                         # pdfSuccess = expVal * densityAtT, sigmaS = albedo * densityAtT (heterogeneous.cpp:600-608)
        assert 0.2 < a[:, 0].mean() < 0.95
        np.testing.assert_allclose(a[ok][:, 11], a[ok][:, 12] * a[ok][:, 5] / 0.9, rtol=1e-5)
    assert np.abs(a - b)[fl][:, 8:14].max() < 2e-3
    if p.rif_mode != P.RIF_CONST:                                 # refractive media also report the exit point and momentum
        assert np.abs(a - b)[fl][:, 1:5].max() < 2e-3 and np.abs(a - b)[fl][:, 14:17].max() < 2e-3


@pytest.mark.parametrize("est", [P.TR_WOODCOCK2, P.TR_RATIO])
@pytest.mark.parametrize("curved", [False, True])
def test_eval_transmittance(ctx, orc, est, curved):
    _eval_transmittance(ctx, orc, dict(tr_estimator=est), curved)


def test_eval_transmittance_simpson(ctx, orc):
    """method = simpson: integrateDensity (heterogeneous.cpp:301-376) -- deterministic, so every ray agrees (libm exp / ulp-level sums)"""
    p = scenes.straight_scene(N=24, method=P.METHOD_SIMPSON)
    sc, vols = ctx.upload_scene(p)
    n = 8192
    o = scenes.rand_points(n, -1.4, 1.4); d = scenes.rand_dirs(n)
    maxt = np.random.RandomState(3).uniform(0.0, 3.0, n).astype(np.float32)
    a = ctx.eval_transmittance(sc, o, d, maxt, 5); b = orc.eval_transmittance(p, o, d, maxt, 5)
    assert np.abs(a - b).max() < 2e-6 and 0.05 < a.mean() < 0.95
    dense = p.copy(density_scale=300.0)                              # the early exit (HETVOL_EARLY_EXIT): +inf => exactly 0
    sc2, vols2 = ctx.upload_scene(dense)
    a2 = ctx.eval_transmittance(sc2, o, d, maxt, 5); b2 = orc.eval_transmittance(dense, o, d, maxt, 5)
    assert np.array_equal(a2 == 0, b2 == 0) and (a2 == 0).mean() > 0.2 and np.abs(a2 - b2).max() < 2e-6


def _eval_transmittance(ctx, orc, kw, curved):
    est = kw["tr_estimator"]
    p = scenes.curved_scene(N=24, tr_estimator=est) if curved else scenes.straight_scene(N=24, tr_estimator=est)
    sc, vols = ctx.upload_scene(p)
    n = 8192
    o = scenes.rand_points(n, -0.9, 0.9)
    d = scenes.rand_dirs(n)
    maxt = np.random.RandomState(3).uniform(0.2, 1.5, n).astype(np.float32)
    a = ctx.eval_transmittance(sc, o, d, maxt, 5)
    b = orc.eval_transmittance(p, o, d, maxt, 5)
    close = np.abs(a - b).max(1) < 1e-4
    assert close.mean() > 0.995
    # both estimators are unbiased for exp(-integral sigma_t): compare means (MC tolerance)
    assert abs(a.mean() - b.mean()) < 5e-3


@pytest.mark.parametrize("kind", ["bspline", "trilinear", "trilinear_sdf"])
def test_curved_ray_connection(ctx, orc, kind):
    """A12: the shooting solver finds the optical momentum at p1 whose eikonal ray passes through p2.  Parity with the
    reference's Ceres iterates is unpinned; pinned here: the converged ray (direction, length, optical
    length) against the oracle's solver, and the physical property itself -- tracing the found ray lands on p2."""
    p = scenes.bspline_scene(N=24) if kind == "bspline" else scenes.curved_scene(N=24, rif="radial", stepper=P.STEP_VERLET)
    if kind == "trilinear_sdf":         # the medium shape as a signed-distance grid (K_connect's BND = 1 code, global loads)
        from mitsubaer_amd import synth
        box = ([-1.2] * 3, [1.2] * 3)
        p = p.copy(boundary=P.BOUNDARY_SDF, sdf=-synth.sphere_sdf(64, radius=0.9, aabb_min=box[0], aabb_max=box[1]), sdf_aabb=box)
    sc, vols = ctx.upload_scene(p)
    rng = np.random.RandomState(0)
    n = 512
    lim = 0.45 if kind == "trilinear_sdf" else 0.6
    p1 = rng.uniform(-lim, lim, (n, 3)).astype(np.float32); p2 = rng.uniform(-lim, lim, (n, 3)).astype(np.float32)
    a = ctx.connect(sc, p1, p2, 1)
    b = orc.connect(p, p1, p2, 1)
    ok = (a[:, 0] == 1) & (b[:, 0] == 1)
    assert (a[:, 0] == 1).mean() > 0.85 and (a[:, 0] == b[:, 0]).mean() > 0.93
    da = a[ok, 2:5] / np.linalg.norm(a[ok, 2:5], axis=1, keepdims=True)
    db = b[ok, 2:5] / np.linalg.norm(b[ok, 2:5], axis=1, keepdims=True)
    assert np.abs(da - db).max() < 5e-3                         # both solvers stop at |r|^2/2 < 1e-6
    assert np.abs(a[ok, 8] - b[ok, 8]).max() < 5e-3 and np.abs(a[ok, 9] - b[ok, 9]).max() < 1e-2
    assert np.all(a[ok, 1] >= 1.0)                              # weight = (#agreeing solutions) / RR probability
    # the connection is a ray: re-trace it with the reference's Verlet trace() and land on p2
    q = p.copy(stepper=P.STEP_VERLET, boundary=P.BOUNDARY_AABB, sdf=None)        # mer_er_trace knows the cube / sphere boundaries
    scq, _ = ctx.upload_scene(q)
    op, ov, ds, oo, okk = ctx.er_trace(scq, p1[ok], da, a[ok, 8])
    assert np.abs(op - p2[ok]).max() < 3e-3                     # sqrt(2 tol2) = 1.4e-3 plus bisection granularity
    # curved, not straight: the arc is longer than the chord and the launch direction differs from it
    chord = np.linalg.norm(p2[ok] - p1[ok], axis=1)
    assert (a[ok, 8] >= chord - 5e-3).all()                      # h = 0.043 here; closest approach within sqrt(2 tol2) of p2
    n1, _ = ctx.rif_value_grad([v for v in vols if v is not None][1 if kind == "trilinear_sdf" else -1], P.RIF_BSPLINE3 if kind == "bspline" else P.RIF_TRILINEAR, p1[ok])
    assert np.abs(np.linalg.norm(a[ok, 2:5], axis=1) - n1).max() < 1e-4       # |v0| = n(p1)


def test_lookups_under_a_rotated_data_box(ctx, orc):
    """A2 with the volume plugin's `toWorld` (gridvolume.cpp:110,188-195): worldToGrid = scale * translate * toWorld^-1 is a full
    affine map.  The integer contract (cell, bounds test, linear index) stays bit-exact against the oracle; values to rounding."""
    rng = np.random.RandomState(5)
    data = rng.rand(17, 19, 23).astype(np.float32)
    rgb = rng.rand(9, 8, 7, 3).astype(np.float32)
    tw = P.rotation([1, 2, 3], 35.0, [0.05, -0.1, 0.08])
    mn, mx = [-1, -0.8, -0.6], [1, 0.9, 0.7]
    pts = scenes.rand_points(200000, -1.3, 1.3, seed=9)
    for lay in (capi.LAYOUT_DENSE, capi.LAYOUT_CELL8):
        v = ctx.upload_volume(data, mn, mx, lay, to_world=tw)
        val, idx = ctx.lookup_trilinear(v, pts)
        rv, ri = orc.lookup_trilinear(data, mn, mx, pts, to_world=tw)
        assert np.array_equal(idx, ri) and (ri[:, 3] >= 0).mean() > 0.2 and (ri[:, 3] < 0).mean() > 0.2
        assert np.array_equal(val.view(np.uint32), rv.view(np.uint32))
        v.destroy()
    v = ctx.upload_volume(rgb, mn, mx, to_world=tw)
    assert np.array_equal(ctx.lookup_trilinear_rgb(v, pts).view(np.uint32), orc.lookup_trilinear_rgb(rgb, mn, mx, pts, to_world=tw).view(np.uint32))
    v.destroy()
    with pytest.raises(capi.MerError, match="not invertible"):
        ctx.upload_volume(data, mn, mx, to_world=np.diag([1e20, 1.0, 1.0, 1.0]))        # its inverse has determinant 1e-20


@pytest.mark.parametrize("interp", ["trilinear", "bspline"])
def test_rif_value_and_gradient_under_a_rotated_data_box(ctx, orc, interp):
    """RIF volumes with a `toWorld`: the point goes through worldToVolume, the gradient comes back through its rotation transposed
    (splinevolume.cpp:343,359); the gradient of a field that is linear in VOLUME space is the rotated constant vector."""
    N = 20
    tw = P.rotation([0, 0, 1], 30.0, [0.1, 0.0, -0.05])
    ax = np.linspace(-1.3, 1.3, N)
    z, y, x = np.meshgrid(ax, ax, ax, indexing="ij")
    data = (1.4 + 0.1 * x - 0.05 * y + 0.02 * z).astype(np.float32)
    p = scenes.bspline_scene(N=16) if interp == "bspline" else scenes.curved_scene(N=16)
    p = p.copy(rif=data, rif_aabb=([-1.3] * 3, [1.3] * 3), rif_to_world=tw)
    sc, vols = ctx.upload_scene(p)
    pts = scenes.rand_points(4096, -0.55, 0.55, seed=3)                  # inside the spline-safe box also after the rotation
    val, grad = ctx.rif_value_grad(vols[-1], P.RIF_BSPLINE3 if interp == "bspline" else P.RIF_TRILINEAR, pts)
    rv, rg, _ = orc.rif_eval(p, pts)
    assert np.abs(val - rv).max() < 2e-5 and np.abs(grad - rg).max() < 2e-4
    g_world = np.asarray(tw)[:3, :3] @ np.array([0.1, -0.05, 0.02])
    assert np.abs(grad - g_world[None, :]).max() < 2e-3
    for v in vols:
        v.destroy()


def test_connection_through_the_boundary(ctx, orc):
    """A12 boundary branch on the GPU (Connector::computefdf / path_lengths with cross = true): the refracted chord of a constant
    index (closed form), and agreement with the oracle on a radial field -- cube, sphere and signed-distance boundaries"""
    from tests.test_oracle_kat import _refracted_chord_check, _outside_pairs
    n0, R, N = 1.4, 0.8, 24
    p = scenes.curved_scene(N=N, rif=np.full((N, N, N), n0, np.float32), boundary=P.BOUNDARY_SPHERE, sph_radius=R, stepper=P.STEP_VERLET)
    p1, p2 = _outside_pairs()
    sc, vols = ctx.upload_scene(p)
    frac, off, dl, do, dn = _refracted_chord_check(ctx.connect(sc, p1, p2, 1), p1, p2, n0, R)
    assert frac > 0.75 and off < 3e-4 and dl < 3e-4 and do < 5e-4 and dn < 1e-6, (frac, off, dl, do, dn)
    for v in vols:
        v.destroy()
    box = ([-1.2] * 3, [1.2] * 3)
    for kw in (dict(boundary=P.BOUNDARY_SPHERE, sph_radius=0.8), dict(), dict(boundary=P.BOUNDARY_SDF, sdf=-synth.sphere_sdf(64, radius=0.8, aabb_min=box[0], aabb_max=box[1]), sdf_aabb=box),
               dict(boundary=P.BOUNDARY_SPHERE, sph_radius=0.8, boundary_bsdf=P.BSDF_HDIELECTRIC)):
        q = scenes.curved_scene(N=N, rif="radial", stepper=P.STEP_VERLET, **kw)
        sc, vols = ctx.upload_scene(q)
        p1, p2 = _outside_pairs(256, seed=3)
        if not kw:
            p2 = p2 * 1.6                                      # outside the cube too
        a = ctx.connect(sc, p1, p2, 2); b = orc.connect(q, p1, p2, 2)
        ok = (a[:, 0] == 1) & (b[:, 0] == 1)
        assert (a[:, 0] == b[:, 0]).mean() > 0.9 and ok.mean() > 0.25, ((a[:, 0] == b[:, 0]).mean(), ok.mean())   # one random start each (Russian roulette 1e-2)
        da = a[ok, 2:5] / np.linalg.norm(a[ok, 2:5], axis=1, keepdims=True); db = b[ok, 2:5] / np.linalg.norm(b[ok, 2:5], axis=1, keepdims=True)
        same = np.abs(da - db).max(1) < 5e-3             # through a refracting boundary two solvers may settle on different rays: both valid
        assert same.mean() > 0.9, same.mean()
        A, B = a[ok][same], b[ok][same]
        assert np.abs(A[:, 8] - B[:, 8]).max() < 5e-3 and np.abs(A[:, 9] - B[:, 9]).max() < 1e-2
        assert np.abs(A[:, 1] - B[:, 1]).max() < 2e-3 * np.abs(B[:, 1]).max()                 # weight: restarts x boundary BSDF
        for v in vols:
            v.destroy()
