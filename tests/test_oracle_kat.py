"""Pins the CPU oracle on every fixture the reference offers for this path, plus the
analytic known answers of SURVEY section 7.3.  CPU only."""
import numpy as np
import pytest
from scipy import stats
from mitsubaer_amd import params as P, synth
from tests import scenes


# ---------------------------------------------------------------------------------------------------
# A5: the one numeric known answer recorded from the reference's own basisspline.h (SURVEY 8c probe):
# 9x8x7 grid on [-1,1]x[-2,2]x[0,3], data 1.3+0.05i+0.01j^2-0.02k (float ops), query (0.13,-0.4,1.7)
def _probe_data():
    f = np.float32
    k, j, i = np.meshgrid(np.arange(7), np.arange(8), np.arange(9), indexing="ij")
    return ((f(1.3) + f(0.05) * i.astype(f)) + (f(0.01) * (j * j).astype(f))) - (f(0.02) * k.astype(f))


def test_bspline_known_answer_fp64(orc):
    c = orc.bspline_build(_probe_data().astype(np.float32), double=True)
    v, g, h = orc.bspline_eval(c, [-1, -2, 0], [1, 2, 3], np.array([[0.13, -0.4, 1.7]]), hessian=True)
    assert abs(v[0] - 1.53604451) < 2e-8
    assert np.abs(g[0] - [0.199610477, 0.0975018531, -0.0408002379]).max() < 2e-8
    assert abs(h[0][0] - 0.0257299574) < 2e-8


def test_bspline_known_answer_fp32(orc):
    c = orc.bspline_build(_probe_data().astype(np.float32))
    v, g, h = orc.bspline_eval(c, [-1, -2, 0], [1, 2, 3], np.array([[0.13, -0.4, 1.7]], np.float32), hessian=True)
    # fp32 results of the reference are compiler-flag dependent: a few ulp
    assert abs(v[0] - 1.53604436) < 1e-6
    assert np.abs(g[0] - [0.199609831, 0.0975019857, -0.0408001691]).max() < 2e-6
    assert abs(h[0][0] - 0.0257304087) < 2e-6


def test_bspline_interpolates_nodes_and_linear_fields(orc):
    rng = np.random.RandomState(0)
    shape = (11, 12, 13)
    data = rng.rand(*shape).astype(np.float32)
    mn, mx = [0, 0, 0], [1, 2, 3]
    c = orc.bspline_build(data, double=True)
    idx = np.array([[i, j, k] for k in range(2, shape[0] - 2) for j in range(2, shape[1] - 2) for i in range(2, shape[2] - 2)])
    pts = np.stack([mn[a] + (mx[a] - mn[a]) * idx[:, a] / (shape[2 - a] - 1) for a in range(3)], 1)
    v, _ = orc.bspline_eval(c, mn, mx, pts)
    assert np.abs(v - data[idx[:, 2], idx[:, 1], idx[:, 0]]).max() < 1e-7      # value(node) == data
    lin = synth.linear_rif(0, shape=(11, 41, 13))       # mirror boundary: the kink decays as (sqrt(3)-2)^k
    c = orc.bspline_build(lin, double=True)
    q = rng.uniform([0.3, 0.8, 0.8], [0.7, 1.2, 2.2], size=(100, 3))
    v, g = orc.bspline_eval(c, mn, mx, q)
    assert np.abs(v - (1.3 + 0.3 * q[:, 1] / 2)).max() < 1e-6
    assert np.abs(g - [0, 0.15, 0]).max() < 1e-5


# ---------------------------------------------------------------------------------------------------
# A9: the reference's own phase-function fixture: src/tests/test_chisquare.cpp:508-573 on
# data/tests/test_phase.xml (isotropic; hg g=0.9; hg g=-0.3): 10x20 (theta,phi) bins, 20 incident
# directions, 200000 samples each, significance 0.0025 with Sidak correction, cells pooled below 5.
def _chi2_phase(sample_fn, pdf_fn, wi, rng):
    tb, pb, n = 10, 20, 10 * 20 * 1000
    u2 = rng.rand(n, 2).astype(np.float32)
    wo, _ = sample_fn(np.repeat(wi[None], n, 0), u2)
    theta = np.arccos(np.clip(wo[:, 2], -1, 1)); phi = np.arctan2(wo[:, 1], wo[:, 0]); phi[phi < 0] += 2 * np.pi
    ti = np.clip(np.floor(theta * tb / np.pi).astype(int), 0, tb - 1)
    pi_ = np.clip(np.floor(phi * pb / (2 * np.pi)).astype(int), 0, pb - 1)
    table = np.bincount(ti * pb + pi_, minlength=tb * pb).astype(np.float64)
    m = 24                                             # midpoint rule per cell (reference: adaptive NDIntegrator)
    th = (np.arange(tb * m) + 0.5) * np.pi / (tb * m); ph = (np.arange(pb * m) + 0.5) * 2 * np.pi / (pb * m)
    T, Ph = np.meshgrid(th, ph, indexing="ij")
    d = np.stack([np.sin(T) * np.cos(Ph), np.sin(T) * np.sin(Ph), np.cos(T)], -1).reshape(-1, 3).astype(np.float32)
    pdf = pdf_fn(np.repeat(wi[None], d.shape[0], 0), d).reshape(tb * m, pb * m) * np.sin(T)
    ref = pdf.reshape(tb, m, pb, m).sum((1, 3)) * (np.pi / (tb * m)) * (2 * np.pi / (pb * m)) * n
    order = np.argsort(ref.ravel())
    chsq, df, pc, pr = 0.0, 0, 0.0, 0.0
    pooled = 0
    for i in order:
        e, o = ref.ravel()[i], table[i]
        if e == 0:
            assert o <= n * 1e-4
        elif e < 5 or (0 < pr < 5):
            pc += o; pr += e; pooled += 1
        else:
            chsq += (o - e) ** 2 / e; df += 1
    if pooled:
        chsq += (pc - pr) ** 2 / pr; df += 1
    df -= 1
    pval = 1 - stats.chi2.cdf(chsq, df)
    alpha = 1 - (1 - 0.0025) ** (1.0 / 20)
    return pval, alpha


@pytest.mark.parametrize("kind,g", [(P.PHASE_ISOTROPIC, 0.0), (P.PHASE_HG, 0.9), (P.PHASE_HG, -0.3)])
def test_phase_chisquare_reference_fixture(orc, kind, g):
    rng = np.random.RandomState(42)
    wis = scenes.rand_dirs(20, seed=7)
    for wi in wis:                                      # wiSamples = 20 incident directions, as the reference (test_chisquare.cpp:515)
        pval, alpha = _chi2_phase(lambda a, b: orc.phase_sample(kind, g, a, b), lambda a, b: orc.phase_eval(kind, g, a, b), wi, rng)
        assert pval >= alpha, (pval, alpha)


def test_hg_mean_cosine_and_pdf_normalisation(orc):
    rng = np.random.RandomState(1)
    for g in (0.9, -0.3, 0.8, 0.0):
        wi = np.repeat(np.array([[0.3, -0.5, 0.81]], np.float32) / np.linalg.norm([0.3, -0.5, 0.81]), 200000, 0).astype(np.float32)
        wo, pdf = orc.phase_sample(P.PHASE_HG, g, wi, rng.rand(200000, 2).astype(np.float32))
        assert abs((-(wi * wo).sum(1)).mean() - g) < 4e-3          # getMeanCosine() == g
        assert np.allclose(np.linalg.norm(wo, axis=1), 1, atol=1e-5)
        assert np.allclose(orc.phase_eval(P.PHASE_HG, g, wi, wo), pdf, rtol=1e-6)


# ---------------------------------------------------------------------------------------------------
# A1/A2: VOL v3 format (mfiles/Test.m recipe: 20x30x23 grid on [0,1]^3) and trilinear lookup
def test_trilinear_lookup_contract(orc):
    rng = np.random.RandomState(1)
    data = rng.rand(23, 30, 20).astype(np.float32)          # [z][y][x]: the MATLAB array is 20x30x23 (x,y,z)
    mn, mx = [0, 0, 0], [1, 1, 1]
    idx = np.array([[3, 4, 5], [0, 0, 0], [18, 28, 21]])
    pts = np.stack([idx[:, a] / (np.array([20, 30, 23])[a] - 1) for a in range(3)], 1).astype(np.float32)
    v, ii = orc.lookup_trilinear(data, mn, mx, pts + 1e-4)
    assert np.array_equal(ii[:, :3], idx)
    assert np.array_equal(ii[:, 3], (idx[:, 2] * 30 + idx[:, 1]) * 20 + idx[:, 0])
    assert np.abs(v - data[idx[:, 2], idx[:, 1], idx[:, 0]]).max() < 2e-2
    # reject outside and on the max face (x2 >= res): gridvolume.cpp:344-346
    v, ii = orc.lookup_trilinear(data, mn, mx, np.array([[1.0, 0.5, 0.5], [-0.01, 0.5, 0.5], [0.5, 0.5, 1.2]], np.float32))
    assert np.all(v == 0) and np.all(ii[:, 3] == -1)
    # u8 path: m_densityMap[i] = i/255 (gridvolume.cpp:204-214)
    d8 = np.full((4, 4, 4), 255, np.uint8)
    v, _ = orc.lookup_trilinear(d8, mn, mx, np.array([[0.5, 0.5, 0.5]], np.float32))
    assert v[0] == 1.0


def test_trilinear_gradient_is_analytic(orc):
    data = synth.linear_rif(16)
    pts = scenes.rand_points(1000, -0.99, 0.99)
    v, g = orc.trilinear_value_grad(data, [-1] * 3, [1] * 3, pts)
    assert np.abs(v - (1.45 + 0.15 * pts[:, 1])).max() < 1e-5
    assert np.abs(g - [0, 0.15, 0]).max() < 1e-4


# ---------------------------------------------------------------------------------------------------
# A6/A7 analytic invariants
@pytest.mark.parametrize("stepper,tol", [(P.STEP_VERLET, 2e-4), (P.STEP_RK4, 2e-5)])
def test_linear_rif_translation_invariants(orc, stepper, tol):
    p = scenes.curved_scene(N=32, stepper=stepper, stepsize=2e-3)
    n = 256
    p0 = scenes.rand_points(n, -0.3, 0.3); d0 = scenes.rand_dirs(n)
    dist = np.full(n, 0.6, np.float32)
    op, ov, ds, oo, ok = orc.er_trace(p, p0, d0, dist)
    n0 = 1.45 + 0.15 * p0[:, 1]
    assert ok.all()
    assert np.abs(ov[:, 0] - n0 * d0[:, 0]).max() < tol       # n d_x conserved
    assert np.abs(ov[:, 2] - n0 * d0[:, 2]).max() < tol       # n d_z conserved
    n1 = 1.45 + 0.15 * op[:, 1]
    assert np.abs(np.linalg.norm(ov, axis=1) - n1).max() < 5e-4   # |v| = n (eikonal)
    assert np.allclose(ds, 0.6, atol=1e-5)
    assert np.abs(oo - 0.6 * 0.5 * (n0 + n1)).max() < 2e-3    # optical length ~ mean index * arc length


def test_radial_rif_bouguer_invariant(orc):
    p = scenes.curved_scene(N=48, rif="radial", stepper=P.STEP_RK4, stepsize=2e-3)
    n = 256
    p0 = scenes.rand_points(n, -0.4, 0.4); d0 = scenes.rand_dirs(n)
    op, ov, ds, oo, ok = orc.er_trace(p, p0, d0, np.full(n, 0.5, np.float32))
    r0 = np.linalg.norm(p0, axis=1) ** 2
    v0 = d0 * (2 - r0 / 3)[:, None]
    L0 = np.cross(p0, v0); L1 = np.cross(op, ov)
    assert np.abs(L1 - L0).max() < 3e-3                       # |r x n d| = const (trilinear grid error included)


def test_trace_exit_steps_back_inside(orc):
    p = scenes.curved_scene(N=24)
    n = 512
    p0 = scenes.rand_points(n, -0.9, 0.9); d0 = scenes.rand_dirs(n)
    op, ov, ds, oo, ok = orc.er_trace(p, p0, d0, np.full(n, np.inf, np.float32))
    assert not ok.any()
    assert (np.abs(op) <= 1 + 1e-6).all()                     # last inside point
    assert (np.abs(op).max(1) > 1 - 3 * p.stepsize).all()     # within ~a step of the boundary


# ---------------------------------------------------------------------------------------------------
# A3/A4/A8 known answers
def test_homogeneous_free_flight_and_transmittance(orc):
    p = scenes.homogeneous_scene(strategy=P.STRATEGY_SINGLE, channel=0, medium_sampling_weight=1.0,
                                 sigma_a=[0.5, 0.5, 0.5], sigma_s=[1.5, 1.5, 1.5])
    n = 200000
    o = np.zeros((n, 3), np.float32); d = np.tile(np.array([[1, 0, 0]], np.float32), (n, 1))
    rec = orc.sample_distance(p, o, d, np.full(n, 0.8, np.float32), 1)
    succ = rec[:, 0] == 1
    assert abs(succ.mean() - (1 - np.exp(-2 * 0.8))) < 4e-3       # free-flight CDF 1-exp(-sigma_t t)
    assert abs(rec[succ, 1].mean() - (1 / 2 - 0.8 * np.exp(-1.6) / (1 - np.exp(-1.6)))) < 4e-3
    # pdf identities homogeneous.cpp:317-343
    assert np.allclose(rec[succ, 11], 2 * np.exp(-2 * rec[succ, 1]), rtol=1e-5)
    assert np.allclose(rec[~succ, 12], np.exp(-1.6), rtol=1e-5)
    tr = orc.eval_transmittance(p, o[:4], d[:4], np.full(4, 0.8, np.float32), 1)
    assert np.allclose(tr, np.exp(-1.6), rtol=1e-6)


@pytest.mark.parametrize("est", [P.TR_WOODCOCK2, P.TR_RATIO])
def test_delta_tracking_transmittance_is_unbiased(orc, est):
    """E[tau] = exp(-int sigma_t) on a linear ramp sigma_t(x) = 4 * (0.5 + 0.25 x): analytic integral."""
    N = 33
    ramp = np.broadcast_to((0.5 + 0.25 * np.linspace(-1, 1, N, dtype=np.float32))[None, None, :], (N, N, N)).copy()
    p = scenes.straight_scene(N=8, tr_estimator=est)
    p.density = ramp
    n = 400000
    o = np.tile(np.array([[-0.5, 0.1, -0.2]], np.float32), (n, 1)); d = np.tile(np.array([[1, 0, 0]], np.float32), (n, 1))
    tr = orc.eval_transmittance(p, o, d, np.full(n, 1.0, np.float32), 3)[:, 0]
    exact = np.exp(-4 * (0.5 * 1.0 + 0.25 * (0.5 ** 2 - 0.5 ** 2) / 2))      # int_{-0.5}^{0.5} = 0.5
    assert abs(tr.mean() - exact) < 4 * tr.std() / np.sqrt(n) + 1e-4
    if est == P.TR_WOODCOCK2:
        assert set(np.unique(tr)).issubset({0.0, 0.5, 1.0})       # binary estimator, nSamples = 2


@pytest.mark.parametrize("stepper", [P.STEP_VERLET, P.STEP_RK4])
@pytest.mark.parametrize("rifkind", ["trilinear", "bspline"])
def test_fp32_trace_stays_within_the_stated_tolerance_of_fp64(orc, stepper, rifkind):
    """D6: `FLOAT` is double in one reference configuration and float in the other.  The oracle's fp32 trace against its fp64 trace at the
    reference's default step h = 1e-3 (up to 1000 steps): position 6e-5, momentum 2e-4, optical length 5e-5 -- the tolerance the GPU (fp32)
    is held to against the fp64 oracle in tests/test_gpu_leaf.py."""
    mk = scenes.curved_scene if rifkind == "trilinear" else scenes.bspline_scene
    p = mk(N=24, stepper=stepper, stepsize=1e-3)
    n = 1024
    p0 = scenes.rand_points(n, -0.45, 0.45); d0 = scenes.rand_dirs(n)
    dist = np.random.RandomState(2).uniform(0.2, 1.0, n).astype(np.float32)
    a = orc.er_trace(p, p0, d0, dist); b = orc.er_trace(p.copy(rif_double=1), p0, d0, dist)
    same = (a[4] == b[4]) & (a[4] == 1)
    assert (a[4] == b[4]).mean() > 0.999 and same.mean() > 0.6
    assert np.abs(a[0] - b[0])[same].max() < 6e-5 and np.abs(a[1] - b[1])[same].max() < 2e-4 and np.abs(a[3] - b[3])[same].max() < 5e-5


def _ramp_scene(**kw):
    N = 33
    ramp = np.broadcast_to((0.5 + 0.25 * np.linspace(-1, 1, N, dtype=np.float32))[None, None, :], (N, N, N)).copy()
    p = scenes.straight_scene(N=8, **kw)
    p.density = ramp
    return p


def test_simpson_quadrature_known_answers(orc):
    """method = simpson (heterogeneous.cpp:301-376): composite Simpson along the ray is exact for a density that is linear along it.
    sigma_t(x) = 4 (0.5 + 0.25 x): int_{-0.5}^{0.5} = 2, int_{-0.9}^{0.6} over a diagonal-free ray likewise closed form; the early exit
    (HETVOL_EARLY_EXIT, :336-362) returns +inf => transmittance 0 once the running sum passes -log(Epsilon) * 3 / (stepSize * scale)."""
    p = _ramp_scene(method=P.METHOD_SIMPSON)
    o = np.array([[-0.5, 0.1, -0.2], [-0.9, 0.3, 0.4], [-2.0, 0.0, 0.0], [0.0, 0.0, 5.0]], np.float32)
    d = np.array([[1, 0, 0], [1, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    maxt = np.array([1.0, 1.5, 2.95, 3.0], np.float32)             # (a segment that ends ON the far face would read 0 there: lookupFloat's x2 >= res rule)
    tr = orc.eval_transmittance(p, o, d, maxt, 3)[:, 0]
    F = lambda x: 4 * (0.5 * x + 0.125 * x * x)                     # antiderivative of sigma_t
    exact = [np.exp(-(F(0.5) - F(-0.5))), np.exp(-(F(0.6) - F(-0.9))), np.exp(-(F(0.95) - F(-1.0))), 1.0]
    np.testing.assert_allclose(tr, exact, rtol=2e-6)
    # a stepSize of its own changes nothing on a linear density; a dense medium trips the early exit
    tr2 = orc.eval_transmittance(p.copy(het_stepsize=0.11), o, d, maxt, 3)[:, 0]
    np.testing.assert_allclose(tr2, exact, rtol=2e-6)
    assert orc.eval_transmittance(p.copy(density_scale=400.0), o[:1], d[:1], maxt[:1], 3)[0, 0] == 0.0


def test_simpson_agrees_with_delta_tracking(orc):
    """the cross-check SURVEY 7.3 asks for: on a smooth non-linear density the deterministic quadrature and the mean of the ratio-tracking
    estimator agree within the estimator's Monte-Carlo error (plus the quadrature's own error, < 1e-4 at half-voxel steps)."""
    p = scenes.straight_scene(N=32)
    n = 200000
    o = np.tile(np.array([[-0.8, -0.35, 0.2]], np.float32), (n, 1)); dv = np.array([0.8, 0.5, -0.33]); dv /= np.linalg.norm(dv)
    d = np.tile(dv.astype(np.float32)[None], (n, 1))
    mc = orc.eval_transmittance(p.copy(tr_estimator=P.TR_RATIO), o, d, np.full(n, 1.4, np.float32), 7)[:, 0]
    q = orc.eval_transmittance(p.copy(method=P.METHOD_SIMPSON), o[:1], d[:1], np.full(1, 1.4, np.float32), 7)[0, 0]
    assert 0.02 < q < 0.98
    assert abs(mc.mean() - q) < 4 * mc.std() / np.sqrt(n) + 2e-4, (mc.mean(), q)


def test_simpson_free_flight_inverts_the_density_integral(orc):
    """invertDensityIntegral (heterogeneous.cpp:419-544): on constant sigma_t = 2 the sampled distance is -log(1-u)/2 from the box entry for the
    stream's first float u; pdfSuccess = sigma_t e^{-2t}, pdfFailure = transmittance = e^{-2t}; on the ramp the sampled t solves
    F(t) - F(t0) = -log(1-u) (the quadratic fit of a linear density is exact) and the success fraction is 1 - exp(-int)."""
    n = 4096
    p = scenes.straight_scene(N=8, method=P.METHOD_SIMPSON); p.density = np.full((8, 8, 8), 0.5, np.float32)
    o = np.tile(np.array([[-1.5, 0.2, 0.1]], np.float32), (n, 1)); d = np.tile(np.array([[1, 0, 0]], np.float32), (n, 1))
    rec = orc.sample_distance(p, o, d, np.full(n, 2.0, np.float32), 4)
    u = np.array([orc.rng_floats(4, i, 0, 1)[0] for i in range(n)], np.float64)
    want = -np.log1p(-u) / 2.0
    succ = rec[:, 0] == 1
    assert np.array_equal(succ, want < 1.5 - 1e-6) or abs(succ.mean() - (want < 1.5).mean()) < 2e-3
    np.testing.assert_allclose(rec[succ, 1], 0.5 + want[succ], rtol=2e-5, atol=2e-6)          # t measured from the ray origin: entry at 0.5
    np.testing.assert_allclose(rec[succ, 11], 2.0 * np.exp(-2.0 * want[succ]), rtol=1e-4)     # pdfSuccess
    np.testing.assert_allclose(rec[succ, 12], np.exp(-2.0 * want[succ]), rtol=1e-4)           # pdfFailure
    np.testing.assert_allclose(rec[succ, 8], np.exp(-2.0 * want[succ]), rtol=1e-4)            # transmittance
    np.testing.assert_allclose(rec[succ, 5:8], 0.9 * 2.0, rtol=1e-5)                          # sigmaS = albedo * sigma_t
    np.testing.assert_allclose(rec[~succ, 12], np.exp(-3.0), rtol=1e-5)                       # failure: exp(-whole integral)
    # ramp: F(t) - F(-0.9) = desired
    pr = _ramp_scene(method=P.METHOD_SIMPSON)
    o2 = np.tile(np.array([[-0.9, 0.0, 0.0]], np.float32), (n, 1))
    rec = orc.sample_distance(pr, o2, d, np.full(n, 1.5, np.float32), 4)
    succ = rec[:, 0] == 1
    x = -0.9 + rec[succ, 1].astype(np.float64)
    F = lambda x: 4 * (0.5 * x + 0.125 * x * x)
    np.testing.assert_allclose(F(x) - F(-0.9), -np.log1p(-u[succ]), rtol=1e-4, atol=1e-5)
    assert abs(succ.mean() - (1 - np.exp(-(F(0.6) - F(-0.9))))) < 3e-2


def test_simpson_render_agrees_with_woodcock_render(orc):
    """same scene rendered with method = simpson and with delta tracking: two estimators of one image (means within Monte-Carlo error)"""
    base = scenes.straight_scene(N=16, w=12, h=10).copy(rfilter=P.FILTER_BOX, rfilter_param=0.5, max_depth=8)
    a, _ = orc.render(base, 0, 256, 3)
    b, _ = orc.render(base.copy(method=P.METHOD_SIMPSON), 0, 256, 3)
    ma, mb = a[..., :3].mean(), b[..., :3].mean()
    assert abs(ma - mb) < 0.01 * ma, (ma, mb)
    with pytest.raises(RuntimeError, match="simpson"):
        orc.render(scenes.curved_scene(N=16, w=4, h=4).copy(method=P.METHOD_SIMPSON), 0, 1, 1)


def test_woodcock_collision_density(orc):
    """collisions land with density sigma_t(x) T(x); on constant rho = 0.5, scale 4: exponential, rate 2."""
    p = scenes.straight_scene(N=8)
    p.density = np.full((8, 8, 8), 0.5, np.float32)
    n = 200000
    o = np.tile(np.array([[-0.9, 0, 0]], np.float32), (n, 1)); d = np.tile(np.array([[1, 0, 0]], np.float32), (n, 1))
    rec = orc.sample_distance(p, o, d, np.full(n, 1.5, np.float32), 4)
    succ = rec[:, 0] == 1
    assert abs(succ.mean() - (1 - np.exp(-3.0))) < 4e-3
    assert np.allclose(rec[succ, 5:8], 0.9 * 2.0, rtol=1e-5)     # sigmaS = albedo * sigma_t
    assert np.allclose(rec[succ, 8], 0.5, rtol=1e-5)             # transmittance = 1/sigma_t (heterogeneous.cpp:649)


def test_refractive_sample_distance_limits(orc):
    """sigma = const through the composed estimator == heterogeneousrefractive::sampleDistance statistics;
    refRatioSq = n_end^2 / n_start^2 (heterogeneousrefractive.cpp:469,501)."""
    p = scenes.curved_scene(N=24, sigma_mode=P.SIGMA_HOMOGENEOUS, strategy=P.STRATEGY_SINGLE, channel=0,
                            medium_sampling_weight=1.0, sigma_a=[0.2] * 3, sigma_s=[1.8] * 3, stepper=P.STEP_VERLET)
    n = 20000
    o = np.tile(np.array([[0, -0.2, 0]], np.float32), (n, 1)); d = np.tile(np.array([[0, 1, 0]], np.float32), (n, 1))
    rec = orc.sample_distance(p, o, d, np.full(n, 9.0, np.float32), 2)
    succ = rec[:, 0] == 1
    # exit happens when the (straight, along y) ray has travelled 1.2: P(success) = 1 - exp(-2*1.2)
    assert abs(succ.mean() - (1 - np.exp(-2.4))) < 1e-2
    ny = 1.45 + 0.15 * rec[:, 3]
    assert np.allclose(rec[:, 13], (ny / 1.42) ** 2, rtol=2e-3)
    assert np.allclose(np.linalg.norm(rec[:, 14:17], axis=1), ny, rtol=2e-3)      # mRec.d = un-normalised momentum


# ---------------------------------------------------------------------------------------------------
# A10/A11 known answers
def test_non_absorbing_medium_furnace_straight(orc):
    """albedo 1, env 1, n = 1: every pixel receives exactly 1 (NEE + phase MIS weights sum to one)."""
    p = scenes.straight_scene(N=16, w=24, h=24, albedo=[1, 1, 1], rfilter=P.FILTER_BOX, rfilter_param=0.5, rr_depth=1000)
    film, c = orc.render(p, 0, 16, 1)
    img = film[..., :3] / film[..., 4:5]
    assert abs(img.mean() - 1.0) < 2e-2
    assert c[orc.C_PATHS] == 24 * 24 * 16


def test_emission_only_slab(orc):
    """no env, emissive non-scattering medium (albedo 0): L = eps/sigma_t * (1 - T), collision estimator."""
    p = scenes.straight_scene(N=8, w=16, h=16, albedo=[0, 0, 0], env_radiance=[0, 0, 0], emission=[1.0, 0.6, 0.3],
                              rfilter=P.FILTER_BOX, rfilter_param=0.5, fov_x_deg=10.0)
    p.density = np.full((8, 8, 8), 0.5, np.float32)
    film, _ = orc.render(p, 0, 256, 2)
    img = film[..., :3] / film[..., 4:5]
    expect = np.array([1.0, 0.6, 0.3]) * (1 - np.exp(-2.0 * 2.0))      # sigma_t = 2, chord ~2 at fov 10 deg
    assert np.abs(img.mean((0, 1)) - expect).max() < 2e-2


def test_filter_table(orc):
    v, r, s = orc.filter_table(P.FILTER_GAUSSIAN, 0.5)
    assert r == 2.0 and abs(s - 15.5) < 1e-6 and v[31] == 0
    assert abs(2 * r / 31 * v[:31].sum() - 1.0) < 1e-5                 # normalised (rfilter.cpp:50-54)
    v, r, s = orc.filter_table(P.FILTER_BOX, 0.5)
    assert abs(r - 0.50001) < 1e-7 and np.allclose(v[:31], v[0])


def test_rng_stream_properties(orc):
    a = orc.rng_floats(0, 0, 0, 100000)
    assert a.min() >= 0 and a.max() < 1 and abs(a.mean() - 0.5) < 5e-3
    assert np.all((a * 2 ** 23) == np.floor(a * 2 ** 23))               # 23-bit granularity (random.cpp:630-639)
    assert not np.array_equal(a[:8], orc.rng_floats(0, 1, 0, 8))
    assert not np.array_equal(a[:8], orc.rng_floats(0, 0, 1, 8))


# ----------------------------------------------------------------------------- A12 + point-emitter luminaire sampling
def test_connect_constant_index_is_the_chord(orc):
    """A12 known answer: in a constant-index field the eikonal ray is a straight line, so the connection p1 -> p2 is the
    chord: arc length |p2-p1|, optical length n|p2-p1|, direction (p2-p1)/|p2-p1|, exactly one solution (weight 1)."""
    N = 12
    p = scenes.curved_scene(N=N, rif="linear")
    p.rif = np.full((N, N, N), 1.3, np.float32)
    rng = np.random.RandomState(5)
    p1 = rng.uniform(-0.8, 0.8, (64, 3)).astype(np.float32); p2 = rng.uniform(-0.8, 0.8, (64, 3)).astype(np.float32)
    out = orc.connect(p, p1, p2, 9)
    chord = p2 - p1; L = np.linalg.norm(chord, axis=1)
    ok = out[:, 0] == 1
    # the initial momentum is drawn in the hemisphere about the chord (uniformSample, :1078-1084); a draw whose ray leaves the
    # shape before its closest approach needs the boundary branch (not built) and is lost to Russian roulette
    assert ok.mean() > 0.85
    out, chord, L = out[ok], chord[ok], L[ok]
    np.testing.assert_allclose(out[:, 1], 1.0, atol=0)
    np.testing.assert_allclose(out[:, 8], L, rtol=2e-3)                      # arc length, quantised by the step size
    np.testing.assert_allclose(out[:, 9], 1.3 * L, rtol=2e-3)
    d = out[:, 2:5] / np.linalg.norm(out[:, 2:5], axis=1, keepdims=True)
    np.testing.assert_allclose(d, chord / L[:, None], atol=2e-4)


def test_point_emitter_single_scatter_matches_quadrature(orc):
    """Point-emitter luminaire sampling (src/emitters/point.cpp sampleDirect + Scene::evalTransmittance) in a homogeneous,
    isotropic medium with exactly one scattering event (max_depth = 3: the null boundary crossing counts as a depth,
    volpath.cpp:259-262): the radiance along the central camera ray has the closed form
      L = int_0^2 sigma_s exp(-sigma_t t) (1/4pi) I exp(-sigma_t d(t)) / d(t)^2 dt ."""
    sig_a, sig_s = 0.3, 0.9
    pp = np.array([0.1, 0.6, -0.2]); I = np.array([1.0, 0.8, 0.5])
    p = scenes.homogeneous_scene(w=2, h=2, fov_x_deg=0.02, sigma_a=[sig_a] * 3, sigma_s=[sig_s] * 3, env_radiance=[0, 0, 0],
                                 point_position=list(pp), point_intensity=list(I), max_depth=3,
                                 rfilter=P.FILTER_BOX, rfilter_param=0.5)
    film, _ = orc.render(p, 0, 60000, 11)
    got = film[..., :3].sum((0, 1)) / film[..., 4].sum()
    st = sig_a + sig_s
    t = (np.arange(200000) + 0.5) / 200000 * 2.0
    x = np.stack([-1 + t, 0 * t, 0 * t], 1)                                    # camera at (-3,0,0) looking along +x
    d = np.linalg.norm(pp[None] - x, axis=1)
    ref = (sig_s * np.exp(-st * t) / (4 * np.pi) * np.exp(-st * d) / d ** 2).mean() * 2.0
    np.testing.assert_allclose(got, ref * I, rtol=0.03)


def test_point_emitter_curved_equals_straight_in_constant_index(orc):
    """The curved branch (connection solver + transmittance along the connecting ray) must reproduce the straight branch
    when the index field is constant 1 (same expectation; different sampler stream, so compared as film means)."""
    N = 12
    kw = dict(w=4, h=4, env_radiance=[0, 0, 0], point_position=[0.2, 0.3, -0.1], point_intensity=[1.0, 0.8, 0.5], max_depth=3,
              rfilter=P.FILTER_BOX, rfilter_param=0.5)
    ps = scenes.straight_scene(N=N, **kw)
    pc = scenes.curved_scene(N=N, **kw)
    pc.rif = np.ones((N, N, N), np.float32)
    fs, _ = orc.render(ps, 0, 3000, 2)
    fc, _ = orc.render(pc, 0, 3000, 2)
    a = fs[..., :3].sum((0, 1)) / fs[..., 4].sum(); b = fc[..., :3].sum((0, 1)) / fc[..., 4].sum()
    assert a.min() > 0
    np.testing.assert_allclose(b, a, rtol=0.08)


# ----------------------------------------------------------------------------- N1 transient film
def _transient(p, **kw):
    base = dict(decomposition=P.DECOMPOSITION_TRANSIENT, min_bound=0.0, max_bound=24.0, bin_width=0.25)
    base.update(kw)
    return p.copy(**base)


@pytest.mark.parametrize("kind", ["straight", "curved", "homogeneous"])
def test_transient_frames_sum_to_the_steady_state_film(orc, kind):
    """Binning by path length only redistributes radiance: with bounds that hold every path, the frames add up to the
    steady-state film, and alpha / weight are those of the steady state (film.cpp:71-78, bdpt_proc.cpp:449-485)."""
    mk = {"straight": lambda: scenes.straight_scene(N=16, w=8, h=8), "curved": lambda: scenes.curved_scene(N=16, w=8, h=8),
          "homogeneous": lambda: scenes.homogeneous_scene(w=8, h=8)}[kind]
    ps = mk().copy(rfilter=P.FILTER_BOX, rfilter_param=0.5, max_depth=12)
    pt = _transient(ps, max_bound=60.0, bin_width=0.5)
    fs, _ = orc.render(ps, 0, 32, 5)
    ft, _ = orc.render(pt, 0, 32, 5)
    assert ft.shape == (8, 8, 120 * 3 + 2)
    np.testing.assert_array_equal(ft[..., -2:], fs[..., 3:])
    tot = ft[..., :-2].reshape(8, 8, 120, 3).sum(2)
    np.testing.assert_allclose(tot, fs[..., :3], rtol=2e-5, atol=1e-5)


@pytest.mark.parametrize("kind", ["straight", "curved", "homogeneous"])
def test_bounce_frames_are_bounce_orders(orc, kind):
    """decomposition = bounce bins by the number of path edges (film.cpp:66-68, bdpt_proc.cpp:179-187,335-381).  Known answers: the frames add
    up to the steady-state film; raising maxDepth by one populates exactly one more frame and leaves the others bit for bit (the sampler
    streams of the shorter paths are unchanged), so frame k holds the light of one bounce order."""
    mk = {"straight": lambda: scenes.straight_scene(N=16, w=8, h=8), "curved": lambda: scenes.curved_scene(N=16, w=8, h=8),
          "homogeneous": lambda: scenes.homogeneous_scene(w=8, h=8)}[kind]
    base = mk().copy(rfilter=P.FILTER_BOX, rfilter_param=0.5, rr_depth=100)
    films, tops = [], []
    for d in (3, 4, 5, 6):                                     # maxDepth 2 stops at the (null) boundary of the medium shape: nothing scatters yet
        pb = base.copy(max_depth=d, decomposition=P.DECOMPOSITION_BOUNCE, min_bound=0.0, max_bound=12.0, bin_width=1.0)
        fb, _ = orc.render(pb, 0, 32, 5)
        fs, _ = orc.render(base.copy(max_depth=d), 0, 32, 5)
        assert fb.shape == (8, 8, 12 * 3 + 2)
        np.testing.assert_array_equal(fb[..., -2:], fs[..., 3:])
        fr = fb[..., :-2].reshape(8, 8, 12, 3)
        np.testing.assert_allclose(fr.sum(2), fs[..., :3], rtol=2e-5, atol=1e-5)
        films.append(fr); tops.append(int(np.nonzero(fr.sum((0, 1, 3)))[0].max()))
    assert tops == [tops[0] + i for i in range(4)], tops
    for lo, hi, t in zip(films, films[1:], tops):
        np.testing.assert_array_equal(hi[:, :, :t + 1], lo[:, :, :t + 1])
        assert hi[:, :, t + 1].sum() > 0 and hi[:, :, t + 2:].sum() == 0


def test_bounce_film_ignores_calibrated_transient(orc):
    """Known answer from the reference's code: the EBounce loop runs from i = 2 whatever m_calibratedTransient says (bdpt_proc.cpp:179-187),
    only the ETransient branch starts at 3 (:163-170) -- so `calibratedTransient` changes no bin of a bounce film, and shifts a transient
    film by the camera edge."""
    base = scenes.curved_scene(N=16, w=8, h=8).copy(rfilter=P.FILTER_BOX, rfilter_param=0.5, max_depth=6)
    pb = base.copy(decomposition=P.DECOMPOSITION_BOUNCE, min_bound=0.0, max_bound=12.0, bin_width=1.0)
    f0, _ = orc.render(pb, 0, 16, 5)
    f1, _ = orc.render(pb.copy(calibrated_transient=True), 0, 16, 5)
    np.testing.assert_array_equal(f0, f1)
    pt = base.copy(decomposition=P.DECOMPOSITION_TRANSIENT, min_bound=0.0, max_bound=12.0, bin_width=1.0)
    t0, _ = orc.render(pt, 0, 16, 5)
    t1, _ = orc.render(pt.copy(calibrated_transient=True), 0, 16, 5)
    assert not np.array_equal(t0, t1)


def test_transient_single_scatter_profile_matches_quadrature(orc):
    """Time-resolved known answer: homogeneous isotropic medium, point emitter, exactly one scattering event.  A path that
    scatters at depth t along the central camera ray has optical length 2 + t + d(t) (camera edge 2, n = 1), so frame k holds
    the single-scatter integrand integrated over {t : 2 + t + d(t) in bin k}."""
    sig_a, sig_s = 0.3, 0.9
    pp = np.array([0.1, 0.6, -0.2]); I = np.array([1.0, 0.8, 0.5])
    p = scenes.homogeneous_scene(w=2, h=2, fov_x_deg=0.02, sigma_a=[sig_a] * 3, sigma_s=[sig_s] * 3, env_radiance=[0, 0, 0],
                                 point_position=list(pp), point_intensity=list(I), max_depth=3, rfilter=P.FILTER_BOX, rfilter_param=0.5,
                                 decomposition=P.DECOMPOSITION_TRANSIENT, min_bound=2.0, max_bound=6.0, bin_width=0.25)
    film, _ = orc.render(p, 0, 120000, 11)
    frames = 16
    got = film[..., :-2].reshape(2, 2, frames, 3).sum((0, 1)) / film[..., -1].sum()
    st = sig_a + sig_s
    n = 400000
    t = (np.arange(n) + 0.5) / n * 2.0
    x = np.stack([-1 + t, 0 * t, 0 * t], 1)
    d = np.linalg.norm(pp[None] - x, axis=1)
    f = sig_s * np.exp(-st * t) / (4 * np.pi) * np.exp(-st * d) / d ** 2 * (2.0 / n)
    bins = np.floor((2.0 + t + d - 2.0) / 0.25).astype(int)
    ref = np.bincount(bins[(bins >= 0) & (bins < frames)], weights=f[(bins >= 0) & (bins < frames)], minlength=frames)
    assert ref[:2].sum() == 0 and got[:2].sum() == 0                        # nothing can arrive before 2 + |pp - entry| ~ 2.6
    big = ref > 0.02 * ref.max()
    np.testing.assert_allclose(got[big, 0], ref[big] * I[0], rtol=0.06)
    np.testing.assert_allclose(got.sum(0), ref.sum() * I, rtol=0.03)
    # calibrated transients leave the camera edge (length 2) out: the same profile, 8 bins earlier
    pc = p.copy(calibrated_transient=True, min_bound=0.0, max_bound=4.0)
    fc, _ = orc.render(pc, 0, 120000, 11)
    np.testing.assert_allclose(fc[..., :-2], film[..., :-2], rtol=1e-4, atol=1e-7)


def test_transient_rejects_bad_bounds(orc):
    p = scenes.homogeneous_scene(w=2, h=2, decomposition=P.DECOMPOSITION_TRANSIENT, min_bound=1.0, max_bound=1.0, bin_width=0.5)
    with pytest.raises(RuntimeError, match="frames"):
        orc.render(p, 0, 1, 0)


# ----------------------------------------------------------------------------- N1: continuous-wave modulation (PathLengthSampler)
def _mod_scene(m, **kw):
    base = dict(w=2, h=2, decomposition=P.DECOMPOSITION_TRANSIENT, max_bound=8.0, modulation=m, mod_lambda=2.0, mod_phase_deg=30.0, mod_P=8, mod_neighbors=3)
    base.update(kw)
    return scenes.homogeneous_scene(**base)


def test_correlation_functions_known_values(orc):
    """PathLengthSampler::correlationFunction (src/librender/pathlengthsampler.cpp:68-114), closed forms at lambda = 2, phase = 30 deg:
    the phase shifts the argument by phase*lambda/(2 pi) = 1/6."""
    t = np.array([0.0, 0.25, 0.5, 1.0, 1.5, 2.0, 3.3], np.float32)
    u = (t + 1.0 / 6.0).astype(np.float64)
    np.testing.assert_allclose(orc.correlation(_mod_scene(P.MODULATION_SINE), t), np.cos(u * np.pi), atol=2e-6)
    np.testing.assert_allclose(orc.correlation(_mod_scene(P.MODULATION_SQUARE), t), 2.0 * (np.abs(np.fmod(u, 2.0) - 1.0) - 0.5), atol=2e-6)
    w = np.fmod(u, 2.0)
    ham = np.where(w < 1 / 3, 3 * w, np.where(w < 1.0, 1.0, np.where(w < 4 / 3, 1 - (w - 1.0) * 3, 0.0)))
    np.testing.assert_allclose(orc.correlation(_mod_scene(P.MODULATION_HAMILTONIAN), t), ham, atol=3e-6)
    ms = np.where(w < 0.25, 1 - w * 3.5, np.where(w > 1.75, 1 - (2 - w) * 3.5, 0.125))          # P = 8
    np.testing.assert_allclose(orc.correlation(_mod_scene(P.MODULATION_MSEQ), t), ms, atol=3e-6)
    # periodic with period lambda
    for m in (P.MODULATION_SINE, P.MODULATION_SQUARE, P.MODULATION_HAMILTONIAN, P.MODULATION_MSEQ, P.MODULATION_DEPTHSELECTIVE):
        a = orc.correlation(_mod_scene(m), np.array([0.4, 1.1], np.float32)); b = orc.correlation(_mod_scene(m), np.array([2.4, 5.1], np.float32))
        np.testing.assert_allclose(a, b, atol=5e-6)


def test_modulated_film_is_the_transient_film_weighted_by_the_correlation(orc):
    """bdpt_proc.cpp:446-447: with a modulation every contribution is multiplied by correlationFunction(pathLength).  With bins much
    narrower than lambda, sum_k frame_k * corr(centre_k) of the unmodulated transient film reproduces the modulated film."""
    kw = dict(w=4, h=4, fov_x_deg=20.0, sigma_a=[0.3] * 3, sigma_s=[0.9] * 3, env_radiance=[0, 0, 0], point_position=[0.1, 0.6, -0.2],
              point_intensity=[1.0, 0.8, 0.5], max_depth=6, rfilter=P.FILTER_BOX, rfilter_param=0.5, decomposition=P.DECOMPOSITION_TRANSIENT,
              min_bound=0.0, max_bound=16.0)
    pm = scenes.homogeneous_scene(modulation=P.MODULATION_SINE, mod_lambda=3.0, mod_phase_deg=20.0, **kw)
    pt = scenes.homogeneous_scene(bin_width=0.01, **kw)
    fm, _ = orc.render(pm, 0, 400, 3)
    ft, _ = orc.render(pt, 0, 400, 3)
    assert fm.shape == (4, 4, 5)
    centres = (np.arange(1600) + 0.5) * 0.01
    corr = orc.correlation(pm, centres.astype(np.float32))
    want = (ft[..., :-2].reshape(4, 4, 1600, 3) * corr[None, None, :, None]).sum(2)
    np.testing.assert_allclose(fm[..., :3], want, rtol=2e-2, atol=3e-3 * np.abs(want).max())   # bin-centre quadrature: 0.01 / lambda
    np.testing.assert_array_equal(fm[..., 3:], ft[..., -2:])
    assert (fm[..., :3] < 0).any()                                   # a correlation is signed: so is the film


def test_modulation_needs_a_transient_film(orc):
    p = scenes.homogeneous_scene(w=2, h=2, modulation=P.MODULATION_SINE)
    with pytest.raises(RuntimeError, match="needs decomposition = transient"):
        orc.render(p, 0, 1, 0)


# ----------------------------------------------------------------------------- N2: hdielectric boundary
def test_dielectric_fresnel_reflectance_at_normal_incidence(orc):
    """maxDepth = 2 leaves exactly one surface event: the camera ray reflects off the cube face with probability F and then sees
    the unit environment (delta BSDF: weight 1); the refracted part never comes back.  Head-on: F = ((n-1)/(n+1))^2."""
    for n, F in ((1.5, 0.04), (1.33, (0.33 / 2.33) ** 2)):
        p = scenes.homogeneous_scene(w=2, h=2, fov_x_deg=0.05, rif_const=n, boundary_bsdf=P.BSDF_HDIELECTRIC, max_depth=2,
                                     rfilter=P.FILTER_BOX, rfilter_param=0.5)
        film, _ = orc.render(p, 0, 200000, 1)
        got = film[..., 0].sum() / film[..., 4].sum()
        assert abs(got - F) < 4 * np.sqrt(F * (1 - F) / 800000) + 1e-4, (n, got, F)


@pytest.mark.parametrize("kind", ["homogeneous", "grid", "empty_glass"])
def test_dielectric_furnace_constant_index(orc, kind):
    """non-absorbing medium of constant index behind a smooth dielectric boundary, unit environment: radiance 1 along every
    camera ray (the 1/eta^2 on the way in and eta^2 on the way out of hdielectric.cpp:213-216 cancel; total internal reflection
    only redirects)."""
    kw = dict(w=12, h=12, fov_x_deg=40.0, boundary_bsdf=P.BSDF_HDIELECTRIC, rr_depth=100000, rfilter=P.FILTER_BOX, rfilter_param=0.5)
    if kind == "homogeneous":
        p = scenes.homogeneous_scene(sigma_a=[0, 0, 0], sigma_s=[1.0, 1.0, 1.0], rif_const=1.5, **kw)
    elif kind == "grid":
        p = scenes.straight_scene(N=16, albedo=[1, 1, 1], rif_const=1.33, **kw)
    else:
        p = scenes.straight_scene(N=16, albedo=[1, 1, 1], rif_const=1.5, density_scale=1e-6, **kw)
    film, _ = orc.render(p, 0, 400, 3)
    img = film[..., :3] / film[..., 4:5]
    assert abs(img.mean() - 1.0) < 5e-3, img.mean()


def test_dielectric_index_one_is_the_null_boundary_in_expectation(orc):
    """eta = 1: F = 0 and the refracted direction is the incident one -- the dielectric boundary degenerates to the index-matched
    one, except that the environment is then collected on exit instead of by emitter sampling (same expectation)."""
    kw = dict(N=16, w=6, h=6, fov_x_deg=30.0, rfilter=P.FILTER_BOX, rfilter_param=0.5, albedo=[0.8, 0.7, 0.6])
    a, _ = orc.render(scenes.straight_scene(**kw), 0, 4000, 2)
    b, _ = orc.render(scenes.straight_scene(boundary_bsdf=P.BSDF_HDIELECTRIC, rif_const=1.0, **kw), 0, 4000, 2)
    ma = a[..., :3].sum((0, 1)) / a[..., 4].sum(); mb = b[..., :3].sum((0, 1)) / b[..., 4].sum()
    np.testing.assert_allclose(mb, ma, rtol=0.02)


# ----------------------------------------------------------------------------- N2: aggressivetracing
def test_aggressive_tracing_same_image_and_off_switch(orc):
    """`aggressivetracing` (heterogeneousrefractive.cpp:473-493,697-704) walks legs of min(depth below the SDF surface - maxSDFError,
    distance left) without inside tests.  It only re-partitions the steps of a segment: same image in expectation, more
    steps (every leg ends with its own remainder step); with an error bound larger than the shape no leg is ever taken and the
    render is bit-identical to plain tracing."""
    from mitsubaer_amd import synth
    box = ([-1.2] * 3, [1.2] * 3)
    sdf = -synth.sphere_sdf(48, radius=0.9, aabb_min=box[0], aabb_max=box[1])
    kw = dict(N=16, w=8, h=8, rif="radial", fov_x_deg=35.0, rfilter=P.FILTER_BOX, rfilter_param=0.5, boundary=P.BOUNDARY_SDF, sdf=sdf, sdf_aabb=box)
    plain, c0 = orc.render(scenes.curved_scene(**kw), 0, 600, 5)
    aggr, c1 = orc.render(scenes.curved_scene(aggressive_tracing=True, **kw), 0, 600, 5)
    off, c2 = orc.render(scenes.curved_scene(aggressive_tracing=True, sdf_max_error=10.0, **kw), 0, 600, 5)
    assert np.array_equal(plain, off) and np.array_equal(c0, c2)
    assert not np.array_equal(plain, aggr)
    m0 = plain[..., :3].sum() / plain[..., 4].sum(); m1 = aggr[..., :3].sum() / aggr[..., 4].sum()
    assert abs(m1 / m0 - 1.0) < 0.01, (m0, m1)
    # one more (remainder) step per leg, and the tested trace of what is left -- often of length 0 -- still takes its remainder step
    assert c0[orc.C_STEPS] < c1[orc.C_STEPS] < 2 * c0[orc.C_STEPS]
    # homogeneous sigma along curved rays: the reference's own sampleDistance
    kw2 = dict(kw, sigma_mode=P.SIGMA_HOMOGENEOUS, phase=P.PHASE_ISOTROPIC)
    a, _ = orc.render(scenes.curved_scene(**kw2), 0, 600, 6)
    b, _ = orc.render(scenes.curved_scene(aggressive_tracing=True, **kw2), 0, 600, 6)
    ma = a[..., :3].sum() / a[..., 4].sum(); mb = b[..., :3].sum() / b[..., 4].sum()
    assert abs(mb / ma - 1.0) < 0.01, (ma, mb)


# ----------------------------------------------------------------------------- N4: analytic acoustic RIF
@pytest.mark.parametrize("m", [0, 1, 2, 3])
def test_acoustic_rif_value_gradient_hessian(orc, m):
    """acousticrifvolume (src/volume/acousticrifvolume.cpp:224-342): n = n_o + n_max J_m(k_r r) cos(m phi) in the (y, z) plane with
    phi = atan2(y, z); the gradient and Hessian written there must be the derivatives of that value (central differences of the
    closed form in fp64), and the field does not depend on x."""
    from scipy import special
    n_o, n_max, k_r = float(np.float32(1.3333)), float(np.float32(0.05)), 6.0            # the scene struct carries them as float32
    p = scenes.homogeneous_scene(rif_mode=P.RIF_ACOUSTIC, ac_n_o=n_o, ac_n_max=n_max, ac_k_r=k_r, ac_mode=m, rif_double=1, stepsize=0.01)
    rng = np.random.RandomState(3 + m)
    pts = rng.uniform(-0.9, 0.9, (400, 3)).astype(np.float32)
    pts = pts[np.hypot(pts[:, 1], pts[:, 2]) > 0.05]

    def f(q):
        r = np.hypot(q[:, 1], q[:, 2]); phi = np.arctan2(q[:, 1], q[:, 2])
        return n_o + n_max * special.jv(m, k_r * r) * np.cos(m * phi)
    q = pts.astype(np.float64)
    val, grad, hess = orc.rif_eval(p, pts)
    assert np.abs(val - f(q)).max() < 1e-12
    e = 1e-5
    for a in range(3):
        d = np.zeros(3); d[a] = e
        ga = (f(q + d) - f(q - d)) / (2 * e)
        assert np.abs(grad[:, a] - ga).max() < 1e-8, a
        for b in range(3):
            d2 = np.zeros(3); d2[b] = e
            hab = (f(q + d + d2) - f(q + d - d2) - f(q - d + d2) + f(q - d - d2)) / (4 * e * e)
            assert np.abs(hess[:, a, b] - hab).max() < 2e-5, (a, b)
    assert np.all(grad[:, 0] == 0) and np.all(hess[:, 0, :] == 0) and np.all(hess[:, :, 0] == 0)
    # fp32 evaluation (what the GPU path mirrors) against fp64
    p.rif_double = 0
    v32, g32, h32 = orc.rif_eval(p, pts)
    assert np.abs(v32 - val).max() < 5e-7 and np.abs(g32 - grad).max() < 5e-5


def test_acoustic_rif_renders_like_its_sampled_grid(orc):
    """the analytic field and the same field sampled onto a 96^3 trilinear grid (synth.acoustic_rif) give the same image"""
    from mitsubaer_amd import synth
    kw = dict(N=16, w=8, h=8, fov_x_deg=35.0, rfilter=P.FILTER_BOX, rfilter_param=0.5)
    n_o, n_max, k_r, m = 1.33, 0.08, 4.0, 1
    grid = synth.acoustic_rif(96, n0=n_o, nmax=n_max, mode=m, kr=k_r, axis=0)
    # synth's axis=0 field uses (u, v) = (y, z) with phi = atan2(v, u) = atan2(z, y); the reference's phi = atan2(y, z): for m = 1 the
    # two differ by the reflection y <-> z, so compare image means of a field symmetric under it only when m = 0
    a, _ = orc.render(scenes.curved_scene(rif=synth.acoustic_rif(96, n0=n_o, nmax=n_max, mode=0, kr=k_r, axis=0), **kw), 0, 500, 4)
    pa = scenes.curved_scene(**kw); pa.rif = None; pa.rif_mode = P.RIF_ACOUSTIC; pa.ac_n_o, pa.ac_n_max, pa.ac_k_r, pa.ac_mode = n_o, n_max, k_r, 0
    b, _ = orc.render(pa, 0, 500, 4)
    ma = a[..., :3].sum() / a[..., 4].sum(); mb = b[..., :3].sum() / b[..., 4].sum()
    assert abs(mb / ma - 1.0) < 0.01, (ma, mb)


def test_maxexp_distribution_known_answers(orc):
    """MaxExpDist (src/medium/maxexp.h:28-98; strategy = maximum): the density proportional to max_i sigma_i exp(-sigma_i t).
    Known answers: cdf(sample(u)) = u, the pdf returned by sample equals pdf(t), the pdf integrates to one, and it is the upper
    envelope of the three exponentials divided by their integral; equal coefficients are the reference's internal error."""
    sig = np.array([0.55, 3.55, 7.55], np.float32)
    u = np.linspace(0.001, 0.999, 997).astype(np.float32)
    r = orc.maxexp(sig, u)
    t, pdf_s, pdf_t, cdf_t = r.T
    assert np.all(np.diff(t) > 0) and t[0] > 0
    assert np.abs(cdf_t - u).max() < 2e-5 and np.abs(pdf_s - pdf_t).max() < 1e-5 * pdf_t.max()
    tt = np.linspace(0, 40, 400001)
    env = np.max(sig[:, None].astype(np.float64) * np.exp(-sig[:, None].astype(np.float64) * tt[None, :]), axis=0)
    norm = np.trapz(env, tt)
    assert np.abs(np.interp(t, tt, env / norm) - pdf_t).max() < 2e-4
    with pytest.raises(RuntimeError, match="sigmaT must vary across channels"):
        orc.maxexp([1.0, 1.0, 2.0], u)


def test_strategy_maximum_is_unbiased_like_balance(orc):
    """the sampling strategy changes variance, not the expectation: homogeneous medium rendered with `maximum` and with `balance`"""
    from tests import scenes
    pa = scenes.homogeneous_scene(w=24, h=20, strategy=P.STRATEGY_MAXIMUM)
    pb = scenes.homogeneous_scene(w=24, h=20, strategy=P.STRATEGY_BALANCE)
    fa, _ = orc.render(pa, 0, 2048, 3, nthreads=8); fb, _ = orc.render(pb, 0, 2048, 3, nthreads=8)
    ma = fa[..., :3].sum((0, 1)) / fa[..., 4].sum(); mb = fb[..., :3].sum((0, 1)) / fb[..., 4].sum()
    # the densest channel converges slowly under either strategy (heavy-tailed weights): 0.975 vs 0.967 at 2048 spp, 0.931 vs 0.974 at 256
    assert np.all(np.abs(ma / mb - 1.0) < np.array([5e-3, 8e-3, 2.5e-2])), (ma, mb)


def _refracted_chord_check(out, p1, p2, n0, R):
    """closed form for a constant index n0 inside a sphere of radius R and a target outside: straight to the boundary point b,
    Snell's law, straight on -- p2 must lie on the refracted ray; returns (fraction connected, worst distance of p2 from that ray,
    worst error of the inside length, worst error of the optical length)"""
    ok = out[:, 0] == 1
    d = out[ok, 2:5].astype(np.float64); dh = d / np.linalg.norm(d, axis=1, keepdims=True)
    P1 = p1[ok].astype(np.float64); P2 = p2[ok].astype(np.float64)
    bq = (P1 * dh).sum(1); c = (P1 * P1).sum(1) - R * R; t = -bq + np.sqrt(bq * bq - c)
    B = P1 + t[:, None] * dh; Nn = B / np.linalg.norm(B, axis=1, keepdims=True)
    cosi = (dh * Nn).sum(1); cost = np.sqrt(1 - n0 * n0 * (1 - cosi ** 2))
    r = n0 * dh + (cost - n0 * cosi)[:, None] * Nn
    off = np.linalg.norm(np.cross(P2 - B, r), axis=1) / np.linalg.norm(r, axis=1)
    return ok.mean(), off.max(), np.abs(out[ok, 8] - t).max(), np.abs(out[ok, 9] - (n0 * t + np.linalg.norm(P2 - B, axis=1))).max(), np.abs(np.linalg.norm(d, axis=1) - n0).max()


def _outside_pairs(n=64, seed=0):
    rng = np.random.RandomState(seed)
    p1 = rng.uniform(-0.4, 0.4, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    return p1, (d * rng.uniform(1.2, 2.0, (n, 1))).astype(np.float32)


def test_connection_through_the_boundary_is_the_refracted_chord(orc):
    """A12, boundary branch (heterogeneousrefractive.cpp:873-919, 963-992, 1040-1074): a connection whose far end lies outside the
    medium shape is marched to the boundary, refracted by Snell's law (exterior index 1) and continued straight.  Known answer: with a
    constant index inside a sphere the connecting path is the refracted chord."""
    from tests import scenes
    n0, R, N = 1.4, 0.8, 24
    p = scenes.curved_scene(N=N, rif=np.full((N, N, N), n0, np.float32), boundary=P.BOUNDARY_SPHERE, sph_radius=R, stepper=P.STEP_VERLET)
    p1, p2 = _outside_pairs()
    frac, off, dl, do, dn = _refracted_chord_check(orc.connect(p, p1, p2, 1), p1, p2, n0, R)
    assert frac > 0.75 and off < 3e-4 and dl < 3e-4 and do < 5e-4 and dn < 1e-6, (frac, off, dl, do, dn)


# ----------------------------------------------------------------------------- emitter `area` on a `rectangle` (src/emitters/area.cpp, src/shapes/rectangle.cpp)
# the rectangle above the cube, facing down: local (x, y, z) -> world (1.5 x, 2.5 - z, -1.5 y); normal toWorld(0,0,1) = (0, -1, 0)
RECT_ABOVE = np.array([[1.5, 0, 0, 0], [0, 0, -1, 2.5], [0, -1.5, 0, 0]], np.float64)


def test_area_emitter_seen_directly(orc):
    """a camera ray that meets the rectangle before (or instead of) the medium shape returns AreaLight::eval: the radiance on the front side, 0 on the
    back, nothing with hideEmitters (volpath.cpp:203-206), and the rectangle hides the environment behind it"""
    # rectangle x = -2 in front of the camera at (-3, 0, 0), facing it: local (x, y, z) -> world (-2 - z, 0.5 y, 0.5 x)?  normal toWorld(0,0,1) = (-1, 0, 0)
    front = np.array([[0, 0, -1, -2.0], [0, 0.5, 0, 0], [0.5, 0, 0, 0]], np.float64)
    p = scenes.homogeneous_scene(w=16, h=16, fov_x_deg=20.0, rfilter=P.FILTER_BOX, rfilter_param=0.5, env_radiance=[0.25, 0.25, 0.25],
                                 area_to_world=front, area_radiance=[3.0, 2.0, 1.0])
    a = orc.render_paths(p, 0, 1)
    assert np.allclose(a, [3.0, 2.0, 1.0])                    # every ray of the 20-degree view ends on the 1 x 1 rectangle one unit away
    back = front.copy(); back[0, 2] = 1.0                      # flip the normal: the camera sees the back side
    assert np.all(orc.render_paths(p.copy(area_to_world=back), 0, 1) == 0)
    assert np.all(orc.render_paths(p.copy(hide_emitters=True), 0, 1) == 0)


def test_area_emitter_furnace(orc):
    """rectangle of radiance 1 (front side towards the medium) + environment of radiance 1 around a non-absorbing medium: from every point of the
    medium every direction carries radiance 1, so every path returns 1 in expectation -- luminaire sampling of the rectangle, its MIS partner
    (phase sampling that hits the rectangle), the rectangle shadowing the environment, and the environment's own two estimators must add up"""
    for mk in (lambda **kw: scenes.homogeneous_scene(w=12, h=12, sigma_s=[1.0, 2.0, 3.0], sigma_a=[0, 0, 0], **kw),
               lambda **kw: scenes.straight_scene(N=16, w=12, h=12, albedo=[1, 1, 1], phase=P.PHASE_HG, g=0.6, density_scale=3.0, **kw)):
        p = mk(fov_x_deg=30.0, rfilter=P.FILTER_BOX, rfilter_param=0.5, rr_depth=1000, area_to_world=RECT_ABOVE, area_radiance=[1.0, 1.0, 1.0])
        film, _ = orc.render(p, 0, 1200, 3)
        mean = film[..., :3].sum((0, 1)) / film[..., 4].sum()
        assert np.all(np.abs(mean - 1.0) < 1.5e-2), mean          # heavy-tailed throughput (sigma_s up to 3, no absorption): ~0.5 % noise per channel at this sample count


def test_area_emitter_single_scatter_matches_quadrature(orc):
    """homogeneous isotropic medium, single scattering (maxDepth 4, see below), black environment: the radiance along the central camera ray is
    int_0^2 sigma_s e^{-sigma_t t} / (4 pi) int_rect Le cos(theta_y) / d^2 e^{-sigma_t s(x, y)} dA dt, s = the part of the segment inside the cube;
    the estimator is luminaire sampling + phase sampling combined by the power heuristic"""
    sig_a, sig_s = 0.2, 0.8
    Le = np.array([3.0, 2.0, 1.0])
    # maxDepth 4: the luminaire sample of the first scattering event may cross the (null) boundary once (interactions = maxDepth - depth - 1 = 1);
    # a second scattering event is sampled but can reach no emitter any more (interactions = 0)
    p = scenes.homogeneous_scene(w=2, h=2, fov_x_deg=0.02, sigma_a=[sig_a] * 3, sigma_s=[sig_s] * 3, env_radiance=[0, 0, 0], max_depth=4,
                                 rfilter=P.FILTER_BOX, rfilter_param=0.5, area_to_world=RECT_ABOVE, area_radiance=list(Le))
    film, _ = orc.render(p, 0, 60000, 5)
    got = film[..., :3].sum((0, 1)) / film[..., 4].sum()
    st = sig_a + sig_s
    nt, nr = 400, 240
    t = (np.arange(nt) + 0.5) / nt * 2.0
    x = np.stack([-1 + t, 0 * t, 0 * t], 1)                                          # scatter points on the x axis
    u = ((np.arange(nr) + 0.5) / nr * 2 - 1) * 1.5
    yx, yz = np.meshgrid(u, u, indexing="ij")
    y = np.stack([yx.ravel(), np.full(yx.size, 2.5), yz.ravel()], 1)                  # rectangle points (y = 2.5, |x|, |z| <= 1.5)
    d = y[None] - x[:, None]
    dist = np.linalg.norm(d, axis=2)
    cos_y = d[..., 1] / dist                                                         # normal (0,-1,0): cos = (x - y) . n / d = (y_y - x_y) / d
    # inside length: the segment leaves the cube through the first face it reaches (the top face y = 1 or a side face)
    with np.errstate(divide="ignore", invalid="ignore"):
        tx = np.where(d[..., 0] > 0, (1 - x[:, None, 0]) / d[..., 0], np.where(d[..., 0] < 0, (-1 - x[:, None, 0]) / d[..., 0], np.inf))
        ty = (1 - x[:, None, 1]) / d[..., 1]
        tz = np.where(d[..., 2] > 0, (1 - x[:, None, 2]) / d[..., 2], np.where(d[..., 2] < 0, (-1 - x[:, None, 2]) / d[..., 2], np.inf))
    s_in = np.minimum(np.minimum(tx, ty), tz) * dist                                  # parameters are fractions of the segment
    dA = (3.0 / nr) ** 2
    inner = (cos_y / dist ** 2 * np.exp(-st * s_in)).sum(1) * dA
    ref = (sig_s * np.exp(-st * t) / (4 * np.pi) * inner).sum() * (2.0 / nt)
    np.testing.assert_allclose(got, ref * Le, rtol=0.03)


def test_area_emitter_rejects_what_is_not_built(orc):
    p = scenes.curved_scene(N=16, w=4, h=4, area_to_world=RECT_ABOVE, area_radiance=[1, 1, 1])
    with pytest.raises(RuntimeError, match="straight rays"):
        orc.render(p, 0, 1, 0)
    shear = RECT_ABOVE.copy(); shear[0, 1] = 0.7
    with pytest.raises(RuntimeError, match="shear"):
        orc.render(scenes.homogeneous_scene(w=4, h=4, area_to_world=shear, area_radiance=[1, 1, 1]), 0, 1, 0)
