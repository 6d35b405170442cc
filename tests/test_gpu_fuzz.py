"""Seeded random scene configurations (every switch of the scene description drawn independently) through mer_render_paths against the oracle: the
hand-written cases of tests/test_gpu_render.py cover each feature, this covers their COMBINATIONS (boundary x medium kind x phase x strategy x
estimator x stepper x RIF kind x emitters x depth rules x hideEmitters).  Same per-path thresholds as everywhere: >= 99 % of the paths within 1e-4
(>= 92 % when a curved-ray connection solver runs)."""
import numpy as np
import pytest
from mitsubaer_amd import params as P, capi, synth
from tests import scenes
from tests.test_oracle_kat import RECT_ABOVE

pytestmark = pytest.mark.gpu


def _random_scene(seed):
    r = np.random.RandomState(1000 + seed)
    pick = lambda *a: a[r.randint(len(a))]
    N = pick(12, 16, 24)
    kw = dict(width=pick(17, 24, 33), height=pick(13, 20, 24), rfilter=pick(P.FILTER_BOX, P.FILTER_GAUSSIAN), rfilter_param=0.5,
              max_depth=pick(-1, -1, 3, 4, 6), rr_depth=pick(5, 2, 50), hide_emitters=bool(pick(0, 0, 1)),
              phase=pick(P.PHASE_ISOTROPIC, P.PHASE_HG), g=float(pick(0.8, -0.4, 0.3)),
              env_radiance=pick([1.0, 1.0, 1.0], [0.0, 0.0, 0.0], [0.5, 0.7, 0.9]), fov_x_deg=float(pick(95.84, 40.0, 60.0)))
    sphere = pick(0, 0, 1)
    if sphere:
        kw.update(boundary=P.BOUNDARY_SPHERE, sph_radius=float(pick(0.8, 0.9)))
    curved = pick(0, 1, 1)
    grid_sigma = pick(0, 1, 1)
    if grid_sigma:
        kw.update(sigma_mode=P.SIGMA_GRID, density=synth.density_field(N), density_scale=float(pick(2.0, 4.0)), tr_estimator=pick(P.TR_RATIO, P.TR_WOODCOCK2),
                  albedo=pick([0.9, 0.9, 0.9], [0.95, 0.8, 0.6]))
        if pick(0, 0, 1):
            kw.update(albedo_mode=P.ALBEDO_GRID, albedo_grid=scenes.rgb_albedo(N, seed=seed))
        if pick(0, 0, 1):
            kw.update(emission=[0.2, 0.12, 0.06])
    else:
        kw.update(sigma_mode=P.SIGMA_HOMOGENEOUS, sigma_s=pick([0.5, 3.5, 7.5], [1.0, 1.0, 1.0]), sigma_a=pick([0.05] * 3, [0.0, 0.1, 0.3]),
                  strategy=pick(P.STRATEGY_BALANCE, P.STRATEGY_SINGLE, P.STRATEGY_MAXIMUM))
        if kw["strategy"] == P.STRATEGY_MAXIMUM and kw["sigma_s"] == [1.0, 1.0, 1.0] and kw["sigma_a"] == [0.05] * 3:
            kw["strategy"] = P.STRATEGY_BALANCE                     # MaxExpDist needs sigma_t to vary across the channels
    point = pick(0, 0, 1)
    if curved:
        kind = pick("trilinear", "trilinear", "bspline")
        if kind == "trilinear":
            kw.update(rif_mode=P.RIF_TRILINEAR, rif=pick(synth.linear_rif(N), synth.radial_rif(N)), stepper=pick(P.STEP_RK4, P.STEP_VERLET))
        else:
            kw.update(rif_mode=P.RIF_BSPLINE3, rif=synth.radial_rif(N, (-1.3,) * 3, (1.3,) * 3), rif_aabb=([-1.3] * 3, [1.3] * 3), stepper=pick(P.STEP_VERLET, P.STEP_RK4))
        kw.update(stepsize=0.5 * 2.0 / (N - 1))
        if pick(0, 0, 1):
            kw.update(boundary_bsdf=P.BSDF_HDIELECTRIC)
    else:
        if pick(0, 0, 1):
            kw.update(rif_const=1.33, boundary_bsdf=P.BSDF_HDIELECTRIC)
        elif pick(0, 1):
            kw.update(area_to_world=RECT_ABOVE, area_radiance=pick([3.0, 2.0, 1.0], [1.0, 1.0, 1.0]))
        if grid_sigma and pick(0, 0, 0, 1):
            kw.update(method=P.METHOD_SIMPSON)
    if point:
        inside = pick(1, 1, 0)
        kw.update(point_position=[0.2, 0.3, -0.1] if inside else [0.3, 1.6, -0.4], point_intensity=[1.0, 0.8, 0.5])
    layout = pick(capi.LAYOUT_DENSE, capi.LAYOUT_CELL8, capi.LAYOUT_BRICK27, capi.LAYOUT_AUTO)
    return P.SceneParams(**kw), layout, bool(curved and point)


@pytest.mark.parametrize("seed", range(40))
def test_random_scene_configuration_matches_oracle(ctx, orc, seed):
    p, layout, connections = _random_scene(seed)
    sc, vols = ctx.upload_scene(p, layout=layout)
    a = ctx.render_paths(sc, seed % 3, seed=seed)
    b = orc.render_paths(p, seed % 3, seed)
    assert np.isfinite(a).all()
    # method = simpson: the free flight is the root of a quadrature found by Newton / bisection to a tolerance; the GPU's and the oracle's roots differ in
    # the last bits (expf / logf ulps), a CONTINUOUS perturbation that a deep path (scale 4, no depth limit) carries to a few 1e-4 -- not a decision flip
    tol = 2e-3 if p.method == P.METHOD_SIMPSON else 1e-4
    close = np.abs(a - b).max(2) <= tol * np.maximum(1.0, np.abs(b).max(2))
    assert close.mean() > (0.90 if connections else 0.99), (seed, close.mean(), {k: v for k, v in p.__dict__.items() if not hasattr(v, "shape")})
    for v in vols:
        v.destroy()


@pytest.mark.parametrize("seed", range(24))
def test_random_scene_film_does_not_depend_on_the_scheduling(ctx, seed):
    """The same random configurations as FILMS, under the scheduling the render picks by itself (four pipelines, spawned side walks where they apply,
    fitted launch grids, batches of 4 passes) and under its opposite (one pipeline, every walk in the path's lane, full grids, batches of 8, a slot
    pool small enough to be refilled dozens of times; the march list spatially sorted when the rays are curved): the film is the same up to float summation
    order, every sample lands, and the work counters agree -- the scheduler moves work around, it never changes or loses any."""
    p, layout, connections = _random_scene(seed)
    sc, vols = ctx.upload_scene(p, layout=layout)
    spp = 6
    ctx.counters_reset(); fa = ctx.render_to_host(sc, 0, spp, seed=seed); ca = ctx.counters()
    with ctx.options(pipes=1, spawn_walks=0, grid_fit=0, check_every=8, nslots=2048, march_sort=2 if p.rif_mode != P.RIF_CONST else 0):
        ctx.counters_reset(); fb = ctx.render_to_host(sc, 0, spp, seed=seed); cb = ctx.counters()
    assert np.isfinite(fa).all() and np.isfinite(fb).all()
    scale = max(float(np.abs(fb[..., :3]).max()), 1e-6)
    assert np.abs(fa[..., :3] - fb[..., :3]).max() <= 5e-4 * scale, (seed, float(np.abs(fa[..., :3] - fb[..., :3]).max()), scale)
    assert np.allclose(fa[..., 3:], fb[..., 3:], rtol=1e-4, atol=1e-5)                      # alpha and weight
    for k in (capi.C_PATHS, capi.C_REAL):
        assert ca[k] == cb[k], (seed, k, ca[k], cb[k])
    # a side walk whose prefactor is zero (a look-up that the depth limit blocks) is not spawned at all, where the in-lane form walks and then discards
    # the result: the spawning render does at most the other one's marching work (equal when nothing is blocked)
    for k in (capi.C_TENTATIVE, capi.C_STEPS):
        assert ca[k] <= cb[k], (seed, k, ca[k], cb[k])
        if p.max_depth < 0:
            assert ca[k] == cb[k], (seed, k, ca[k], cb[k])
    for v in vols:
        v.destroy()
