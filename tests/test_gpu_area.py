"""Emitter `area` on a `rectangle` shape (src/emitters/area.cpp, src/shapes/rectangle.cpp) -- SURVEY 8b's AREA_RECT -- as a luminaire-sampling
target with the phase-function sample as its MIS partner (volpath.cpp:120-173,370-428), straight rays.  GPU vs the oracle per path (the oracle's
estimator is pinned on closed forms in tests/test_oracle_kat.py: direct view, furnace, single-scatter quadrature), plus the furnace on the GPU."""
import numpy as np
import pytest
from mitsubaer_amd import params as P, capi
from tests import scenes
from tests.test_oracle_kat import RECT_ABOVE

pytestmark = pytest.mark.gpu
SIDE = np.array([[0, 0, 1, -2.2], [0, 1.2, 0, 0.3], [0.8, 0, 0, -0.4]], np.float64)        # left of the cube, tilted sizes, facing +x; partly in the camera's view
A = dict(area_to_world=RECT_ABOVE, area_radiance=[3.0, 2.0, 1.0])
CASES = {
    "homogeneous_rect_only": lambda: scenes.homogeneous_scene(w=32, h=24, env_radiance=[0, 0, 0], **A),
    "homogeneous_rect_env_hg": lambda: scenes.homogeneous_scene(w=32, h=24, phase=P.PHASE_HG, g=0.7, strategy=P.STRATEGY_SINGLE, **A),
    "grid_rect_env_ratio": lambda: scenes.straight_scene(N=24, w=32, h=24, **A),
    "grid_rect_only_woodcock2": lambda: scenes.straight_scene(N=24, w=32, h=24, env_radiance=[0, 0, 0], tr_estimator=P.TR_WOODCOCK2, **A),
    "grid_rect_simpson": lambda: scenes.straight_scene(N=24, w=32, h=24, method=P.METHOD_SIMPSON, **A),
    "grid_rect_point_emissive_rgb_albedo": lambda: scenes.straight_scene(N=24, w=32, h=24, albedo_mode=P.ALBEDO_GRID, albedo_grid=scenes.rgb_albedo(24), emission=[0.2, 0.12, 0.06],
                                                                      point_position=[0.2, 0.3, -0.1], point_intensity=[1.0, 0.8, 0.5], **A),
    "sphere_rect_in_view": lambda: scenes.straight_scene(N=24, w=48, h=40, boundary=P.BOUNDARY_SPHERE, sph_radius=0.9, area_to_world=SIDE, area_radiance=[2.0, 2.0, 4.0]),
    "rect_in_view_hidden": lambda: scenes.straight_scene(N=24, w=48, h=40, hide_emitters=True, area_to_world=SIDE, area_radiance=[2.0, 2.0, 4.0]),
    "max_depth_4": lambda: scenes.straight_scene(N=24, w=32, h=24, max_depth=4, **A),
    "transient_rect": lambda: scenes.homogeneous_scene(w=24, h=20, env_radiance=[0, 0, 0], decomposition=P.DECOMPOSITION_TRANSIENT, min_bound=0.0, max_bound=64.0, bin_width=4.0, **A),     # 16 frames; no path is longer than 64
}


@pytest.mark.parametrize("name", sorted(n for n in CASES if not n.startswith("transient")))
def test_area_emitter_paths_match_oracle(ctx, orc, name):
    p = CASES[name]()
    sc, vols = ctx.upload_scene(p)
    for s in (0, 1):
        a = ctx.render_paths(sc, s, seed=3); b = orc.render_paths(p, s, 3)
        assert np.isfinite(a).all()
        close = np.abs(a - b).max(2) <= 1e-4 * np.maximum(1.0, np.abs(b).max(2))
        assert close.mean() > 0.99, (name, close.mean())
    with ctx.options(inline_walks=0):                           # the two-kernel form walks the same paths
        assert np.array_equal(ctx.render_paths(sc, 1, seed=3), a)
    for v in vols:
        v.destroy()


@pytest.mark.parametrize("name", ["homogeneous_rect_env_hg", "grid_rect_env_ratio", "sphere_rect_in_view", "transient_rect"])
def test_area_emitter_film_matches_oracle(ctx, orc, name):
    p = CASES[name]()
    sc, vols = ctx.upload_scene(p)
    film = ctx.render_to_host(sc, 0, 8, seed=5)
    ref, _ = orc.render(p, 0, 8, 5)
    assert film.shape == ref.shape
    assert np.allclose(film[..., -2:], ref[..., -2:], rtol=1e-5, atol=1e-5)
    rel = np.linalg.norm(film[..., :-2].astype(np.float64) - ref[..., :-2]) / np.linalg.norm(ref[..., :-2].astype(np.float64))
    assert rel < 2e-2, rel
    if name == "transient_rect":                                # frames add up to the steady-state film
        ss, v2 = ctx.upload_scene(p.copy(decomposition=P.DECOMPOSITION_NONE))
        steady = ctx.render_to_host(ss, 0, 8, seed=5)
        np.testing.assert_allclose(film[..., :-2].reshape(p.height, p.width, 16, 3).sum(2), steady[..., :3], rtol=1e-4, atol=1e-5)
    for v in vols:
        v.destroy()


def test_area_emitter_furnace_on_the_gpu(ctx):
    """rectangle of radiance 1 facing a non-absorbing medium + environment of radiance 1: every path carries radiance 1 in expectation (luminaire sampling
    of the rectangle + phase sampling that hits it + the rectangle shadowing the environment + the environment's two estimators)"""
    p = scenes.straight_scene(N=16, w=64, h=64, fov_x_deg=30.0, rfilter=P.FILTER_BOX, rfilter_param=0.5, albedo=[1, 1, 1], phase=P.PHASE_HG, g=0.6, density_scale=3.0,
                              rr_depth=1000, area_to_world=RECT_ABOVE, area_radiance=[1.0, 1.0, 1.0])
    sc, vols = ctx.upload_scene(p)
    film = ctx.render_to_host(sc, 0, 256, seed=3)
    mean = film[..., :3].sum((0, 1)) / film[..., 4].sum()
    assert np.all(np.abs(mean - 1.0) < 4e-3), mean
    for v in vols:
        v.destroy()


def test_area_emitter_errors(ctx):
    for bad, msg in ((scenes.curved_scene(N=16, **A), "straight rays"),
                     (scenes.straight_scene(N=16, boundary_bsdf=P.BSDF_HDIELECTRIC, rif_const=1.3, **A), "index-matched"),
                     (scenes.straight_scene(N=16, area_radiance=[1, 1, 1], area_to_world=np.array([[0.3, 0, 0, 0], [0, 0.3, 0, 0], [0, 0, 1, 0.0]])), "outside the medium shape"),
                     (scenes.straight_scene(N=16, area_radiance=[1, 1, 1], area_to_world=np.array([[1, 0.6, 0, 0], [0, 1, 0, 3.0], [0, 0, 1, 0.0]])), "shear")):
        sc, vols = ctx.upload_scene(bad)
        with pytest.raises(capi.MerError, match=msg):
            ctx.render_paths(sc, 0)
        for v in vols:
            v.destroy()
