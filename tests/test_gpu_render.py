"""GPU parity of the full hot path (mer_render / mer_render_paths through the C-ABI) against the CPU oracle.

Both sides consume the same counter-based sampler stream per (pixel, sample), so most paths agree to
rounding; a few diverge where a libm ulp flips an accept/reject decision.  Tolerances are stated per test:
per-path agreement fraction, and per-pixel relative L2 of the film at equal spp.
"""
import numpy as np
import pytest
from mitsubaer_amd import params as P, capi
from tests import scenes

pytestmark = pytest.mark.gpu


def _rel_l2(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b.astype(np.float64)), 1e-30))


CASES = {
    "cfg2_straight_ratio": lambda: scenes.straight_scene(N=32),
    "cfg2_straight_woodcock2": lambda: scenes.straight_scene(N=32, tr_estimator=P.TR_WOODCOCK2),
    # method = simpson of the heterogeneous medium: deterministic quadrature for free flights, NEE and look-up transmittance
    "straight_simpson": lambda: scenes.straight_scene(N=32, method=P.METHOD_SIMPSON),
    "straight_simpson_point_rgb_albedo": lambda: scenes.straight_scene(N=24, method=P.METHOD_SIMPSON, het_stepsize=0.02, albedo_mode=P.ALBEDO_GRID, albedo_grid=scenes.rgb_albedo(24),
                                                                      point_position=[0.2, 0.3, -0.1], point_intensity=[1.0, 0.8, 0.5]),
    "cfg1_homogeneous_isotropic": lambda: scenes.homogeneous_scene(),
    "cfg1_homogeneous_single": lambda: scenes.homogeneous_scene(strategy=P.STRATEGY_SINGLE, phase=P.PHASE_HG, g=0.7),
    "cfg1_homogeneous_maximum": lambda: scenes.homogeneous_scene(strategy=P.STRATEGY_MAXIMUM),
    "refractive_homogeneous_sigma_maximum": lambda: scenes.curved_scene(N=24, sigma_mode=P.SIGMA_HOMOGENEOUS, stepper=P.STEP_VERLET, strategy=P.STRATEGY_MAXIMUM),
    # `toWorld` on the volume plugins (gridvolume.cpp:110,188-195; splinevolume.cpp:320-376): rotated / shifted data boxes around a sphere
    "toworld_straight": lambda: scenes.straight_scene(N=24, boundary=P.BOUNDARY_SPHERE, sph_radius=0.7, density_to_world=P.rotation([1, 2, 3], 35.0, [0.05, -0.1, 0.08])),
    "toworld_curved_trilinear": lambda: scenes.curved_scene(N=24, rif="radial", boundary=P.BOUNDARY_SPHERE, sph_radius=0.7, rif_to_world=P.rotation([0, 0, 1], 30.0, [0.1, 0.0, -0.05]),
                                                            density_to_world=P.rotation([1, 0, 0], -20.0)),
    "toworld_curved_bspline": lambda: scenes.bspline_scene(N=24, boundary=P.BOUNDARY_SPHERE, sph_radius=0.7, rif_to_world=P.rotation([1, 1, 0], 25.0, [0.05, 0.05, 0.0])),
    "toworld_point_curved": lambda: scenes.curved_scene(N=24, w=32, h=24, rif="radial", boundary=P.BOUNDARY_SPHERE, sph_radius=0.7, rif_to_world=P.rotation([0, 1, 0], 40.0),
                                                        env_radiance=[0, 0, 0], point_position=[0.2, 0.3, -0.1], point_intensity=[1.0, 0.8, 0.5]),
    # a point emitter OUTSIDE the medium shape reached through curved rays: the connection crosses the boundary (A12 boundary branch)
    "point_curved_outside_sphere": lambda: scenes.curved_scene(N=24, w=32, h=24, rif="radial", boundary=P.BOUNDARY_SPHERE, sph_radius=0.8, env_radiance=[0, 0, 0],
                                                               point_position=[0.3, 1.6, -0.4], point_intensity=[4.0, 3.0, 2.0]),
    "point_curved_outside_dielectric": lambda: scenes.curved_scene(N=24, w=32, h=24, rif="radial", boundary=P.BOUNDARY_SPHERE, sph_radius=0.8, boundary_bsdf=P.BSDF_HDIELECTRIC,
                                                                   env_radiance=[0, 0, 0], point_position=[0.3, 1.6, -0.4], point_intensity=[4.0, 3.0, 2.0]),
    "cfg3_curved_rk4_trilinear": lambda: scenes.curved_scene(N=32),
    "cfg3_curved_verlet_trilinear": lambda: scenes.curved_scene(N=32, stepper=P.STEP_VERLET),
    "cfg4_radial_rk4": lambda: scenes.curved_scene(N=32, rif="radial"),
    "parity_verlet_bspline": lambda: scenes.bspline_scene(N=32),
    "curved_woodcock2": lambda: scenes.curved_scene(N=24, tr_estimator=P.TR_WOODCOCK2),
    "refractive_homogeneous_sigma": lambda: scenes.curved_scene(N=24, sigma_mode=P.SIGMA_HOMOGENEOUS, stepper=P.STEP_VERLET),
    "sphere_boundary": lambda: scenes.curved_scene(N=24, boundary=P.BOUNDARY_SPHERE, sph_radius=0.9),
    "max_depth_3": lambda: scenes.straight_scene(N=24, max_depth=3),
    "point_straight_inside": lambda: scenes.straight_scene(N=24, env_radiance=[0, 0, 0], point_position=[0.2, 0.3, -0.1], point_intensity=[1.0, 0.8, 0.5]),
    "point_straight_outside_plus_env": lambda: scenes.straight_scene(N=24, point_position=[0.0, 3.0, 0.5], point_intensity=[9.0, 7.0, 5.0], tr_estimator=P.TR_WOODCOCK2),
    "point_homogeneous": lambda: scenes.homogeneous_scene(env_radiance=[0, 0, 0], point_position=[0.2, 0.3, -0.1], point_intensity=[1.0, 0.8, 0.5]),
    "point_curved_trilinear": lambda: scenes.curved_scene(N=24, w=32, h=24, env_radiance=[0, 0, 0], point_position=[0.2, 0.3, -0.1], point_intensity=[1.0, 0.8, 0.5]),
    "point_curved_bspline": lambda: scenes.bspline_scene(N=24, w=32, h=24, env_radiance=[0, 0, 0], point_position=[0.2, 0.3, -0.1], point_intensity=[1.0, 0.8, 0.5]),
    "point_curved_homogeneous_sigma": lambda: scenes.curved_scene(N=24, w=32, h=24, sigma_mode=P.SIGMA_HOMOGENEOUS, stepper=P.STEP_VERLET, point_position=[-0.3, 0.1, 0.4], point_intensity=[1.0, 0.8, 0.5]),
    "cfg5_rgb_albedo_grid_emissive": lambda: scenes.curved_scene(N=24, albedo_mode=P.ALBEDO_GRID, albedo_grid=scenes.rgb_albedo(24), emission=[0.2, 0.12, 0.06]),
    "point_curved_cfg5_rgb_albedo_emissive": lambda: scenes.curved_scene(N=24, w=32, h=24, albedo_mode=P.ALBEDO_GRID, albedo_grid=scenes.rgb_albedo(24), emission=[0.2, 0.12, 0.06],
                                                                         env_radiance=[0, 0, 0], point_position=[0.2, 0.3, -0.1], point_intensity=[1.0, 0.8, 0.5]),
    "dielectric_homogeneous": lambda: scenes.homogeneous_scene(rif_const=1.5, boundary_bsdf=P.BSDF_HDIELECTRIC),
    "dielectric_straight_grid": lambda: scenes.straight_scene(N=24, rif_const=1.33, boundary_bsdf=P.BSDF_HDIELECTRIC),
    "dielectric_curved_trilinear": lambda: scenes.curved_scene(N=24, boundary_bsdf=P.BSDF_HDIELECTRIC),
    "dielectric_curved_sphere_radial": lambda: scenes.curved_scene(N=24, rif="radial", boundary=P.BOUNDARY_SPHERE, sph_radius=0.9, boundary_bsdf=P.BSDF_HDIELECTRIC),
    "dielectric_curved_bspline": lambda: scenes.bspline_scene(N=24, boundary_bsdf=P.BSDF_HDIELECTRIC),
    "point_curved_dielectric": lambda: scenes.curved_scene(N=24, w=32, h=24, boundary_bsdf=P.BSDF_HDIELECTRIC, point_position=[0.2, 0.3, -0.1], point_intensity=[1.0, 0.8, 0.5]),
    "emissive_rgb": lambda: scenes.curved_scene(N=24, env_radiance=[0, 0, 0], emission=[1.0, 0.6, 0.3], albedo=[0.95, 0.9, 0.8]),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_per_path_radiance_matches_oracle(ctx, orc, name):
    p = CASES[name]()
    sc, vols = ctx.upload_scene(p)
    agree = []
    for s in (0, 1):
        a = ctx.render_paths(sc, s, seed=3)
        b = orc.render_paths(p, s, 3)
        assert np.isfinite(a).all()
        close = np.abs(a - b).max(2) <= 1e-4 * np.maximum(1.0, np.abs(b).max(2))
        agree.append(close.mean())
    # stated tolerance: >= 99% of paths identical to 1e-4; the rest are decision flips from libm ulps.  Curved-ray connections
    # (point_curved_*) run an iterative solver per scattering event whose accept/reject decisions flip more often (its trajectory from a random
    # initial direction is sensitive to the last bit of the field evaluation): >= 92 %; observed 0.947 (B-spline) ... 0.99.
    assert min(agree) > (0.92 if "point_curved" in name else 0.99), agree
    for v in vols:
        v.destroy()


@pytest.mark.parametrize("name", ["point_curved_trilinear", "point_curved_bspline", "point_curved_outside_sphere", "point_curved_outside_dielectric"])
def test_paths_that_disagree_with_the_oracle_are_unbiased(ctx, orc, name):
    """Curved-ray connections run an iterative solver per scattering event; a last-bit difference in the field evaluation can flip
    one of its accept / reject decisions, and 1 - 8 % of the paths then differ from the oracle's.  Such a path is still a valid
    sample of the same estimator -- if so, the GPU's and the oracle's means over the DISAGREEING paths agree within their Monte
    Carlo error (a biased divergence, e.g. connections lost on one side only, would show here), and so do the means over all paths."""
    p = CASES[name]()
    sc, vols = ctx.upload_scene(p)
    a = np.stack([ctx.render_paths(sc, s, seed=11) for s in range(12)]).astype(np.float64).sum(-1)
    b = np.stack([orc.render_paths(p, s, 11) for s in range(12)]).astype(np.float64).sum(-1)
    differ = np.abs(a - b) > 1e-4 * np.maximum(1.0, np.abs(b))
    n = int(differ.sum())
    assert 0.0 < differ.mean() < 0.08, differ.mean()
    ga, gb = a[differ], b[differ]
    # the two values of a disagreeing path are samples of (nearly) the same distribution: compare means with the error of their difference
    err = np.sqrt((ga.var() + gb.var()) / n)
    assert abs(ga.mean() - gb.mean()) < 4.0 * err + 1e-12, (ga.mean(), gb.mean(), err, n)
    allerr = np.sqrt((ga.var() + gb.var()) / n) * n / a.size            # only the disagreeing paths contribute to the difference of the totals
    assert abs(a.mean() - b.mean()) < 4.0 * allerr + 1e-12, (a.mean(), b.mean(), allerr)
    for v in vols:
        v.destroy()


@pytest.mark.parametrize("layout", ["dense", "cell8", "brick27"])
@pytest.mark.parametrize("name", ["cfg4_radial_rk4", "cfg3_curved_verlet_trilinear", "point_curved_trilinear"])
def test_global_load_kernels_match_oracle(ctx, orc, name, layout):
    """configs[3]'s code paths: a field of 4 GiB or more cannot be read through a buffer descriptor and the march lists are left
    unsorted above 2^28 nodes.  Option buffer_loads = 0 selects those kernels (global loads: RIFK_DENSE / CELL8 / BRICK27 without
    _BUF) for a small field, mq_sort = 0 the unsorted lists: same per-path contract against the oracle."""
    p = CASES[name]()
    lay = {"dense": capi.LAYOUT_DENSE, "cell8": capi.LAYOUT_CELL8, "brick27": capi.LAYOUT_BRICK27}[layout]
    with ctx.options(buffer_loads=0, mq_sort=0):
        sc, vols = ctx.upload_scene(p, layout=lay)
        ref, vr = None, []
        for s in (0, 1):
            a = ctx.render_paths(sc, s, seed=3)
            b = orc.render_paths(p, s, 3)
            close = np.abs(a - b).max(2) <= 1e-4 * np.maximum(1.0, np.abs(b).max(2))
            assert close.mean() > (0.92 if name.startswith("point_curved") else 0.99), close.mean()
        a = ctx.render_paths(sc, 0, seed=3)
    sb, vb = ctx.upload_scene(p, layout=lay)                  # the same field through buffer loads and sorted lists: no bit differs
    assert np.array_equal(ctx.render_paths(sb, 0, seed=3), a)
    for v in vols + vb:
        v.destroy()


def test_full_frame_camera_does_not_overrun_the_hit_ring(ctx):
    """Every camera sample reaches the medium (hit fraction 1, against ~0.2 in the bench scene): K_gen's launches push as many work ids
    as they reserve.  The ring must hold the throttle's backlog plus a whole launch, or unread ids are overwritten: samples rendered
    twice and samples lost, with C_PATHS still adding up.  Checked on the film's weight channel (box filter: exactly spp per pixel)
    and on 1 vs 4 pipelines."""
    N = 16
    p = scenes.straight_scene(N=N, w=512, h=512, fov_x_deg=20.0, rfilter=P.FILTER_BOX, rfilter_param=0.5, density_scale=1.0, max_depth=3)
    sc, vols = ctx.upload_scene(p)
    spp = 32                                                  # 8.4 M samples: 4 pipelines x 2.1 M
    ctx.counters_reset()
    f4 = ctx.render_to_host(sc, 0, spp, seed=1)
    c4 = ctx.counters()
    assert c4[capi.C_PATHS] == 512 * 512 * spp
    def weights_ok(f):
        # every sample adds the same weight w0 = (box-filter table value)^2 to its own pixel, and to a neighbour too when it falls
        # within 1e-5 of the pixel's edge (box.cpp:39): a pixel BELOW spp * w0 has lost a sample, one well above has one twice
        w = f[..., 4].astype(np.float64); w0 = np.median(w) / spp
        assert abs(w0 - 1.0) < 1e-3
        assert (w > (spp - 0.5) * w0).all(), "samples lost: min weight %g of %g" % (w.min(), spp * w0)
        assert (w > (spp + 0.5) * w0).mean() < 5e-3 and abs(w.sum() / (512 * 512 * spp * w0) - 1.0) < 1e-4
    weights_ok(f4)
    with ctx.options(pipes=1, nslots=262144):                 # a small slot pool: the ring is sized by the launch, not by the slots
        f1 = ctx.render_to_host(sc, 0, spp, seed=1)
    weights_ok(f1)
    assert np.allclose(f1, f4, rtol=1e-4, atol=1e-4)
    for v in vols:
        v.destroy()


@pytest.mark.parametrize("name", ["cfg2_straight_ratio", "cfg3_curved_rk4_trilinear", "parity_verlet_bspline", "cfg1_homogeneous_isotropic",
                                  "dielectric_straight_grid", "dielectric_curved_trilinear"])
def test_film_matches_oracle_at_equal_spp(ctx, orc, name):
    p = CASES[name]()
    sc, vols = ctx.upload_scene(p)
    spp = 8
    film = ctx.render_to_host(sc, 0, spp, seed=5)
    ref, cref = orc.render(p, 0, spp, 5, nthreads=8)
    # weight channel: same sample positions => identical up to atomic summation order
    assert np.allclose(film[..., 4], ref[..., 4], rtol=1e-5, atol=1e-5)
    assert np.allclose(film[..., 3], ref[..., 3], rtol=1e-5, atol=1e-5)
    # stated per-pixel L2 tolerance at equal spp: 2% relative (decision flips in <1% of paths)
    assert _rel_l2(film[..., :3], ref[..., :3]) < 2e-2
    img = film[..., :3] / np.maximum(film[..., 4:5], 1e-12)
    imr = ref[..., :3] / np.maximum(ref[..., 4:5], 1e-12)
    assert abs(img.mean() - imr.mean()) < 2e-3
    for v in vols:
        v.destroy()


def test_counters_match_oracle(ctx, orc):
    p = scenes.curved_scene(N=24)
    sc, vols = ctx.upload_scene(p)
    ctx.counters_reset()
    ctx.render_to_host(sc, 0, 4, seed=9)
    c = ctx.counters()
    _, co = orc.render(p, 0, 4, 9, nthreads=8)
    assert c[capi.C_PATHS] == co[orc.C_PATHS] == p.width * p.height * 4
    for k in (capi.C_STEPS, capi.C_RIF_EVALS, capi.C_TENTATIVE, capi.C_REAL, capi.C_SEGMENTS, capi.C_NEE):
        assert abs(float(c[k]) - float(co[k])) <= 0.02 * float(co[k]) + 5, (k, c[k], co[k])


def test_sharding_is_exact_partition(ctx):
    """sample-interleaved and tile shards add up to the unsharded film (multi-GPU contract)."""
    p = scenes.curved_scene(N=24, w=70, h=45)        # partial edge tiles
    sc, vols = ctx.upload_scene(p)
    full = ctx.render_to_host(sc, 0, 6, seed=2)
    parts = sum(ctx.render_to_host(sc, r, 3, seed=2, spp_stride=2) for r in range(2))
    assert np.allclose(full, parts, rtol=1e-4, atol=1e-5)
    tiles = sum(ctx.render_to_host(sc, 0, 6, seed=2, tile_rank=r, tile_count=3) for r in range(3))
    assert np.allclose(full, tiles, rtol=1e-4, atol=1e-5)


def test_tile_shards_are_dealt_on_diagonals(ctx):
    """SHARD_TILES: the tiles rank r renders are exactly those mitsubaer_amd.dist.tile_owner deals to it (box filter: a sample stays in its
    pixel, so the weight channel shows who rendered what); with 8 ranks and 16 tile columns no rank owns a whole column (option
    tile_deal = 0, the plain row-major deal, does), and every rank has tiles in every tile row and column."""
    from mitsubaer_amd import dist as mdist
    p = scenes.straight_scene(N=16, w=512, h=480, rfilter=P.FILTER_BOX, rfilter_param=0.5, max_depth=2)
    sc, vols = ctx.upload_scene(p)
    whole = ctx.render_to_host(sc, 0, 1, seed=4)[..., 4]                    # one box-filter weight (0.99996: the 32-entry table) per pixel
    for world in (2, 3, 8):
        owner = mdist.tile_owner(p.width, p.height, world)
        total = np.zeros((p.height, p.width), np.float32)
        for r in range(world):
            w = ctx.render_to_host(sc, 0, 1, seed=4, tile_rank=r, tile_count=world)[..., 4]
            tiles = w.reshape(p.height // 32, 32, p.width // 32, 32).sum((1, 3)) > 0
            assert np.array_equal(tiles, owner == r), (world, r)
            total += w
        assert np.allclose(total, whole, atol=1e-6)
    owner = mdist.tile_owner(p.width, p.height, 8)
    for r in range(8):
        assert (owner == r).any(0).all() and (owner == r).any(1).all()
    with ctx.options(tile_deal=0):
        w = ctx.render_to_host(sc, 0, 1, seed=4, tile_rank=3, tile_count=8)[..., 4]
        cols = (w.reshape(p.height // 32, 32, p.width // 32, 32).sum((1, 3)) > 0).all(0)
        assert cols.sum() == 2                                   # the plain deal: two whole tile columns
    for v in vols:
        v.destroy()


@pytest.mark.parametrize("name", ["cfg2_straight_ratio", "cfg2_straight_woodcock2", "point_straight_inside", "dielectric_straight_grid", "max_depth_3", "toworld_straight"])
def test_inline_walks_change_no_path(ctx, name):
    """straight rays in a gridded sigma_t: K_event runs the walks itself (option inline_walks = 1, the default: persistent lanes, no hand-over to
    K_march) or parks the lane for K_march (0) -- the same sampler draws in the same order, so no bit of any path differs; films agree to summation
    order and the counters are equal"""
    p = CASES[name]()
    sc, vols = ctx.upload_scene(p)
    a = ctx.render_paths(sc, 1, seed=6)
    ctx.counters_reset(); fa = ctx.render_to_host(sc, 0, 8, seed=6); ca = ctx.counters()
    with ctx.options(inline_walks=0):
        assert np.array_equal(ctx.render_paths(sc, 1, seed=6), a)
        ctx.counters_reset(); fb = ctx.render_to_host(sc, 0, 8, seed=6); cb = ctx.counters()
    assert np.allclose(fa, fb, rtol=1e-4, atol=1e-5)
    for k in (capi.C_PATHS, capi.C_TENTATIVE, capi.C_REAL, capi.C_SEGMENTS, capi.C_NEE):
        assert ca[k] == cb[k], (k, ca[k], cb[k])
    for v in vols:
        v.destroy()


@pytest.mark.parametrize("name", ["cfg3_curved_rk4_trilinear", "cfg4_radial_rk4", "curved_woodcock2", "refractive_homogeneous_sigma", "parity_verlet_bspline", "sphere_boundary"])
def test_spawned_side_walks_change_no_film(ctx, orc, name):
    """curved rays, steady-state film: the transmittance walks of luminaire samples and emitter look-ups are handed to side-walk slots while the path
    goes on (option spawn_walks = 1, the default) or run in the path's own lane (0).  Both draw from the same forked sampler streams, so the film is
    the same up to float summation order -- also when nearly every walk finds its side-walk slot busy (a tiny slot pool) -- and equals the oracle's."""
    p = CASES[name]()
    sc, vols = ctx.upload_scene(p)
    spp = 8
    ctx.counters_reset(); fa = ctx.render_to_host(sc, 0, spp, seed=5); ca = ctx.counters()
    assert ca[capi.C_SIDE_SPAWNED] > 0
    with ctx.options(spawn_walks=0):
        ctx.counters_reset(); fb = ctx.render_to_host(sc, 0, spp, seed=5); cb = ctx.counters()
    assert cb[capi.C_SIDE_SPAWNED] == 0 and cb[capi.C_SIDE_INLINE] == 0
    assert np.allclose(fa, fb, rtol=2e-4, atol=2e-5)
    assert ca[capi.C_PATHS] == cb[capi.C_PATHS] and ca[capi.C_REAL] == cb[capi.C_REAL] and ca[capi.C_TENTATIVE] == cb[capi.C_TENTATIVE]
    ref, _ = orc.render(p, 0, spp, 5, nthreads=8)
    assert _rel_l2(fa[..., :3], ref[..., :3]) < 2e-2
    for v in vols:
        v.destroy()


@pytest.mark.parametrize("name", ["cfg3_curved_rk4_trilinear", "cfg4_radial_rk4", "curved_woodcock2", "sphere_boundary"])
@pytest.mark.parametrize("bits,major", [(1, 0), (2, 0), (3, 1)])
def test_spatially_sorted_march_list_changes_no_path(ctx, name, bits, major):
    """option march_sort: between K_event and K_march the pass's march list is counting-sorted by (cell of the lane's position, exit-time class)
    and K_march sweeps it in XCD-contiguous chunks.  Only the ORDER in which lanes are processed changes: per-path radiance is bit-identical, the
    work counters are equal, and the film agrees up to float summation order (several pipelines, side walks spawned)."""
    p = CASES[name]()
    sc, vols = ctx.upload_scene(p)
    a = ctx.render_paths(sc, 0, seed=3)
    with ctx.options(march_sort=bits, march_sort_major=major):
        b = ctx.render_paths(sc, 0, seed=3)
    assert np.array_equal(a, b)
    spp = 8
    ctx.counters_reset(); fa = ctx.render_to_host(sc, 0, spp, seed=5); ca = ctx.counters()
    with ctx.options(march_sort=bits, march_sort_major=major):
        ctx.counters_reset(); fb = ctx.render_to_host(sc, 0, spp, seed=5); cb = ctx.counters()
        with ctx.options(pipes=1):
            fc = ctx.render_to_host(sc, 0, spp, seed=5)
    assert np.allclose(fa, fb, rtol=2e-4, atol=2e-5) and np.allclose(fa, fc, rtol=2e-4, atol=2e-5)
    for k in (capi.C_PATHS, capi.C_REAL, capi.C_TENTATIVE, capi.C_STEPS):
        assert ca[k] == cb[k]
    for v in vols:
        v.destroy()


@pytest.mark.parametrize("name", ["cfg3_curved_rk4_trilinear", "curved_woodcock2", "cfg2_straight_ratio", "point_curved_trilinear"])
def test_fitted_launch_grids_change_no_path(ctx, name):
    """option grid_fit (default 1): K_event / K_connect / K_march are launched with as many blocks as their lists can still hold (live path slots x
    records per path + the side walks in flight at the last read-back) instead of one block per 256 records.  The bound is never below a list's
    length, so nothing changes but the number of blocks that find no work: per-path radiance bit-identical, equal work counters, the same film --
    also under the two-walk Woodcock estimator (side walks return to K_event) and with the connection kernel in the pass."""
    p = CASES[name]()
    sc, vols = ctx.upload_scene(p)
    a = ctx.render_paths(sc, 0, seed=3)
    with ctx.options(grid_fit=0):
        b = ctx.render_paths(sc, 0, seed=3)
    assert np.array_equal(a, b)
    spp = 12
    ctx.counters_reset(); fa = ctx.render_to_host(sc, 0, spp, seed=5); ca = ctx.counters()
    with ctx.options(grid_fit=0):
        ctx.counters_reset(); fb = ctx.render_to_host(sc, 0, spp, seed=5); cb = ctx.counters()
    assert np.allclose(fa, fb, rtol=2e-4, atol=2e-5)
    for k in (capi.C_PATHS, capi.C_REAL, capi.C_TENTATIVE, capi.C_STEPS, capi.C_CONNECT_UNITS):
        assert ca[k] == cb[k]
    assert fa[..., 4].min() >= 0 and abs(fa[..., 4].sum() - fb[..., 4].sum()) < 1e-3 * fb[..., 4].sum()     # every sample landed
    for v in vols:
        v.destroy()


def test_options_are_range_checked(ctx):
    """mer_context_set_option refuses unknown names and values outside the documented range, and leaves the option unchanged"""
    bad = dict(pipes=0, ksteps=0, nslots=100, mq_sort=2, adaptive_k=3, grid_fit=2, check_every=0, march_sort=5, march_sort_major=2, spawn_walks=2, connect_launches=0, prefilter=9)
    for name, v in bad.items():
        before = ctx.get_option(name)
        with pytest.raises(capi.MerError):
            ctx.set_option(name, v)
        assert ctx.get_option(name) == before
    with pytest.raises(capi.MerError):
        ctx.set_option("no_such_option", 1)
    assert ctx.get_option("check_every") == 4 and ctx.get_option("grid_fit") == 1 and ctx.get_option("march_sort") == 0 and ctx.get_option("spawn_walks") == 1


def test_determinism(ctx):
    p = scenes.straight_scene(N=24)
    sc, vols = ctx.upload_scene(p)
    a = ctx.render_paths(sc, 0, seed=1)
    b = ctx.render_paths(sc, 0, seed=1)
    assert np.array_equal(a, b)
    c = ctx.render_paths(sc, 0, seed=2)
    assert not np.array_equal(a, c)


def test_cell8_layout_is_bit_identical(ctx):
    p = scenes.curved_scene(N=24)
    sc, vols = ctx.upload_scene(p, layout=capi.LAYOUT_DENSE)
    a = ctx.render_paths(sc, 0, seed=1)
    sc2, vols2 = ctx.upload_scene(p, layout=capi.LAYOUT_CELL8)
    b = ctx.render_paths(sc2, 0, seed=1)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("shape", [(24, 24, 24), (25, 24, 23), (9, 12, 17), (3, 2, 5)])
@pytest.mark.parametrize("stepper", [P.STEP_RK4, P.STEP_VERLET])
@pytest.mark.parametrize("layout", [capi.LAYOUT_BRICK27, capi.LAYOUT_BRICK125])
def test_brick_layouts_are_bit_identical(ctx, shape, stepper, layout):
    """BRICK27 (3x3x3 corners of every 2x2x2-cell brick per 128-byte record) is a storage choice: the corners a cell reads are the same
    floats, so the render is bit-identical to the dense layout -- even and odd cell counts (padded bricks), ragged and tiny grids."""
    rng = np.random.RandomState(5)
    rif = (1.3 + 0.3 * rng.rand(*shape)).astype(np.float32)
    p = scenes.curved_scene(N=16, rif=rif, stepper=stepper, stepsize=0.03)
    sc, vols = ctx.upload_scene(p, layout=capi.LAYOUT_DENSE)
    sc2, vols2 = ctx.upload_scene(p, layout=layout)
    for s in (0, 1):
        assert np.array_equal(ctx.render_paths(sc, s, seed=1), ctx.render_paths(sc2, s, seed=1))
    pts = rng.uniform(-1.1, 1.1, (4096, 3)).astype(np.float32)
    v0, g0 = ctx.rif_value_grad(vols[-1], P.RIF_TRILINEAR, pts)
    v1, g1 = ctx.rif_value_grad(vols2[-1], P.RIF_TRILINEAR, pts)
    assert np.array_equal(v0, v1) and np.array_equal(g0, g1)
    for v in vols + vols2:
        v.destroy()


def test_auto_layout_and_list_sorting_change_nothing_per_path(ctx, monkeypatch):
    """MER_LAYOUT_AUTO picks a record layout by grid size, and the work lists are sorted by event class / estimated exit time:
    both are scheduling choices, the per-path radiance stays bit-identical to the dense layout with unsorted march lists."""
    p = scenes.curved_scene(N=24, w=48, h=40)
    sc, vols = ctx.upload_scene(p, layout=capi.LAYOUT_DENSE)
    sc2, vols2 = ctx.upload_scene(p, layout=capi.LAYOUT_AUTO)
    with ctx.options(mq_sort=0):
        a = [ctx.render_paths(sc, s, seed=2) for s in (0, 1)]
    with ctx.options(mq_sort=1):
        b = [ctx.render_paths(sc2, s, seed=2) for s in (0, 1)]
    c = [ctx.render_paths(sc2, s, seed=2) for s in (0, 1)]
    for x, y, z in zip(a, b, c):
        assert np.array_equal(x, y) and np.array_equal(x, z)
    for v in vols + vols2:
        v.destroy()


@pytest.mark.parametrize("stepper", [P.STEP_RK4, P.STEP_VERLET])
@pytest.mark.parametrize("shape", [(24, 24, 24), (9, 12, 17)])
def test_lds_staged_bricks_are_bit_identical(ctx, stepper, shape):
    """lds_bricks=1: K_march DMAs every lane's current BRICK27 record into LDS and reads the cells of that brick from there.  Storage
    again: the film and the per-path radiance are bit-identical to the register cell cache."""
    rng = np.random.RandomState(11)
    rif = (1.3 + 0.3 * rng.rand(*shape)).astype(np.float32)
    p = scenes.curved_scene(N=16, rif=rif, stepper=stepper, stepsize=0.03, w=48, h=40)
    sc, vols = ctx.upload_scene(p, layout=capi.LAYOUT_BRICK27)
    with ctx.options(lds_bricks=0):
        a = [ctx.render_paths(sc, s, seed=3) for s in (0, 1)]; fa = ctx.render_to_host(sc, 0, 4, seed=3)
    with ctx.options(lds_bricks=1):
        b = [ctx.render_paths(sc, s, seed=3) for s in (0, 1)]; fb = ctx.render_to_host(sc, 0, 4, seed=3)
    assert all(np.array_equal(x, y) for x, y in zip(a, b)) and a[0].max() > 0
    assert np.allclose(fa, fb, rtol=1e-5, atol=1e-6)           # film: atomic accumulation order
    for v in vols:
        v.destroy()


def test_auto_layout_keeps_a_transformed_rif_dense(ctx):
    """The CELL8 / BRICK27 records carry no toWorld transform: MER_LAYOUT_AUTO must leave a rotated RIF volume in the dense layout (and render
    it), while asking for a record layout explicitly is an error naming the reason."""
    p = scenes.curved_scene(N=24, rif="radial", boundary=P.BOUNDARY_SPHERE, sph_radius=0.7, rif_to_world=P.rotation([0, 0, 1], 30.0, [0.1, 0.0, -0.05]))
    sc, vols = ctx.upload_scene(p, layout=capi.LAYOUT_DENSE)
    sc2, vols2 = ctx.upload_scene(p, layout=capi.LAYOUT_AUTO)
    assert np.array_equal(ctx.render_paths(sc, 0, seed=4), ctx.render_paths(sc2, 0, seed=4))
    sc3, vols3 = ctx.upload_scene(p, layout=capi.LAYOUT_BRICK27)
    with pytest.raises(RuntimeError, match="dense layout"):
        ctx.render_paths(sc3, 0, seed=4)
    for v in vols + vols2 + vols3:
        v.destroy()


@pytest.mark.parametrize("name", ["curved", "straight", "point_curved"])
def test_concurrent_pipelines_render_the_same_film(ctx, monkeypatch, name):
    """mer_render cuts a shard into `pipes` (option) independent pipelines (own slots, lists, stream; shared film): same samples, same paths --
    the film differs by float summation order only, the counters not at all; a pipeline without samples is skipped"""
    p = {"curved": lambda: scenes.curved_scene(N=24, w=70, h=45), "straight": lambda: scenes.straight_scene(N=24, w=70, h=45),
         "point_curved": lambda: scenes.curved_scene(N=16, w=24, h=20, rif="radial", env_radiance=[0, 0, 0], point_position=[0.2, 0.3, -0.1],
                                                     point_intensity=[1.0, 0.8, 0.5])}[name]()
    sc, vols = ctx.upload_scene(p)
    films, counts = [], []
    for n in (1, 2, 3, 4):
        with ctx.options(pipes=n):
            ctx.counters_reset()
            films.append(ctx.render_to_host(sc, 1, 7, seed=3, spp_stride=2))
            counts.append(np.array(ctx.counters()[:7]))
    for f, c in zip(films[1:], counts[1:]):
        assert np.allclose(f, films[0], rtol=1e-4, atol=1e-5)
        assert np.array_equal(c, counts[0])
    one = ctx.render_to_host(sc, 5, 1, seed=3)                      # a single sample per pixel: one pipeline has all the work
    with ctx.options(pipes=1):
        assert np.allclose(one, ctx.render_to_host(sc, 5, 1, seed=3), rtol=1e-4, atol=1e-5)
    for v in vols:
        v.destroy()


def test_brick27_is_for_the_rif_only(ctx):
    p = scenes.straight_scene(N=16)
    with pytest.raises(RuntimeError, match="refractive-index field only"):
        dens = ctx.upload_volume(p.density, p.density_aabb[0], p.density_aabb[1], capi.LAYOUT_BRICK27)
        sc = ctx.scene_desc(p, dens)
        ctx.render_paths(sc, 0)


def test_constant_rif_reproduces_straight_rays(ctx):
    """a constant RIF through the curved code path = the straight-ray estimator (same expectation)."""
    N = 24
    ps = scenes.straight_scene(N=N, w=32, h=32, rfilter=P.FILTER_BOX, rfilter_param=0.5)
    pc = scenes.curved_scene(N=N, w=32, h=32, rfilter=P.FILTER_BOX, rfilter_param=0.5)
    pc.rif = np.ones((N, N, N), np.float32)
    s1, _ = ctx.upload_scene(ps)
    s2, _ = ctx.upload_scene(pc)
    a = ctx.render_to_host(s1, 0, 64, seed=1)
    b = ctx.render_to_host(s2, 0, 64, seed=1)
    ma = a[..., :3].sum() / a[..., 4].sum(); mb = b[..., :3].sum() / b[..., 4].sum()
    assert abs(ma - mb) < 5e-3


def test_errors_are_loud(ctx):
    p = scenes.straight_scene(N=16)
    sc, vols = ctx.upload_scene(p)
    sc.density = 999
    with pytest.raises(capi.MerError, match="No density specified"):
        ctx.render_to_host(sc, 0, 1)
    sc, vols = ctx.upload_scene(p)
    sc.rr_depth = 0
    with pytest.raises(capi.MerError, match="rrDepth"):
        ctx.render_to_host(sc, 0, 1)
    sc.rr_depth = 5; sc.g = 1.5
    with pytest.raises(capi.MerError, match="asymmetry"):
        ctx.render_to_host(sc, 0, 1)
    with pytest.raises(capi.MerError):
        ctx.upload_volume(np.zeros((4, 4, 4, 2), np.float32), [-1] * 3, [1] * 3)     # 2 channels
