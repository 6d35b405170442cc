"""Edge cases of the hot path through the C-ABI: empty and ragged inputs, minimum sizes, depth limits, shards without work."""
import numpy as np
import pytest
from mitsubaer_amd import params as P, synth, capi
from tests import scenes

pytestmark = pytest.mark.gpu


def _agree(a, b, tol=1e-4):
    return (np.abs(a - b).max(-1) <= tol * np.maximum(1.0, np.abs(b).max(-1))).mean()


def test_empty_leaf_inputs(ctx):
    p = scenes.curved_scene(N=8)
    sc, vols = ctx.upload_scene(p)
    e3 = np.zeros((0, 3), np.float32); e1 = np.zeros((0,), np.float32)
    v, idx = ctx.lookup_trilinear(vols[0], e3)
    assert v.shape == (0,) and idx.shape == (0, 4)
    assert ctx.er_trace(sc, e3, e3, e1)[0].shape == (0, 3)
    assert ctx.sample_distance(sc, e3, e3, e1, 1).shape == (0, 20)
    assert ctx.connect(sc, e3, e3, 1).shape[0] == 0
    for vv in vols:
        vv.destroy()


def test_zero_samples_and_empty_shards_leave_the_film_untouched(ctx):
    p = scenes.curved_scene(N=8, w=40, h=24)                       # 2 x 1 tiles of 32 x 32
    sc, vols = ctx.upload_scene(p)
    assert not ctx.render_to_host(sc, 0, 0, seed=1).any()          # spp_count = 0
    full = ctx.render_to_host(sc, 0, 2, seed=1)
    parts = [ctx.render_to_host(sc, 0, 2, seed=1, tile_rank=r, tile_count=5) for r in range(5)]   # ranks 2..4 own no tile
    assert not parts[2].any() and not parts[3].any() and not parts[4].any()
    np.testing.assert_allclose(sum(parts), full, rtol=1e-6, atol=1e-6)
    for vv in vols:
        vv.destroy()


@pytest.mark.parametrize("w,h", [(1, 1), (3, 5), (33, 31)])
def test_odd_film_sizes_match_the_oracle(ctx, orc, w, h):
    p = scenes.curved_scene(N=12, w=w, h=h, fov_x_deg=30.0)
    sc, vols = ctx.upload_scene(p)
    a = ctx.render_to_host(sc, 0, 8, seed=2); b, _ = orc.render(p, 0, 8, 2)
    assert a.shape == b.shape == (h, w, 5)
    np.testing.assert_allclose(a[..., 3:], b[..., 3:], rtol=1e-5, atol=1e-5)
    assert np.linalg.norm(a - b) / np.linalg.norm(b) < 2e-2
    for vv in vols:
        vv.destroy()


def test_ragged_grids_with_their_own_boxes(ctx, orc):
    """non-cubic grids whose AABBs differ from each other and from the shape (gridvolume `min`/`max`, gridvolume.cpp:112-117)"""
    rng = np.random.RandomState(3)
    dens = rng.rand(7, 9, 11).astype(np.float32)                  # [z][y][x]
    yy = np.linspace(0, 1, 6, dtype=np.float32)[None, :, None]
    rif = (1.3 + 0.3 * yy + np.zeros((5, 6, 7), np.float32)).astype(np.float32)
    p = scenes.curved_scene(N=8, w=24, h=20, density=dens, density_aabb=([-1.2, -1.0, -1.1], [1.1, 1.3, 1.0]),
                            rif=rif, rif_aabb=([-1.5, -1.4, -1.3], [1.2, 1.6, 1.4]), stepsize=0.05)
    sc, vols = ctx.upload_scene(p)
    for s in (0, 1):
        assert _agree(ctx.render_paths(sc, s, seed=5), orc.render_paths(p, s, 5)) > 0.99
    for vv in vols:
        vv.destroy()


def test_minimum_grid_and_depth_limits(ctx, orc):
    d2 = np.array([[[0.2, 0.9], [0.5, 0.1]], [[0.7, 0.3], [1.0, 0.6]]], np.float32)       # 2 x 2 x 2: one cell
    r2 = np.array([[[1.3, 1.3], [1.5, 1.5]], [[1.3, 1.3], [1.5, 1.5]]], np.float32)
    for md in (1, 2, 3, -1):
        p = scenes.curved_scene(N=8, w=16, h=12, density=d2, rif=r2, stepsize=0.1, max_depth=md)
        sc, vols = ctx.upload_scene(p)
        a = ctx.render_paths(sc, 0, seed=6); b = orc.render_paths(p, 0, 6)
        assert _agree(a, b) > 0.99, md
        if md in (1, 2):
            assert np.array_equal(a, b)                             # no scattering is possible: closed-form paths, identical
        for vv in vols:
            vv.destroy()


def test_large_sample_indices_are_distinct_streams(ctx, orc):
    p = scenes.straight_scene(N=8, w=8, h=8)
    sc, vols = ctx.upload_scene(p)
    a = ctx.render_paths(sc, 2**31 - 2, seed=1); b = orc.render_paths(p, 2**31 - 2, 1)
    assert _agree(a, b) > 0.98 and not np.array_equal(a, ctx.render_paths(sc, 0, seed=1))
    for vv in vols:
        vv.destroy()


def test_lookups_at_infinite_and_huge_coordinates_are_outside_the_grid(ctx, orc):
    """float -> int conversion saturates on the GPU (+inf and anything >= 2^31 become INT_MAX): the bounds test of lookupFloat /
    lookupSpectrum must not wrap to "inside" and fetch from a wild address.  Such points are outside the grid: value 0, linear index -1."""
    rng = np.random.default_rng(5)
    d = rng.random((6, 7, 8)).astype(np.float32); rgb = rng.random((6, 7, 8, 3)).astype(np.float32)
    lo, hi = [0, 0, 0], [1, 1, 1]
    big = [np.inf, 3e9, 1e30, 3.4e38, -np.inf, -3e9, -1e30]
    pts = np.array([[b if a == k else 0.5 for a in range(3)] for b in big for k in range(3)] + [[np.inf] * 3, [0.5, 0.5, 0.5]], np.float32)
    v = ctx.upload_volume(d, lo, hi); c = ctx.upload_volume(rgb, lo, hi)
    val, idx = ctx.lookup_trilinear(v, pts)
    oval, oidx = orc.lookup_trilinear(d, lo, hi, pts)
    assert (val[:-1] == 0).all() and (idx[:-1, 3] == -1).all() and val[-1] == oval[-1] and idx[-1, 3] == oidx[-1, 3] >= 0
    assert (oval[:-1] == 0).all()
    out = ctx.lookup_trilinear_rgb(c, pts)
    assert (out[:-1] == 0).all() and np.array_equal(out[-1], orc.lookup_trilinear_rgb(rgb, lo, hi, pts)[-1])
    v.destroy(); c.destroy()
