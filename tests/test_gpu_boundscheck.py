"""The bounds-checking build of the library (libmer_check.so = the same sources with -DMER_BOUNDS_CHECK): every index a kernel forms
into a device buffer -- path-state slots, work-list segments and items, hit ring, film, per-path output, grid payloads, cell / brick
records, spline coefficients -- is compared with the buffer's extent, the first violation is recorded and the access redirected.
An out-of-range index is thus REPORTED here even when, in the product build, it would silently read mapped memory (and fault only
when the allocator happens to place the buffer next to an unmapped page: the intermittent abort of round 1).  Every kernel family
of mer_render runs once under the checks; results are bit-identical to the product build."""
import numpy as np
import pytest
from mitsubaer_amd import params as P, capi
from tests import scenes
from tests.test_gpu_sdf import CASES as SDF_CASES
from tests.test_gpu_render import CASES as RENDER_CASES

pytestmark = pytest.mark.gpu
KINDS = {1: "slot", 2: "queue segment overflow", 3: "queue item", 4: "hit ring", 5: "film", 6: "path_out", 7: "dense grid", 8: "cell / brick record",
         9: "spline coefficients", 10: "rgb grid", 11: "live row"}


@pytest.fixture(scope="module")
def cctx():
    c = capi.Context(0, check=True)
    en, *_ = c.debug_bounds()
    assert en, "libmer_check.so was built without -DMER_BOUNDS_CHECK"
    yield c
    c.close()


def _clean(c, what):
    en, n, kind, idx, lim = c.debug_bounds()
    assert n == 0, "%s: %d out-of-range accesses; first: %s index %d, limit %d" % (what, n, KINDS.get(kind, kind), idx, lim)


def test_product_build_has_no_checks(ctx):
    en, n, *_ = ctx.debug_bounds()
    assert not en and n == 0


@pytest.mark.parametrize("name", sorted(SDF_CASES))
def test_sdf_boundary_kernels_stay_in_bounds(cctx, ctx, name):
    """the signed-distance kernels (BND = 1) read the RIF with GLOBAL loads: an out-of-range index is not clamped by a buffer descriptor"""
    p = SDF_CASES[name]()
    sc, vols = cctx.upload_scene(p)
    s2, v2 = ctx.upload_scene(p)
    for s in (0, 1):
        a = cctx.render_paths(sc, s, seed=3)
        _clean(cctx, name)
        assert np.array_equal(a, ctx.render_paths(s2, s, seed=3))
    f = cctx.render_to_host(sc, 0, 5, seed=1)
    _clean(cctx, name + " (film, 4 pipelines)")
    assert np.isfinite(f).all()
    for v in vols + v2:
        v.destroy()


@pytest.mark.parametrize("name", sorted(n for n in RENDER_CASES if n.startswith(("point_curved", "cfg", "dielectric_curved", "parity"))))
@pytest.mark.parametrize("buffer_loads", [1, 0])
def test_render_kernels_stay_in_bounds(cctx, name, buffer_loads):
    p = RENDER_CASES[name]()
    layouts = [capi.LAYOUT_DENSE] if p.rif_mode != P.RIF_TRILINEAR else [capi.LAYOUT_DENSE, capi.LAYOUT_CELL8, capi.LAYOUT_BRICK27]
    with cctx.options(buffer_loads=buffer_loads):
        for lay in layouts:
            sc, vols = cctx.upload_scene(p, layout=lay)
            cctx.render_paths(sc, 0, seed=3)
            _clean(cctx, "%s layout %d paths" % (name, lay))
            cctx.render_to_host(sc, 0, 6, seed=2, spp_stride=1)
            _clean(cctx, "%s layout %d film" % (name, lay))
            for v in vols:
                v.destroy()


def test_transient_and_odd_films_stay_in_bounds(cctx):
    p = scenes.curved_scene(N=24, w=33, h=17, decomposition=P.DECOMPOSITION_TRANSIENT, min_bound=0.0, max_bound=12.0, bin_width=0.5)
    sc, vols = cctx.upload_scene(p)
    f = cctx.render_to_host(sc, 0, 4, seed=1)
    _clean(cctx, "transient film")
    assert f.shape[2] == 24 * 3 + 2
    for v in vols:
        v.destroy()
    p = scenes.straight_scene(N=16, w=1, h=1)
    sc, vols = cctx.upload_scene(p)
    cctx.render_to_host(sc, 0, 3, seed=1)
    _clean(cctx, "1x1 film")
    for v in vols:
        v.destroy()


@pytest.mark.parametrize("opts", [dict(march_sort=2), dict(march_sort=3, march_sort_major=1), dict(march_sort=4, mq_sort=0), dict(grid_fit=1, nslots=4096), dict(grid_fit=0, nslots=4096)])
def test_sorted_march_list_and_fitted_grids_stay_in_bounds(cctx, ctx, opts):
    """the counting sort of the march list (keys written with the pushes, histogram / scan / scatter kernels, the XCD-contiguous sweep) and launch
    grids sized by the lists' bound, under the index checks: spawned side walks, four pipelines, a slot pool small enough to be refilled many times"""
    p = scenes.curved_scene(N=32, w=40, h=28, rfilter=P.FILTER_BOX)
    sc, vols = cctx.upload_scene(p)
    s2, v2 = ctx.upload_scene(p)
    with cctx.options(**opts):
        a = cctx.render_paths(sc, 0, seed=3)
        _clean(cctx, "paths %s" % opts)
        f = cctx.render_to_host(sc, 0, 16, seed=2)
        _clean(cctx, "film %s" % opts)
    assert np.array_equal(a, ctx.render_paths(s2, 0, seed=3))
    g = ctx.render_to_host(s2, 0, 16, seed=2)
    assert np.allclose(f, g, rtol=2e-4, atol=2e-5)
    assert abs(float(f[..., 4].sum()) - 40 * 28 * 16) < 1e-2 * 40 * 28 * 16          # every sample landed (box filter: weight 1 each)
    for v in vols + v2:
        v.destroy()


def test_connect_leaf_with_degenerate_pairs_stays_in_bounds(cctx):
    """mer_connect on pairs that make the shooting problem singular or hopeless: coincident points, points a rounding error apart,
    the far corners of the shape, a target outside it (reached through the boundary).  Rejected or solved -- never out of range, never non-finite."""
    for p in (SDF_CASES["point_curved_sdf"](), scenes.curved_scene(N=24, rif="radial", stepper=P.STEP_VERLET), scenes.bspline_scene(N=24)):
        sc, vols = cctx.upload_scene(p)
        a = np.array([[0.2, 0.3, -0.1], [0.2, 0.3, -0.1], [0.0, 0.0, 0.0], [-0.62, -0.62, -0.62], [0.1, 0.1, 0.1], [0.5, 0.0, 0.0]], np.float32)
        b = np.array([[0.2, 0.3, -0.1], [0.2 + 1e-7, 0.3, -0.1], [1e-30, 0.0, 0.0], [0.62, 0.62, 0.62], [3.0, 3.0, 3.0], [-0.5, 1e-4, 0.0]], np.float32)
        out = cctx.connect(sc, a, b, 7)
        _clean(cctx, "mer_connect degenerate pairs")
        assert np.isfinite(out[:, [0, 1, 8, 9]]).all()
        ok = out[:, 0] == 1
        assert np.isfinite(out[ok]).all()
        for v in vols:
            v.destroy()


def test_configs3_at_1024_cubed_stays_in_bounds(cctx, ctx):
    """The size where index WIDTH matters (round 2's fault: 2^27 bricks x 32 words overflowed a 32-bit record index): configs[3]'s 1024^3 fields
    in both record layouts -- CELL8 (32 GiB: global loads, 64-bit addresses) and BRICK27 (16 GiB, 2^32 record words) -- under the bounds-checking
    build: zero violations, and the same per-path image as the product build.  Also the connection queues with many K_connect launches per pass
    (event-queue segments are sized by the number of producer launches)."""
    import bench
    NN, W = 1024, 1024
    p, _ = bench.scene_params("cfg4", NN, W, with_fields=False)
    ref = None
    for lay in (capi.LAYOUT_CELL8, capi.LAYOUT_BRICK27):
        sc, vols = bench.upload(cctx, "cfg4", NN, p, lay)
        a = cctx.render_paths(sc, 0, seed=5)
        _clean(cctx, "1024^3 layout %d paths" % lay)
        cctx.render_to_host(sc, 0, 1, seed=5, tile_rank=3, tile_count=8)
        _clean(cctx, "1024^3 layout %d tile shard" % lay)
        for v in vols:
            v.destroy()
        if ref is None:
            s2, v2 = bench.upload(ctx, "cfg4", NN, p, lay)
            ref = ctx.render_paths(s2, 0, seed=5)
            for v in v2:
                v.destroy()
        assert np.array_equal(a, ref)
    pc = RENDER_CASES["point_curved_cfg5_rgb_albedo_emissive"]()
    with cctx.options(connect_launches=8):
        sc, vols = cctx.upload_scene(pc, layout=capi.LAYOUT_BRICK27)
        cctx.render_to_host(sc, 0, 16, seed=2)
        _clean(cctx, "8 K_connect launches per pass")
        for v in vols:
            v.destroy()
