"""N2 (second half): the medium shape as the negative region of a signed-distance grid (`sdf` child of heterogeneousrefractive).
GPU vs the oracle per path, and the SDF-grid sphere against the analytic sphere boundary."""
import numpy as np
import pytest
from mitsubaer_amd import params as P, synth, capi
from tests import scenes

pytestmark = pytest.mark.gpu
BOX = ([-1.2] * 3, [1.2] * 3)


def _sdf(N=64, radius=0.9):
    return -synth.sphere_sdf(N, radius=radius, aabb_min=BOX[0], aabb_max=BOX[1])       # negative inside


SD = dict(boundary=P.BOUNDARY_SDF, sdf_aabb=BOX)
CASES = {
    "straight_null": lambda: scenes.straight_scene(N=24, sdf=_sdf(), **SD),
    "straight_dielectric": lambda: scenes.straight_scene(N=24, sdf=_sdf(), rif_const=1.33, boundary_bsdf=P.BSDF_HDIELECTRIC, **SD),
    "homogeneous_dielectric": lambda: scenes.homogeneous_scene(sdf=_sdf(), rif_const=1.5, boundary_bsdf=P.BSDF_HDIELECTRIC, **SD),
    "curved_null_rk4": lambda: scenes.curved_scene(N=24, rif="radial", sdf=_sdf(), **SD),
    "curved_dielectric_verlet": lambda: scenes.curved_scene(N=24, rif="radial", stepper=P.STEP_VERLET, sdf=_sdf(), boundary_bsdf=P.BSDF_HDIELECTRIC, **SD),
    "curved_bspline_dielectric": lambda: scenes.bspline_scene(N=24, sdf=_sdf(), boundary_bsdf=P.BSDF_HDIELECTRIC, **SD),
    # `aggressivetracing` (heterogeneousrefractive.cpp:473-493): untested legs of min(depth, distance left) while deep inside the shape
    "aggressive_curved_rk4": lambda: scenes.curved_scene(N=24, rif="radial", sdf=_sdf(), aggressive_tracing=True, **SD),
    "aggressive_dielectric_verlet": lambda: scenes.curved_scene(N=24, rif="radial", stepper=P.STEP_VERLET, sdf=_sdf(), boundary_bsdf=P.BSDF_HDIELECTRIC,
                                                                 aggressive_tracing=True, **SD),
    "aggressive_homogeneous_sigma": lambda: scenes.curved_scene(N=24, rif="radial", sigma_mode=P.SIGMA_HOMOGENEOUS, phase=P.PHASE_ISOTROPIC, sdf=_sdf(),
                                                                 aggressive_tracing=True, sdf_max_error=0.01, **SD),
    "point_curved_sdf": lambda: scenes.curved_scene(N=24, w=32, h=24, rif="radial", sdf=_sdf(), env_radiance=[0, 0, 0], point_position=[0.2, 0.3, -0.1],
                                                   point_intensity=[1.0, 0.8, 0.5], **SD),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_sdf_boundary_paths_match_oracle(ctx, orc, name):
    p = CASES[name]()
    sc, vols = ctx.upload_scene(p)
    for s in (0, 1):
        a = ctx.render_paths(sc, s, seed=3); b = orc.render_paths(p, s, 3)
        assert np.isfinite(a).all()
        close = np.abs(a - b).max(2) <= 1e-4 * np.maximum(1.0, np.abs(b).max(2))
        assert close.mean() > (0.92 if name.startswith("point_curved") else 0.99), close.mean()
    for v in vols:
        v.destroy()


@pytest.mark.parametrize("layout,buffer_loads", [(capi.LAYOUT_BRICK27, 1), (capi.LAYOUT_BRICK27, 0), (capi.LAYOUT_CELL8, 0), (capi.LAYOUT_AUTO, 1)])
@pytest.mark.parametrize("name", ["curved_null_rk4", "aggressive_dielectric_verlet", "point_curved_sdf"])
def test_sdf_boundary_with_the_record_layouts_of_large_fields(ctx, orc, name, layout, buffer_loads):
    """the signed-distance boundary with the RIF in BRICK27 records (buffer and global loads) and CELL8 records read with global loads (what a
    field of 4 GiB or more selects; option buffer_loads = 0 selects it for a small one): per path against the oracle, and no bit different
    from the dense layout where no connection solver runs"""
    p = CASES[name]()
    with ctx.options(buffer_loads=buffer_loads):
        sc, vols = ctx.upload_scene(p, layout=layout)
        a = ctx.render_paths(sc, 0, seed=3)
    b = orc.render_paths(p, 0, 3)
    close = np.abs(a - b).max(2) <= 1e-4 * np.maximum(1.0, np.abs(b).max(2))
    assert close.mean() > (0.92 if name.startswith("point_curved") else 0.99), close.mean()
    sd, vd = ctx.upload_scene(p, layout=capi.LAYOUT_DENSE)
    assert np.array_equal(ctx.render_paths(sd, 0, seed=3), a)
    for v in vols + vd:
        v.destroy()


def test_sdf_sphere_reproduces_the_analytic_sphere(ctx):
    """a 96^3 signed-distance grid of a sphere against boundary = sphere: same image to the grid's resolution"""
    kw = dict(N=24, w=32, h=24, rif="radial", boundary_bsdf=P.BSDF_HDIELECTRIC, fov_x_deg=40.0, rfilter=P.FILTER_BOX, rfilter_param=0.5)
    pa = scenes.curved_scene(boundary=P.BOUNDARY_SPHERE, sph_radius=0.9, **kw)
    pb = scenes.curved_scene(sdf=_sdf(96), **SD, **kw)
    sa, va = ctx.upload_scene(pa); sb, vb = ctx.upload_scene(pb)
    fa = ctx.render_to_host(sa, 0, 256, seed=1); fb = ctx.render_to_host(sb, 0, 256, seed=1)
    ma = fa[..., :3].sum() / fa[..., 4].sum(); mb = fb[..., :3].sum() / fb[..., 4].sum()
    assert abs(mb / ma - 1.0) < 5e-3, (ma, mb)
    for v in va + vb:
        v.destroy()


def test_aggressive_tracing_changes_the_step_partition_not_the_image(ctx):
    """legs without inside tests re-partition the steps of a segment (each leg ends with its own remainder step), so paths differ in
    the last bits but the image is the same; an error bound larger than the shape switches the legs off: bit-identical to plain tracing"""
    kw = dict(N=24, w=32, h=24, rif="radial", fov_x_deg=40.0, rfilter=P.FILTER_BOX, rfilter_param=0.5, sdf=_sdf(), **SD)
    pa = scenes.curved_scene(**kw)
    pb = scenes.curved_scene(aggressive_tracing=True, **kw)
    pc = scenes.curved_scene(aggressive_tracing=True, sdf_max_error=10.0, **kw)
    sa, va = ctx.upload_scene(pa); sb, vb = ctx.upload_scene(pb); sc, vc = ctx.upload_scene(pc)
    a = ctx.render_paths(sa, 0, seed=4); b = ctx.render_paths(sb, 0, seed=4); c = ctx.render_paths(sc, 0, seed=4)
    assert np.array_equal(a, c) and not np.array_equal(a, b)
    fa = ctx.render_to_host(sa, 0, 256, seed=1); fb = ctx.render_to_host(sb, 0, 256, seed=1)
    ma = fa[..., :3].sum() / fa[..., 4].sum(); mb = fb[..., :3].sum() / fb[..., 4].sum()
    assert abs(mb / ma - 1.0) < 5e-3, (ma, mb)
    for v in va + vb + vc:
        v.destroy()
    pd = scenes.curved_scene(N=24, aggressive_tracing=True)                       # no sdf volume
    sd, vd = ctx.upload_scene(pd)
    with pytest.raises(RuntimeError, match="aggressivetracing needs"):
        ctx.render_paths(sd, 0)
    for v in vd:
        v.destroy()


def test_sdf_boundary_is_refused_where_it_is_not_built(ctx):
    p = CASES["curved_null_rk4"]()
    sc, vols = ctx.upload_scene(p)
    with pytest.raises(RuntimeError, match="mer_render only"):
        ctx.er_trace(sc, np.zeros((1, 3), np.float32), np.array([[1, 0, 0]], np.float32), np.array([0.1], np.float32))
    sc.sdf = 0
    with pytest.raises(RuntimeError, match="no sdf volume"):
        ctx.render_paths(sc, 0)
    for v in vols:
        v.destroy()
