"""N4: the analytic ultrasound RIF of `acousticrifvolume` (src/volume/acousticrifvolume.cpp:101-106,224-342) evaluated inside the kernels
(rif_mode = MER_RIF_ACOUSTIC, no grid): GPU vs the oracle per ray and per path."""
import numpy as np
import pytest
from mitsubaer_amd import params as P
from tests import scenes

pytestmark = pytest.mark.gpu
AC = dict(rif_mode=P.RIF_ACOUSTIC, ac_n_o=1.33, ac_n_max=0.08, ac_k_r=4.0)


def _scene(mode, **kw):
    base = dict(N=24, w=48, h=40, stepsize=0.5 * 2.0 / 23, ac_mode=mode, **AC)
    base.update(kw)
    p = scenes.straight_scene(**base)
    return p


@pytest.mark.parametrize("stepper", [P.STEP_VERLET, P.STEP_RK4])
@pytest.mark.parametrize("mode", [0, 2])
def test_er_trace_through_the_analytic_field(ctx, orc, stepper, mode):
    p = _scene(mode, stepper=stepper)
    sc, vols = ctx.upload_scene(p)
    n = 2048
    p0 = scenes.rand_points(n, -0.8, 0.8, seed=5); d0 = scenes.rand_dirs(n, seed=6)
    dist = np.random.RandomState(7).uniform(0.05, 1.5, n).astype(np.float32)
    gp, gv, gd, go, gok = ctx.er_trace(sc, p0, d0, dist)
    op, ov, od, oo, ook = orc.er_trace(p, p0, d0, dist)
    assert np.array_equal(gok, ook)
    assert np.abs(gp - op).max() < 5e-5 and np.abs(gv - ov).max() < 5e-5 and np.abs(go - oo).max() < 5e-5
    bent = np.linalg.norm(gv / np.linalg.norm(gv, axis=1, keepdims=True) - d0, axis=1)
    assert bent.max() > 1e-3                                  # the field does bend the rays
    for v in vols:
        v.destroy()


CASES = {
    "grid_sigma_rk4_m0": lambda: _scene(0, stepper=P.STEP_RK4),
    "grid_sigma_verlet_m2": lambda: _scene(2, stepper=P.STEP_VERLET),
    "homogeneous_sigma_m1": lambda: _scene(1, sigma_mode=P.SIGMA_HOMOGENEOUS, phase=P.PHASE_ISOTROPIC),
    "dielectric_m1": lambda: _scene(1, stepper=P.STEP_RK4, boundary_bsdf=P.BSDF_HDIELECTRIC),
    "point_emitter_m1": lambda: _scene(1, w=24, h=20, stepper=P.STEP_RK4, env_radiance=[0, 0, 0], point_position=[0.2, 0.3, -0.1],
                                       point_intensity=[1.0, 0.8, 0.5]),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_per_path_radiance_matches_oracle(ctx, orc, name):
    p = CASES[name]()
    sc, vols = ctx.upload_scene(p)
    for s in (0, 1):
        a = ctx.render_paths(sc, s, seed=3); b = orc.render_paths(p, s, 3)
        assert np.isfinite(a).all()
        close = np.abs(a - b).max(2) <= 1e-4 * np.maximum(1.0, np.abs(b).max(2))
        assert close.mean() > (0.92 if name.startswith("point") else 0.99), close.mean()
    for v in vols:
        v.destroy()


def test_acoustic_parameters_are_checked(ctx):
    p = _scene(0); p.ac_k_r = 0.0
    sc, vols = ctx.upload_scene(p)
    with pytest.raises(RuntimeError, match="acousticrifvolume"):
        ctx.render_paths(sc, 0)
    for v in vols:
        v.destroy()
