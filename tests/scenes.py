"""Small seeded scenes shared by the parity tests (sizes the oracle finishes in seconds)."""
import numpy as np
from mitsubaer_amd import params as P, synth


def straight_scene(N=32, w=48, h=40, **kw):
    d = synth.density_field(N)
    base = dict(width=w, height=h, density=d, rfilter=P.FILTER_GAUSSIAN, rfilter_param=0.5)
    base.update(kw)
    return P.SceneParams(**base)


def curved_scene(N=32, w=48, h=40, rif="linear", **kw):
    d = synth.density_field(N)
    r = rif if isinstance(rif, np.ndarray) else (synth.linear_rif(N) if rif == "linear" else synth.radial_rif(N))
    base = dict(width=w, height=h, density=d, rif_mode=P.RIF_TRILINEAR, rif=r, stepsize=0.5 * 2.0 / (N - 1),
                rfilter=P.FILTER_GAUSSIAN, rfilter_param=0.5)
    base.update(kw)
    return P.SceneParams(**base)


def bspline_scene(N=32, w=48, h=40, **kw):
    """RIF grid extends past the medium so that every query stays inside the spline-safe box."""
    d = synth.density_field(N)
    r = synth.radial_rif(N, (-1.3, -1.3, -1.3), (1.3, 1.3, 1.3))
    base = dict(width=w, height=h, density=d, rif_mode=P.RIF_BSPLINE3, rif=r, rif_aabb=([-1.3] * 3, [1.3] * 3),
                stepsize=0.5 * 2.0 / (N - 1), stepper=P.STEP_VERLET, rfilter=P.FILTER_BOX, rfilter_param=0.5)
    base.update(kw)
    return P.SceneParams(**base)


def homogeneous_scene(w=48, h=40, **kw):
    base = dict(width=w, height=h, sigma_mode=P.SIGMA_HOMOGENEOUS, phase=P.PHASE_ISOTROPIC,
                rfilter=P.FILTER_GAUSSIAN, rfilter_param=0.5)
    base.update(kw)
    return P.SceneParams(**base)


def rgb_albedo(N, seed=4):
    """RGB albedo grid (gridvolume, 3 channels: src/volume/gridvolume.cpp:390-421), values in [0.5, 0.95]"""
    rng = np.random.RandomState(seed)
    return (0.5 + 0.45 * rng.rand(N, N, N, 3)).astype(np.float32)


def rand_points(n, lo=-1.05, hi=1.05, seed=1):
    rng = np.random.RandomState(seed)
    return rng.uniform(lo, hi, size=(n, 3)).astype(np.float32)


def rand_dirs(n, seed=2):
    rng = np.random.RandomState(seed)
    v = rng.normal(size=(n, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    return v.astype(np.float32)
