"""ctypes binding of the CPU oracle (oracle/libmer_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (mitsubaer_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

VOL_F32, VOL_U8 = 1, 3
SIGMA_HOMOGENEOUS, SIGMA_GRID = 0, 1
RIF_CONST, RIF_TRILINEAR, RIF_BSPLINE3 = 0, 1, 2
STEP_VERLET, STEP_RK4 = 0, 1
BOUNDARY_AABB, BOUNDARY_SPHERE = 0, 1
PHASE_ISOTROPIC, PHASE_HG = 0, 1
TR_WOODCOCK2, TR_RATIO = 0, 1
STRATEGY_BALANCE, STRATEGY_SINGLE, STRATEGY_MANUAL, STRATEGY_MAXIMUM = 0, 1, 2, 3
FILTER_BOX, FILTER_GAUSSIAN = 0, 1
ALBEDO_CONST, ALBEDO_GRID = 0, 1
C_PATHS, C_STEPS, C_RIF_EVALS, C_TENTATIVE, C_REAL, C_SEGMENTS, C_NEE = range(7)
C_COUNT = 16


class Grid(C.Structure):
    _fields_ = [("res", C.c_int32 * 3), ("channels", C.c_int32), ("dtype", C.c_int32),
                ("aabb_min", C.c_float * 3), ("aabb_max", C.c_float * 3), ("world_to_volume", C.c_float * 12), ("data", C.c_void_p)]


class Scene(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32),
        ("fov_x_deg", C.c_float), ("near_clip", C.c_float), ("far_clip", C.c_float),
        ("cam_to_world", C.c_float * 12),
        ("rfilter", C.c_int32), ("rfilter_param", C.c_float),
        ("max_depth", C.c_int32), ("rr_depth", C.c_int32), ("hide_emitters", C.c_int32),
        ("boundary", C.c_int32), ("bmin", C.c_float * 3), ("bmax", C.c_float * 3),
        ("sph_center", C.c_float * 3), ("sph_radius", C.c_float),
        ("sigma_mode", C.c_int32), ("sigma_a", C.c_float * 3), ("sigma_s", C.c_float * 3),
        ("strategy", C.c_int32), ("channel", C.c_int32), ("sampling_density", C.c_float),
        ("medium_sampling_weight", C.c_float),
        ("density", Grid), ("density_scale", C.c_float),
        ("albedo_mode", C.c_int32), ("albedo", C.c_float * 3), ("albedo_grid", Grid),
        ("rif_mode", C.c_int32), ("rif_const", C.c_float), ("rif", Grid),
        ("stepper", C.c_int32), ("stepsize", C.c_float), ("rif_double", C.c_int32),
        ("phase", C.c_int32), ("g", C.c_float),
        ("tr_estimator", C.c_int32),
        ("env_radiance", C.c_float * 3), ("emission", C.c_float * 3),
        ("point_position", C.c_float * 3), ("point_intensity", C.c_float * 3),
        ("decomposition", C.c_int32), ("min_bound", C.c_float), ("max_bound", C.c_float), ("bin_width", C.c_float),
        ("calibrated_transient", C.c_int32),
        ("modulation", C.c_int32), ("mod_lambda", C.c_float), ("mod_phase_deg", C.c_float), ("mod_P", C.c_int32), ("mod_neighbors", C.c_int32),
        ("boundary_bsdf", C.c_int32),
        ("sdf", Grid),
        ("aggressive_tracing", C.c_int32),
        ("sdf_max_error", C.c_float),
        ("ac_n_o", C.c_float), ("ac_n_max", C.c_float), ("ac_k_r", C.c_float), ("ac_mode", C.c_int32),
        ("method", C.c_int32), ("het_stepsize", C.c_float),
        ("area_to_world", C.c_float * 12), ("area_radiance", C.c_float * 3),
    ]


def build(force=False):
    so = os.path.join(_HERE, "libmer_oracle.so")
    src = os.path.join(_HERE, "mer_oracle.cpp")
    if force or not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.environ.get("ORC_LIB")                  # e.g. a sanitizer build of the oracle (CPU only)
        if not so:
            so = os.path.join(_HERE, "libmer_oracle.so")
            if not os.path.exists(so):
                build()
        _LIB = C.CDLL(so)
        _LIB.orc_last_error.restype = C.c_char_p
        _LIB.orc_render.restype = C.c_int
        _LIB.orc_render_paths.restype = C.c_int
    return _LIB


def _fp(a):
    return a.ctypes.data_as(C.c_void_p)


def make_grid(data, aabb_min, aabb_max, keep, to_world=None):
    """data: numpy array indexed [z][y][x] (or [z][y][x][c]); float32 or uint8; to_world: the plugin's toWorld (None = identity)."""
    g = Grid()
    if data is None:
        return g
    if to_world is not None:
        m = np.eye(4); t = np.asarray(to_world, np.float64); m[:t.shape[0], :4] = t
        g.world_to_volume[:] = [float(v) for v in np.linalg.inv(m)[:3, :4].astype(np.float32).reshape(-1)]
    a = np.ascontiguousarray(data)
    keep.append(a)
    ch = 1 if a.ndim == 3 else a.shape[3]
    g.res[:] = [a.shape[2], a.shape[1], a.shape[0]]
    g.channels = ch
    g.dtype = VOL_U8 if a.dtype == np.uint8 else VOL_F32
    g.aabb_min[:] = [float(v) for v in aabb_min]
    g.aabb_max[:] = [float(v) for v in aabb_max]
    g.data = a.ctypes.data
    return g


def _sdf_max_error(p):
    """maxSDFError() of the signed-distance volume: one voxel diagonal (src/volume/splinevolume.cpp:282), unless the scene gives it"""
    e = getattr(p, "sdf_max_error", None)
    if e is not None:
        return float(e)
    if getattr(p, "sdf", None) is None:
        return 0.0
    nz, ny, nx = np.asarray(p.sdf).shape[:3]
    lo, hi = np.asarray(p.sdf_aabb[0], np.float64), np.asarray(p.sdf_aabb[1], np.float64)
    st = (hi - lo) / np.array([nx - 1, ny - 1, nz - 1], np.float64)
    return float(np.sqrt((st * st).sum()))


def make_scene(p):
    """p: mitsubaer_amd.scene.SceneParams (plain attribute bag).  Returns (Scene, keepalive list)."""
    keep = []
    s = Scene()
    s.width, s.height = p.width, p.height
    s.fov_x_deg, s.near_clip, s.far_clip = p.fov_x_deg, p.near_clip, p.far_clip
    s.cam_to_world[:] = [float(v) for v in np.asarray(p.cam_to_world, np.float32).reshape(-1)]
    s.rfilter, s.rfilter_param = p.rfilter, p.rfilter_param
    s.max_depth, s.rr_depth, s.hide_emitters = p.max_depth, p.rr_depth, int(p.hide_emitters)
    s.boundary = p.boundary
    s.bmin[:] = p.bmin; s.bmax[:] = p.bmax
    s.sph_center[:] = p.sph_center; s.sph_radius = p.sph_radius
    s.sigma_mode = p.sigma_mode
    s.sigma_a[:] = p.sigma_a; s.sigma_s[:] = p.sigma_s
    s.strategy, s.channel, s.sampling_density = p.strategy, p.channel, p.sampling_density
    s.medium_sampling_weight = p.medium_sampling_weight
    s.density = make_grid(p.density, p.density_aabb[0], p.density_aabb[1], keep, getattr(p, "density_to_world", None)) if p.density is not None else Grid()
    s.density_scale = p.density_scale
    s.albedo_mode = p.albedo_mode
    s.albedo[:] = p.albedo
    s.albedo_grid = make_grid(p.albedo_grid, p.albedo_aabb[0], p.albedo_aabb[1], keep, getattr(p, "albedo_to_world", None)) if p.albedo_grid is not None else Grid()
    s.rif_mode, s.rif_const = p.rif_mode, p.rif_const
    s.rif = make_grid(p.rif, p.rif_aabb[0], p.rif_aabb[1], keep, getattr(p, "rif_to_world", None)) if (p.rif is not None and p.rif_mode != 8) else Grid()
    s.ac_n_o, s.ac_n_max, s.ac_k_r, s.ac_mode = (float(getattr(p, "ac_n_o", 1.0)), float(getattr(p, "ac_n_max", 0.0)),
                                                 float(getattr(p, "ac_k_r", 1.0)), int(getattr(p, "ac_mode", 0)))
    s.method = int(getattr(p, "method", 0)); s.het_stepsize = float(getattr(p, "het_stepsize", 0.0))
    s.sdf = make_grid(p.sdf, p.sdf_aabb[0], p.sdf_aabb[1], keep, getattr(p, "sdf_to_world", None)) if p.sdf is not None else Grid()
    s.stepper, s.stepsize, s.rif_double = p.stepper, p.stepsize, int(getattr(p, "rif_double", 0))
    s.phase, s.g = p.phase, p.g
    s.tr_estimator = p.tr_estimator
    s.env_radiance[:] = p.env_radiance
    s.emission[:] = p.emission
    s.point_position[:] = p.point_position; s.point_intensity[:] = p.point_intensity
    s.decomposition = p.decomposition; s.min_bound = p.min_bound; s.max_bound = p.max_bound; s.bin_width = p.bin_width
    s.calibrated_transient = int(p.calibrated_transient)
    s.modulation = p.modulation; s.mod_lambda = p.mod_lambda; s.mod_phase_deg = p.mod_phase_deg; s.mod_P = p.mod_P; s.mod_neighbors = p.mod_neighbors
    s.boundary_bsdf = p.boundary_bsdf
    s.aggressive_tracing = int(getattr(p, "aggressive_tracing", False)); s.sdf_max_error = _sdf_max_error(p)
    a2w = getattr(p, "area_to_world", None)
    m = np.eye(4); t = np.asarray(a2w if a2w is not None else np.eye(4), np.float64); m[:t.shape[0], :4] = t
    s.area_to_world[:] = [float(v) for v in m[:3, :4].astype(np.float32).reshape(-1)]
    s.area_radiance[:] = getattr(p, "area_radiance", [0.0, 0.0, 0.0])
    return s, keep


def lookup_trilinear(data, aabb_min, aabb_max, pts, to_world=None):
    keep = []
    g = make_grid(data, aabb_min, aabb_max, keep, to_world)
    pts = np.ascontiguousarray(pts, np.float32)
    n = pts.shape[0]
    val = np.empty(n, np.float32)
    idx = np.empty((n, 4), np.int32)
    lib().orc_lookup_trilinear(C.byref(g), _fp(pts), C.c_int64(n), _fp(val), _fp(idx))
    return val, idx


def lookup_trilinear_rgb(data, aabb_min, aabb_max, pts, to_world=None):
    keep = []
    g = make_grid(data, aabb_min, aabb_max, keep, to_world)
    pts = np.ascontiguousarray(pts, np.float32)
    n = pts.shape[0]
    out = np.empty((n, 3), np.float32)
    lib().orc_lookup_trilinear_rgb(C.byref(g), _fp(pts), C.c_int64(n), _fp(out))
    return out


def trilinear_value_grad(data, aabb_min, aabb_max, pts):
    keep = []
    g = make_grid(data, aabb_min, aabb_max, keep)
    pts = np.ascontiguousarray(pts, np.float32)
    n = pts.shape[0]
    val = np.empty(n, np.float32)
    grad = np.empty((n, 3), np.float32)
    lib().orc_trilinear_value_grad(C.byref(g), _fp(pts), C.c_int64(n), _fp(val), _fp(grad))
    return val, grad


def bspline_build(data, double=False):
    """data[z][y][x] float32 -> coefficient array of the same shape (float32 or float64)."""
    a = np.ascontiguousarray(data, np.float32)
    N = (C.c_int32 * 3)(a.shape[2], a.shape[1], a.shape[0])
    out = np.empty(a.shape, np.float64 if double else np.float32)
    (lib().orc_bspline_build_f64 if double else lib().orc_bspline_build_f32)(_fp(a), N, _fp(out))
    return out


def bspline_eval(coeff, xmin, xmax, pts, hessian=False):
    double = coeff.dtype == np.float64
    dt = np.float64 if double else np.float32
    c = np.ascontiguousarray(coeff)
    N = (C.c_int32 * 3)(c.shape[2], c.shape[1], c.shape[0])
    mn = (C.c_float * 3)(*[float(v) for v in xmin]); mx = (C.c_float * 3)(*[float(v) for v in xmax])
    pts = np.ascontiguousarray(pts, dt)
    n = pts.shape[0]
    val = np.empty(n, dt); grad = np.empty((n, 3), dt)
    hess = np.empty((n, 9), dt) if hessian else None
    fn = lib().orc_bspline_eval_f64 if double else lib().orc_bspline_eval_f32
    fn(_fp(c), N, mn, mx, _fp(pts), C.c_int64(n), _fp(val), _fp(grad), _fp(hess) if hessian else None)
    return (val, grad, hess) if hessian else (val, grad)


def rif_eval(p, pts):
    """the scene's RIF (any rif_mode) at pts: value[n], gradient[n][3], Hessian[n][3][3] (float64 arrays; fp64 arithmetic when p.rif_double)"""
    s, keep = make_scene(p)
    pts = np.ascontiguousarray(pts, np.float32); n = pts.shape[0]
    val = np.empty(n, np.float64); grad = np.empty((n, 3), np.float64); hess = np.empty((n, 3, 3), np.float64)
    f = lib().orc_rif_eval; f.restype = None
    f(C.byref(s), _fp(pts), C.c_int64(n), val.ctypes.data_as(C.c_void_p), grad.ctypes.data_as(C.c_void_p), hess.ctypes.data_as(C.c_void_p))
    return val, grad, hess


def er_trace(p, p0, d0, dist):
    s, keep = make_scene(p)
    p0 = np.ascontiguousarray(p0, np.float32); d0 = np.ascontiguousarray(d0, np.float32)
    dist = np.ascontiguousarray(dist, np.float32)
    n = p0.shape[0]
    op = np.empty((n, 3), np.float32); ov = np.empty((n, 3), np.float32)
    ds = np.empty(n, np.float32); oo = np.empty(n, np.float32); ok = np.empty(n, np.int32)
    lib().orc_er_trace(C.byref(s), _fp(p0), _fp(d0), _fp(dist), C.c_int64(n), _fp(op), _fp(ov), _fp(ds), _fp(oo), _fp(ok))
    return op, ov, ds, oo, ok


def sample_distance(p, o, d, maxt, seed):
    s, keep = make_scene(p)
    o = np.ascontiguousarray(o, np.float32); d = np.ascontiguousarray(d, np.float32)
    maxt = np.ascontiguousarray(maxt, np.float32)
    n = o.shape[0]
    rec = np.empty((n, 20), np.float32)
    lib().orc_sample_distance(C.byref(s), _fp(o), _fp(d), _fp(maxt), C.c_int64(n), C.c_uint64(seed), _fp(rec))
    return rec


def eval_transmittance(p, o, d, maxt, seed):
    s, keep = make_scene(p)
    o = np.ascontiguousarray(o, np.float32); d = np.ascontiguousarray(d, np.float32)
    maxt = np.ascontiguousarray(maxt, np.float32)
    n = o.shape[0]
    out = np.empty((n, 3), np.float32)
    lib().orc_eval_transmittance(C.byref(s), _fp(o), _fp(d), _fp(maxt), C.c_int64(n), C.c_uint64(seed), _fp(out))
    return out


def connect(p, p1, p2, seed):
    s, keep = make_scene(p)
    p1 = np.ascontiguousarray(p1, np.float32); p2 = np.ascontiguousarray(p2, np.float32)
    n = p1.shape[0]
    out = np.zeros((n, 12), np.float32)
    lib().orc_connect(C.byref(s), _fp(p1), _fp(p2), C.c_int64(n), C.c_uint64(seed), _fp(out))
    return out


def phase_sample(kind, g, wi, u2):
    wi = np.ascontiguousarray(wi, np.float32); u2 = np.ascontiguousarray(u2, np.float32)
    n = wi.shape[0]
    wo = np.empty((n, 3), np.float32); pdf = np.empty(n, np.float32)
    lib().orc_phase_sample(C.c_int32(kind), C.c_float(g), _fp(wi), _fp(u2), C.c_int64(n), _fp(wo), _fp(pdf))
    return wo, pdf


def phase_eval(kind, g, wi, wo):
    wi = np.ascontiguousarray(wi, np.float32); wo = np.ascontiguousarray(wo, np.float32)
    n = wi.shape[0]
    val = np.empty(n, np.float32)
    lib().orc_phase_eval(C.c_int32(kind), C.c_float(g), _fp(wi), _fp(wo), C.c_int64(n), _fp(val))
    return val


def camera_rays(p, pos2):
    s, keep = make_scene(p)
    pos2 = np.ascontiguousarray(pos2, np.float32)
    n = pos2.shape[0]
    o = np.empty((n, 3), np.float32); d = np.empty((n, 3), np.float32)
    lib().orc_camera_rays(C.byref(s), _fp(pos2), C.c_int64(n), _fp(o), _fp(d))
    return o, d


def filter_table(kind, param):
    v = np.zeros(33, np.float32)
    r = C.c_float(); sc = C.c_float()
    lib().orc_filter_table(C.c_int32(kind), C.c_float(param), _fp(v), C.byref(r), C.byref(sc))
    return v, r.value, sc.value


def correlation(p, path_length):
    s, keep = make_scene(p)
    t = np.ascontiguousarray(path_length, np.float32)
    out = np.empty(t.shape[0], np.float32)
    lib().orc_correlation(C.byref(s), _fp(t), C.c_int64(t.shape[0]), _fp(out))
    return out


def maxexp(sigma_t, u):
    """MaxExpDist (src/medium/maxexp.h): rows (t = sample(u), pdf from sample, pdf(t), cdf(t))"""
    u = np.ascontiguousarray(u, np.float32); out = np.empty((u.shape[0], 4), np.float32)
    st = (C.c_float * 3)(*[float(v) for v in sigma_t])
    f = lib().orc_maxexp; f.restype = C.c_int
    if f(st, _fp(u), C.c_int64(u.shape[0]), _fp(out)) != 0:
        raise RuntimeError(lib().orc_last_error().decode())
    return out


def rng_floats(seed, pixel, sample, n):
    out = np.empty(n, np.float32)
    lib().orc_rng_floats(C.c_uint64(seed), C.c_uint32(pixel), C.c_uint32(sample), C.c_int32(n), _fp(out))
    return out


def render(p, spp_begin, spp_count, seed, nthreads=8, rows=None):
    s, keep = make_scene(p)
    ch = lib().orc_film_channels(C.byref(s))
    if ch < 0:
        raise RuntimeError(lib().orc_last_error().decode())
    film = np.zeros((p.height, p.width, ch), np.float32)
    counters = np.zeros(C_COUNT, np.uint64)
    y0, y1 = rows if rows else (0, p.height)
    rc = lib().orc_render(C.byref(s), C.c_int32(spp_begin), C.c_int32(spp_count), C.c_uint64(seed),
                          C.c_int32(y0), C.c_int32(y1), C.c_int32(nthreads), _fp(film), _fp(counters))
    if rc != 0:
        raise RuntimeError(lib().orc_last_error().decode())
    return film, counters


def render_paths(p, sample_index, seed, nthreads=8):
    s, keep = make_scene(p)
    out = np.zeros((p.height, p.width, 3), np.float32)
    rc = lib().orc_render_paths(C.byref(s), C.c_int32(sample_index), C.c_uint64(seed), C.c_int32(nthreads), _fp(out))
    if rc != 0:
        raise RuntimeError(lib().orc_last_error().decode())
    return out
