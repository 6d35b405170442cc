"""Flat scene parameters for the hot path (what the C-ABI descs carry).

Vocabulary follows the reference's plugins: a `heterogeneous` /
`homogeneous` / `heterogeneousrefractive` medium with `density` / `albedo` / `rif` volumes,
an `hg` / `isotropic` phase function, a `perspective` sensor with an `hdrfilm`, a
`volpath` integrator and a `constant` environment emitter.
"""
import numpy as np

# enums shared with include/mer.h
VOL_F32, VOL_U8 = 1, 3
SIGMA_HOMOGENEOUS, SIGMA_GRID = 0, 1
RIF_CONST, RIF_TRILINEAR, RIF_BSPLINE3 = 0, 1, 2
RIF_ACOUSTIC = 8          # acousticrifvolume, evaluated analytically: n_o + n_max J_m(k_r r) cos(m phi) in the (y, z) plane
STEP_VERLET, STEP_RK4 = 0, 1
BOUNDARY_AABB, BOUNDARY_SPHERE, BOUNDARY_SDF = 0, 1, 2
PHASE_ISOTROPIC, PHASE_HG = 0, 1
TR_WOODCOCK2, TR_RATIO = 0, 1
STRATEGY_BALANCE, STRATEGY_SINGLE, STRATEGY_MANUAL, STRATEGY_MAXIMUM = 0, 1, 2, 3
FILTER_BOX, FILTER_GAUSSIAN = 0, 1
ALBEDO_CONST, ALBEDO_GRID = 0, 1
DECOMPOSITION_NONE, DECOMPOSITION_TRANSIENT, DECOMPOSITION_BOUNCE = 0, 1, 2
METHOD_WOODCOCK, METHOD_SIMPSON = 0, 1
BSDF_NULL, BSDF_HDIELECTRIC = 0, 1
MODULATION_NONE, MODULATION_SINE, MODULATION_SQUARE, MODULATION_HAMILTONIAN, MODULATION_MSEQ, MODULATION_DEPTHSELECTIVE = 0, 1, 2, 3, 4, 5


def look_at(origin, target, up):
    """Transform::lookAt (reference src/libcore/transform.cpp:191-214), float32, left-handed.
    Returns the row-major 3x4 camera-to-world matrix with columns (left, newUp, dir, origin)."""
    f = np.float32
    p = np.asarray(origin, f); t = np.asarray(target, f); u = np.asarray(up, f)
    d = (t - p).astype(f)
    d = (d / f(np.sqrt(f(np.dot(d, d))))).astype(f)
    left = np.cross(u, d).astype(f)
    left = (left / f(np.sqrt(f(np.dot(left, left))))).astype(f)
    new_up = np.cross(d, left).astype(f)
    m = np.zeros((3, 4), f)
    m[:, 0] = left; m[:, 1] = new_up; m[:, 2] = d; m[:, 3] = p
    return m


def world_to_volume(to_world):
    """inverse of a volume plugin's `toWorld` (GridDataSource::configure, src/volume/gridvolume.cpp:188-189) as the row-major 3x4
    float32 matrix the grid descs carry; None = identity (all zeros in the desc)."""
    if to_world is None:
        return np.zeros((3, 4), np.float32)
    m = np.eye(4); t = np.asarray(to_world, np.float64); m[:t.shape[0], :4] = t
    return np.linalg.inv(m)[:3, :4].astype(np.float32)


def rotation(axis, angle_deg, translate=(0, 0, 0)):
    """Transform::translate(t) * Transform::rotate(axis, angle) as a 4x4 (src/libcore/transform.cpp)"""
    a = np.asarray(axis, np.float64); a = a / np.linalg.norm(a); th = np.deg2rad(angle_deg); c, s_ = np.cos(th), np.sin(th)
    x, y, z = a
    r = np.array([[c + x * x * (1 - c), x * y * (1 - c) - z * s_, x * z * (1 - c) + y * s_],
                  [y * x * (1 - c) + z * s_, c + y * y * (1 - c), y * z * (1 - c) - x * s_],
                  [z * x * (1 - c) - y * s_, z * y * (1 - c) + x * s_, c + z * z * (1 - c)]])
    m = np.eye(4); m[:3, :3] = r; m[:3, 3] = translate
    return m


class SceneParams:
    """Attribute bag; defaults follow the reference plugin defaults."""

    def __init__(self, **kw):
        # sensor perspective + film hdrfilm (scenes/volumetric/BoundedScatteringVolume_directionalsource.xml:27-49)
        self.width = 512; self.height = 512
        self.fov_x_deg = 95.8402; self.near_clip = 1e-2; self.far_clip = 1e4
        self.cam_to_world = look_at([-3, 0, 0], [-2, 0, 0], [0, 1, 0])
        self.rfilter = FILTER_GAUSSIAN; self.rfilter_param = 0.5
        # integrator volpath (src/librender/integrator.cpp:190-225)
        self.max_depth = -1; self.rr_depth = 5; self.hide_emitters = False
        # shape: cube [-1,1]^3 (scenes/volumetric/bounds.obj), null BSDF
        self.boundary = BOUNDARY_AABB
        self.boundary_bsdf = BSDF_NULL                            # BSDF_HDIELECTRIC: smooth dielectric, eta = RIF at the hit point
        self.bmin = [-1.0, -1.0, -1.0]; self.bmax = [1.0, 1.0, 1.0]
        self.sph_center = [0.0, 0.0, 0.0]; self.sph_radius = 1.0
        self.sdf = None; self.sdf_aabb = ([-1, -1, -1], [1, 1, 1])    # BOUNDARY_SDF: signed-distance grid, negative inside
        self.aggressive_tracing = False; self.sdf_max_error = None    # `aggressivetracing`: untested legs while deep inside the SDF shape;
        #                                                               None = the volume's maxSDFError(): one voxel diagonal (splinevolume.cpp:282)
        # medium
        self.sigma_mode = SIGMA_GRID
        self.sigma_a = [0.05, 0.05, 0.05]; self.sigma_s = [0.5, 3.5, 7.5]
        self.strategy = STRATEGY_BALANCE; self.channel = -1; self.sampling_density = 0.0
        self.medium_sampling_weight = -1.0
        self.density = None; self.density_aabb = ([-1, -1, -1], [1, 1, 1]); self.density_scale = 4.0
        # `toWorld` of the volume plugins (3x4 / 4x4, None = identity): src/volume/gridvolume.cpp:110,188-195
        self.density_to_world = None; self.albedo_to_world = None; self.rif_to_world = None; self.sdf_to_world = None
        self.albedo_mode = ALBEDO_CONST; self.albedo = [0.9, 0.9, 0.9]
        self.albedo_grid = None; self.albedo_aabb = ([-1, -1, -1], [1, 1, 1])
        self.rif_mode = RIF_CONST; self.rif_const = 1.0
        # RIF_ACOUSTIC (src/volume/acousticrifvolume.cpp:101-106): n_o, n_max, k_r = 2 pi freq / speed, mode
        self.ac_n_o = 1.3333; self.ac_n_max = 0.0; self.ac_k_r = 2.0 * 3.14159265358979323846 * 832000.0 / 1500.0; self.ac_mode = 0
        self.rif = None; self.rif_aabb = ([-1, -1, -1], [1, 1, 1])
        self.stepper = STEP_RK4; self.stepsize = 1e-3
        self.rif_double = 0
        self.phase = PHASE_HG; self.g = 0.8
        self.tr_estimator = TR_RATIO
        # heterogeneous `method` (woodcock | simpson) and its `stepSize` (0 = inferred from the grids): src/medium/heterogeneous.cpp:183-202,245-257
        self.method = METHOD_WOODCOCK; self.het_stepsize = 0.0
        self.env_radiance = [1.0, 1.0, 1.0]
        self.emission = [0.0, 0.0, 0.0]
        self.point_position = [0.0, 0.0, 0.0]; self.point_intensity = [0.0, 0.0, 0.0]     # emitter `point`
        # emitter `area` on a `rectangle` shape (src/emitters/area.cpp, src/shapes/rectangle.cpp): the image of [-1,1]^2 x {0} under area_to_world
        # (3x4 or 4x4, no shear; None = identity), radiance into the half space of its normal toWorld(0,0,1); zero radiance = none
        self.area_to_world = None; self.area_radiance = [0.0, 0.0, 0.0]
        # film decomposition (src/librender/film.cpp:56-84): 0 none | 1 transient | 2 bounce (bins by edge count); frames = ceil((max-min)/binWidth)
        self.decomposition = DECOMPOSITION_NONE; self.min_bound = 0.0; self.max_bound = 0.0; self.bin_width = 1.0
        self.calibrated_transient = False
        # path-length modulation (src/librender/pathlengthsampler.cpp:12-40): lambda, phase [deg], P, neighbors
        self.modulation = MODULATION_NONE; self.mod_lambda = 1.0; self.mod_phase_deg = 0.0; self.mod_P = 32; self.mod_neighbors = 3
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError("unknown scene parameter '%s'" % k)
            setattr(self, k, v)

    def copy(self, **kw):
        q = SceneParams()
        q.__dict__.update(self.__dict__)
        for k, v in kw.items():
            if not hasattr(q, k):
                raise AttributeError("unknown scene parameter '%s'" % k)
            setattr(q, k, v)
        return q


def sdf_max_error(p):
    """maxSDFError() of the scene's signed-distance volume: the diagonal of one voxel (src/volume/splinevolume.cpp:282), unless given."""
    if p.sdf_max_error is not None:
        return float(p.sdf_max_error)
    if p.sdf is None:
        return 0.0
    import numpy as np
    nz, ny, nx = np.asarray(p.sdf).shape[:3]
    lo, hi = np.asarray(p.sdf_aabb[0], np.float64), np.asarray(p.sdf_aabb[1], np.float64)
    st = (hi - lo) / np.array([nx - 1, ny - 1, nz - 1], np.float64)
    return float(np.sqrt((st * st).sum()))
