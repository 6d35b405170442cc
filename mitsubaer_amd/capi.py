"""ctypes binding of libmer.so (include/mer.h): the HIP hot path.  No CPU fallback exists --
loading fails loudly if the library is missing, and context creation fails without a GPU."""
import ctypes as C
import os
import numpy as np
from . import params as P

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MER_LIB", os.path.join(_HERE, "libmer.so"))
CHECK_LIB_PATH = os.path.join(_HERE, "libmer_check.so")     # same sources, -DMER_BOUNDS_CHECK (Context(check=True))
_LIBS = {}

C_PATHS, C_STEPS, C_RIF_EVALS, C_TENTATIVE, C_REAL, C_SEGMENTS, C_NEE, C_LOOP_ITERS, C_ACTIVE_LANES, C_CONNECT_UNITS, C_CONNECT_STEPS, C_CONNECT_LANE_SLOTS, C_SIDE_SPAWNED, C_SIDE_INLINE = range(14)
C_COUNT = 16
LAYOUT_DENSE, LAYOUT_CELL8, LAYOUT_BRICK27, LAYOUT_BRICK125, LAYOUT_AUTO = 0, 1, 2, 3, 4

# every symbol include/mer.h declares (checked by tests/test_abi.py against the header text)
SYMBOLS = [
    "mer_abi_version", "mer_context_create", "mer_context_destroy", "mer_last_error", "mer_context_set_stream",
    "mer_device_info", "mer_context_set_option", "mer_context_get_option", "mer_debug_bounds", "mer_volume_upload", "mer_volume_upload_dev", "mer_volume_build_spline",
    "mer_volume_download_spline", "mer_volume_destroy", "mer_film_channels", "mer_film_alloc_n", "mer_film_zero_n",
    "mer_film_download_n", "mer_film_alloc", "mer_film_zero", "mer_film_download",
    "mer_film_free", "mer_render", "mer_synchronize", "mer_last_kernel_ms", "mer_last_render_stats", "mer_counters_read",
    "mer_counters_reset", "mer_lookup_trilinear", "mer_lookup_trilinear_rgb", "mer_rif_value_grad", "mer_er_trace",
    "mer_sample_distance", "mer_connect", "mer_eval_transmittance", "mer_phase_sample", "mer_phase_eval", "mer_camera_rays",
    "mer_correlation", "mer_render_paths", "mer_rng_floats", "mer_synth_field_dev", "mer_device_free",
    "mer_multi_create", "mer_multi_destroy", "mer_multi_last_error", "mer_multi_size", "mer_multi_context", "mer_multi_set_option",
    "mer_multi_volume_upload", "mer_multi_volume_build_spline", "mer_multi_volume_destroy", "mer_multi_render", "mer_multi_last_stats",
]
SHARD_SAMPLES, SHARD_TILES = 0, 1
REDUCE_NONE, REDUCE_RCCL, REDUCE_PEER_COPY = 0, 1, 2


class GridDesc(C.Structure):
    _fields_ = [("res", C.c_int32 * 3), ("channels", C.c_int32), ("dtype", C.c_int32),
                ("aabb_min", C.c_float * 3), ("aabb_max", C.c_float * 3), ("world_to_volume", C.c_float * 12)]


class SceneDesc(C.Structure):
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
        ("density", C.c_int32), ("density_scale", C.c_float),
        ("albedo_mode", C.c_int32), ("albedo", C.c_float * 3), ("albedo_grid", C.c_int32),
        ("rif_mode", C.c_int32), ("rif_const", C.c_float), ("rif", C.c_int32),
        ("stepper", C.c_int32), ("stepsize", C.c_float),
        ("phase", C.c_int32), ("g", C.c_float),
        ("tr_estimator", C.c_int32),
        ("env_radiance", C.c_float * 3), ("emission", C.c_float * 3),
        ("point_position", C.c_float * 3), ("point_intensity", C.c_float * 3),
        ("decomposition", C.c_int32), ("min_bound", C.c_float), ("max_bound", C.c_float), ("bin_width", C.c_float),
        ("calibrated_transient", C.c_int32),
        ("modulation", C.c_int32), ("mod_lambda", C.c_float), ("mod_phase_deg", C.c_float), ("mod_P", C.c_int32), ("mod_neighbors", C.c_int32),
        ("boundary_bsdf", C.c_int32),
        ("sdf", C.c_int32),
        ("aggressive_tracing", C.c_int32),
        ("sdf_max_error", C.c_float),
        ("ac_n_o", C.c_float), ("ac_n_max", C.c_float), ("ac_k_r", C.c_float), ("ac_mode", C.c_int32),
        ("method", C.c_int32), ("het_stepsize", C.c_float),
        ("area_to_world", C.c_float * 12), ("area_radiance", C.c_float * 3),
    ]


class Shard(C.Structure):
    _fields_ = [("spp_begin", C.c_int32), ("spp_count", C.c_int32), ("spp_stride", C.c_int32),
                ("tile_rank", C.c_int32), ("tile_count", C.c_int32)]


class MerError(RuntimeError):
    """Mirrors the reference's Log(EError) -> std::runtime_error (src/libcore/logger.cpp:100-147)."""


def lib(path=None):
    """The C-ABI library (ctypes).  path = None: libmer.so; Context(check=True) loads libmer_check.so beside it."""
    path = path or LIB_PATH
    if path not in _LIBS:
        if not os.path.exists(path):
            raise MerError("%s is not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "-- there is no CPU fallback for the hot path" % (os.path.basename(path), path))
        L = C.CDLL(path)
        L.mer_last_error.restype = C.c_char_p
        L.mer_last_error.argtypes = [C.c_void_p]
        for s in SYMBOLS:
            if s not in ("mer_last_error", "mer_context_destroy", "mer_multi_destroy", "mer_multi_last_error", "mer_multi_context"):
                getattr(L, s).restype = C.c_int
        L.mer_context_destroy.restype = None
        L.mer_multi_destroy.restype = None
        L.mer_multi_destroy.argtypes = [C.c_void_p]
        L.mer_multi_last_error.restype = C.c_char_p
        L.mer_multi_last_error.argtypes = [C.c_void_p]
        L.mer_multi_context.restype = C.c_void_p
        L.mer_multi_context.argtypes = [C.c_void_p, C.c_int32]
        _LIBS[path] = L
    return _LIBS[path]


def _fp(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, np.float32)


class Volume:
    def __init__(self, ctx, handle, desc, layout):
        self.ctx, self.handle, self.desc, self.layout = ctx, handle, desc, layout
        self.has_spline = False

    def build_spline(self):
        self.ctx._check(self.ctx.lib.mer_volume_build_spline(self.ctx.h, C.c_int32(self.handle)))
        self.has_spline = True
        return self

    def download_spline(self):
        out = np.empty((self.desc.res[2], self.desc.res[1], self.desc.res[0]), np.float32)
        self.ctx._check(self.ctx.lib.mer_volume_download_spline(self.ctx.h, C.c_int32(self.handle), _fp(out)))
        return out

    def destroy(self):
        if self.handle:
            if isinstance(self.ctx, MultiContext):
                self.ctx.destroy_volume(self)
            else:
                self.ctx.lib.mer_volume_destroy(self.ctx.h, C.c_int32(self.handle))
            self.handle = 0


class Context:
    """One context per (process, GPU)."""

    def __init__(self, device_id=0, check=False, _borrowed=None, **options):
        """check=True: the bounds-checking build of the library (libmer_check.so); options: mer_context_set_option names."""
        self.lib = lib(CHECK_LIB_PATH if check else None)
        self.owned = _borrowed is None
        if _borrowed is not None:                      # a context owned by a MultiContext
            self.h = C.c_void_p(_borrowed)
        else:
            self.h = C.c_void_p()
            rc = self.lib.mer_context_create(C.c_int32(device_id), C.byref(self.h))
            if rc != 0:
                raise MerError(self.lib.mer_last_error(None).decode())
        self.device_id = device_id
        for k, v in options.items():
            self.set_option(k, v)

    def _check(self, rc):
        if rc != 0:
            raise MerError(self.lib.mer_last_error(self.h).decode())

    def close(self):
        if self.h and self.owned:
            self.lib.mer_context_destroy(self.h)
        self.h = C.c_void_p()

    def set_option(self, name, value):
        self._check(self.lib.mer_context_set_option(self.h, name.encode(), C.c_int64(int(value))))

    def get_option(self, name):
        v = C.c_int64()
        self._check(self.lib.mer_context_get_option(self.h, name.encode(), C.byref(v)))
        return v.value

    def options(self, **kw):
        """context manager: set options for the duration of a `with` block, then restore them"""
        ctx = self

        class _Scope:
            def __enter__(self_):
                self_.old = {k: ctx.get_option(k) for k in kw}
                for k, v in kw.items():
                    ctx.set_option(k, v)
                return ctx

            def __exit__(self_, *a):
                for k, v in self_.old.items():
                    ctx.set_option(k, v)
        return _Scope()

    def debug_bounds(self):
        """-> (enabled, violations, kind, index, limit) of the bounds-checking build; resets the record"""
        en = C.c_int32(); out = (C.c_uint64 * 4)()
        self._check(self.lib.mer_debug_bounds(self.h, C.byref(en), out))
        return bool(en.value), int(out[0]), int(out[1]), int(out[2]), int(out[3])

    def set_stream(self, stream_ptr):
        self._check(self.lib.mer_context_set_stream(self.h, C.c_void_p(stream_ptr)))

    def device_info(self):
        name = C.create_string_buffer(256)
        cu = C.c_int32(); hbm = C.c_int64()
        self._check(self.lib.mer_device_info(self.h, name, C.c_int32(256), C.byref(cu), C.byref(hbm)))
        return name.value.decode(), cu.value, hbm.value

    # ---- volumes -------------------------------------------------------------------------------
    @staticmethod
    def _desc(shape, channels, dtype, aabb_min, aabb_max, to_world=None):
        d = GridDesc()
        d.world_to_volume[:] = [float(v) for v in P.world_to_volume(to_world).reshape(-1)]
        d.res[:] = [shape[2], shape[1], shape[0]]
        d.channels = channels
        d.dtype = dtype
        d.aabb_min[:] = [float(v) for v in aabb_min]
        d.aabb_max[:] = [float(v) for v in aabb_max]
        return d

    def upload_volume(self, data, aabb_min, aabb_max, layout=LAYOUT_DENSE, to_world=None):
        """data[z][y][x](,c): float32 or uint8 numpy array; to_world: the volume plugin's `toWorld` (3x4 or 4x4), None = identity."""
        a = np.ascontiguousarray(data)
        if a.dtype != np.uint8:
            a = a.astype(np.float32, copy=False)
        ch = 1 if a.ndim == 3 else a.shape[3]
        d = self._desc(a.shape, ch, P.VOL_U8 if a.dtype == np.uint8 else P.VOL_F32, aabb_min, aabb_max, to_world)
        h = C.c_int32()
        self._check(self.lib.mer_volume_upload(self.h, C.byref(d), _fp(a), C.c_int32(layout), C.byref(h)))
        return Volume(self, h.value, d, layout)

    def upload_volume_dev(self, dev_ptr, shape, aabb_min, aabb_max, layout=LAYOUT_DENSE):
        d = self._desc(shape, 1, P.VOL_F32, aabb_min, aabb_max)
        h = C.c_int32()
        self._check(self.lib.mer_volume_upload_dev(self.h, C.byref(d), C.c_void_p(dev_ptr), C.c_int32(layout), C.byref(h)))
        return Volume(self, h.value, d, layout)

    def synth_volume(self, kind, N, layout=LAYOUT_DENSE, aabb_min=(-1, -1, -1), aabb_max=(1, 1, 1)):
        """Synthetic field generated in HBM (0 = sigma_t density, 1 = linear RIF, 2 = radial RIF)."""
        ptr = C.c_void_p()
        self._check(self.lib.mer_synth_field_dev(self.h, C.c_int32(kind), C.c_int32(N), C.byref(ptr)))
        try:
            v = self.upload_volume_dev(ptr.value, (N, N, N), aabb_min, aabb_max, layout)
        finally:
            self.lib.mer_device_free(self.h, ptr)
        return v

    # ---- scene ---------------------------------------------------------------------------------
    def scene_desc(self, p, density=None, albedo_grid=None, rif=None, sdf=None):
        """p: params.SceneParams; volumes as Volume objects."""
        s = SceneDesc()
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
        s.density = density.handle if density is not None else 0
        s.density_scale = p.density_scale
        s.albedo_mode = p.albedo_mode
        s.albedo[:] = p.albedo
        s.albedo_grid = albedo_grid.handle if albedo_grid is not None else 0
        s.rif_mode, s.rif_const = p.rif_mode, p.rif_const
        s.rif = rif.handle if rif is not None else 0
        s.ac_n_o, s.ac_n_max, s.ac_k_r, s.ac_mode = float(p.ac_n_o), float(p.ac_n_max), float(p.ac_k_r), int(p.ac_mode)
        s.method = int(p.method); s.het_stepsize = float(p.het_stepsize)
        s.stepper, s.stepsize = p.stepper, p.stepsize
        s.phase, s.g = p.phase, p.g
        s.tr_estimator = p.tr_estimator
        s.env_radiance[:] = p.env_radiance
        s.emission[:] = p.emission
        s.point_position[:] = p.point_position; s.point_intensity[:] = p.point_intensity
        s.decomposition = p.decomposition; s.min_bound = p.min_bound; s.max_bound = p.max_bound; s.bin_width = p.bin_width
        s.calibrated_transient = int(p.calibrated_transient)
        s.modulation = p.modulation; s.mod_lambda = p.mod_lambda; s.mod_phase_deg = p.mod_phase_deg; s.mod_P = p.mod_P; s.mod_neighbors = p.mod_neighbors
        s.boundary_bsdf = p.boundary_bsdf
        s.sdf = sdf.handle if sdf is not None else 0
        s.aggressive_tracing = int(p.aggressive_tracing); s.sdf_max_error = P.sdf_max_error(p)
        m = np.eye(4); t = np.asarray(p.area_to_world if p.area_to_world is not None else np.eye(4), np.float64); m[:t.shape[0], :4] = t
        s.area_to_world[:] = [float(v) for v in m[:3, :4].astype(np.float32).reshape(-1)]
        s.area_radiance[:] = p.area_radiance
        return s

    def upload_scene(self, p, layout=LAYOUT_DENSE, rif_layout=None):
        """Uploads the numpy fields referenced by p and returns (SceneDesc, [Volume...])."""
        vols = []
        dens = alb = rif = None
        if p.sigma_mode == P.SIGMA_GRID and p.density is not None:
            dl = LAYOUT_CELL8 if layout in (LAYOUT_BRICK27, LAYOUT_BRICK125, LAYOUT_AUTO) else layout   # bricks are the RIF's layout; sigma_t keeps its cell records
            dens = self.upload_volume(p.density, p.density_aabb[0], p.density_aabb[1], dl if np.asarray(p.density).dtype != np.uint8 else LAYOUT_DENSE, p.density_to_world)
            vols.append(dens)
        if p.albedo_mode == P.ALBEDO_GRID and p.albedo_grid is not None:
            alb = self.upload_volume(p.albedo_grid, p.albedo_aabb[0], p.albedo_aabb[1], to_world=p.albedo_to_world)
            vols.append(alb)
        if p.rif_mode not in (P.RIF_CONST, P.RIF_ACOUSTIC) and p.rif is not None:
            rl = layout if rif_layout is None else rif_layout
            rif = self.upload_volume(p.rif, p.rif_aabb[0], p.rif_aabb[1], rl if p.rif_mode == P.RIF_TRILINEAR else LAYOUT_DENSE, p.rif_to_world)
            if p.rif_mode == P.RIF_BSPLINE3:
                rif.build_spline()
            vols.append(rif)
        sdf = None
        if p.boundary == P.BOUNDARY_SDF and p.sdf is not None:
            sdf = self.upload_volume(p.sdf, p.sdf_aabb[0], p.sdf_aabb[1], LAYOUT_DENSE, p.sdf_to_world)
            vols.append(sdf)
        return self.scene_desc(p, dens, alb, rif, sdf), vols

    # ---- film + render -------------------------------------------------------------------------
    def film_channels(self, scene):
        """frames*3 + 2: RGB per frame, alpha, weight (5 in steady state)"""
        ch = C.c_int32()
        self._check(self.lib.mer_film_channels(self.h, C.byref(scene), C.byref(ch)))
        return ch.value

    def film_alloc(self, w, h, channels=5):
        ptr = C.c_void_p()
        self._check(self.lib.mer_film_alloc_n(self.h, C.c_int32(w), C.c_int32(h), C.c_int32(channels), C.byref(ptr)))
        return ptr

    def film_zero(self, ptr, w, h, channels=5):
        self._check(self.lib.mer_film_zero_n(self.h, ptr, C.c_int32(w), C.c_int32(h), C.c_int32(channels)))

    def film_download(self, ptr, w, h, channels=5):
        out = np.empty((h, w, channels), np.float32)
        self._check(self.lib.mer_film_download_n(self.h, ptr, C.c_int32(w), C.c_int32(h), C.c_int32(channels), _fp(out)))
        return out

    def film_free(self, ptr):
        self._check(self.lib.mer_film_free(self.h, ptr))

    def render(self, scene, film_ptr, spp_begin, spp_count, seed=0, spp_stride=1, tile_rank=0, tile_count=1):
        """Asynchronous on the context stream.  film_ptr: c_void_p / int device pointer."""
        sh = Shard(spp_begin, spp_count, spp_stride, tile_rank, tile_count)
        fp = film_ptr if isinstance(film_ptr, C.c_void_p) else C.c_void_p(int(film_ptr))
        self._check(self.lib.mer_render(self.h, C.byref(scene), C.byref(sh), C.c_uint64(seed), fp))

    def render_to_host(self, scene, spp_begin, spp_count, seed=0, **kw):
        ch = self.film_channels(scene)
        f = self.film_alloc(scene.width, scene.height, ch)
        try:
            self.render(scene, f, spp_begin, spp_count, seed, **kw)
            return self.film_download(f, scene.width, scene.height, ch)
        finally:
            self.film_free(f)

    def synchronize(self):
        self._check(self.lib.mer_synchronize(self.h))

    def last_kernel_ms(self):
        ms = C.c_float()
        self._check(self.lib.mer_last_kernel_ms(self.h, C.byref(ms)))
        return ms.value

    def last_render_stats(self):
        n = C.c_int32(); a = C.c_float(); b = C.c_float()
        self._check(self.lib.mer_last_render_stats(self.h, C.byref(n), C.byref(a), C.byref(b)))
        return n.value, a.value, b.value

    def counters(self):
        out = np.zeros(C_COUNT, np.uint64)
        self._check(self.lib.mer_counters_read(self.h, _fp(out)))
        return out

    def counters_reset(self):
        self._check(self.lib.mer_counters_reset(self.h))

    # ---- leaf entry points ---------------------------------------------------------------------
    def lookup_trilinear(self, vol, pts):
        pts = _f32(pts); n = pts.shape[0]
        val = np.empty(n, np.float32); idx = np.empty((n, 4), np.int32)
        self._check(self.lib.mer_lookup_trilinear(self.h, C.c_int32(vol.handle), _fp(pts), C.c_int64(n), _fp(val), _fp(idx)))
        return val, idx

    def lookup_trilinear_rgb(self, vol, pts):
        pts = _f32(pts); n = pts.shape[0]
        out = np.empty((n, 3), np.float32)
        self._check(self.lib.mer_lookup_trilinear_rgb(self.h, C.c_int32(vol.handle), _fp(pts), C.c_int64(n), _fp(out)))
        return out

    def rif_value_grad(self, vol, interp, pts):
        pts = _f32(pts); n = pts.shape[0]
        val = np.empty(n, np.float32); grad = np.empty((n, 3), np.float32)
        self._check(self.lib.mer_rif_value_grad(self.h, C.c_int32(vol.handle), C.c_int32(interp), _fp(pts), C.c_int64(n), _fp(val), _fp(grad)))
        return val, grad

    def er_trace(self, scene, p0, d0, dist):
        p0 = _f32(p0); d0 = _f32(d0); dist = _f32(dist); n = p0.shape[0]
        op = np.empty((n, 3), np.float32); ov = np.empty((n, 3), np.float32)
        ds = np.empty(n, np.float32); oo = np.empty(n, np.float32); ok = np.empty(n, np.int32)
        self._check(self.lib.mer_er_trace(self.h, C.byref(scene), _fp(p0), _fp(d0), _fp(dist), C.c_int64(n),
                                       _fp(op), _fp(ov), _fp(ds), _fp(oo), _fp(ok)))
        return op, ov, ds, oo, ok

    def sample_distance(self, scene, o, d, maxt, seed):
        o = _f32(o); d = _f32(d); maxt = _f32(maxt); n = o.shape[0]
        rec = np.empty((n, 20), np.float32)
        self._check(self.lib.mer_sample_distance(self.h, C.byref(scene), _fp(o), _fp(d), _fp(maxt), C.c_int64(n), C.c_uint64(seed), _fp(rec)))
        return rec

    def connect(self, scene, p1, p2, seed):
        p1 = _f32(p1); p2 = _f32(p2); n = p1.shape[0]
        out = np.zeros((n, 12), np.float32)
        self._check(self.lib.mer_connect(self.h, C.byref(scene), _fp(p1), _fp(p2), C.c_int64(n), C.c_uint64(seed), _fp(out)))
        return out

    def eval_transmittance(self, scene, o, d, maxt, seed):
        o = _f32(o); d = _f32(d); maxt = _f32(maxt); n = o.shape[0]
        out = np.empty((n, 3), np.float32)
        self._check(self.lib.mer_eval_transmittance(self.h, C.byref(scene), _fp(o), _fp(d), _fp(maxt), C.c_int64(n), C.c_uint64(seed), _fp(out)))
        return out

    def phase_sample(self, kind, g, wi, u2):
        wi = _f32(wi); u2 = _f32(u2); n = wi.shape[0]
        wo = np.empty((n, 3), np.float32); pdf = np.empty(n, np.float32)
        self._check(self.lib.mer_phase_sample(self.h, C.c_int32(kind), C.c_float(g), _fp(wi), _fp(u2), C.c_int64(n), _fp(wo), _fp(pdf)))
        return wo, pdf

    def phase_eval(self, kind, g, wi, wo):
        wi = _f32(wi); wo = _f32(wo); n = wi.shape[0]
        val = np.empty(n, np.float32)
        self._check(self.lib.mer_phase_eval(self.h, C.c_int32(kind), C.c_float(g), _fp(wi), _fp(wo), C.c_int64(n), _fp(val)))
        return val

    def camera_rays(self, scene, pos2):
        pos2 = _f32(pos2); n = pos2.shape[0]
        o = np.empty((n, 3), np.float32); d = np.empty((n, 3), np.float32)
        self._check(self.lib.mer_camera_rays(self.h, C.byref(scene), _fp(pos2), C.c_int64(n), _fp(o), _fp(d)))
        return o, d

    def correlation(self, scene, path_length):
        t = _f32(path_length); n = t.shape[0]
        out = np.empty(n, np.float32)
        self._check(self.lib.mer_correlation(self.h, C.byref(scene), _fp(t), C.c_int64(n), _fp(out)))
        return out

    def render_paths(self, scene, sample_index, seed=0):
        out = np.zeros((scene.height, scene.width, 3), np.float32)
        self._check(self.lib.mer_render_paths(self.h, C.byref(scene), C.c_int32(sample_index), C.c_uint64(seed), _fp(out)))
        return out

    def rng_floats(self, seed, pixel, sample, n):
        out = np.empty(n, np.float32)
        self._check(self.lib.mer_rng_floats(self.h, C.c_uint64(seed), C.c_uint32(pixel), C.c_uint32(sample), C.c_int32(n), _fp(out)))
        return out


class MultiContext:
    """Several GPUs in one process (include/mer.h: mer_multi_*): one context and, during a render, one host thread per listed device;
    volumes replicated; films sum-reduced onto the first device with RCCL (distinct devices) or peer copy + add (a device listed twice)."""

    def __init__(self, device_ids, check=False, **options):
        self.lib = lib(CHECK_LIB_PATH if check else None)
        ids = (C.c_int32 * len(device_ids))(*[int(d) for d in device_ids])
        self.h = C.c_void_p()
        if self.lib.mer_multi_create(ids, C.c_int32(len(device_ids)), C.byref(self.h)) != 0:
            raise MerError(self.lib.mer_multi_last_error(None).decode())
        self.device_ids = list(device_ids)
        self.contexts = [Context(d, check=check, _borrowed=self.lib.mer_multi_context(self.h, C.c_int32(i))) for i, d in enumerate(device_ids)]
        for k, v in options.items():
            self.set_option(k, v)

    def _check(self, rc):
        if rc != 0:
            raise MerError(self.lib.mer_multi_last_error(self.h).decode())

    def close(self):
        if self.h:
            self.lib.mer_multi_destroy(self.h)
            self.h = C.c_void_p()

    def set_option(self, name, value):
        self._check(self.lib.mer_multi_set_option(self.h, name.encode(), C.c_int64(int(value))))

    def upload_volume(self, data, aabb_min, aabb_max, layout=LAYOUT_DENSE, to_world=None):
        a = np.ascontiguousarray(data)
        if a.dtype != np.uint8:
            a = a.astype(np.float32, copy=False)
        ch = 1 if a.ndim == 3 else a.shape[3]
        d = Context._desc(a.shape, ch, P.VOL_U8 if a.dtype == np.uint8 else P.VOL_F32, aabb_min, aabb_max, to_world)
        h = C.c_int32()
        self._check(self.lib.mer_multi_volume_upload(self.h, C.byref(d), _fp(a), C.c_int32(layout), C.byref(h)))
        return Volume(self, h.value, d, layout)

    def upload_scene(self, p, layout=LAYOUT_DENSE):
        """the replicated-volume form of Context.upload_scene: every device receives every grid, one handle each"""
        vols = []
        dens = alb = rif = sdf = None
        if p.sigma_mode == P.SIGMA_GRID and p.density is not None:
            dl = LAYOUT_CELL8 if layout in (LAYOUT_BRICK27, LAYOUT_BRICK125, LAYOUT_AUTO) else layout
            dens = self.upload_volume(p.density, p.density_aabb[0], p.density_aabb[1], dl if np.asarray(p.density).dtype != np.uint8 else LAYOUT_DENSE, p.density_to_world)
            vols.append(dens)
        if p.albedo_mode == P.ALBEDO_GRID and p.albedo_grid is not None:
            alb = self.upload_volume(p.albedo_grid, p.albedo_aabb[0], p.albedo_aabb[1], to_world=p.albedo_to_world); vols.append(alb)
        if p.rif_mode not in (P.RIF_CONST, P.RIF_ACOUSTIC) and p.rif is not None:
            rif = self.upload_volume(p.rif, p.rif_aabb[0], p.rif_aabb[1], layout if p.rif_mode == P.RIF_TRILINEAR else LAYOUT_DENSE, p.rif_to_world)
            if p.rif_mode == P.RIF_BSPLINE3:
                self._check(self.lib.mer_multi_volume_build_spline(self.h, C.c_int32(rif.handle)))
            vols.append(rif)
        if p.boundary == P.BOUNDARY_SDF and p.sdf is not None:
            sdf = self.upload_volume(p.sdf, p.sdf_aabb[0], p.sdf_aabb[1], LAYOUT_DENSE, p.sdf_to_world); vols.append(sdf)
        return self.contexts[0].scene_desc(p, dens, alb, rif, sdf), vols

    def destroy_volume(self, vol):
        if vol.handle:
            self._check(self.lib.mer_multi_volume_destroy(self.h, C.c_int32(vol.handle)))
            vol.handle = 0

    def render_to_host(self, scene, spp_begin, spp_count, seed=0, shard=SHARD_SAMPLES, rccl=1):
        ch = self.contexts[0].film_channels(scene)
        out = np.empty((scene.height, scene.width, ch), np.float32)
        self._check(self.lib.mer_multi_render(self.h, C.byref(scene), C.c_int32(shard), C.c_int32(spp_begin), C.c_int32(spp_count), C.c_uint64(seed),
                                              C.c_int32(rccl), _fp(out)))
        return out

    def last_stats(self):
        """-> (reduce path REDUCE_*, [render ms per context], reduce ms, counters summed over the contexts)"""
        path = C.c_int32(); ms = (C.c_float * len(self.contexts))(); red = C.c_float(); cnt = np.zeros(C_COUNT, np.uint64)
        self._check(self.lib.mer_multi_last_stats(self.h, C.byref(path), ms, C.byref(red), _fp(cnt)))
        return path.value, list(ms), red.value, cnt
