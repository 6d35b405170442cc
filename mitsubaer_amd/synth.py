"""Deterministic synthetic fields of BASELINE.json's configs.

No RNG-library dependence: the noise term is the lowbias32 integer hash.  Arrays are [z][y][x].
"""
import numpy as np


def lowbias32(x):
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16); x *= np.uint32(0x7feb352d)
    x ^= x >> np.uint32(15); x *= np.uint32(0x846ca68b)
    x ^= x >> np.uint32(16)
    return x


def density_field(N):
    """rho = clamp(0.5 + 0.35 sin(3 pi x) sin(3 pi y) sin(3 pi z) + 0.15 (hash32((i+N(j+Nk)) ^ 0x5EED)/2^32 - 0.5), 0, 1)
    at node centres x = -1 + 2 i/(N-1)."""
    ax = (-1.0 + 2.0 * np.arange(N, dtype=np.float64) / (N - 1))
    s = np.sin(3.0 * np.pi * ax)
    out = np.empty((N, N, N), np.float32)
    ii = np.arange(N, dtype=np.uint32)[None, :]
    jj = np.arange(N, dtype=np.uint32)[:, None]
    with np.errstate(over="ignore"):
        for k in range(N):
            lin = (ii + np.uint32(N) * (jj + np.uint32(N) * np.uint32(k))) ^ np.uint32(0x5EED)
            h = lowbias32(lin).astype(np.float64) / 4294967296.0
            v = 0.5 + 0.35 * (s[None, :] * s[:, None] * s[k]) + 0.15 * (h - 0.5)
            out[k] = np.clip(v, 0.0, 1.0).astype(np.float32)
    return out


def linear_rif(N, nmin=1.3, nmax=1.6, shape=None):
    """mfiles/createLinearRIFWithBox.m:6-20: n = nmin + (nmax-nmin)/(Ny-1)*j along y."""
    nz, ny, nx = shape if shape else (N, N, N)
    col = (nmin + (nmax - nmin) / (ny - 1) * np.arange(ny, dtype=np.float64)).astype(np.float32)
    return np.ascontiguousarray(np.broadcast_to(col[None, :, None], (nz, ny, nx)))


def radial_rif(N, aabb_min=(-1, -1, -1), aabb_max=(1, 1, 1), shape=None):
    """mfiles/createRadialRIFWithBox.m:15-23: n = 2 - (r/R)^2, R = half diagonal of the box."""
    nz, ny, nx = shape if shape else (N, N, N)
    mn = np.asarray(aabb_min, np.float64); mx = np.asarray(aabb_max, np.float64)
    c = (mx + mn) / 2
    R = max(np.linalg.norm(mx - c), np.linalg.norm(mn - c))
    x = mn[0] + (mx[0] - mn[0]) * np.arange(nx) / (nx - 1) - c[0]
    y = mn[1] + (mx[1] - mn[1]) * np.arange(ny) / (ny - 1) - c[1]
    z = mn[2] + (mx[2] - mn[2]) * np.arange(nz) / (nz - 1) - c[2]
    out = np.empty((nz, ny, nx), np.float32)
    xy = x[None, :] ** 2 + y[:, None] ** 2
    for k in range(nz):
        out[k] = (2.0 - (xy + z[k] ** 2) / (R * R)).astype(np.float32)
    return out


def rif_from_sdf(sdf, nmin=1.10, nmax=1.50, r=1.0, flip=False):
    """mfiles/createRIFFromSD.m:12-38: d = max(+-sdf, 0) (depth below the surface; `flip` for grids that store the
    inside as negative), h = max(d), n = nmin + (nmax - nmin) / h^r * d^r -- the index rises from nmin at the surface to nmax at
    the deepest voxel; r = 1 is linear, the script writes r in (1, 1.5, 2, 3, 10)."""
    d = np.asarray(sdf, np.float64)
    if flip:
        d = -d
    d = np.maximum(d, 0.0)
    h = d.max()
    if not h > 0:
        raise ValueError("the signed distance grid has no interior (max depth is 0)")
    k = (nmax - nmin) / h ** r
    return (nmin + k * d ** r).astype(np.float32)


def sphere_sdf(N, radius=0.75, aabb_min=(-1, -1, -1), aabb_max=(1, 1, 1)):
    """signed distance to a centred sphere sampled on an N^3 grid, positive inside (the convention createRIFFromSD.m ends up with)"""
    ax = [np.linspace(aabb_min[i], aabb_max[i], N) for i in range(3)]
    z, y, x = np.meshgrid(ax[2], ax[1], ax[0], indexing="ij")
    return (radius - np.sqrt(x * x + y * y + z * z)).astype(np.float32)


def acoustic_rif(N, n0=1.33, nmax=1e-3, mode=0, kr=None, aabb_min=(-1, -1, -1), aabb_max=(1, 1, 1), axis=2):
    """Sampled form of the reference's analytic ultrasound field (src/volume/acousticrifvolume.cpp:101-105,224-342): a standing
    Bessel mode in a cylinder, n = n0 + nmax * J_m(k_r r) * cos(m phi), r and phi in the plane normal to `axis`.  k_r defaults to
    the first zero of J_m over the half width of the box (a node on the cylinder wall)."""
    from scipy import special
    ax = [np.linspace(aabb_min[i], aabb_max[i], N) for i in range(3)]
    z, y, x = np.meshgrid(ax[2], ax[1], ax[0], indexing="ij")
    u, v = [(y, z), (x, z), (x, y)][axis]
    r = np.sqrt(u * u + v * v); phi = np.arctan2(v, u)
    if kr is None:
        half = 0.5 * min(aabb_max[i] - aabb_min[i] for i in range(3) if i != axis)
        kr = special.jn_zeros(mode, 1)[0] / half
    return (n0 + nmax * special.jv(mode, kr * r) * np.cos(mode * phi)).astype(np.float32)
