"""ctypes binding of libmer_host.so: the C++ host mirror of the reference's plugin interface (scene-XML subset ->
plugin objects -> flat scene -> C-ABI).  Parsing / validation needs no GPU; rendering does."""
import ctypes as C
import os
import numpy as np
from . import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class HostError(RuntimeError):
    pass


def lib():
    global _LIB
    if _LIB is None:
        capi.lib()                                           # libmer.so first (same directory, rpath $ORIGIN)
        path = os.path.join(_HERE, "libmer_host.so")
        if not os.path.exists(path):
            raise HostError("libmer_host.so is not built; run __graft_entry__.build()")
        _LIB = C.CDLL(path)
        _LIB.merhost_last_error.restype = C.c_char_p
    return _LIB


def _defs(defines):
    return ";".join("%s=%s" % (k, v) for k, v in (defines or {}).items()).encode()


def flatten_xml(path, defines=None):
    """Parse + validate a scene file; returns (capi.SceneDesc with zero volume handles, sampleCount)."""
    d = capi.SceneDesc(); spp = C.c_int32()
    if lib().merhost_flatten_xml(path.encode(), _defs(defines), C.byref(d), C.byref(spp)) != 0:
        raise HostError(lib().merhost_last_error().decode())
    return d, spp.value


def render_xml(path, defines=None, device=0, spp=0, seed=0, layout=capi.LAYOUT_AUTO, devices=None, shard=capi.SHARD_SAMPLES):
    """devices: a list of GPU indices renders on all of them (mer_multi_*: shard = capi.SHARD_SAMPLES | SHARD_TILES); None = `device` alone"""
    d, _ = flatten_xml(path, defines)
    frames = int(np.ceil((d.max_bound - d.min_bound) / d.bin_width)) if (d.decomposition and not d.modulation) else 1
    film = np.zeros((d.height, d.width, frames * 3 + 2), np.float32)
    if devices is None:
        rc = lib().merhost_render_xml(path.encode(), _defs(defines), C.c_int32(device), C.c_int32(spp), C.c_uint64(seed),
                                      C.c_int32(layout), film.ctypes.data_as(C.c_void_p))
    else:
        ids = (C.c_int32 * len(devices))(*devices)
        rc = lib().merhost_render_xml_multi(path.encode(), _defs(defines), ids, C.c_int32(len(devices)), C.c_int32(shard), C.c_int32(spp), C.c_uint64(seed),
                                            C.c_int32(layout), film.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise HostError(lib().merhost_last_error().decode())
    return film


def write_exr(path, rgb):
    """rgb: float32 [h][w][3] -> uncompressed scan-line OpenEXR (merhost::writeExr)"""
    a = np.ascontiguousarray(rgb, np.float32)
    if lib().merhost_write_exr(path.encode(), a.ctypes.data_as(C.c_void_p), C.c_int32(a.shape[0]), C.c_int32(a.shape[1])) != 0:
        raise HostError(lib().merhost_last_error().decode())


def read_exr_uncompressed(path):
    """Minimal reader for the subset write_exr produces (OpenEXR 2 file layout: magic, version, attributes, offset table,
    scan lines); returns ({attribute name: (type, bytes)}, float32 [h][w][3] RGB)."""
    import struct
    b = open(path, "rb").read()
    magic, version = struct.unpack_from("<II", b, 0)
    if magic != 20000630 or (version & 0xFF) != 2:
        raise ValueError("not an OpenEXR 2 file")
    pos = 8; attrs = {}
    while b[pos] != 0:
        e = b.index(b"\0", pos); name = b[pos:e].decode(); pos = e + 1
        e = b.index(b"\0", pos); typ = b[pos:e].decode(); pos = e + 1
        size, = struct.unpack_from("<I", b, pos); pos += 4
        attrs[name] = (typ, b[pos:pos + size]); pos += size
    pos += 1
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    if attrs["compression"][1] != b"\0":
        raise ValueError("compressed EXR")
    names = []; c = attrs["channels"][1]; q = 0
    while c[q] != 0:
        e = c.index(b"\0", q); names.append(c[q:e].decode()); q = e + 1 + 16
    offsets = struct.unpack_from("<%dQ" % h, b, pos)
    img = np.empty((h, w, 3), np.float32)
    for y in range(h):
        yy, size = struct.unpack_from("<iI", b, offsets[y])
        rows = np.frombuffer(b, "<f4", len(names) * w, offsets[y] + 8).reshape(len(names), w)
        for k, n in enumerate(names):
            img[yy - y0, :, "RGB".index(n)] = rows[k]
    return attrs, img
