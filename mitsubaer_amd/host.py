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


def render_xml(path, defines=None, device=0, spp=0, seed=0, layout=capi.LAYOUT_CELL8):
    d, _ = flatten_xml(path, defines)
    frames = int(np.ceil((d.max_bound - d.min_bound) / d.bin_width)) if (d.decomposition and not d.modulation) else 1
    film = np.zeros((d.height, d.width, frames * 3 + 2), np.float32)
    if lib().merhost_render_xml(path.encode(), _defs(defines), C.c_int32(device), C.c_int32(spp), C.c_uint64(seed),
                                C.c_int32(layout), film.ctypes.data_as(C.c_void_p)) != 0:
        raise HostError(lib().merhost_last_error().decode())
    return film
