"""VOL v3 grid files (reference format: src/volume/gridvolume.cpp:54-89,217-287; writer mfiles/writeGridToVol.m).

48-byte little-endian header: 'V','O','L',3, int32 type (1=float32, 3=uint8), int32 xres,yres,zres,
int32 channels, 6 x float32 AABB (min xyz, max xyz); payload at byte 48, x fastest:
data[((z*yres+y)*xres+x)*channels+c].  Arrays here are indexed [z][y][x] (or [z][y][x][c]).
"""
import struct
import numpy as np

VOL_F32, VOL_F16, VOL_U8, VOL_QDIR = 1, 2, 3, 4


def write_vol(path, data, aabb_min, aabb_max):
    a = np.ascontiguousarray(data)
    if a.dtype == np.uint8:
        typ = VOL_U8
    else:
        a = a.astype("<f4")
        typ = VOL_F32
    ch = 1 if a.ndim == 3 else a.shape[3]
    zres, yres, xres = a.shape[:3]
    with open(path, "wb") as f:
        f.write(b"VOL" + bytes([3]))
        f.write(struct.pack("<i", typ))
        f.write(struct.pack("<iii", xres, yres, zres))
        f.write(struct.pack("<i", ch))
        f.write(struct.pack("<6f", *[float(v) for v in aabb_min], *[float(v) for v in aabb_max]))
        f.write(a.tobytes())


def read_vol(path, mmap=True):
    """Returns (data[z][y][x](,c), aabb_min, aabb_max).  Errors mirror GridDataSource::loadFromFile."""
    with open(path, "rb") as f:
        hdr = f.read(48)
    if len(hdr) < 48 or hdr[0:3] != b"VOL":
        raise RuntimeError("Encountered an invalid volume data file (incorrect header identifier)")
    if hdr[3] != 3:
        raise RuntimeError("Encountered an invalid volume data file (incorrect file version)")
    typ, xres, yres, zres, ch = struct.unpack("<5i", hdr[4:24])
    bb = struct.unpack("<6f", hdr[24:48])
    if typ == VOL_F32:
        dt = np.dtype("<f4")
    elif typ == VOL_U8:
        dt = np.dtype("u1")
    elif typ == VOL_F16:
        raise RuntimeError("Error: float16 volumes are not yet supported!")
    else:
        raise RuntimeError("Encountered a volume data file of unknown type (type=%i, channels=%i)!" % (typ, ch))
    if ch not in (1, 3):
        raise RuntimeError("Encountered an unsupported volume data file (%i channels, only 1 and 3 are supported)" % ch)
    shape = (zres, yres, xres) if ch == 1 else (zres, yres, xres, ch)
    if mmap:
        data = np.memmap(path, dtype=dt, mode="r", offset=48, shape=shape)
    else:
        data = np.fromfile(path, dtype=dt, offset=48, count=int(np.prod(shape))).reshape(shape)
    return data, np.array(bb[:3], np.float32), np.array(bb[3:], np.float32)
