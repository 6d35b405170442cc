"""Host-side logic: VOL v3 files, synthetic fields, scene parameters, sharding arithmetic (CPU only)."""
import os
import struct
import numpy as np
import pytest
from mitsubaer_amd import volio, synth, params as P, dist as mdist


def test_vol_roundtrip_reference_recipe(tmp_path):
    """mfiles/Test.m:6-16: 20x30x23 grid on [0,1]^3 written by writeGridToVol, read back."""
    rng = np.random.RandomState(1)
    data = rng.rand(23, 30, 20).astype(np.float32)
    f = str(tmp_path / "testVol.vol")
    volio.write_vol(f, data, [0, 0, 0], [1, 1, 1])
    raw = open(f, "rb").read()
    assert raw[:4] == b"VOL\x03" and len(raw) == 48 + data.size * 4
    assert struct.unpack("<5i", raw[4:24]) == (1, 20, 30, 23, 1)           # type, xres, yres, zres, channels
    assert struct.unpack("<f", raw[48:52])[0] == data[0, 0, 0]               # payload at byte 48, x fastest
    assert struct.unpack("<f", raw[52:56])[0] == data[0, 0, 1]
    d2, mn, mx = volio.read_vol(f)
    assert np.array_equal(np.asarray(d2), data) and np.array_equal(mn, [0, 0, 0]) and np.array_equal(mx, [1, 1, 1])


def test_vol_reference_box_shape_and_u8_rgb(tmp_path):
    """The MATLAB generators' own grid: 226x226x51 on [-225,225]^2 x [25,125] (createLinearRIFWithBox.m)."""
    rif = synth.linear_rif(0, shape=(51, 226, 226))
    f = str(tmp_path / "BoxRIF.vol")
    volio.write_vol(f, rif, [-225, -225, 25], [225, 225, 125])
    d, mn, mx = volio.read_vol(f)
    assert d.shape == (51, 226, 226) and d[0, 0, 0] == np.float32(1.3) and abs(d[0, 225, 0] - 1.6) < 1e-6
    rgb = (np.random.RandomState(0).rand(4, 5, 6, 3) * 255).astype(np.uint8)
    f2 = str(tmp_path / "rgb.vol")
    volio.write_vol(f2, rgb, [0] * 3, [1] * 3)
    d, _, _ = volio.read_vol(f2, mmap=False)
    assert d.dtype == np.uint8 and np.array_equal(d, rgb)


def test_vol_errors_mirror_reference(tmp_path):
    f = str(tmp_path / "bad.vol")
    open(f, "wb").write(b"VOX\x03" + b"\0" * 44)
    with pytest.raises(RuntimeError, match="incorrect header identifier"):
        volio.read_vol(f)
    open(f, "wb").write(b"VOL\x02" + b"\0" * 44)
    with pytest.raises(RuntimeError, match="incorrect file version"):
        volio.read_vol(f)
    open(f, "wb").write(b"VOL\x03" + struct.pack("<5i", 2, 2, 2, 2, 1) + b"\0" * 24)
    with pytest.raises(RuntimeError, match="float16"):
        volio.read_vol(f)
    open(f, "wb").write(b"VOL\x03" + struct.pack("<5i", 1, 2, 2, 2, 2) + b"\0" * 24 + b"\0" * 64)
    with pytest.raises(RuntimeError, match="channels"):
        volio.read_vol(f)


def test_synthetic_fields_are_deterministic_and_in_range():
    d = synth.density_field(16)
    assert d.dtype == np.float32 and d.min() >= 0 and d.max() <= 1 and np.array_equal(d, synth.density_field(16))
    # hash known answers (lowbias32)
    assert synth.lowbias32(np.array([0, 1, 0x5EED, 12345], np.uint32)).tolist() == [0, 1753845952, 611984374, 2435775735]
    lin = synth.linear_rif(9)
    assert lin[0, 0, 0] == np.float32(1.3) and abs(lin[0, 8, 0] - 1.6) < 1e-6 and np.all(lin[:, 4, :] == lin[0, 4, 0])
    rad = synth.radial_rif(9)
    assert abs(rad[4, 4, 4] - 2.0) < 1e-6 and abs(rad[0, 0, 0] - 1.0) < 1e-6       # n = 2 - (r/R)^2


def test_look_at_matches_reference_convention():
    m = P.look_at([-3, 0, 0], [-2, 0, 0], [0, 1, 0])       # scenes/volumetric/...xml:31
    assert np.allclose(m[:, 2], [1, 0, 0]) and np.allclose(m[:, 3], [-3, 0, 0])
    assert np.allclose(m[:, 0], np.cross([0, 1, 0], [1, 0, 0]))           # left = up x dir (left-handed)
    assert np.allclose(m[:, 1], [0, 1, 0])
    with pytest.raises(AttributeError):
        P.SceneParams(no_such_parameter=1)


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_shard_arithmetic_partitions_the_job(world):
    spp = 37
    seen = []
    for r in range(world):
        a = mdist.shard_args(mdist.SHARD_SAMPLES, r, world, spp)
        seen += [a["spp_begin"] + k * a["spp_stride"] for k in range(a["spp_count"])]
    assert sorted(seen) == list(range(spp))
    tiles = []
    for r in range(world):
        a = mdist.shard_args(mdist.SHARD_TILES, r, world, spp)
        assert a["spp_count"] == spp
        tiles += [t for t in range(100) if t % a["tile_count"] == a["tile_rank"]]
    assert sorted(tiles) == list(range(100))
    with pytest.raises(ValueError):
        mdist.shard_args(mdist.SHARD_SAMPLES, world, world, spp)


def test_rif_from_sdf_follows_the_reference_recipe():
    """mfiles/createRIFFromSD.m: n = nmin at and outside the surface, nmax at the deepest voxel, power law in between"""
    from mitsubaer_amd import synth
    sdf = synth.sphere_sdf(33, radius=0.75)
    for r in (1.0, 2.0, 10.0):
        n = synth.rif_from_sdf(sdf, nmin=1.10, nmax=1.50, r=r)
        assert n.dtype == np.float32 and n.shape == sdf.shape
        assert abs(n.max() - 1.50) < 1e-6 and abs(n.min() - 1.10) < 1e-6
        assert np.all(n[sdf <= 0] == np.float32(1.10))
        mid = sdf == sdf.max()
        assert np.allclose(n[mid], 1.50)
        d = np.maximum(sdf.astype(np.float64), 0); h = d.max()
        np.testing.assert_allclose(n, 1.10 + 0.40 * (d / h) ** r, atol=2e-7)
    flipped = synth.rif_from_sdf(-sdf, flip=True)
    np.testing.assert_array_equal(flipped, synth.rif_from_sdf(sdf))
    with pytest.raises(ValueError):
        synth.rif_from_sdf(-np.abs(sdf))


def test_exr_writer_round_trip(tmp_path):
    """the host's OpenEXR writer (uncompressed scan lines, float32 B,G,R): header attributes and pixels read back"""
    from mitsubaer_amd import host
    rng = np.random.RandomState(2)
    img = rng.rand(7, 11, 3).astype(np.float32) * 10
    f = str(tmp_path / "a.exr")
    host.write_exr(f, img)
    attrs, back = host.read_exr_uncompressed(f)
    assert np.array_equal(back, img)
    assert attrs["channels"][0] == "chlist" and attrs["compression"] == ("compression", b"\0") and attrs["lineOrder"][1] == b"\0"
    assert struct.unpack("<4i", attrs["dataWindow"][1]) == (0, 0, 10, 6) == struct.unpack("<4i", attrs["displayWindow"][1])
    assert os.path.getsize(f) > 7 * 11 * 12


def test_acoustic_rif_is_a_bessel_mode():
    from scipy import special
    n = synth.acoustic_rif(33, n0=1.33, nmax=2e-3, mode=0)
    assert n.dtype == np.float32 and n.shape == (33, 33, 33)
    assert abs(n[:, 16, 16].max() - (1.33 + 2e-3)) < 1e-6                      # J_0(0) = 1 on the axis
    assert np.allclose(n[5], n[20])                                             # constant along the cylinder axis
    assert abs(n[0, 16, 32] - 1.33) < 2e-6                                      # node of J_0 on the wall (k_r = j_01 / half width)
    m2 = synth.acoustic_rif(33, mode=2, nmax=1e-3)
    assert abs(m2[0, 16, 16] - 1.33) < 1e-7 and m2.max() - 1.33 < 1e-3 * special.jv(2, special.jnp_zeros(2, 1)[0]) + 1e-6
