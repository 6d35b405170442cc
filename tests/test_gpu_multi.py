"""Several GPUs behind the C-ABI in one process (mer_multi_*, include/mer.h): one context + one host thread per listed device, replicated
volumes, films sum-reduced onto the first device.  A one-GPU box lists its device twice -- {0, 0} runs every line of the host code
(threads, shards, per-context films, peer copy + add kernel, counters) except the RCCL call itself, which refuses duplicate devices;
the RCCL binding (dlopen of librccl.so, ncclCommInitAll, grouped ncclReduce) is exercised with a one-rank communicator (rccl = 2), and
with distinct devices where the box has them.  Replaces the N workers + film->put of src/librender/renderproc.cpp:142-149."""
import numpy as np
import pytest
from mitsubaer_amd import params as P, capi
from tests import scenes

pytestmark = pytest.mark.gpu


def _ndev():
    import torch
    return torch.cuda.device_count()


@pytest.mark.parametrize("shard", [capi.SHARD_SAMPLES, capi.SHARD_TILES])
@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0]])
def test_same_device_listed_twice_equals_one_context(ctx, devices, shard):
    p = scenes.curved_scene(N=24, w=70, h=45)           # partial edge tiles
    sc, vols = ctx.upload_scene(p)
    ref = ctx.render_to_host(sc, 0, 6, seed=2)
    m = capi.MultiContext(devices)
    try:
        msc, mv = m.upload_scene(p)
        film = m.render_to_host(msc, 0, 6, seed=2, shard=shard)
        assert np.allclose(film, ref, rtol=1e-4, atol=1e-5)                          # float summation order only
        path, ms, red, cnt = m.last_stats()
        assert path == capi.REDUCE_PEER_COPY and len(ms) == len(devices) and all(t > 0 for t in ms)
        assert cnt[capi.C_PATHS] == p.width * p.height * 6
        again = m.render_to_host(msc, 0, 6, seed=2, shard=shard)                     # same paths; float atomics inside a context sum in any order
        assert np.allclose(again, film, rtol=1e-5, atol=1e-6)
        for v in mv:
            v.destroy()
    finally:
        m.close()
    for v in vols:
        v.destroy()


def test_transient_film_and_spline_volume_through_multi(ctx):
    p = scenes.bspline_scene(N=24, w=40, h=33, decomposition=P.DECOMPOSITION_TRANSIENT, min_bound=0.0, max_bound=12.0, bin_width=1.5)
    sc, vols = ctx.upload_scene(p)
    ref = ctx.render_to_host(sc, 0, 4, seed=3)
    m = capi.MultiContext([0, 0])
    try:
        msc, mv = m.upload_scene(p)
        film = m.render_to_host(msc, 0, 4, seed=3, shard=capi.SHARD_TILES)
        assert film.shape == ref.shape and film.shape[2] == 8 * 3 + 2
        assert np.allclose(film, ref, rtol=1e-4, atol=1e-5)
    finally:
        m.close()
    for v in vols:
        v.destroy()


def test_rccl_binding_with_a_one_rank_communicator(ctx):
    """rccl = 2: the reduction goes through librccl.so even for a single context -- dlopen, ncclCommInitAll({0}), ncclGroupStart, ncclReduce
    (float, sum, root 0, on the context's stream), ncclGroupEnd -- and must leave the film untouched."""
    p = scenes.straight_scene(N=16, w=64, h=40)
    sc, vols = ctx.upload_scene(p)
    ref = ctx.render_to_host(sc, 0, 3, seed=8)
    m = capi.MultiContext([0])
    try:
        msc, mv = m.upload_scene(p)
        film = m.render_to_host(msc, 0, 3, seed=8, rccl=2)
        path, _, _, _ = m.last_stats()
        assert path == capi.REDUCE_RCCL
        assert np.allclose(film, ref, rtol=1e-5, atol=1e-6)
        film1 = m.render_to_host(msc, 0, 3, seed=8)                                  # default: one context, nothing to reduce
        assert m.last_stats()[0] == capi.REDUCE_NONE and np.allclose(film1, ref, rtol=1e-5, atol=1e-6)
    finally:
        m.close()
    for v in vols:
        v.destroy()


def test_errors_are_loud():
    with pytest.raises(capi.MerError):
        capi.MultiContext([])
    with pytest.raises(capi.MerError, match="device"):
        capi.MultiContext([0, 4096])
    m = capi.MultiContext([0, 0])
    try:
        p = scenes.straight_scene(N=16, w=32, h=32)
        msc, mv = m.upload_scene(p)
        with pytest.raises(capi.MerError, match="listed twice"):
            m.render_to_host(msc, 0, 2, rccl=2)                                      # RCCL refuses duplicate devices: said, not papered over
        with pytest.raises(capi.MerError, match="shard mode"):
            m.render_to_host(msc, 0, 2, shard=7)
        with pytest.raises(capi.MerError):
            m.set_option("no_such_option", 1)
    finally:
        m.close()


@pytest.mark.skipif(_ndev() < 2, reason="needs two GPUs (the driver's 8-GPU node; the RCCL reduce over xGMI)")
@pytest.mark.parametrize("shard", [capi.SHARD_SAMPLES, capi.SHARD_TILES])
def test_distinct_devices_reduce_with_rccl(ctx, shard):
    n = min(_ndev(), 8)
    p = scenes.curved_scene(N=24, w=70, h=45)
    sc, vols = ctx.upload_scene(p)
    ref = ctx.render_to_host(sc, 0, 8, seed=2)
    m = capi.MultiContext(list(range(n)))
    try:
        msc, mv = m.upload_scene(p)
        film = m.render_to_host(msc, 0, 8, seed=2, shard=shard)
        assert m.last_stats()[0] == capi.REDUCE_RCCL
        assert np.allclose(film, ref, rtol=1e-4, atol=1e-5)
        film2 = m.render_to_host(msc, 0, 8, seed=2, shard=shard, rccl=0)             # the peer-copy path between real devices
        assert m.last_stats()[0] == capi.REDUCE_PEER_COPY
        assert np.allclose(film2, ref, rtol=1e-4, atol=1e-5)
    finally:
        m.close()
    for v in vols:
        v.destroy()
