"""Multi-process path on CPU: world_size 2 over gloo.  Each rank renders its shard of the job with the CPU
oracle (test infrastructure), the film is sum-reduced with torch.distributed exactly as bench.py does with
RCCL, and the result must equal the unsharded render."""
import os
import socket
import sys
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, mode, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from mitsubaer_amd import dist as mdist
    from oracle import orc
    from tests import scenes
    r, w, _ = mdist.init_process_group("gloo")
    assert (r, w) == (rank, world)
    p = scenes.curved_scene(N=16, w=40, h=36)
    sh = mdist.shard_args(mode, rank, world, 4)
    film = np.zeros((p.height, p.width, 5), np.float32)
    if mode == mdist.SHARD_SAMPLES:
        for k in range(sh["spp_count"]):
            f, _ = orc.render(p, sh["spp_begin"] + k * sh["spp_stride"], 1, 7, nthreads=2)
            film += f
    else:   # tiles: the oracle shards by rows; emulate 32x32 tile ownership by masking the per-pixel box-filtered film
        p = p.copy(rfilter=1 - 1, rfilter_param=0.5)       # box filter: splats stay inside their pixel
        f, _ = orc.render(p, 0, 4, 7, nthreads=2)
        owner = np.kron(mdist.tile_owner(p.width, p.height, world), np.ones((32, 32), int))[:p.height, :p.width]   # per pixel
        film = f * (owner == rank)[..., None]
    t = torch.from_numpy(film)
    mdist.reduce_film(t)
    c = torch.tensor([float(rank + 1)] * 4, dtype=torch.float64)
    mdist.reduce_counters(c)
    assert c[0].item() == world * (world + 1) / 2
    if rank == 0:
        np.save(os.path.join(out_dir, "film_%s.npy" % mode), t.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["samples", "tiles"])
def test_world_size_2_film_reduce_equals_unsharded(tmp_path, orc, mode):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, mode, str(tmp_path)), nprocs=world, join=True)
    got = np.load(str(tmp_path / ("film_%s.npy" % mode)))
    from tests import scenes
    p = scenes.curved_scene(N=16, w=40, h=36)
    if mode == "tiles":
        p = p.copy(rfilter=0, rfilter_param=0.5)
    ref, _ = orc.render(p, 0, 4, 7, nthreads=2)
    assert np.allclose(got, ref, rtol=1e-5, atol=1e-6)
