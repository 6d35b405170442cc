"""Multi-GPU sharding of the hot path: one process per GPU, paths are independent, the
only shared state is the film -> one RCCL all-reduce(sum) of float[H][W][5] per render.

Replaces the reference's image-block work queue + film->put under a mutex
(src/librender/renderproc.cpp:142-149) and its TCP/SSH RemoteWorker protocol (src/libcore/sched_remote.cpp).
torch.distributed is plumbing only (backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests).
"""
import os

SHARD_SAMPLES = "samples"     # rank r renders sample indices s = r (mod world): perfect balance, result independent of world
SHARD_TILES = "tiles"         # rank r renders the 32x32 image tiles (the reference's block size) that tile_owner() deals to it
TILE = 32


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init_process_group(backend=None):
    """Initialise torch.distributed from the torchrun environment (no-op for a single process)."""
    import torch.distributed as dist
    rank, world, local = env_rank_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)           # one process per GPU; "nccl" is RCCL on ROCm
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_args(mode, rank, world, spp_total):
    """-> kwargs of capi.Context.render / oracle shards for this rank.  spp_total = samples per pixel of the
    whole job.  SHARD_SAMPLES needs no divisibility: rank r takes ceil((spp_total - r) / world) samples."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    if mode == SHARD_SAMPLES:
        count = max(0, (spp_total - rank + world - 1) // world)
        return dict(spp_begin=rank, spp_count=count, spp_stride=world, tile_rank=0, tile_count=1)
    if mode == SHARD_TILES:
        return dict(spp_begin=0, spp_count=spp_total, spp_stride=1, tile_rank=rank, tile_count=world)
    raise ValueError("unknown shard mode %r" % mode)


def tile_skew(tiles_x, world):
    """Column rotation per tile row of libmer's tile deal (csrc/mer_render.hip: tile_skew_for): 0 for an unsharded film, else the smallest of
    3, 5, 7, 11, 13 coprime to the number of tile columns."""
    import math
    if world <= 1 or tiles_x <= 1:
        return 0
    for s in (3, 5, 7, 11, 13):
        if math.gcd(s % tiles_x, tiles_x) == 1:
            return s % tiles_x
    return 1 % tiles_x


def tile_owner(width, height, world):
    """-> int array [tiles_y][tiles_x]: the rank that renders each 32x32 tile in SHARD_TILES mode.  Tiles are dealt round-robin in row-major
    order after row ty has been rotated by tile_skew * ty columns, so that a rank's tiles lie on diagonals of the tile grid instead of in whole
    columns (the columns through the medium carry the work): the host-side mirror of decode_work (csrc/mer_device.hpp)."""
    import numpy as np
    tiles_x, tiles_y = (width + TILE - 1) // TILE, (height + TILE - 1) // TILE
    sk = tile_skew(tiles_x, world)
    ty, tx = np.meshgrid(np.arange(tiles_y), np.arange(tiles_x), indexing="ij")
    tx0 = (tx - sk * ty) % tiles_x
    return (ty * tiles_x + tx0) % world


def reduce_film(film_tensor):
    """Sum-reduce the film across ranks in place (untouched pixels are 0, filter borders overlap-add)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(film_tensor, op=dist.ReduceOp.SUM)
    return film_tensor


def reduce_counters(counters_tensor):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(counters_tensor, op=dist.ReduceOp.SUM)
    return counters_tensor
