"""Multi-GPU sharding of the hot path: one process per GPU, paths are independent, the
only shared state is the film -> one RCCL all-reduce(sum) of float[H][W][5] per render.

Replaces the reference's image-block work queue + film->put under a mutex
(src/librender/renderproc.cpp:142-149) and its TCP/SSH RemoteWorker protocol (src/libcore/sched_remote.cpp).
torch.distributed is plumbing only (backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests).
"""
import os

SHARD_SAMPLES = "samples"     # rank r renders sample indices s = r (mod world): perfect balance, result independent of world
SHARD_TILES = "tiles"         # rank r renders 32x32 image tiles t = r (mod world) (the reference's block size)


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
