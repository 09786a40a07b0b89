"""Stream sharding over the GPUs of one node: every stream is an independent unit with
private state (one RtResynth each, rt.resynth.lib.cpp:907-910,1010), so ranks own contiguous blocks of
ceil(streams / world) streams and exchange nothing on the data path — no collective, no RCCL."""
import copy


def shard_streams(n_total, world, rank):
    """Contiguous block [lo, hi) of rank `rank`; the blocks tile [0, n_total) exactly."""
    per = -(-n_total // world)
    lo = min(n_total, rank * per)
    hi = min(n_total, lo + per)
    return lo, hi


def shard_config(cfg, lo):
    """Per-stream policy seeds are cfg.seed + global stream index: a shard starting at stream `lo` gets seed + lo."""
    c = copy.copy(cfg)
    c.seed = int(cfg.seed) + int(lo)
    return c


def max_over_ranks(value, dist=None, device=None):
    """Slowest rank's wall time (the bench contract's timing rule); identity for a single process."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
