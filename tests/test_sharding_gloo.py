"""N > 1 path on CPU: two gloo ranks shard a stream batch with the product's sharding helpers, run their shard,
and the gathered result must equal the single-process result stream for stream.  The per-rank compute here is the
CPU oracle (tests may use it); on the GPU box the same helpers drive api.Context (tests/test_gpu_parity.py::
test_streams_are_independent checks the GPU side of the same property)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KW = dict(analysis_volume=1.0, stereo_spread=0.0, carrier_volume=0.0, seed=1234)
N_STREAMS, N_FRAMES = 5, 12000


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib
    from cpp_audio_amd import sharding, signals
    x = signals.make_streams(N_STREAMS, N_FRAMES)
    lo, hi = sharding.shard_streams(N_STREAMS, world, rank)
    cfg = sharding.shard_config(oracle_lib.default_config(**KW), lo)   # seed + first global stream index
    local = oracle_lib.run_offline_batch(cfg, x[lo:hi]) if hi > lo else np.zeros((0, N_FRAMES, 2), np.float32)
    per = -(-N_STREAMS // world)
    pad = np.zeros((per, N_FRAMES, 2), np.float32)
    pad[: hi - lo] = local
    bufs = [torch.zeros(per, N_FRAMES, 2) for _ in range(world)]
    dist.all_gather(bufs, torch.from_numpy(pad))          # test-only gather; the data path itself has no collective
    # the bench contract's timing rule: slowest rank
    t = sharding.max_over_ranks(0.5 + rank, dist)
    counts = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([lo, hi]))
    if rank == 0:
        q.put((torch.cat(bufs)[:N_STREAMS].numpy(), t, [c.tolist() for c in counts]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_two_rank_sharding_matches_single_process(oracle, world):
    from cpp_audio_amd import sharding, signals
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, t, ranges = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    x = signals.make_streams(N_STREAMS, N_FRAMES)
    ref = oracle.run_offline_batch(oracle.default_config(**KW), x)
    assert np.array_equal(got, ref)
    assert t == 0.5 + (world - 1)
    # shards tile [0, N) exactly once
    assert ranges[0][0] == 0 and ranges[-1][1] == N_STREAMS and all(a[1] == b[0] for a, b in zip(ranges[:-1], ranges[1:]))


def test_shard_partition_properties():
    from cpp_audio_amd import sharding
    for n in (1, 7, 8, 256, 8000):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                lo, hi = sharding.shard_streams(n, world, r)
                assert 0 <= lo <= hi <= n
                cover += list(range(lo, hi))
            assert cover == list(range(n))
