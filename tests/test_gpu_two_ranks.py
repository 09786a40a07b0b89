"""The N > 1 path on the product: two ranks (torch.distributed.run, gloo) each drive api.Context on their shard of the streams — both on
the one GPU of the test box — and bench.py runs under the same launcher.  The rank processes are started by the session's helper process
(conftest.run_program), which has never touched a GPU."""
import json
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENV = {"HSA_ENABLE_IPC_MODE_LEGACY": "0", "OMP_NUM_THREADS": "2"}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _torchrun(nproc, script, *args):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
            "--master-port", str(_free_port()), script, *[str(a) for a in args]]


def test_two_ranks_drive_the_product_on_their_shards(run_program, tmp_path):
    from cpp_audio_amd import api, signals
    S, n = 5, 30000
    rc, so, se = run_program(_torchrun(2, os.path.join(ROOT, "tests", "host", "two_rank_worker.py"), tmp_path, S, n), timeout=900, env=ENV)
    assert rc == 0, so[-2000:] + se[-4000:]
    parts, ranges = [], []
    for r in range(2):
        parts.append(np.load(tmp_path / f"rank{r}.npy"))
        ranges.append(np.load(tmp_path / f"range{r}.npy").tolist())
    assert ranges[0][0] == 0 and ranges[0][1] == ranges[1][0] and ranges[1][1] == S   # the shards tile [0, S)
    got = np.concatenate(parts)
    x = signals.make_streams(S, n)
    with api.Context(api.default_config(analysis_volume=1.0, stereo_spread=0.0, carrier_volume=0.0, seed=1234), S, 0) as ctx:
        ref = ctx.process_host(x)
    assert np.sqrt(np.mean(ref.astype(np.float64) ** 2)) > 1e-3
    assert np.array_equal(got, ref), "a stream's result does not depend on which rank ran it (per-stream seeds = seed + global index)"


@pytest.mark.parametrize("extra", [["--streams", "8"], ["--streams-total", "11"], ["--streams-total", "2000"]])   # 2000 = configs[4]'s 1000 streams per rank
def test_bench_runs_with_two_ranks(run_program, extra):
    rc, so, se = run_program(_torchrun(2, os.path.join(ROOT, "bench.py"), "--gpus", 2, "--steps", 2, "--warmup", 1, "--seconds", 1, *extra),
                             timeout=900, env=ENV, cwd=ROOT)
    assert rc == 0, so[-2000:] + se[-4000:]
    line = [l for l in so.splitlines() if l.startswith('{"metric"')]
    assert len(line) == 1, so[-2000:]
    d = json.loads(line[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["checks"]["error_flags"] == 0
    total = 16 if extra[0] == "--streams" else int(extra[1])
    assert d["config"]["streams_total"] == total
    assert abs(d["value"] - total * 48000 * d["steps"] / (d["ms_per_step"] * d["steps"] * 1e-3)) < 1e-6 * d["value"]
    assert "cpu_baseline" not in d and "other_configs" not in d   # N = 1 only
