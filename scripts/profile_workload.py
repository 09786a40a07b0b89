#!/usr/bin/env python3
"""One named workload of the bench line (bench.workloads()) as `steps` equal blocks and nothing else, for rocprofv3:
    rocprofv3 --kernel-trace --stats -- python3 scripts/profile_workload.py <name> <steps>
    rocprofv3 --kernel-trace --pmc <counters> -- python3 scripts/profile_workload.py <name> <steps>
Every ra_* launch of the run belongs to one of the `steps` blocks (no set-up block), so a kernel's per-step figures are those of its last
(launches / steps) dispatches (scripts/profile_aggregate.py).  Prints one JSON line: name, steps, ms per step of the steps after the first."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from cpp_audio_amd import api  # noqa: E402


def main():
    name = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    dev = torch.device("cuda", 0)
    w = bench.workloads()[name]
    n = 10 * bench.SR
    if name.startswith("voice_"):
        which, S = w
        cfg, rows, desc = bench.voice_workload(api, which, S, 10.0, steps + 1)
        out = torch.empty(S, n, 2, dtype=torch.float32, device=dev)
        st = torch.cuda.Stream()
        with api.Voice(cfg, S, 0) as v:
            v.set_hip_stream(st.cuda_stream)
            v.events(rows)
            v.process_device(n, out.data_ptr())
            t0 = time.perf_counter()
            for _ in range(steps - 1):
                v.process_device_async(n, out.data_ptr())
            v.synchronize()
            dt = (time.perf_counter() - t0) / max(steps - 1, 1)
        desc = "gen.voice " + desc
    else:
        desc, kw, S, carrier = w
        r = bench.Runner(api, torch, dev, 0, kw, S, n, 4321 if name != "configs1" else 1234, carrier=carrier, total_steps=steps + 1, setup_block=False)
        r.step(); r.finish(); r.ctx.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps - 1):
            r.step()
        r.finish(); r.ctx.synchronize()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / max(steps - 1, 1)
        r.close()
    print(json.dumps({"workload": name, "description": desc, "streams": S, "frames_per_stream_per_step": n, "steps": steps, "ms_per_step": dt * 1e3,
                      "csrc_sha256": bench.csrc_sha256()}))


if __name__ == "__main__":
    main()
