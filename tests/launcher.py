"""Helper process of the test session (tests/conftest.py starts it before any test runs, i.e. before this session can have touched a
GPU): runs the external programs tests need — compilers, compiled host drivers, torch.distributed launches — on the tests' behalf.
A process that has initialised the GPU must not fork + exec on the GPU pool; this one never initialises it, so its children may.
Protocol: one JSON object per line on stdin {"argv": [...], "timeout": s, "env": {...}, "cwd": ...} -> one per line on stdout
{"rc": int, "out": str, "err": str}."""
import json
import os
import subprocess
import sys

for line in sys.stdin:
    try:
        req = json.loads(line)
        env = dict(os.environ)
        env.update(req.get("env") or {})
        p = subprocess.run(req["argv"], capture_output=True, text=True, timeout=req.get("timeout", 600), env=env, cwd=req.get("cwd"))
        res = {"rc": p.returncode, "out": p.stdout, "err": p.stderr}
    except subprocess.TimeoutExpired as e:
        res = {"rc": -9, "out": (e.stdout or b"").decode(errors="replace") if isinstance(e.stdout, bytes) else (e.stdout or ""), "err": "timeout"}
    except Exception as e:  # noqa: BLE001
        res = {"rc": -1, "out": "", "err": repr(e)}
    sys.stdout.write(json.dumps(res) + "\n")
    sys.stdout.flush()
