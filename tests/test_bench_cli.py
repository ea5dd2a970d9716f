"""bench.py's launcher contract that can be checked without a GPU: `--gpus N` with fewer than N visible devices must
exit non-zero BEFORE starting anything (the driver relies on a failing exit code rather than a silent 1-GPU run), and a
WORLD_SIZE that contradicts --gpus is refused."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)


def test_more_gpus_than_devices_is_an_error():
    import torch
    n = torch.cuda.device_count()
    r = _run(["--gpus", str(max(2, n + 1)), "--steps", "1", "--warmup", "0"])
    assert r.returncode == 2 and "GPU(s) visible" in r.stderr and r.stdout.strip() == ""


def test_world_size_must_match_gpus():
    r = _run(["--gpus", "4", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr
