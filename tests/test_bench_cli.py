"""bench.py's launcher contract that can be checked without a GPU: `--gpus N` with fewer than N visible devices must
exit non-zero BEFORE starting anything (the driver relies on a failing exit code rather than a silent 1-GPU run), and a
WORLD_SIZE that contradicts --gpus is refused."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=900)


def test_more_gpus_than_devices_is_an_error():
    import torch
    n = torch.cuda.device_count()
    r = _run(["--gpus", str(max(2, n + 1)), "--steps", "1", "--warmup", "0"])
    assert r.returncode == 2 and "GPU(s) visible" in r.stderr and r.stdout.strip() == ""


def test_world_size_must_match_gpus():
    r = _run(["--gpus", "4", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr


def test_visible_gpus_counts_without_hip(monkeypatch):
    """The launcher's device count reads the KFD topology and the *_VISIBLE_DEVICES variables; it never calls HIP."""
    sys.path.insert(0, ROOT)
    import bench
    n = bench.visible_gpus()
    assert n >= 0
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert bench.visible_gpus() == 0
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0")
    assert bench.visible_gpus() == min(n, 1)


def test_a_failing_rank_fails_the_run_and_its_stderr_is_shown(tmp_path):
    """One rank dies before the rendezvous: the launcher reports ITS stderr tail, kills whatever is left after the
    grace period (exact child PIDs) and exits non-zero; per-rank logs are kept."""
    r = _run(["--gpus", "2", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"],
             {"LD_BENCH_SHARE_GPU": "1", "LD_BENCH_FAIL_RANK": "1", "LD_BENCH_HANG_RANK": "0", "LD_BENCH_GRACE": "2",
              "LD_BENCH_LOG_DIR": str(tmp_path)})
    assert r.returncode == 1, (r.returncode, r.stderr)
    assert "injected failure on rank 1" in r.stderr and "bench_rank1.err" in r.stderr
    assert "did not finish within" in r.stderr
    assert os.path.exists(tmp_path / "bench_rank0.err") and os.path.exists(tmp_path / "bench_rank1.err")


def test_a_hung_run_times_out_and_is_killed(tmp_path):
    r = _run(["--gpus", "2", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"],
             {"LD_BENCH_SHARE_GPU": "1", "LD_BENCH_HANG_RANK": "all", "LD_BENCH_RANK_TIMEOUT": "3", "LD_BENCH_LOG_DIR": str(tmp_path)})
    assert r.returncode == 1 and "timeout" in r.stderr and r.stdout.strip() == ""


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_over_gloo_print_one_line():
    """The N > 1 control flow of bench.py end to end on a 1-GPU box: two rank processes share the GPU, gloo carries the
    collectives (RCCL refuses two ranks on one device), rank 0 prints ONE JSON line with n_gpus == 2."""
    import json
    args = ["--gpus", "2", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-roofline", "--no-other-dtype"]
    # the launcher's own timeout (it kills its ranks by PID and prints the first failing rank's stderr tail) sits below
    # the test's; two processes bringing up HIP + gloo on one shared, freshly booted GPU box were once seen to take
    # minutes (round 4: one 300 s timeout in ~10 runs, never reproduced): one retry, with the first attempt's tail shown
    env = {"LD_BENCH_SHARE_GPU": "1", "LD_BENCH_RANK_TIMEOUT": "400"}
    r = _run(args, env)
    if r.returncode != 0 and "timeout" in r.stderr:
        print("first attempt timed out:\n" + r.stderr[-3000:])
        r = _run(args, env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]       # (gloo itself prints a "[Gloo] Rank 0 is connected" line)
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 4 and j["value"] > 0 and j["scaling"] == "weak"
    assert "functional test" in j["config"]["parallelism"]


@pytest.mark.gpu
def test_cfg5_one_image_on_two_ranks_spreads_its_branch_units_and_equals_the_one_rank_image(tmp_path):
    """cfg5 with FEWER images than ranks (`--workload cfg5 --images 1 --gpus 2`): whole images cannot fill the node, so the
    launcher's ranks take the K = 2 branch-patches of the image (dist.sample_kmask_sharded: units over ranks, ONE all-gather of
    [x_t, x0_hat] at the fusion step, images over ranks, one final gather; /root/reference/ddpm.py:1021-1042 is the fusion it
    shards).  Two rank processes share the one GPU over gloo; the image they finish must be the image ONE rank samples
    (`--gpus 1`, the reference's two-branch path): every unit draws its slice of the one noise stream."""
    import json
    import numpy as np
    # fp32 storage: on the random-init net the last DDIM steps amplify ANY difference, so the equality of the
    # two schedules is pinned in the parity mode; fp16 (cfg5's stated dtype) runs through the same code in the default line
    common = ["--workload", "cfg5", "--dtype", "fp32", "--steps", "50", "--no-roofline", "--no-cpu-baseline"]
    one, two = str(tmp_path / "one.npy"), str(tmp_path / "two.npy")
    r1 = _run(["--gpus", "1"] + common, {"LD_BENCH_DUMP": one})
    assert r1.returncode == 0, r1.stderr[-2000:]
    env = {"LD_BENCH_SHARE_GPU": "1", "LD_BENCH_RANK_TIMEOUT": "500", "LD_BENCH_DUMP": two}
    r2 = _run(["--gpus", "2", "--images", "1"] + common, env)
    if r2.returncode != 0 and "timeout" in r2.stderr:
        print("first attempt timed out:\n" + r2.stderr[-3000:])
        r2 = _run(["--gpus", "2", "--images", "1"] + common, env)
    assert r2.returncode == 0, r2.stderr[-2000:]
    j = json.loads([l for l in r2.stdout.splitlines() if l.startswith("{")][0])
    assert j["n_gpus"] == 2 and j["config"]["images_in_job"] == 1 and "sample_kmask_sharded" in j["config"]["parallelism"]
    a, b = np.load(one), np.load(two)
    assert a.shape == b.shape == (1, 1, 512, 512)
    d = np.abs(a - b)
    print(f"cfg5 fp32, one image: 2 ranks (branch units sharded) vs 1 rank: max-abs {float(d.max()):.3e} mean-abs {float(d.mean()):.3e}")
    # the unit runner and sample() step a branch with the same kernels on the same shapes; what differs is the order of a few
    # fp32 sums (tests/test_hip_dist.py pins the same equality at 1e-5 on small maps)
    # measured: 4.8e-5 / 1.4e-6
    assert float(d.max()) < 1e-3 and float(d.mean()) < 2e-5, (float(d.max()), float(d.mean()))


@pytest.mark.gpu
def test_default_line_carries_every_baseline_config():
    """The default bench line (what the driver runs, N = 1): the headline cfg3 measurement first, then the compact `cfg4_share`
    (64 patches per GPU) and `cfg5` (512^2, fp16, DDIM 50, branch + fusion) legs with throughput, step time and their dominant
    family's roofline fraction, the other storage dtype, the dtype's end-to-end distance -- ONE JSON line."""
    import json
    r = _run(["--steps", "6", "--warmup", "3", "--no-cpu-baseline"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["metric"].startswith("local patches/sec") and j["n_gpus"] == 1 and j["steps"] == 6 and j["dtype"] == "bf16"
    assert j["roofline"]["bound"] in ("hbm", "mfma") and 0 < j["roofline"]["frac"] < 1 and "resblock_conv_path" in j["roofline"]
    c4, c5 = j["cfg4_share"], j["cfg5"]
    assert c4["patches_per_gpu"] == 64 and c4["unit"] == "patches/s" and c4["value"] > 0 and 0 < c4["dominant"]["frac"] < 1
    assert c5["unit"] == "images/s" and c5["dtype"] == "fp16" and c5["value"] > 0 and c5["dominant"]["kernel"].startswith("conv3x3<f16")
    # a second sampler object must not serialise its sub-batches behind the first one's streams (round 6): cfg5 alone reads ~15
    assert c5["value"] > 11.0, c5
    assert j["other_dtype"]["dtype"] == "fp16" and "dtype_end_to_end" in j
