"""bench.py's own launch path on CPU: `python bench.py --gpus 2 --dry-run` must start two ranks by itself (child
torch.distributed.run, gloo), run the sharded gather / timing / report control flow and print ONE JSON line; the
exit code of a failing child propagates."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=timeout)


def test_gpus_2_launches_its_own_ranks_and_prints_one_json_line():
    p = _run(["--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1", "--batch", "3"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dry_run"] and out["gather_ok"] and out["global_batch"] == 6
    assert out["ranks_seen"] == 2 and out["slowest_over_fastest_rank"] >= 1.0
    assert out["steps"] == 3 and out["warmup"] == 1 and out["value"] is None


def test_gpus_8_dry_run_the_scale_the_driver_launches():
    """the N = 8 control flow (BASELINE config 4: batch 64 = 8 images x 8 ranks): eight gloo ranks shard 64 detections
    sets, all_gather them in order, agree on the MAX-over-ranks time, rank 0 alone prints the line"""
    p = _run(["--gpus", "8", "--dry-run", "--steps", "2", "--warmup", "1", "--batch", "8"], timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["gather_ok"] and out["global_batch"] == 64 and out["scaling"] == "weak"
    assert out["ranks_seen"] == 8   # every rank answered the all_reduce of ones


def test_single_rank_dry_run_needs_no_launcher():
    p = _run(["--dry-run", "--steps", "2", "--warmup", "0", "--batch", "2"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert json.loads(p.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_child_failure_propagates():
    # an empty shard makes every rank raise inside the child launcher: the parent must not report success
    p = _run(["--gpus", "2", "--dry-run", "--batch", "0"])
    assert p.returncode != 0


def test_without_gpu_the_real_bench_fails_loudly():
    import torch

    if torch.cuda.is_available():
        return
    p = _run(["--steps", "1", "--warmup", "0"])
    assert p.returncode != 0 and "MI355X" in (p.stderr + p.stdout)
