"""bench.py's multi-rank plumbing on CPU: `python bench.py --gpus N` must START N ranks itself,
report the world size it saw, and run the shard / gather / barrier / max-over-ranks path (gloo, --dry-run: no GPU work)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *map(str, args)], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0]), p.stderr


@pytest.mark.parametrize("gpus,batch", [(1, 1), (2, 1), (2, 32), (3, 2)])
def test_bench_spawns_its_ranks_dry_run(gpus, batch):
    out, err = run_bench("--gpus", gpus, "--dry-run", "--steps", 2, "--warmup", 1, "--tokens", 64, "--batch-per-gpu", batch)
    assert out["dry_run"] is True and out["n_gpus"] == gpus and out["world_size_seen"] == gpus
    # the per-rank diagnostics a first real multi-GPU run needs: every rank reported its wall and gather time
    pr = out["per_rank"]
    assert pr["ranks_reporting"] == gpus and 0 <= pr["wall_s"]["min"] <= pr["wall_s"]["max"]
    assert 0 <= pr["gather_ms_per_step"]["min"] <= pr["gather_ms_per_step"]["max"]
    assert out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak" and out["higher_is_better"] is True
    assert out["config"]["batch_per_gpu"] == batch
    assert out["value"] > 0
    if gpus > 1:
        assert f"launching {gpus} ranks" in err


def test_bench_eight_ranks_config3_dry_run():
    """The driver's 8-GPU line (`bench.py --gpus 8 --config 3`): 8 ranks start, every one reports, the shard is configs[3]'s (VERDICT r4
    item 7).  No GPU work: the fabricated streams travel through the same shard / gather / barrier / max-over-ranks code."""
    out, err = run_bench("--gpus", 8, "--dry-run", "--config", 3, "--steps", 1, "--warmup", 1)
    assert out["dry_run"] is True and out["n_gpus"] == 8 and out["world_size_seen"] == 8
    assert out["per_rank"]["ranks_reporting"] == 8
    assert out["config"]["batch_per_gpu"] == 32 and out["scaling"] == "weak"
    assert out["value"] > 0 and "launching 8 ranks" in err


def test_bench_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--layers", "2"], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode != 0 and ("HIP device" in p.stderr or "no CPU fallback" in p.stderr)
