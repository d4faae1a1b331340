"""bench.py started as plain `python bench.py --gpus N`: it must start its own N ranks (torch.distributed.run),
get them through process-group initialisation and the gather, and have rank 0 print the one JSON line.  The dry
run does all of that over gloo on CPU tensors, without a GPU and without rendering."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _run(*flags):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), *flags], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_two_ranks_start_themselves_and_reassemble_the_frame():
    out = _run("--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1", "--nx", "40", "--ny", "24")
    assert out["dry_run"] is True and out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["frame_ok"] is True and out["ms_per_step"] > 0


def test_one_rank_needs_no_launcher():
    out = _run("--gpus", "1", "--dry-run", "--steps", "2", "--warmup", "0", "--nx", "37", "--ny", "21")
    assert out["n_gpus"] == 1 and out["frame_ok"] is True
