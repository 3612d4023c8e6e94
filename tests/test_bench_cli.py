"""`python bench.py --gpus N` from a plain shell must become N ranks (self-spawn under torch.distributed.run) and run the
distillation leg on every rank with the real flat-buffer all-reduce.  Checked here on CPU with the bench's own dry-run mode
(tiny widths, gloo, emulated kernels): the launch / rendezvous / collective plumbing of the path the driver takes on an 8-GPU node."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus_2_self_spawns_two_ranks_and_all_reduces():
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-cpu", "1", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly ONE JSON line"
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "weak" and "DRY RUN" in out["data"]
    d = out["distill_step"]
    assert d["n_gpus"] == 2 and d["finite"] and d["allreduce_ms"] > 0 and "gloo" in d["grad_exchange"]
    assert abs(d["samples_per_s"] - 2e3 / d["ms_per_step"]) < 1e-2 * d["samples_per_s"] + 1e-3   # global = ranks x per-rank
