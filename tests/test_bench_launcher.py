"""CPU: bench.py's multi-rank plumbing. `python bench.py --gpus 2` with no launcher around it must become two ranks by
itself (`--gpus` used to be parsed and ignored), shard / gather / max-reduce over them, and refuse a world
size that does not match --gpus. Driven with `--stub` (fake engine on CPU tensors, gloo) where there is no GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def test_gpus2_spawns_two_ranks_itself(lib_built):
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--stub", "--steps", "3", "--warmup", "1", "--batch", "5"],
                       env=_env(), cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["collective_backend"] == "gloo"
    assert d["config"]["global_batch"] == 10 and d["config"]["parallelism"] == "dp2"
    assert len(d["rank_devices"]) == 2 and len(d["rank_crops_per_s"]) == 2
    assert d["stub_gather_ok"] is True, "records of both ranks must arrive rank-major in the gathered tensor"
    # value is whole-job throughput from the max-over-ranks time
    assert abs(d["value"] - 10 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]
    assert d["max_rank_ms_per_step"] >= 1e3 * 5 * 3 / max(d["rank_crops_per_s"]) / 3 - 1e-9


def test_world_size_mismatch_is_refused(lib_built):
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--stub", "--steps", "1", "--warmup", "0"],
                       env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "refusing" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")], "no JSON line for a job that did not run"


def test_gpus_beyond_the_node_is_refused(lib_built):
    """Without --stub the launcher checks the visible GPU count first (0 in this container, 1 on the gpurun box)."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "64", "--steps", "1", "--warmup", "0"], env=_env(), cwd=ROOT,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "refusing" in r.stderr
