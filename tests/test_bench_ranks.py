"""bench.py's multi-rank control flow on CPU: the launcher that starts its own ranks, the world-size
assertion, strided sharding with a ragged / empty shard, the gather rounds, barrier + max-over-ranks timing, per-rank
self-diagnosis and the teardown order - under gloo with a stub hot path (FDBM_BENCH_STUB=1, nothing is measured)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*argv, env_extra=None, timeout=300):
    env = dict(os.environ, FDBM_BENCH_BACKEND="gloo", FDBM_BENCH_STUB="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=timeout)


def test_launch_ranks_weak_scaling_world2():
    r = run_bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "3", "--no-cpu-baseline", "--no-extras")
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["rccl_world_size"] == 2 and line["clips_per_step"] == 6
    pr = line["per_rank"]
    assert [p["rank"] for p in pr] == [0, 1]
    # every rank gathered one batch of 3 spectrograms per step: 2 steps x 3 x 257 x 256 x 8 bytes
    assert all(p["gather_calls"] == 2 and p["gather_bytes_sent"] == 2 * 3 * 257 * 256 * 8 for p in pr)
    assert all(p["elapsed_s"] > 0 and p["gather_s"] > 0 for p in pr)


def test_launch_ranks_sharded_list_with_empty_and_ragged_shards():
    """configs[3] control flow: 3 clips over 2 ranks in batches of 2 (rank 0: clips 0, 2 - one batch; rank 1: clip 1 - a
    ragged batch), then 1 clip over 2 ranks (rank 1's shard is EMPTY: it must still join every gather round)."""
    r = run_bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--clips", "3", "--batch", "2", "--no-cpu-baseline", "--no-extras")
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["clips_per_step"] == 3
    assert [p["gather_bytes_sent"] for p in line["per_rank"]] == [2 * 257 * 256 * 8, 1 * 257 * 256 * 8]
    r = run_bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--clips", "1", "--batch", "2", "--no-cpu-baseline", "--no-extras")
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert [p["gather_bytes_sent"] for p in line["per_rank"]] == [257 * 256 * 8, 0]


def test_world_size_mismatch_is_refused():
    """Started under a launcher whose world size differs from --gpus: refused before any work."""
    r = run_bench("--gpus", "2", "--steps", "1", "--no-cpu-baseline", "--no-extras", env_extra=dict(WORLD_SIZE="3", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE=3" in (r.stderr + r.stdout)
