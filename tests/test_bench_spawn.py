"""bench.py without a launcher: `python bench.py --gpus N` starts its own N rank processes (multiprocessing spawn), and the
N-rank control flow - rendezvous on 127.0.0.1, ranks_seen all_reduce, 8-row band partition, gather to rank 0,
de-interleave - runs here over gloo on synthetic band buffers (`--spawn-selftest`: no GPU, no tracing; the traced path
itself is `-m gpu`).  A failing rank must fail the whole run."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


@pytest.mark.parametrize("n", [2, 3])
def test_self_spawned_ranks(n):
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--backend", "gloo", "--spawn-selftest"],
                       capture_output=True, text=True, timeout=300, env=_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                          # rank 0 alone prints
    out = json.loads(lines[0])
    assert out["spawn_selftest"] and out["ranks_seen"] == n and out["n_gpus"] == n and out["launcher"] == "self-spawned"


def test_failing_rank_fails_the_run():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--spawn-selftest", "--selftest-fail-rank", "1"],
                       capture_output=True, text=True, timeout=300, env=_env(), cwd=ROOT)
    assert r.returncode != 0
    assert "bench-rank1" in r.stderr and "code 3" in r.stderr


def test_external_launcher_still_works():
    port = 29600 + os.getpid() % 300
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), BENCH, "--gpus", "2", "--backend", "gloo", "--spawn-selftest"],
                       capture_output=True, text=True, timeout=300, env=_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["ranks_seen"] == 2 and out["launcher"] == "external"


def test_gpus_mismatch_is_refused():
    env = _env()
    env.update({"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--spawn-selftest"], capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


@pytest.mark.gpu
def test_two_ranks_trace_pack_gather_on_the_gpu():
    """The product's N > 1 chain end to end with 2 self-spawned ranks: each rank generates the world on the device, traces its
    8-row bands (svo_trace_rows_frames), packs them (svo_gbuffer_pack), rank 0 gathers and de-interleaves - and bench.py's
    mandatory self-check compares the gathered frame with a single-GPU trace of the whole image.  The box has one GPU, so
    both ranks share it and the exchange is staged through the host over gloo (RCCL refuses two ranks on one device); what
    runs between the ranks is the same control flow as over RCCL."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--workload", "c3small_1080p_depth10_4x1x4_shadow",
                        "--steps", "24", "--warmup", "8"], capture_output=True, text=True, timeout=600, env=_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["config"]["ranks_seen"] == 2 and out["config"]["gather"] is True
    assert out["config"]["launcher"] == "self-spawned" and out["value"] > 0 and out["diagnostics"]["trace_only_mrays"] > out["value"]
