"""``python bench.py --gpus 2`` must start its two ranks itself (no external torchrun), rendezvous, all-reduce and print
ONE JSON line that says two ranks formed.  Rehearsed on CPU: gloo backend, a toy torch.nn model (``--rehearse-cpu`` -- the
launch / setup_replica / reducer / barrier / JSON plumbing is the code the GPU run uses; the HIP kernels are not involved
and the line is marked as a rehearsal).  Reference behaviour matched: accelerate launch + accelerator.prepare,
/root/reference/train.py:167-169,211."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["MI355SEG_DIST_BACKEND"] = "gloo"
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rehearse-cpu", "--steps", "3", "--warmup", "1"] + extra,
                       capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout                     # exactly one JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_self_launches_two_ranks():
    r = _run(["--gpus", "2"])
    assert r["n_gpus"] == 2 and r["rccl_ranks"] == 2 and r["dist_backend"] == "gloo"
    assert r["config"]["parallelism"] == "dp2" and r["config"]["global_batch"] == 4
    assert len(r["rank_devices"]) == 2 and r["rank_devices"][0].startswith("rank0:") and r["rank_devices"][1].startswith("rank1:")
    assert r["steps"] == 3 and r["warmup"] == 1 and r["value"] > 0 and r["scaling"] == "weak"
    assert "rehearsal" in r
    # the `comm` block: what a step sends and how long it waited for it (so that a scaling shortfall can be attributed)
    c = r["comm"]
    assert c["grad_bytes_per_step"] == 4 * (4 * 27 + 4 + 4 + 4 + 2 * 4 + 2) and c["buckets"] >= 1
    assert c["allreduce_wait_ms_per_step"] > 0.0 and c["broadcast_buffers_ms_per_step"] > 0.0
    assert c["buffer_broadcast_collectives_per_step"] == 1.0 and 0.0 < c["allreduce_wait_frac_of_step"] < 1.0


def test_bench_single_rank_does_not_launch():
    r = _run(["--gpus", "1"])
    assert r["n_gpus"] == 1 and r["rccl_ranks"] == 1 and r["dist_backend"] is None and "comm" not in r


def test_bench_under_an_external_launcher_is_one_of_the_ranks():
    """The driver's form: torch.distributed.run starts the ranks; bench.py must not launch again."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["MI355SEG_DIST_BACKEND"] = "gloo"
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29631", os.path.join(ROOT, "bench.py"), "--rehearse-cpu", "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    assert json.loads(lines[0])["n_gpus"] == 2
