"""CPU: ``bench.py``'s multi-rank control flow, executed before a driver with an 8-GPU node ever does (VERDICT round 3,
missing-1).  ``--device cpu --tiny`` swaps ONLY the device and the backend: the ranks run the product's own host path
(MODEL.DEVICE cpu, BASELINE configs[0]) over gloo through the same code as on GPUs -- the launcher of ``--gpus N``, the env://
rendezvous, the parameter broadcast, the barrier-bracketed region, the ``--min-seconds`` calibration of a COMMON step count,
the MAX-reduce of the elapsed time, the replay on every rank with live all-reduces, the per-rank gather, the secondary
(teacher) region -- and print a line flagged ``dry_run``.  The reference's contract: one process per GPU + DDP all-reduce,
tools/train_net.py:66-71,187-195."""
import json
import os
import socket
import subprocess
import sys

import pytest

from cvpr22_cross_modal_pseudo_labeling_amd.engine import launch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "2"
    return env


def _json_lines(stdout):
    return [ln for ln in stdout.splitlines() if ln.startswith("{")]


def test_bench_two_ranks_through_its_own_launcher():
    res = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--device", "cpu", "--tiny", "--steps", "2", "--warmup", "1",
                          "--min-seconds", "4", "--secondary-steps", "2"], capture_output=True, text=True, env=_env(), timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = _json_lines(res.stdout)
    assert len(lines) == 1, res.stdout  # rank 0 alone on stdout, ONE line
    d = json.loads(lines[0])
    assert d["dry_run"] is True and "DRY RUN" in d["metric"] and d["device"] == "cpu" and d["backend"] == "gloo"
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 4 and d["config"]["parallelism"] == "dp2"
    assert d["rccl_ranks"] == 0  # a gloo group is not RCCL: the field counts ranks of an nccl-backend group only
    assert d["scaling"] == "weak" and d["unit"] == "images/sec" and d["config"]["losses_finite"] is True
    # every bucket's all-reduce left from a backward hook (overlapped with the rest of the backward), on both ranks
    assert d["allreduce"]["buckets"] >= 3 and d["allreduce"]["issued_from_backward_hooks"] == d["allreduce"]["buckets"]
    assert d["allreduce"]["exposed_wait_ms_per_step"] > 0
    assert [r["rank"] for r in d["ranks"]] == [0, 1]
    for r in d["ranks"]:
        assert r["issued_from_backward_hooks"] == d["allreduce"]["buckets"]
        assert r["steps"] == d["steps"]  # the calibrated count is common to the ranks (MAX-reduced probe)
        assert 0 < r["elapsed_s"] <= d["ms_per_step"] * d["steps"] / 1e3 + 1e-3  # the line's time is the MAX over ranks
    assert d["steps"] >= 2 and abs(d["value"] - 4 * d["steps"] / (d["ms_per_step"] * d["steps"] / 1e3)) <= 1e-6 * d["value"]
    assert d["replay_steps"] >= 3  # the post-region replay ran (on every rank: it all-reduces)
    sec = d["secondary"]
    assert sec["steps"] == 2 and sec["images_per_s"] > 0 and sec["losses_finite"] is True and "zeroshot_mask" in sec["workload"]
    assert sec["allreduce_payload_MB"] > d["allreduce"]["payload_MB"]  # the teacher trains the trunk as well
    assert "cpu_baseline" not in d
    # the driver keeps `config`: BASELINE config 2's number rides in it; the steps took their batches from the staging thread
    assert d["config"]["secondary_ms_per_step"] == sec["ms_per_step"] and d["config"]["secondary_images_per_s"] == sec["images_per_s"]
    assert "zeroshot_mask" in d["config"]["secondary_workload"]
    assert d["config"]["h2d_in_timed_region"] is True and d["config"]["h2d_MB_per_step"] > 0
    # every rank pinned itself to a private share of the cores before it started (engine/launch.py)
    shares = [launch._parse_cpulist(r["cpus"]) for r in d["ranks"]]
    assert all(shares) and all(r["cpus_from"] == "launcher" for r in d["ranks"])
    if len(os.sched_getaffinity(0)) >= 2:
        assert not (shares[0] & shares[1])


def test_bench_eight_ranks_dry_run():
    """World 8 (the node the driver's scaling run uses), not only 2: bucket order on eight ranks, the common calibrated step
    count, the gather of eight ``ranks[*]`` with their CPU shares, one line.  Dry run on the host path over gloo."""
    env = _env()
    env["OMP_NUM_THREADS"] = "1"
    res = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--device", "cpu", "--tiny", "--steps", "2", "--warmup", "1",
                          "--min-seconds", "2", "--secondary-steps", "0", "--workload", "teacher"], capture_output=True, text=True,
                         env=env, timeout=1500)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = _json_lines(res.stdout)
    assert len(lines) == 1, res.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["config"]["global_batch"] == 16 and d["config"]["parallelism"] == "dp8" and d["rccl_ranks"] == 0
    assert [r["rank"] for r in d["ranks"]] == list(range(8))
    assert len({r["steps"] for r in d["ranks"]}) == 1 and d["ranks"][0]["steps"] == d["steps"] >= 2
    assert all(r["issued_from_backward_hooks"] == d["allreduce"]["buckets"] >= 4 for r in d["ranks"])
    assert abs(d["value"] - 16 * d["steps"] / (d["ms_per_step"] * d["steps"] / 1e3)) <= 1e-6 * d["value"]
    shares = [launch._parse_cpulist(r["cpus"]) for r in d["ranks"]]
    assert all(shares)
    if len(os.sched_getaffinity(0)) >= 8 * launch.MIN_CORES_PER_RANK:
        assert sum(len(s) for s in shares) == len(set().union(*shares))  # private shares
    else:  # too few cores for eight private shares of a useful size: the ranks say so and stay unpinned
        assert all(r["cpus_from"].startswith("unpinned") for r in d["ranks"])


def test_bench_under_torch_distributed_run_as_the_driver_launches_it():
    """The driver's form for N > 1: ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N --steps K --warmup W`` -- here with the dry-run switches appended."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--device", "cpu", "--tiny", "--workload", "teacher"], capture_output=True, text=True, env=_env(), timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = _json_lines(res.stdout)
    assert len(lines) == 1, res.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["config"]["global_batch"] == 4
    assert d["secondary"] is None and "teacher" in d["metric"] and d["config"]["secondary_ms_per_step"] is None
    # no OVIS_RANK_CPUS from this launcher: the ranks planned their own shares from LOCAL_RANK / LOCAL_WORLD_SIZE
    assert all(r["cpus"] and r["cpus_from"] in ("even split", "numa") for r in d["ranks"])
    assert d["allreduce"]["issued_from_backward_hooks"] == d["allreduce"]["buckets"] >= 4


def test_bench_rank_lost_inside_the_timed_region_fails_the_job():
    res = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--device", "cpu", "--tiny", "--steps", "3", "--warmup", "1",
                          "--workload", "teacher", "--fault-inject", "5:1"], capture_output=True, text=True, env=_env(), timeout=900)
    assert res.returncode != 0
    assert not _json_lines(res.stdout)  # no line that a driver could mistake for a result
    assert "rank 5" in res.stderr and res.returncode == 13


def test_bench_flag_mismatch_and_missing_gpu_are_refused_with_reasons():
    env = _env()
    env.update(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--device", "cpu", "--tiny"], capture_output=True, text=True, env=env, timeout=300)
    assert res.returncode != 0 and "WORLD_SIZE=2" in res.stderr
    res = subprocess.run([sys.executable, BENCH, "--steps", "1"], capture_output=True, text=True, env=_env(), timeout=300)
    assert res.returncode != 0 and ("GPU" in res.stderr or "MI355X" in res.stderr)  # no CPU fallback of the measured path


@pytest.mark.gpu
def test_bench_two_ranks_sharing_the_gpu_over_gloo_device_side_dry_run():
    """The DEVICE side of the N > 1 path on a one-GPU box: two ranks on the same MI355X (``--share-gpu``; RCCL refuses that, so the
    collectives go over gloo), HIP ops, two-stream pipelined trainer with its worker thread next to live all-reduces issued from
    backward hooks, event-timed exposed wait, replay, secondary region.  What stays unexecuted without a multi-GPU node: RCCL itself
    (``ReduceOp.AVG``, ``device_id`` init).
    Several processes time-slicing one device is not a configuration the product targets (one process per GPU); on this pool two
    ranks ran 12 / 12 times, four ranks hang and eight ranks aborted once with a queue error.  A platform-level
    failure of the shared-device run (time-out, queue abort) is therefore retried once and then reported as a SKIP with its
    reason; a run that completes is held to every assertion below."""
    cmd = [sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--share-gpu", "--tiny", "--steps", "4", "--warmup", "2",
           "--secondary-steps", "2", "--burn-seconds", "0", "--no-cpu-baseline"]
    res, why = None, ""
    for _ in range(2):
        try:
            res = subprocess.run(cmd, capture_output=True, text=True, env=_env(), timeout=240)
        except subprocess.TimeoutExpired:
            res, why = None, "timed out after 240 s (ranks time-slicing one device)"
            continue
        if res.returncode == 0:
            break
        platform = [m for m in ("HSA_STATUS_ERROR", "Memory access fault", "Connection closed by peer", "hipError") if m in res.stderr]
        if not platform:
            break  # a failure of OUR code: falls through to the assertion below
        res, why = None, f"platform error in the shared-device run: {platform}"
    if res is None:
        pytest.skip(f"--share-gpu dry run did not complete on this box: {why}")
    assert res.returncode == 0, res.stderr[-3000:]
    lines = _json_lines(res.stdout)
    assert len(lines) == 1, res.stdout
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["device"] == "cuda" and d["backend"] == "gloo" and d["n_gpus"] == 2 and d["rccl_ranks"] == 0
    assert d["config"]["pipelined"] is True and d["config"]["losses_finite"] is True and d["config"]["global_batch"] == 4
    assert d["allreduce"]["issued_from_backward_hooks"] == d["allreduce"]["buckets"] >= 3
    assert all(r["issued_from_backward_hooks"] == d["allreduce"]["buckets"] and r["steps"] == d["steps"] for r in d["ranks"])
    assert d["allreduce"]["exposed_wait_ms_per_step"] > 0 and d["replay_steps"] >= 3
    assert d["roofline"] is not None and d["kernels"]  # the HIP ops ran: per-kernel figures from the replay
    assert d["secondary"]["losses_finite"] is True and d["secondary"]["images_per_s"] > 0
