"""The single-node launcher behind ``bench.py --gpus N`` (engine/launch.py): rendezvous environment of every rank,
rank 0 alone on stdout, failure of one rank ends the job with its code, and ``bench.py`` itself turns into the launcher
before torch is imported (CPU only: the children here are tiny scripts, two of them form a real gloo group)."""
import json
import os
import subprocess
import sys
import textwrap

from cvpr22_cross_modal_pseudo_labeling_amd.engine import launch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_needs_spawn_only_without_launcher_env():
    assert launch.needs_spawn(2, env={})
    assert not launch.needs_spawn(1, env={})
    assert not launch.needs_spawn(8, env={"WORLD_SIZE": "8"})  # torch.distributed.run already started the ranks


def test_rank_env_is_env_rendezvous_on_loopback():
    env = launch.rank_env(3, 8, 29512, base={"PATH": "/bin"})
    assert env["RANK"] == env["LOCAL_RANK"] == "3" and env["WORLD_SIZE"] == env["LOCAL_WORLD_SIZE"] == "8"
    assert env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "29512"
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["PATH"] == "/bin"


def _run(script, nproc, tmp_path, extra_env=None):
    path = tmp_path / "child.py"
    path.write_text(textwrap.dedent(script))
    driver = ("import sys; sys.path.insert(0, %r)\n"
              "from cvpr22_cross_modal_pseudo_labeling_amd.engine import launch\n"
              "sys.exit(launch.spawn_ranks([%r, %r], %d))\n" % (ROOT, str(path), str(tmp_path), nproc))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(extra_env or {})
    return subprocess.run([sys.executable, "-c", driver], capture_output=True, text=True, env=env, timeout=300)


def test_spawn_two_ranks_form_a_group_and_only_rank0_prints(tmp_path):
    res = _run("""
        import json, os, sys
        import torch, torch.distributed as dist
        dist.init_process_group("gloo", init_method="env://")
        t = torch.tensor([float(dist.get_rank() + 1)])
        dist.all_reduce(t)
        print(json.dumps({"rank": dist.get_rank(), "world": dist.get_world_size(), "sum": float(t),
                          "local": os.environ["LOCAL_RANK"]}))
        dist.destroy_process_group()
    """, 2, tmp_path)
    assert res.returncode == 0, res.stderr
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]  # (gloo prints a connection banner of its own)
    assert len(lines) == 1, res.stdout  # rank 1's line went to stderr
    assert json.loads(lines[0]) == {"rank": 0, "world": 2, "sum": 3.0, "local": "0"}
    assert '"rank": 1' in res.stderr


def test_spawn_failure_of_one_rank_ends_the_job(tmp_path):
    res = _run("""
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(120)   # would hang the job: the launcher must end it
    """, 2, tmp_path)
    assert res.returncode == 7
    assert "rank 1" in res.stderr


def test_bench_becomes_the_launcher_before_importing_torch(tmp_path):
    """``python bench.py --gpus 2`` without a launcher environment: the parent spawns two ranks and never imports torch;
    the ranks (no GPU in this container) stop at the device check with a message naming the flag."""
    probe = tmp_path / "sitecustomize.py"
    probe.write_text("import atexit, os, sys\n"
                     "atexit.register(lambda: open(os.path.join(%r, 'torch_seen_' + os.environ.get('RANK', 'parent')), 'w')"
                     ".write(str('torch' in sys.modules)))\n" % str(tmp_path))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["PYTHONPATH"] = str(tmp_path) + os.pathsep + env.get("PYTHONPATH", "")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, env=env, timeout=600)
    assert res.returncode != 0
    assert "--gpus 2" in res.stderr and "visible" in res.stderr, res.stderr[-2000:]
    assert (tmp_path / "torch_seen_parent").read_text() == "False"
    assert (tmp_path / "torch_seen_0").read_text() == "True" and (tmp_path / "torch_seen_1").exists()


def test_sigterm_to_the_launcher_ends_every_rank(tmp_path):
    """a launcher killed by ``timeout`` / a scheduler must not orphan its ranks (they would keep their GPUs and
    the rendezvous port)."""
    import signal
    import time

    child = tmp_path / "child.py"
    child.write_text("import os, sys, time\n"
                     "open(os.path.join(sys.argv[1], 'pid_' + os.environ['RANK']), 'w').write(str(os.getpid()))\n"
                     "time.sleep(300)\n")
    driver = ("import sys; sys.path.insert(0, %r)\n"
              "from cvpr22_cross_modal_pseudo_labeling_amd.engine import launch\n"
              "sys.exit(launch.spawn_ranks([%r, %r], 2))\n" % (ROOT, str(child), str(tmp_path)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    parent = subprocess.Popen([sys.executable, "-c", driver], env=env, stderr=subprocess.PIPE, text=True)
    deadline = time.time() + 60
    while time.time() < deadline and not all((tmp_path / f"pid_{r}").exists() and (tmp_path / f"pid_{r}").read_text() for r in (0, 1)):
        time.sleep(0.1)
    pids = [int((tmp_path / f"pid_{r}").read_text()) for r in (0, 1)]
    parent.send_signal(signal.SIGTERM)
    assert parent.wait(timeout=30) == 128 + signal.SIGTERM
    time.sleep(0.5)
    for pid in pids:  # both ranks are gone (signal 0 = existence probe of the exact PID)
        try:
            os.kill(pid, 0)
            alive = open(f"/proc/{pid}/stat").read().split()[2] != "Z"
        except (ProcessLookupError, FileNotFoundError):
            alive = False
        assert not alive, pid
    assert "stopping the ranks" in parent.stderr.read()
