"""Single-node launcher: one child process per GPU with the env:// rendezvous variables of
``torch.distributed.launch`` (the reference's launcher contract, tools/train_net.py:187-195 + README.md:73:
``RANK`` / ``LOCAL_RANK`` / ``WORLD_SIZE`` / ``MASTER_ADDR`` / ``MASTER_PORT``; the script reads them itself).

The parent must NOT have initialised the GPU (no HIP call, no ``torch.cuda.is_available()``): it only counts
devices, forks the children as ordinary subprocesses and waits -- nothing is exec'ed over a process that has
touched the device.  Rank 0 inherits the parent's stdout (so ``bench.py``'s ONE JSON line stays one line);
the other ranks' stdout is joined to stderr.  The first child that fails ends the job: the remaining
children are terminated by PID and the parent returns that child's exit code.
"""
import os
import signal
import socket
import subprocess
import sys
import time


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_env(rank, nproc, master_port, base=None, master_addr="127.0.0.1"):
    """Environment of local rank ``rank`` of ``nproc`` (single node: RANK == LOCAL_RANK)."""
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(nproc), "LOCAL_WORLD_SIZE": str(nproc),
                "MASTER_ADDR": master_addr, "MASTER_PORT": str(master_port)})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this driver: RCCL needs it
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // max(nproc, 1))))
    return env


def needs_spawn(requested, env=None):
    """True when the caller asked for more than one rank and no launcher has set the rendezvous variables yet."""
    env = os.environ if env is None else env
    return requested > 1 and "WORLD_SIZE" not in env


class _Terminated(Exception):
    """SIGTERM reached the launcher (``timeout``, a scheduler): unwinds ``spawn_ranks`` so that its ``finally`` ends the ranks."""


def _stop(children):
    """Terminate, then kill, exactly the processes this launcher started (by PID, never by pattern)."""
    live = [c for c in children if c.poll() is None]
    for c in live:
        try:
            c.terminate()
        except OSError:
            pass
    deadline = time.time() + 10
    for c in live:
        try:
            c.wait(timeout=max(0.1, deadline - time.time()))
        except subprocess.TimeoutExpired:
            try:
                c.kill()
            except OSError:
                pass
            c.wait()


def spawn_ranks(argv, nproc, master_port=None, env=None, poll_seconds=0.2, python=None):
    """Run ``python argv...`` as ``nproc`` ranks and wait for all of them.  Returns the job's exit code: 0 when every
    rank exited 0, else the code of the first rank seen failing (the others are terminated).  Whatever ends the wait --
    a failing rank, SIGTERM / SIGINT to the launcher, a ``Popen`` that raises half-way through the spawn -- no started
    rank is left behind holding its GPU and the rendezvous port."""
    if nproc < 1:
        raise ValueError("nproc must be >= 1")
    port = master_port or free_port()
    cmd = [python or sys.executable] + list(argv)
    children = []

    def on_term(signum, frame):
        raise _Terminated(signum)

    old_term = None
    try:
        old_term = signal.signal(signal.SIGTERM, on_term)
    except ValueError:  # not the main thread: the caller's handlers stay
        old_term = None
    code = 0
    try:
        for r in range(nproc):
            children.append(subprocess.Popen(cmd, env=rank_env(r, nproc, port, env),
                                             stdout=None if r == 0 else sys.stderr))
        alive = set(range(nproc))
        while alive and code == 0:
            for r in sorted(alive):
                rc = children[r].poll()
                if rc is None:
                    continue
                alive.discard(r)
                if rc != 0:
                    code = rc if rc > 0 else 128 - rc  # killed by signal s: 128 + s, as a shell reports it
                    sys.stderr.write(f"launch: rank {r} (pid {children[r].pid}) exited with {rc}; stopping the other ranks\n")
                    break
            if alive and code == 0:
                time.sleep(poll_seconds)
    except _Terminated as e:
        code = 128 + int(e.args[0])
        sys.stderr.write(f"launch: signal {e.args[0]} received; stopping the ranks\n")
    except KeyboardInterrupt:
        code = 130
        sys.stderr.write("launch: interrupted; stopping the ranks\n")
    finally:
        _stop(children)
        if old_term is not None:
            signal.signal(signal.SIGTERM, old_term)
    return code
