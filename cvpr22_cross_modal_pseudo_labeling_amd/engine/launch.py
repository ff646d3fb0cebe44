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


def rank_env(rank, nproc, master_port, base=None, master_addr="127.0.0.1", gpu_ranks=True):
    """Environment of local rank ``rank`` of ``nproc`` (single node: RANK == LOCAL_RANK).  ``gpu_ranks`` False (the CPU-only
    configuration over gloo, whose step IS OpenMP / torch-CPU math): OMP_NUM_THREADS is left to the rank's core share."""
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(nproc), "LOCAL_WORLD_SIZE": str(nproc),
                "MASTER_ADDR": master_addr, "MASTER_PORT": str(master_port)})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this driver: RCCL needs it
    # a SMALL OpenMP team per rank: the GPU path does almost no CPU math, and a team as wide as the rank's core share spins
    # after every host op and starves the rank's own staging / loader threads (profiles/r5_cli_input_path.txt)
    if gpu_ranks:
        env.setdefault("OMP_NUM_THREADS", str(max(1, min(8, (os.cpu_count() or 1) // max(nproc, 1) // 4))))
    else:
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // max(nproc, 1))))
    return env


# ---- per-rank CPU placement ------------------------------------------------------------------------------------------
# A rank drives its GPU from two host threads that issue ~530 launches per step between host reads (engine/trainer.py);
# on a two-socket 8-GPU node an unpinned rank migrates between sockets and its launches cross the inter-socket link.
# Every rank therefore pins itself -- BEFORE its first GPU call, so the runtime's helper threads inherit the mask -- to a
# private share of the cores of the NUMA node its GPU hangs off (sysfs), or to an even share of the allowed cores when
# the topology cannot be read.  The reference leaves placement to the OS (tools/train_net.py:187-195 only sets the device).
MIN_CORES_PER_RANK = 4


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_numa_nodes(sysfs_root="/sys"):
    """NUMA node of every AMD GPU (render node), in PCI-address order -- the order HIP enumerates devices in by default.
    [] when sysfs has no answer (containers without /sys/class/drm, a node reporting -1)."""
    base = os.path.join(sysfs_root, "class", "drm")
    found = []
    try:
        entries = sorted(e for e in os.listdir(base) if e.startswith("renderD"))
    except OSError:
        return []
    for e in entries:
        dev = os.path.join(base, e, "device")
        try:
            with open(os.path.join(dev, "vendor")) as f:
                if f.read().strip().lower() != "0x1002":
                    continue
        except OSError:
            continue  # not a PCI function: the compute-partition render nodes (amdgpu_xcp_*) an MI300 / MI355X also lists
        try:
            with open(os.path.join(dev, "numa_node")) as f:
                node = int(f.read().strip())
            addr = os.path.basename(os.path.realpath(dev))
        except (OSError, ValueError):
            return []
        if node < 0:
            return []
        found.append((addr, node))
    return [n for _, n in sorted(found)]


def _smt_order(cpus, sysfs_root="/sys"):
    """``cpus`` ordered so that the hardware threads of one core are adjacent (key: the lowest sibling id): contiguous shares
    then own whole cores instead of sharing them with the rank that got the sibling ids (node0 = "0-63,128-191" on the MI355X
    hosts: 128-191 are the second threads of 0-63)."""
    def key(c):
        try:
            with open(os.path.join(sysfs_root, "devices", "system", "cpu", f"cpu{c}", "topology", "thread_siblings_list")) as f:
                return (min(_parse_cpulist(f.read())), c)
        except (OSError, ValueError):
            return (c, c)
    return sorted(cpus, key=key)


def plan_affinity(nproc, allowed=None, sysfs_root="/sys", visible=None):
    """[set of cpu ids] per local rank.  ``visible`` = the GPU indices the ranks use (HIP/ROCR_VISIBLE_DEVICES order;
    default 0..nproc-1).  NUMA-aware when every rank's GPU has a known node with allowed cores, else an even contiguous
    split of ``allowed`` (default: this process' affinity mask).  A rank never gets an empty set."""
    allowed = sorted(os.sched_getaffinity(0) if allowed is None else allowed)
    nodes = gpu_numa_nodes(sysfs_root)
    visible = list(range(nproc)) if visible is None else list(visible)
    plan = None
    if nodes and len(visible) >= nproc and all(0 <= v < len(nodes) for v in visible[:nproc]):
        by_node = {}
        for r in range(nproc):
            by_node.setdefault(nodes[visible[r]], []).append(r)
        plan = [None] * nproc
        for node, ranks in by_node.items():
            try:
                with open(os.path.join(sysfs_root, "devices", "system", "node", f"node{node}", "cpulist")) as f:
                    cpus = _smt_order(_parse_cpulist(f.read()) & set(allowed), sysfs_root)
            except (OSError, ValueError):
                cpus = []
            if len(cpus) < len(ranks):
                plan = None
                break
            share = len(cpus) // len(ranks)
            for i, r in enumerate(ranks):
                plan[r] = set(cpus[i * share:(i + 1) * share])
    if plan is None:
        if len(allowed) < nproc:  # fewer cores than ranks: everyone shares everything
            return [set(allowed) for _ in range(nproc)]
        share = len(allowed) // nproc
        ordered = _smt_order(allowed, sysfs_root)
        plan = [set(ordered[r * share:(r + 1) * share]) for r in range(nproc)]
    return plan


def format_cpus(cpus):
    """{0,1,2,3,8} -> '0-3,8'."""
    cpus = sorted(cpus)
    out, i = [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        out.append(str(cpus[i]) if i == j else f"{cpus[i]}-{cpus[j]}")
        i = j + 1
    return ",".join(out)


def _visible_devices(env):
    """GPU indices behind local ranks 0, 1, ... when HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES re-orders or restricts them
    (None: rank r uses GPU r)."""
    hip, rocr = env.get("HIP_VISIBLE_DEVICES"), env.get("ROCR_VISIBLE_DEVICES")
    if hip and rocr:
        # HIP's indices are then relative to the ROCR-filtered list: mapping them onto sysfs order would pin ranks to the
        # wrong socket while the log says "numa" -- an empty list makes plan_affinity fall back to the even split
        return []
    vis = hip or rocr
    if vis and all(v.strip().isdigit() for v in vis.split(",")):
        return [int(v) for v in vis.split(",")]
    return None


def apply_rank_affinity(env=None, sysfs_root="/sys"):
    """Pin the calling rank (call before anything touches the GPU).  ``OVIS_RANK_CPUS`` (set by ``spawn_ranks``) wins;
    under another launcher (torch.distributed.run) the share is planned here from LOCAL_RANK / LOCAL_WORLD_SIZE.
    ``OVIS_NO_AFFINITY=1`` leaves placement to the OS.  Returns what was applied: {"cpus": "0-15", "source": ...}."""
    env = os.environ if env is None else env
    if env.get("OVIS_NO_AFFINITY") == "1" or not hasattr(os, "sched_setaffinity"):
        return {"cpus": None, "source": "off"}
    try:
        if env.get("OVIS_RANK_CPUS"):
            cpus, source = _parse_cpulist(env["OVIS_RANK_CPUS"]), "launcher"
        else:
            world = int(env.get("LOCAL_WORLD_SIZE", env.get("WORLD_SIZE", "1")))
            rank = int(env.get("LOCAL_RANK", env.get("RANK", "0")))
            if world <= 1:
                return {"cpus": format_cpus(os.sched_getaffinity(0)), "source": "single rank: unchanged"}
            cpus = plan_affinity(world, sysfs_root=sysfs_root, visible=_visible_devices(env))[rank]
            vis = _visible_devices(env)
            source = "numa" if (gpu_numa_nodes(sysfs_root) and (vis is None or len(vis) >= world)) else "even split"
        cpus = set(cpus) & os.sched_getaffinity(0) or set(cpus)
        if len(cpus) < MIN_CORES_PER_RANK and len(cpus) < len(os.sched_getaffinity(0)):
            # a rank runs its training thread, the frozen-half worker, the staging thread and RCCL's proxy threads: squeezing
            # them onto one or two cores (an 8-core container shared by 8 ranks) costs more than migration does
            return {"cpus": format_cpus(os.sched_getaffinity(0)), "source": f"unpinned: a share of {len(cpus)} core(s) is too small"}
        os.sched_setaffinity(0, cpus)
        return {"cpus": format_cpus(cpus), "source": source}
    except (OSError, ValueError) as e:  # a cpuset that forbids it: run unpinned, say so
        return {"cpus": None, "source": f"failed: {e}"}


def needs_spawn(requested, env=None):
    """True when the caller asked for more than one rank and no launcher has set the rendezvous variables yet."""
    env = os.environ if env is None else env
    return requested > 1 and "WORLD_SIZE" not in env


class _Terminated(Exception):
    """SIGTERM reached the launcher (``timeout``, a scheduler).  Raised by ``spawn_ranks`` itself between two polls -- the
    signal handler only leaves a note -- so that it can never fire inside ``Popen`` (a started rank missing from
    ``children``) or inside the clean-up."""


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


def spawn_ranks(argv, nproc, master_port=None, env=None, poll_seconds=0.2, python=None, gpu_ranks=None):
    """Run ``python argv...`` as ``nproc`` ranks and wait for all of them.  Returns the job's exit code: 0 when every
    rank exited 0, else the code of the first rank seen failing (the others are terminated).  Whatever ends the wait --
    a failing rank, SIGTERM / SIGINT to the launcher, a ``Popen`` that raises half-way through the spawn -- no started
    rank is left behind holding its GPU and the rendezvous port."""
    if nproc < 1:
        raise ValueError("nproc must be >= 1")
    if gpu_ranks is None:  # the CPU-only configuration (bench.py --device cpu, train_net.py MODEL.DEVICE cpu) does its math on the host
        a = [str(x) for x in argv]
        gpu_ranks = not any(a[i:i + 2] in (["--device", "cpu"], ["MODEL.DEVICE", "cpu"]) for i in range(len(a)))
    port = master_port or free_port()
    cmd = [python or sys.executable] + list(argv)
    children = []
    got = []  # signals seen: the handler only notes them; the poll loop (and the spawn loop) act on the note

    def on_term(signum, frame):
        got.append(signum)

    old_term = None
    try:
        old_term = signal.signal(signal.SIGTERM, on_term)
    except ValueError:  # not the main thread: the caller's handlers stay
        old_term = None
    code = 0
    base_env = os.environ if env is None else env
    no_pin = base_env.get("OVIS_NO_AFFINITY") == "1"
    plan = None if no_pin else plan_affinity(nproc, visible=_visible_devices(base_env))
    try:
        for r in range(nproc):
            if got:
                raise _Terminated(got[0])
            renv = rank_env(r, nproc, port, env, gpu_ranks=gpu_ranks)
            if plan is not None:
                renv.setdefault("OVIS_RANK_CPUS", format_cpus(plan[r]))
            children.append(subprocess.Popen(cmd, env=renv, stdout=None if r == 0 else sys.stderr))
        alive = set(range(nproc))
        while alive and code == 0:
            if got:
                raise _Terminated(got[0])
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
        # further SIGTERMs (timeout -k, a scheduler retrying) are only noted: nothing may cut the clean-up short
        for attempt in range(3):
            try:
                _stop(children)
                break
            except BaseException as e:  # KeyboardInterrupt in the middle of a wait: try again, the ranks must go
                sys.stderr.write(f"launch: clean-up interrupted ({type(e).__name__}); retrying\n")
        # whatever interrupted the retries: no rank may outlive the launcher (it would keep its GPU and the rendezvous port)
        for c in children:
            try:
                if c.poll() is None:
                    c.kill()
                    c.wait(timeout=10)
            except BaseException as e:  # noqa: BLE001 -- keep going: the other ranks still have to go
                sys.stderr.write(f"launch: could not reap pid {c.pid} ({type(e).__name__})\n")
        for c in children:
            if c.poll() is None:
                sys.stderr.write(f"launch: rank process {c.pid} SURVIVED the clean-up\n")
        if old_term is not None:
            signal.signal(signal.SIGTERM, old_term)
    return code
