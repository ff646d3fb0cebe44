"""Which kernel SOURCES a committed counter summary was collected from, so that a summary older than the kernel it
describes cannot be quoted as current (``bench.py`` read whatever file was committed).

``tools/pmc_*_reduce.py`` store ``stamp()`` in every summary they write: a digest per ``csrc/`` source file, the
``__global__`` function -> file map, and the kernel template instances the profiled run launched (the names the
profiler printed, template arguments included).  ``bench.py`` calls ``stale_reason()`` for the kernel family it is
about to quote: the summary is current only while the file that holds the family's kernel and every header it includes
still have the digests of the collection run.  No torch, no GPU: plain file reads.
"""
import glob
import hashlib
import os
import re
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc")
_GLOBAL = re.compile(r"__global__[^;{]*?void\s+(\w+)\s*\(", re.S)


def source_digests(csrc=CSRC):
    """{file name: first 16 hex digits of its sha256} for every kernel source and shared header of ``csrc/``."""
    out = {}
    for path in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h"))):
        with open(path, "rb") as f:
            out[os.path.basename(path)] = hashlib.sha256(f.read()).hexdigest()[:16]
    return out


def kernel_files(csrc=CSRC):
    """{__global__ function name: the ``.hip`` file that defines it}."""
    out = {}
    for path in sorted(glob.glob(os.path.join(csrc, "*.hip"))):
        with open(path) as f:
            for name in _GLOBAL.findall(f.read()):
                out[name] = os.path.basename(path)
    return out


def local_includes(name, csrc=CSRC, _seen=None):
    """Transitive closure of the ``#include "x.h"`` lines of ``csrc/<name>`` over the headers that live in ``csrc/``."""
    seen = set() if _seen is None else _seen
    try:
        with open(os.path.join(csrc, name)) as f:
            text = f.read()
    except OSError:
        return seen
    for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', text, re.M):
        if inc not in seen and os.path.exists(os.path.join(csrc, inc)):
            seen.add(inc)
            local_includes(inc, csrc, seen)
    return seen


def git_head(root=os.path.dirname(os.path.dirname(CSRC))):
    """``git rev-parse HEAD`` (+ ``-dirty`` when csrc/ differs from it).  The GPU box's snapshot has no history: there the
    value ``write_git_head()`` left in ``.git_head`` before the snapshot was taken is returned (None without one)."""
    def from_file():
        try:
            with open(os.path.join(root, ".git_head")) as f:
                return f.read().strip() or None
        except OSError:
            return None

    if not os.path.isdir(os.path.join(root, ".git")):
        return from_file()
    try:
        head = subprocess.run(["git", "-C", root, "rev-parse", "HEAD"], capture_output=True, text=True, timeout=10)
        if head.returncode != 0:
            return from_file()
        dirty = subprocess.run(["git", "-C", root, "status", "--porcelain", "--", "cvpr22_cross_modal_pseudo_labeling_amd/csrc"],
                               capture_output=True, text=True, timeout=10).stdout.strip()
        return head.stdout.strip() + ("-dirty" if dirty else "")
    except (OSError, subprocess.SubprocessError):
        return None


def write_git_head(root=os.path.dirname(os.path.dirname(CSRC))):
    """Leave the current HEAD in ``.git_head`` (git-ignored, travels with the gpurun snapshot); called by ``build()``."""
    head = git_head(root) if os.path.isdir(os.path.join(root, ".git")) else None
    if head:
        with open(os.path.join(root, ".git_head"), "w") as f:
            f.write(head + "\n")
    return head


def stamp(kernel_instances=None):
    """What a counter summary records about the build it was collected on."""
    return {"git_head": git_head(), "source_sha256_16": source_digests(), "kernel_file": kernel_files(),
            "kernel_instances": sorted(kernel_instances) if kernel_instances is not None else None}


def stale_reason(summary, family):
    """None while ``summary`` (a dict loaded from a profiles/*.json) still describes kernel ``family`` as it is in the
    tree; otherwise one sentence saying what differs."""
    st = summary.get("provenance")
    if not st:
        return "the summary carries no provenance stamp (collected before round 4)"
    then, now = st.get("source_sha256_16", {}), source_digests()
    src = kernel_files().get(family)
    if src is None:
        return f"no __global__ {family} in csrc/ any more"
    if st.get("kernel_file", {}).get(family) != src:
        return f"{family} lived in {st.get('kernel_file', {}).get(family)} at collection time, now in {src}"
    changed = [f for f in [src] + sorted(local_includes(src)) if then.get(f) != now.get(f)]
    if changed:
        return (f"{', '.join(changed)} changed since the counters were collected"
                + (f" (at {st['git_head'][:12]})" if st.get("git_head") else ""))
    return None
