"""Can RCCL put two ranks on ONE device on this box?  (If so, bench.py --share-gpu could exercise the RCCL-specific branches --
ReduceOp.AVG inside the collective, init with device_id -- without a multi-GPU node.)  Spawns two ranks on device 0."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

CHILD = r'''
import os, sys, torch, torch.distributed as dist
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
try:
    dist.init_process_group("nccl", init_method="env://", device_id=dev)
    t = torch.full((1024,), float(dist.get_rank() + 1), device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.AVG)
    torch.cuda.synchronize()
    print("rank", dist.get_rank(), "avg", float(t[0]), flush=True)
    dist.destroy_process_group()
except Exception as e:
    print("rank", os.environ["RANK"], "FAILED:", type(e).__name__, str(e)[:300], flush=True)
    sys.exit(3)
'''

if __name__ == "__main__":
    from cvpr22_cross_modal_pseudo_labeling_amd.engine import launch
    path = "/tmp/rccl_same_gpu_child.py"
    open(path, "w").write(CHILD)
    for extra in ({}, {"NCCL_IGNORE_DUPLICATE_GPU": "1", "RCCL_IGNORE_DUPLICATE_GPU": "1"}):
        env = dict(os.environ, **extra)
        print("env", extra, flush=True)
        code = launch.spawn_ranks([path], 2, env=env)
        print("exit code", code, flush=True)
