"""How much of the machine do the kernels of a step fill?  From a rocprofv3 --kernel-trace CSV of a SEQUENTIAL run
(bench.py --no-pipeline: every kernel alone on the GPU), per step (cut at the batched NMS reduce):
  sum of kernel durations, and the same sum weighted with each launch's FILL = min(1, workgroups / resident capacity), the
  capacity from the launch's own VGPR / LDS / workgroup size (256 CUs, 4 SIMDs x 512 VGPRs x 8 waves, 160 KB LDS).
The weighted sum is what the step would take if launches of less than one round of workgroups shared the machine
perfectly: the lower bound for any multi-stream schedule of the same kernels.
    python tools/step_fill_report.py trace.csv"""
import csv
import re
import sys
from collections import defaultdict

CUS, LDS_CU = 256, 160 * 1024


def capacity(r):
    wg = int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1)
    waves = max(1, (wg + 63) // 64)
    regs = int(r.get("VGPR_Count", 0) or 0) + int(r.get("Accum_VGPR_Count", 0) or 0)
    per_simd = 8 if regs <= 0 else max(1, min(8, 512 // max(regs, 1)))
    by_regs = max(1, 4 * per_simd // waves)
    lds = int(r.get("LDS_Block_Size", 0) or 0)
    by_lds = LDS_CU // lds if lds > 0 else 64
    # the split GEMMs use DYNAMIC LDS, which the trace does not report: their residency is known from the source
    # (csrc/split_gemm.hip: the TN kernel 2 x 64 KB per CU; split_gemm_kernel<WM, WN, MODE, NS, OCC, ...>: OCC workgroups per
    # CU by __launch_bounds__, 3 for the halo-tile mode whose stage is larger)
    name = r["Kernel_Name"]
    if "split_gemm_tn_kernel" in name:
        by_lds = 2
    m = re.search(r"split_gemm_kernel<(\d+), (\d+), (\d+), (\d+), (\d+)", name)
    if m:
        by_lds = 3 if m.group(3) == "2" else int(m.group(5))
    return CUS * max(1, min(by_regs, by_lds, 32))


def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "")
    m = re.match(r"([\w:]+(?:<[^(]{0,40})?)", n)
    return (m.group(1) if m else n)[:52]


rows = []
with open(sys.argv[1], newline="") as f:
    rd = csv.DictReader(f)
    need = ["Grid_Size_X", "Workgroup_Size_X", "Start_Timestamp", "End_Timestamp", "Kernel_Name"]
    miss = [c for c in need if c not in rd.fieldnames]
    if miss:
        raise SystemExit(f"columns missing: {miss}; have {rd.fieldnames}")
    for r in rd:
        grid = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
        wg = int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1)
        nwg = max(1, grid // max(wg, 1))
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], min(1.0, nwg / capacity(r))))
rows.sort()
marks = [i for i, r in enumerate(rows) if "nms_reduce" in r[2]]
print(f"{len(marks)} steps in the trace")
print(" step  wall_ms  sum_ms  filled_ms  launches  launches<1round  their_ms  their_filled_ms")
agg = defaultdict(lambda: [0, 0.0, 0.0])
last = None
for si in range(len(marks) - 1):
    seg = rows[marks[si]:marks[si + 1]]
    tot = sum(e - s for s, e, _, _ in seg)
    filled = sum((e - s) * f for s, e, _, f in seg)
    small = [(e - s, f) for s, e, _, f in seg if f < 1.0]
    print(f"{si:5d} {(seg[-1][1] - seg[0][0]) / 1e6:8.2f} {tot / 1e6:7.2f} {filled / 1e6:10.2f} {len(seg):9d} {len(small):16d} "
          f"{sum(d for d, _ in small) / 1e6:9.2f} {sum(d * f for d, f in small) / 1e6:16.2f}")
    last = seg
if last:
    for s, e, n, f in last:
        if f < 1.0:
            a = agg[short(n)]
            a[0] += 1
            a[1] += (e - s) / 1e3
            a[2] += (e - s) * f / 1e3
    print("last step, launches of less than one round by kernel: count, us, filled us")
    for k, (c, us, fu) in sorted(agg.items(), key=lambda kv: -(kv[1][1] - kv[1][2]))[:25]:
        print(f"  {c:4d} {us:9.1f} {fu:9.1f}  {k}")
