"""Matrix-pipe busy fraction per hand-written kernel template from the rocprofv3 counter passes of tools/pmc_mfma.sh.

    mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8)

This is rocprofv3's own ``MfmaUtil`` expression (``reduce(SQ_VALU_MFMA_BUSY_CYCLES,sum) / (reduce(GRBM_GUI_ACTIVE,max) *
SIMD_NUM)``, printed by ``rocprofv3 -L``): SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles summed over every SIMD (16 per
v_mfma_f32_16x16x32_bf16: busy cycles / SQ_INSTS_MFMA = 16.0 on every split-GEMM template); GRBM_GUI_ACTIVE in the
counter CSV is the SUM over the 8 XCDs' instances, so the per-dispatch active clocks are a eighth of it (cross-check
printed per kernel: clocks / kernel-trace duration = the shader clock, 1.9-2.4 GHz).  Kernels are grouped by their
template instance (the split GEMM's forms are template arguments)."""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cvpr22_cross_modal_pseudo_labeling_amd.utils import provenance  # noqa: E402
CUS, SIMDS, XCDS = 256, 4, 8
OURS = set()
for src in glob.glob(os.path.join(ROOT, "cvpr22_cross_modal_pseudo_labeling_amd", "csrc", "*.hip")):
    OURS.update(re.findall(r"__global__[^;{]*?void\s+(\w+)\s*\(", open(src).read(), re.S))


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    m = re.match(r"(\w+)(<[^(]*>)?\(", name)
    if not m or m.group(1) not in OURS:
        return None
    return m.group(1) + (m.group(2) or "")


def load(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            k = short(r.get("Kernel_Name", ""))
            if k:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def durations(path):
    """Mean kernel duration in microseconds per template instance from a rocprofv3 kernel trace."""
    agg = collections.defaultdict(list)
    if not os.path.exists(path):
        return {}
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            k = short(r.get("Kernel_Name", ""))
            if k:
                agg[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return {k: sum(v) / len(v) for k, v in agg.items()}


def main():
    out_dir, wl = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "student")
    dur = durations(os.path.join(out_dir, "pass1_kernel_trace.csv"))
    passes = [load(p) for p in sorted(glob.glob(os.path.join(out_dir, "pass[0-9].csv")))]
    res = {"workload": wl,
           "command": "bash tools/pmc_mfma.sh (rocprofv3 --kernel-trace --pmc <group> -- python3 bench.py --steps 2 --warmup 2 "
                      "--no-pipeline --no-cpu-baseline), every launch of the run",
           "formula": f"mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / ({SIMDS} x {CUS} x GRBM_GUI_ACTIVE / {XCDS}) (rocprofv3's "
                      f"MfmaUtil; the CSV sums GRBM_GUI_ACTIVE over the {XCDS} XCDs); cu_busy_frac = SQ_BUSY_CU_CYCLES / "
                      f"({CUS} x GRBM_GUI_ACTIVE / {XCDS})",
           "kernels": {}}
    names = sorted({k for p in passes for k in p})
    tot = collections.defaultdict(float)
    for k in names:
        row = {}
        for p in passes:
            for c, v in p.get(k, {}).items():
                row[c] = {"launches": len(v), "sum": sum(v)}
        e = {"launches": max((x["launches"] for x in row.values()), default=0)}
        for c, x in row.items():
            e[c + "_per_launch"] = x["sum"] / max(x["launches"], 1)
        act = row.get("GRBM_GUI_ACTIVE", {}).get("sum", 0.0) / XCDS
        if act > 0 and "SQ_VALU_MFMA_BUSY_CYCLES" in row:
            e["mfma_busy_frac"] = round(row["SQ_VALU_MFMA_BUSY_CYCLES"]["sum"] / (SIMDS * CUS * act), 4)
            fam = k.split("<")[0]
            tot[fam + ":busy"] += row["SQ_VALU_MFMA_BUSY_CYCLES"]["sum"]
            tot[fam + ":act"] += act
        if k in dur:
            e["avg_us_under_profiler"] = round(dur[k], 1)
            if act > 0:
                e["implied_clock_GHz"] = round(act / max(row["GRBM_GUI_ACTIVE"]["launches"], 1) / dur[k] / 1e3, 2)
        if row.get("SQ_INSTS_MFMA", {}).get("sum", 0) > 0 and "SQ_VALU_MFMA_BUSY_CYCLES" in row:
            e["busy_cycles_per_mfma"] = round(row["SQ_VALU_MFMA_BUSY_CYCLES"]["sum"] / row["SQ_VALU_MFMA_BUSY_CYCLES"]["launches"]
                                              / (row["SQ_INSTS_MFMA"]["sum"] / row["SQ_INSTS_MFMA"]["launches"]), 2)
        if act > 0 and "SQ_BUSY_CU_CYCLES" in row:
            e["cu_busy_frac"] = round(row["SQ_BUSY_CU_CYCLES"]["sum"] / (CUS * act), 4)
        res["kernels"][k] = {kk: (round(vv, 1) if isinstance(vv, float) and kk.endswith("_per_launch") else vv) for kk, vv in e.items()}
    res["families"] = {f.split(":")[0]: {"mfma_busy_frac": round(tot[f] / (SIMDS * CUS * tot[f.split(':')[0] + ':act']), 4)}
                       for f in tot if f.endswith(":busy") and tot[f.split(":")[0] + ":act"] > 0}
    res["provenance"] = provenance.stamp(names)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
