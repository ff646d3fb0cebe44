#!/bin/bash
# kernel durations of the fg / bg sampler at the sizes of tools/experiments/sampler_time.py (rocprofv3 kernel trace)
set -euo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)}"
cd "$ROOT"
export TMPDIR=/tmp
OUT=/tmp/sampler_prof
rm -rf $OUT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o s -- python tools/experiments/sampler_time.py > /tmp/sampler_prof.log 2>&1 || { tail -20 /tmp/sampler_prof.log; exit 1; }
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] || { echo "no kernel trace"; exit 1; }
python - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "sample_fg_bg" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
for k in range(0, len(d), 55):
    seg = sorted(d[k:k + 55])
    print(f"launches {k}..{k + len(seg) - 1}: median {seg[len(seg) // 2] / 1e3:.1f} us, LDS {rows[k]['LDS_Block_Size']}, VGPR {rows[k]['VGPR_Count']}")
PY
