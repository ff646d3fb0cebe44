#!/bin/bash
# Matrix-pipe utilisation of the hand-written kernels inside the benchmark step, from hardware counters: one rocprofv3
# --pmc pass (only ever combined with --kernel-trace) over a short sequential bench.py run, reduced per kernel template by
# tools/pmc_mfma_reduce.py.  Usage on the GPU box:   bash tools/pmc_mfma.sh <outdir> [student|teacher]
#   pass 1: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE      (busy cycles of the matrix pipe, CU-busy cycles, active clocks)
#   pass 2: SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY   (when the counters exist)
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
OUT=${1:-gpurun_out/pmc_mfma}; WL=${2:-student}
case "$OUT" in /*) ;; *) OUT="$ROOT/$OUT";; esac
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$OUT/counters_available.txt" 2>&1 || true
have() { grep -qw "$1" "$OUT/counters_available.txt"; }
pick() { local out=""; for c in "$@"; do if have "$c"; then out="$out $c"; else echo "counter $c not offered by this rocprofv3" >&2; fi; done; echo $out; }
P1=$(pick SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE)
P2=$(pick SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU)
i=0
for grp in "$P1" "$P2"; do
  i=$((i+1))
  [ -n "$grp" ] || continue
  rm -rf /tmp/pmcmfma_$i
  timeout 900 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmcmfma_$i -o p -- python3 $ROOT/bench.py --workload $WL --steps 2 --warmup 2 --no-cpu-baseline --no-pipeline --burn-seconds 0 --secondary-steps 0 > /tmp/pmcmfma_$i.log 2>&1
  f=$(find /tmp/pmcmfma_$i -name "*counter_collection.csv" | head -1)
  if [ -z "$f" ]; then echo "pass $i ($grp): no output"; tail -5 /tmp/pmcmfma_$i.log; continue; fi
  cp "$f" "$OUT/pass$i.csv"
  k=$(find /tmp/pmcmfma_$i -name "*kernel_trace.csv" | head -1)
  [ -n "$k" ] && cp "$k" "$OUT/pass${i}_kernel_trace.csv"
done
python3 $ROOT/tools/pmc_mfma_reduce.py "$OUT" $WL > "$OUT/mfma_busy_$WL.json" && cat "$OUT/mfma_busy_$WL.json" | head -120
rm -f "$OUT"/pass*.csv   # (incl. the kernel traces: only the reduced summary is kept)
