cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python tools/bench_ops.py --ops roi_bwd 2>&1 | tail -2
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bwd -o bwd -- python tools/bench_ops.py --ops roi_bwd --iters 10 > /dev/null 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/prof_bwd/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:8]:
        print(r['Name'][:70], r['Calls'], r['AverageNs'], r['Percentage'])
PY
