cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_head -o t -- python tools/experiments/res5_head_probe.py 1024 > /dev/null 2>&1
python tools/trace_summary.py gpurun_out/prof_head/t_kernel_trace.csv --last-ms 280 --top 28 | python -c "
import csv,sys
for r in csv.reader(sys.stdin):
    print('%-80s %5s %10s %9s %6s' % (r[0][:80], r[1], r[2], r[3], r[4]))
"
