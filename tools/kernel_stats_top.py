"""Top kernels of a rocprofv3 --stats run: python tools/kernel_stats_top.py <dir> [n]"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 15
for r in list(csv.DictReader(open(f)))[:n]:
    name = re.sub(r"\(.*", "", r["Name"].replace("void ", "").replace("(anonymous namespace)::", ""))[:80]
    print("%6.2f%%  calls=%5s avg_us=%10.1f  %s" % (float(r["Percentage"]), r["Calls"], float(r["AverageNs"]) / 1e3, name))
