"""Kernel by kernel (start, end, us, queue, stream, name) through the trunk backward of one steady-state teacher step, from a
rocprofv3 --kernel-trace CSV: shows which launches of the two streams actually overlap and what they cost beside each other.
python tools/experiments/step_tail_trace.py trace.csv"""
import csv, sys, re
rows=[]
with open(sys.argv[1], newline="") as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id","?"), r.get("Stream_Id", r.get("Stream_ID","?"))))
rows.sort()
cuts=[i for i,r in enumerate(rows) if "sgd_momentum_multi_kernel" in r[2]]
ends=[i for k,i in enumerate(cuts) if k+1==len(cuts) or rows[cuts[k+1]][0]-rows[i][0]>5e6]
a,b=ends[-3]+1, ends[-2]+1
step=rows[a:b]; t0=step[0][0]
def short(n):
    n=n.replace("void ","").replace("(anonymous namespace)::","").replace("at::native::","")
    m=re.match(r"([\w:]+(?:<[^(]{0,40})?)",n); return (m.group(1) if m else n)[:48]
# print the last 2.2 ms before SGD (trunk backward)
tend=step[-1][0]
for s,e,n,q,st in step:
    if s>tend-2.6e6 and s<tend-1.2e6:
        print(f"{(s-t0)/1e3:9.1f} {(e-t0)/1e3:9.1f} {(e-s)/1e3:7.1f} q{q} s{st} {short(n)}")
