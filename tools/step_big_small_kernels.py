"""Non-GEMM kernels of one steady step that take more than `min_us` each (rocprofv3 --kernel-trace CSV, the step between two
batched-NMS launches): python tools/step_big_small_kernels.py trace.csv [min_us]"""
import csv
import re
import sys

rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1])))
marks = [i for i, r in enumerate(rows) if "nms_reduce" in r[2]]
spans = sorted((b - a, a, b) for a, b in zip(marks[:-1], marks[1:]))
n, a, b = spans[len(spans) // 4]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
tot = 0.0
for s, e, name in rows[a:b]:
    us = (e - s) / 1e3
    if us >= min_us and "split_gemm" not in name:
        short = re.sub(r"at::native::|\(anonymous namespace\)::|void ", "", name)[:110]
        print(f"{us:8.1f} us  {short}")
        tot += us
print(f"total {tot / 1e3:.2f} ms")
