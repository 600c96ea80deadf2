"""GPU box: timeline of one steady-state bench step from a rocprofv3 --kernel-trace csv.
usage: python tools/lab/gaps.py <dir with *kernel_trace.csv> [step index, default 10]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
k1 = [i for i, e in enumerate(ev) if "harris_kernel" in e[2]]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
a, b = k1[n], k1[n + 1]
t0 = ev[a][0]
prev_end = None
busy = 0
for s, e, name in ev[a:b + 1]:
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    short = name.replace("okvfe::(anonymous namespace)::", "").replace("void ", "")[:44]
    print("%9.1f us  +%6.1f gap  %7.1f us  %s" % ((s - t0) / 1e3, gap, (e - s) / 1e3, short))
    prev_end = e
    busy += e - s
print("step span %.1f us" % ((ev[b][0] - t0) / 1e3))
