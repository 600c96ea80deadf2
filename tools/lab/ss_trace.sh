#!/bin/bash
# GPU box: per-dispatch durations of one steady-state BriskFeatureDetector + extractor call (tools/bench_scalespace.py),
# in launch order -- which layer costs what
R=$PWD; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/sst
rocprofv3 --kernel-trace --output-format csv -d /tmp/sst -o p -- python $R/tools/bench_scalespace.py 512 2 > /tmp/sst.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob("/tmp/sst/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
names=[r["Kernel_Name"] for r in rows]
# last occurrence of the first kernel of a call = start of the last call
first=[i for i,n in enumerate(names) if "twothird" in n][-1]
t0=int(rows[first]["Start_Timestamp"])
for r in rows[first:first+40]:
    n=r["Kernel_Name"].replace("okvfe::(anonymous namespace)::","").replace("void ","").split("(")[0][:44]
    print("%-46s start %8.1f us  dur %8.1f us  grid %s"%(n,(int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r.get("Grid_Size_X","?")))
PY
