#!/bin/bash
# GPU box: kernel split of tools/bench_scalespace.py (env passes through, e.g. SS_UPRIGHT=1, OKVFE_LIB=...)
R=$PWD; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/ss
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ss -o p -- python $R/tools/bench_scalespace.py 512 2 > /tmp/ss.log 2>&1
tail -1 /tmp/ss.log
python - <<'PY'
import csv,glob
f=glob.glob("/tmp/ss/**/*kernel_stats.csv",recursive=True)[0]
for x in list(csv.DictReader(open(f)))[:14]:
    print("%-50s calls %4s avg %9.1f us"%(x["Name"].replace("okvfe::(anonymous namespace)::","")[:50], x["Calls"], float(x["AverageNs"])/1e3))
PY
