#!/bin/bash
# GPU box: rocprofv3 kernel stats of the default bench command with the LAB library under environment variants.
# usage: bash tools/lab/kstats_env.sh "-" "ENV=VAL[,ENV=VAL]" ...   (top 6 kernels + fps per variant)
R=$PWD; cd /tmp && export TMPDIR=/tmp
export OKVFE_LIB=$R/okvis2_amd/libokvfe_lab.so
i=0
for a in "$@"; do
  i=$((i+1)); envs=""; [ "$a" != "-" ] && envs=$(echo "$a" | tr ',' ' ')
  rm -rf /tmp/kse$i
  env $envs rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kse$i -o p -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras ${BENCH_ARGS:-} > /tmp/kse$i.log 2>&1
  python - "$a" /tmp/kse$i /tmp/kse$i.log <<'PY'
import csv,glob,sys,json
f=glob.glob(sys.argv[2]+"/**/*kernel_stats.csv", recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "at::native" not in r["Name"] and "copyBuffer" not in r["Name"]][:7]
fps=""
for l in open(sys.argv[3]):
    if l.startswith('{"metric"'):
        r=json.loads(l); fps="fps %d ms %.3f"%(r["value"], r["ms_per_step"])
print(sys.argv[1], fps, " ".join("%s=%.1f" % (r["Name"].replace("okvfe::(anonymous namespace)::","").replace("void ","").split("(")[0][:24], float(r["AverageNs"])/1e3) for r in rows))
PY
done
