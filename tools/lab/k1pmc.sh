#!/bin/bash
# K1 experiment helper (GPU box): SQ counters of the fused score kernel for the env settings given
# as arguments ("-" = defaults).  usage: bash tools/lab/k1pmc.sh "-" "OKVFE_K1_TH=31" ...
R=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
for v in "$@"; do
  i=$((i+1))
  [ "$v" = "-" ] && v="OKVFE_DUMMY=1"
  rm -rf /tmp/prof_sq$i
  env $v rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d /tmp/prof_sq$i -o p -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /tmp/prof_sq$i.log 2>&1
  echo "== $v"
  python $R/tools/pmc_summary.py $(find /tmp/prof_sq$i -name '*counter_collection.csv' | head -1) | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items():
    if 'harris_kernel' in k and v['dispatches']>3:
        m=v['mean_per_dispatch']; w=m['SQ_WAVES']
        print(k, 'vgpr',v['vgpr'],'waves',w,'VALU/wave',m['SQ_INSTS_VALU']/w, 'active_valu/wave_cycles',m['SQ_ACTIVE_INST_VALU']/m['SQ_WAVE_CYCLES'], 'wait_inst/wc', m['SQ_WAIT_INST_ANY']/m['SQ_WAVE_CYCLES'],'wait_any/wc',m['SQ_WAIT_ANY']/m['SQ_WAVE_CYCLES'],'busy_cyc',m['SQ_BUSY_CYCLES'],'gui',m['GRBM_GUI_ACTIVE'], 'act_valu', m['SQ_ACTIVE_INST_VALU'], 'wave_cycles', m['SQ_WAVE_CYCLES'])
"
done
