set -x
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc1 -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /tmp/pmc1.log 2>&1
tail -2 /tmp/pmc1.log
f=$(find /tmp/pmc1 -name '*counter_collection.csv' | head -1)
python $R/tools/pmc_summary.py $f $R/gpurun_out/pmc_all_sq.json > /dev/null
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM --output-format csv -d /tmp/pmc2 -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /tmp/pmc2.log 2>&1
tail -2 /tmp/pmc2.log
f=$(find /tmp/pmc2 -name '*counter_collection.csv' | head -1)
python $R/tools/pmc_summary.py $f $R/gpurun_out/pmc_all_sq2.json > /dev/null
ls -la $R/gpurun_out
