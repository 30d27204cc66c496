#!/bin/bash
# SQ instruction counters of one bench-frame launch for each library in $LIBS (A/B of kernel variants)
export TMPDIR=/tmp
for lib in ${LIBS:-ab/*.so}; do
  n=$(basename $lib .so)
  RT_LIB=$lib RT_SAH=1 RT_CULL=1 timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmc_ab/$n -o $n -- python3 tools/prof_run.py ${PASSES:-64} > gpurun_out/pmc_ab/$n.log 2>&1 || { echo "$n failed"; tail -3 gpurun_out/pmc_ab/$n.log; }
done
python3 - <<'PY'
import csv,glob,collections
for f in sorted(glob.glob('gpurun_out/pmc_ab/**/*counter_collection.csv', recursive=True)):
    agg=collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if 'k_trace' in r['Kernel_Name']: agg[r['Counter_Name']]+=float(r['Counter_Value'])
    print(f.split('/')[2], {k: f'{v:.4g}' for k,v in sorted(agg.items())}, 'lane util %.1f%%' % (100*agg['SQ_THREAD_CYCLES_VALU']/(64*agg['SQ_ACTIVE_INST_VALU']) if agg.get('SQ_ACTIVE_INST_VALU') else 0))
PY
