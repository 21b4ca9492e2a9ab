#!/bin/bash
# one PMC pass over the headline launch shape (8 frames per launch): lane utilisation and instruction counts per frame
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmcq
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmcq -- python3 $R/bench.py --no-cpu-baseline --no-diagnostics --steps 64 --warmup 32 --streams 1 ${PMCQ_ARGS} > $R/gpurun_out/pmcq.log 2>&1
python3 - <<'PY'
import csv,glob,collections,os
R=os.environ['GRAFT_REPO_ROOT']
f=glob.glob(R+'/gpurun_out/pmcq/*/*_counter_collection.csv')[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'k_trace_stack' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
m={k:sum(v)/len(v) for k,v in agg.items()}
n=len(next(iter(agg.values())))
print("launches",n)
for k,v in sorted(m.items()): print(k,'%.4g'%v)
print('lane utilisation %.3f'%(m['SQ_THREAD_CYCLES_VALU']/(64*m['SQ_ACTIVE_INST_VALU'])))
print('wait_any %.3f wait_inst %.3f active %.3f'%(m['SQ_WAIT_ANY']/m['SQ_WAVE_CYCLES'],m['SQ_WAIT_INST_ANY']/m['SQ_WAVE_CYCLES'],m['SQ_ACTIVE_INST_ANY']/m['SQ_WAVE_CYCLES']))
PY
