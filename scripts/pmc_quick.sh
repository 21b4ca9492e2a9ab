R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/pmcq -- python3 $R/bench.py --no-cpu-baseline --steps 8 --warmup 4 --streams 1 > $R/gpurun_out/pmcq.log 2>&1
python3 - <<'PY'
import csv,glob,collections,os
R=os.environ['GRAFT_REPO_ROOT']
f=glob.glob(R+'/gpurun_out/pmcq/*/*_counter_collection.csv')[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'k_trace_stack' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()): print(k,'%.4g'%(sum(v)/len(v)))
PY
