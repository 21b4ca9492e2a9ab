R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmcl2
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $R/gpurun_out/pmcl2 -- python3 $R/bench.py --no-cpu-baseline --steps 4 --warmup 1 --streams 1 > $R/gpurun_out/pmcl2.log 2>&1
python3 - <<'PY'
import csv,glob,collections,os
R=os.environ['GRAFT_REPO_ROOT']
f=glob.glob(R+'/gpurun_out/pmcl2/*/*_counter_collection.csv')[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'k_trace_stack' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
m={k:sum(v)/len(v) for k,v in agg.items()}
for k,v in sorted(m.items()): print(k,'%.4g'%v)
if 'TCC_HIT_sum' in m: print('L2 hit rate %.3f'%(m['TCC_HIT_sum']/(m['TCC_HIT_sum']+m['TCC_MISS_sum'])))
PY
