#!/bin/bash
# L2 hit / miss counts per one-frame launch (mean over the 32-camera path) per variant library; usage (on the box): bash scripts/pmc_l2_ab.sh name1 name2 ...
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for name in "$@"; do
  rm -rf $R/gpurun_out/pmcl2_$name
  SVO_AMD_LIB=$R/octree-raymarcher_amd/build/libsvo_$name.so rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $R/gpurun_out/pmcl2_$name -- python3 $R/scripts/l2_probe.py > $R/gpurun_out/pmcl2_$name.log 2>&1
  python3 - "$name" <<'PY'
import csv,glob,collections,os,sys
R=os.environ['GRAFT_REPO_ROOT']; name=sys.argv[1]
f=glob.glob(f'{R}/gpurun_out/pmcl2_{name}/*/*_counter_collection.csv')
if not f: print(name,'no csv'); sys.exit(0)
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f[0])):
    if 'k_trace_stack' in r['Kernel_Name']: agg[r['Counter_Name']][r['Dispatch_Id']]+=float(r['Counter_Value'])
m={k:sum(v.values())/len(v) for k,v in agg.items()}
print('==',name,' '.join(f"{k}={v:.4g}" for k,v in sorted(m.items())),'L2 hit rate %.3f'%(m['TCC_HIT_sum']/(m['TCC_HIT_sum']+m['TCC_MISS_sum'])))
PY
  rm -rf $R/gpurun_out/pmcl2_$name
done
