#!/bin/bash
# PMC pass per variant library (octree-raymarcher_amd/build/libsvo_<name>.so) over the headline launch shape, serialized:
# instruction counts, wave cycles, waits per k_trace_stack launch of 8 frames.  usage (on the box): bash scripts/pmc_ab.sh name1 name2 ...
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for name in "$@"; do
for pass in 1 2; do
  if [ $pass = 1 ]; then C="SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; else C="SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"; fi
  rm -rf $R/gpurun_out/pmc_$name.$pass
  SVO_AMD_LIB=$R/octree-raymarcher_amd/build/libsvo_$name.so rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_$name.$pass -- python3 $R/bench.py --no-cpu-baseline --no-diagnostics --steps 32 --warmup 16 --streams 1 > $R/gpurun_out/pmc_$name.$pass.log 2>&1
done
python3 - "$name" <<'PY'
import csv,glob,collections,os,sys
R=os.environ['GRAFT_REPO_ROOT']; name=sys.argv[1]
m={}
for p in (1,2):
    fs=glob.glob(f'{R}/gpurun_out/pmc_{name}.{p}/*/*_counter_collection.csv')
    if not fs: print(name,'pass',p,'no csv'); continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if 'k_trace_stack' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items(): m[k]=sum(v)/len(v)
print('==',name,' '.join(f"{k}={v:.4g}" for k,v in sorted(m.items())))
if 'SQ_ACTIVE_INST_VALU' in m: print('   lane utilisation %.3f  wait_any %.3f wait_inst %.3f active_any %.3f'%(m['SQ_THREAD_CYCLES_VALU']/(64*m['SQ_ACTIVE_INST_VALU']),m['SQ_WAIT_ANY']/m['SQ_WAVE_CYCLES'],m['SQ_WAIT_INST_ANY']/m['SQ_WAVE_CYCLES'],m['SQ_ACTIVE_INST_ANY']/m['SQ_WAVE_CYCLES']))
PY
done
