"""Static instruction count of the march step's loop body in k_trace_stack<10,...> (runs anywhere).
usage: python scripts/isa_step.py <device-hip-amdgcn-amd-amdhsa-gfx950.s> [...]
The step loop = the smallest loop (backward branch) that contains the descent's v_med3_i32."""
import re, sys, collections

def body(path, sym='_ZN3svo13k_trace_stackILi10E'):
    L = open(path).read().split('\n')
    s = [i for i, l in enumerate(L) if l.startswith(sym) and l.rstrip().endswith(':') or (l.startswith(sym) and ': ' in l)][0]
    e = [i for i in range(s, len(L)) if 's_endpgm' in L[i]][0]
    K = L[s:e]
    labels = {}
    for i, l in enumerate(K):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m: labels[m.group(1)] = i
    med = [i for i, l in enumerate(K) if 'v_med3_i32' in l][0]
    best = None
    for i, l in enumerate(K):
        m = re.match(r'\s*s_c?branch\w*\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] <= med < i:
            span = (labels[m.group(1)], i)
            if best is None or span[1] - span[0] < best[1] - best[0]: best = span
    k = collections.Counter()
    for l in K[best[0]:best[1] + 1]:
        t = l.strip()
        if not t or t[0] in '.;' or t.endswith(':'): continue
        k[t.split()[0]] += 1
    return k, best

for p in sys.argv[1:]:
    k, span = body(p)
    v = sum(c for o, c in k.items() if o.startswith('v_')); sa = sum(c for o, c in k.items() if o.startswith('s_'))
    mem = sum(c for o, c in k.items() if o.startswith(('global_', 'ds_', 'buffer_', 'scratch_', 'flat_')))
    print(f"{p.split('/')[-2]:12s} step loop lines {span[0]}-{span[1]}: total {v+sa+mem}  VALU {v}  SALU {sa} (s_nop {k['s_nop']}, exec ops {sum(c for o,c in k.items() if 'exec' in o or o in ('s_or_b64','s_and_b64','s_andn2_b64','s_xor_b64'))}, branches {sum(c for o,c in k.items() if 'branch' in o)}, waitcnt {k['s_waitcnt']})  mem {mem}  v_mov {k['v_mov_b32_e32']+k['v_mov_b64_e32']}  cndmask {k['v_cndmask_b32_e32']+k['v_cndmask_b32_e64']}  scratch {sum(c for o,c in k.items() if o.startswith('scratch_'))}")
