"""Per-source-line VALU instruction histogram of k_trace_stack<10,...> from a -gline-tables-only .s file."""
import re, collections, sys
S = sys.argv[1]
thr = int(sys.argv[2]) if len(sys.argv) > 2 else 6
lines = open(S).read().split('\n')
start = [i for i, l in enumerate(lines) if l.startswith('_ZN3svo13k_trace_stackILi10E') and ': ' in l or l.startswith('_ZN3svo13k_trace_stackILi10E') and l.endswith(':')][0]
end = [i for i in range(start, len(lines)) if 's_endpgm' in lines[i]][0]
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m: files[int(m.group(1))] = (m.group(3) or m.group(2))
cur = None
cnt = collections.Counter(); kinds = collections.Counter(); salu = 0
for l in lines[start:end]:
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if m: cur = (files.get(int(m.group(1)), '?').split('/')[-1], int(m.group(2))); continue
    t = l.strip()
    if not t or t.startswith(('.', ';', '//')) or t.endswith(':'): continue
    op = t.split()[0]
    if op.startswith('v_'): cnt[cur] += 1; kinds[op] += 1
    elif op.startswith('s_'): salu += 1
print('total VALU', sum(cnt.values()), 'SALU', salu)
byfile = collections.Counter()
for (f, ln), c in cnt.items(): byfile[f] += c
print(dict(byfile))
import os
d = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'octree-raymarcher_amd', 'csrc')
for fn in ('kernel_stack.hip.h', 'march.hip.h'):
    src = open(os.path.join(d, fn)).read().split('\n')
    print('----', fn)
    for (f, ln), c in sorted(cnt.items(), key=lambda x: x[0][1] if x[0] else 0):
        if f == fn and c >= thr: print(f"{ln:4d} {c:4d}  {src[ln-1].strip()[:120]}")
print(kinds.most_common(45))
