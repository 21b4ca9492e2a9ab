"""Static instruction counts of k_trace_stack<10,...,false> per source region of the pass loop, from a -gline-tables-only .s
(hipcc ... -gline-tables-only --cuda-device-only -S csrc/device.hip).  Inlined callees are attributed to the kernel_stack.hip.h
line of their call site where the .loc carries it, else to their own file.  Runs anywhere."""
import collections, re, sys, os
S = sys.argv[1]
lines = open(S).read().split('\n')
name = '_ZN3svo13k_trace_stackILi10ELi8ELi6ELb0ELb0EEEvNS_9TraceArgsE'
start = [i for i, l in enumerate(lines) if l.startswith(name + ':')][0]
end = [i for i in range(start, len(lines)) if 's_endpgm' in lines[i]][0]
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m: files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'octree-raymarcher_amd', 'csrc', 'kernel_stack.hip.h')).read().split('\n')
def find(text):
    return [i + 1 for i, l in enumerate(src) if text in l][0]
marks = [("prologue", 1), ("refill", find("==== refill retired lanes")), ("votes + guard", find("==== votes: which of the rare blocks")),
         ("chunk step", find("---- chunk step: src/Traverse.cpp:142-156")), ("step (asm call + loop control)", find("---- one step of the current level")),
         ("creep block", find("#include \"creep_block.inc\"")), ("hit blocks", find("---- hits.  A shadow ray only sets a flag")), ("epilogue", find("unsigned total = rays_marched;"))]
def region(ln):
    r = marks[0][0]
    for nm, first in marks:
        if ln >= first: r = nm
    return r
cur = None
tot = collections.Counter(); kinds = collections.defaultdict(collections.Counter)
for l in lines[start:end]:
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if m:
        f = files.get(int(m.group(1)), '?'); ln = int(m.group(2))
        cur = region(ln) if f == 'kernel_stack.hip.h' else ('(' + f + ')')
        continue
    t = l.strip()
    if not t or t.startswith(('.', ';', '//')) or t.endswith(':'): continue
    op = t.split()[0]
    cls = 'valu' if op.startswith('v_') else 'salu' if op.startswith('s_') else 'mem' if op.startswith(('global_', 'ds_', 'scratch_', 'buffer_', 'flat_')) else 'other'
    tot[cur] += 1; kinds[cur][cls] += 1
for k, v in sorted(tot.items(), key=lambda x: -x[1]):
    print(f"{v:6d}  {str(k):40s} {dict(kinds[k])}")
print("total", sum(tot.values()))
