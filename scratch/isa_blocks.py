"""Basic blocks of a kernel from the compiled ISA (/tmp/lt_isa.s, written by isa_timeline.py): instruction mix per block, loop depth.
usage: python scratch/isa_blocks.py <mangled-name-substring> [min_depth]"""
import re, sys
src = open('/tmp/lt_isa.s').read().split('\n')
starts = [i for i, l in enumerate(src) if re.match(r'^_Z\w+:', l)]
beg = [i for i in starts if sys.argv[1] in src[i]][0]
end = min([i for i in starts if i > beg] + [len(src)])
body = src[beg:end]
open('/tmp/kernel.s', 'w').write('\n'.join(body))
mind = int(sys.argv[2]) if len(sys.argv) > 2 else 0
cur = None; info = []
for l in body:
    s = l.strip()
    m = re.match(r'^(\.LBB\d+_\d+):', s)
    if m:
        d = re.search(r'Depth=(\d+)', l)
        cur = {'name': m.group(1), 'n': 0, 'valu': 0, 'salu': 0, 'ds': 0, 'wait': 0, 'vmem': 0, 'br': [], 'depth': int(d.group(1)) if d else 0}; info.append(cur); continue
    if cur and s and not s.startswith((';', '.')):
        op = s.split()[0]; cur['n'] += 1
        if op.startswith('v_'): cur['valu'] += 1
        elif op.startswith('s_waitcnt'): cur['wait'] += 1
        elif op.startswith(('s_cbranch', 's_branch')): cur['br'].append(s.split()[-1]); cur['salu'] += 1
        elif op.startswith('s_'): cur['salu'] += 1
        elif op.startswith('ds_'): cur['ds'] += 1
        elif op.startswith(('global_', 'scratch_', 'buffer_')): cur['vmem'] += 1
tot = {}
for c in info:
    if c['depth'] >= mind:
        print(f"{c['name']:12s} depth {c['depth']} n {c['n']:4d} valu {c['valu']:4d} salu {c['salu']:3d} ds {c['ds']:3d} wait {c['wait']:2d} vmem {c['vmem']:3d} -> {c['br']}")
        t = tot.setdefault(c['depth'], dict(n=0, valu=0, salu=0, ds=0, wait=0, vmem=0, blocks=0))
        for k in ('n', 'valu', 'salu', 'ds', 'wait', 'vmem'): t[k] += c[k]
        t['blocks'] += 1
print(tot)
