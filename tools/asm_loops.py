"""Instruction mix of the loops of one kernel in a hipcc -S listing: python tools/asm_loops.py FILE.s KERNEL_PREFIX [min_insts]."""
import re, sys, collections
txt = open(sys.argv[1]).read().split('\n')
pref = sys.argv[2]; mn = int(sys.argv[3]) if len(sys.argv) > 3 else 150
start = next(i for i, l in enumerate(txt) if l.startswith(pref) and ':' in l)
end = next(i for i in range(start, len(txt)) if txt[i].startswith('.Lfunc_end'))
lines = txt[start:end]
labels = {}
for i, l in enumerate(lines):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = i
loops = []
for i, l in enumerate(lines):
    m = re.search(r's_cbranch\w*\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)', l)
    if m:
        t = m.group(1) or m.group(2)
        if t in labels and labels[t] < i: loops.append((labels[t], i))
print('kernel lines', len(lines), 'loops', loops)
for a, b in loops:
    c = collections.Counter()
    for l in lines[a:b]:
        l = l.strip()
        if not l or l.startswith(('.', ';')) or l.endswith(':'): continue
        c[l.split()[0]] += 1
    n = sum(c.values())
    if n < mn: continue
    g = lambda p, ex=(): sum(v for k, v in c.items() if k.startswith(p) and not k.startswith(ex) if True)
    print('loop', a, b, 'insts', n, ' valu', sum(v for k, v in c.items() if k.startswith('v_') and not k.startswith('v_mfma')),
          'mfma', g('v_mfma'), 'ds', g('ds_'), 'vmem', g('global') + g('buffer') + g('scratch'), 'salu', g('s_'))
    print('   ', ', '.join('%s %d' % kv for kv in c.most_common(45)))
