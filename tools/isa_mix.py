"""Instruction mix per basic block of one kernel in a hipcc -S dump: python tools/isa_mix.py file.s kernel_substring"""
import sys, collections
txt = open(sys.argv[1]).read()
i = txt.index(sys.argv[2] + ":") if (sys.argv[2] + ":") in txt else txt.index(sys.argv[2])
body = txt[i:]
body = body[:body.index('s_endpgm')]
lines = [l.split(';')[0].strip() for l in body.split('\n')]
lines = [l for l in lines if l and (l.endswith(':') or not l.startswith('.'))]
print(len(lines), 'lines')
blocks = []
for l in lines:
    if l.endswith(':'):
        blocks.append([l, collections.Counter()]); continue
    if not blocks: continue
    blocks[-1][1][l.split()[0]] += 1
for name, c in blocks:
    if sum(v for k, v in c.items() if k.startswith('v_mfma')) > 0:
        print(name, sum(c.values()), dict(c.most_common(40)))
