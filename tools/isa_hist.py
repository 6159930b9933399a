#!/usr/bin/env python3
"""Instruction mix of the loops of one kernel in a hipcc -S listing: isa_hist.py file.s mangled_kernel_name [min_mfma]"""
import re, sys, collections
txt = open(sys.argv[1]).read().split('\n')
name = sys.argv[2]
minm = int(sys.argv[3]) if len(sys.argv) > 3 else 16
a = next(i for i, l in enumerate(txt) if l.startswith(name + ':'))
b = next(i for i in range(a, len(txt)) if 's_endpgm' in txt[i])
lines = txt[a:b + 1]
labels = {}
for i, l in enumerate(lines):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = i
loops = []
for i, l in enumerate(lines):
    m = re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)', l)
    if m and labels.get(m.group(1), 1e9) < i: loops.append((labels[m.group(1)], i))
def hist(x, y):
    c = collections.Counter()
    for l in lines[x:y + 1]:
        l = l.strip()
        if not l or l.startswith(';') or l.startswith('.'): continue
        op = l.split()[0]
        if op.startswith('v_mfma'): c['MFMA'] += 1
        elif op.startswith('v_pk'): c['VALU_pk'] += 1
        elif op.startswith('v_'): c['VALU:' + op] += 1
        elif op.startswith('ds_'): c['LDS:' + op] += 1
        elif op.startswith('s_'): c['SALU'] += 1
        elif op.startswith('global') or op.startswith('buffer'): c['VMEM'] += 1
    return c
for x, y in loops:
    h = hist(x, y)
    if h['MFMA'] >= minm:
        valu = sum(v for k, v in h.items() if k.startswith('VALU'))
        print(f"lines {x}-{y}: MFMA {h['MFMA']}  VALU {valu}  {dict(h)}")
