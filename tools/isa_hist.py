#!/usr/bin/env python3
"""Instruction histogram + register counts of kernels in a hipcc -save-temps .s file.
usage: isa_hist.py file.s substring [substring ...]"""
import collections, re, sys
s = open(sys.argv[1]).read()
meta = s[s.find('amdhsa.kernels'):]
for pat in sys.argv[2:]:
    for m in re.finditer(r'^(_Z\S*%s\S*):.*\n' % re.escape(pat), s, re.M):
        name = m.group(1)
        body = s[m.end():s.find('s_endpgm', m.end())]
        c = collections.Counter()
        for l in body.splitlines():
            l = l.strip()
            if not l or l.startswith(('.', ';')) or l.endswith(':'):
                continue
            c[l.split()[0]] += 1
        k = meta.find('.name:           ' + name)
        blk = meta[max(0, k - 1500):k + 600]
        vg = re.findall(r'\.vgpr_count:\s+(\d+)', blk); sg = re.findall(r'\.sgpr_count:\s+(\d+)', blk)
        sp = re.findall(r'\.vgpr_spill_count:\s+(\d+)', blk)
        valu = sum(v for k2, v in c.items() if k2.startswith('v_'))
        print('%s\n  total %d, VALU %d, vgpr %s sgpr %s spill %s' % (name, sum(c.values()), valu, vg[-1:] , sg[-1:], sp[-1:]))
        print('  ' + ', '.join('%d×%s' % (v, k2) for k2, v in c.most_common(30)))
