#!/usr/bin/env python3
"""Static instruction mix of one kernel in a hipcc -S listing.  Usage: isa_count.py <file.s> <mangled-name-prefix>"""
import sys
from collections import Counter
path, prefix = sys.argv[1], sys.argv[2]
s = open(path).read()
names = [l.split(':')[0] for l in s.split('\n') if l.startswith(prefix) and ':' in l]
for name in names[:1]:
    i = s.index('\n' + name + ':')
    j = s.index('s_endpgm', i)
    c = Counter()
    for l in s[i:j].split('\n'):
        t = l.split()
        if not l.startswith('\t') or not t or t[0].startswith(('.', ';')):
            continue
        x = t[0]
        k = 'v_pk' if x.startswith('v_pk') else 'valu' if x.startswith('v_') else 'salu' if x.startswith('s_') else 'lds' if x.startswith('ds_') else 'vmem' if x.startswith(('global', 'buffer', 'flat')) else 'other'
        c[k] += 1
    print(name[:70], dict(c))
k = s.index('.name:           ' + names[0]) if names else -1
if k >= 0:
    print([l.strip() for l in s[k:k + 600].split('\n') if 'vgpr_count' in l or 'sgpr_count' in l])
