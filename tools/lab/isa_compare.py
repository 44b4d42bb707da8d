#!/usr/bin/env python3
"""LAB (round 6): are two device-assembly files (hipcc -S --offload-device-only) of one translation unit the same code?  Per kernel base
name the multiset of instruction streams is compared twice: verbatim, and with immediates / offsets / labels blanked (a kernel whose
argument list lost a parameter reads its remaining arguments at other kernarg offsets: the streams then differ in those immediates only).
usage: isa_compare.py before.s after.s"""
import collections
import hashlib
import re
import sys


def kernels(path):
    out, cur = {}, None
    for l in open(path):
        m = re.match(r'^(_ZN\w+):', l)
        if m:
            cur = []
            out[m.group(1)] = cur
            continue
        if l.startswith('.Lfunc_end'):
            cur = None
            continue
        if cur is None:
            continue
        t = l.split(';')[0].strip()
        if t and not t.startswith('.') and not t.endswith(':'):
            cur.append(re.sub(r'\.LBB\d+_\d+', 'L', t))
    return out


def blank(v):
    o = []
    for t in v:
        t = re.sub(r'offset:\d+', 'offset:N', t)
        t = re.sub(r'\b0x[0-9a-f]+\b', 'IMM', t)
        o.append(re.sub(r'(?<![\w\[:])-?\d+\b', 'IMM', t))
    return o


def main(a_path, b_path):
    a, b = kernels(a_path), kernels(b_path)
    base = lambda k: re.match(r'_ZN7vattn_k\d+(\w+?)I', k).group(1) if re.match(r'_ZN7vattn_k\d+(\w+?)I', k) else k
    ga, gb = collections.defaultdict(list), collections.defaultdict(list)
    for k, v in a.items():
        ga[base(k)].append(v)
    for k, v in b.items():
        gb[base(k)].append(v)
    h = lambda v: hashlib.md5('\n'.join(v).encode()).hexdigest()[:8] + ':%d' % len(v)
    for name in sorted(set(ga) | set(gb)):
        la, lb = ga.get(name, []), gb.get(name, [])
        exact = sorted(map(h, la)) == sorted(map(h, lb))
        loose = sorted(h(blank(v)) for v in la) == sorted(h(blank(v)) for v in lb)
        print('%-34s %d -> %d instantiations, instructions %s : %s' % (name, len(la), len(lb), sorted(len(v) for v in lb),
              'IDENTICAL' if exact else 'identical up to immediates / offsets' if loose else 'DIFFERENT'))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
