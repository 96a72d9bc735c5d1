#!/usr/bin/env python3
"""Locate the LIVE (last) top-level definition of each function name in a reference module.

Dev tool only (build container): prints 'name start end' or the source range of one name.
Usage: livedefs.py <module.py>            -> table
       livedefs.py <module.py> <name>     -> prints the live body with line numbers
"""
import re, sys

def scan(path):
    lines = open(path, encoding='utf-8', errors='replace').read().split('\n')
    defs = []  # (name, start_idx)
    for i, ln in enumerate(lines):
        m = re.match(r'def\s+([A-Za-z_0-9]+)\s*\(', ln)
        if m:
            defs.append((m.group(1), i))
    out = {}
    for k, (name, st) in enumerate(defs):
        # body ends before next top-level statement that is not indented / blank / comment / decorator
        en = len(lines)
        for j in range(st + 1, len(lines)):
            l = lines[j]
            if l and not l[0].isspace() and not l.startswith('#'):
                en = j
                break
        # include decorators above
        s = st
        while s > 0 and lines[s - 1].startswith('@'):
            s -= 1
        out[name] = (s + 1, en)  # 1-based start, end exclusive->inclusive line number en
    return lines, out

if __name__ == '__main__':
    lines, out = scan(sys.argv[1])
    if len(sys.argv) == 2:
        for n, (s, e) in sorted(out.items(), key=lambda kv: kv[1]):
            print(f'{n}\t{s}\t{e}')
    else:
        for name in sys.argv[2:]:
            s, e = out[name]
            for i in range(s, e + 1):
                if i - 1 < len(lines):
                    print(f'{i}\t{lines[i-1]}')
