"""Instruction counts per basic block of one kernel in a hipcc -save-temps .s file (VALU / SALU / memory+LDS, branches): what a loop costs per trip.
    python tools/ubench/isa_blocks.py file.s kernel_name_substring [first_label]"""
import re, sys
txt = open(sys.argv[1]).read().split('\n')
name = sys.argv[2]
st = [i for i, l in enumerate(txt) if re.match(r'^_Z\w*' + re.escape(name) + r'\w*:', l)][0]
blk = 'entry'; order = [blk]; C = {blk: [0, 0, 0, [], '']}
for l in txt[st + 1:]:
    if 's_endpgm' in l: break
    m = re.match(r'^(\.LBB\d+_\d+):(.*)', l)
    if m:
        blk = m.group(1); order.append(blk); C[blk] = [0, 0, 0, [], m.group(2).strip()[:60]]; continue
    t = l.strip()
    if not t or t[0] in ';.': continue
    op = t.split()[0]
    if op.startswith('v_'): C[blk][0] += 1
    elif op.startswith('s_cbranch') or op.startswith('s_branch'): C[blk][3].append(t.split()[-1])
    elif op.startswith('s_'): C[blk][1] += 1
    else: C[blk][2] += 1
on = len(sys.argv) <= 3
for b in order:
    if not on and b == sys.argv[3]: on = True
    if on: print('%-12s valu %3d salu %3d mem %2d  -> %s  %s' % (b, C[b][0], C[b][1], C[b][2], ','.join(C[b][3]), C[b][4]))
