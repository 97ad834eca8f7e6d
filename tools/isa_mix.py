"""Instruction mix per basic block of one kernel in `hipcc -S` output (developer tool).
python tools/isa_mix.py file.s <kernel name substring> [min_mfma]   -- prints the blocks that hold MFMAs, and totals for
the blocks inside backward-branch loops."""
import re
import sys
from collections import Counter

src, key = sys.argv[1], sys.argv[2]
min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 1
lines = open(src).read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\S*:', l) and key in l)
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith('.Lfunc_end'))
blocks, cur, name = [], [], 'entry'
for l in lines[start + 1:end]:
    m = re.match(r'^(\.LBB\S+):', l)
    if m:
        blocks.append((name, cur))
        name, cur = m.group(1), []
        continue
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.'):
        continue
    cur.append(t.split()[0])
blocks.append((name, cur))


def cls(op):
    if op.startswith('v_mfma'):
        return 'mfma'
    if op.startswith('v_accvgpr'):
        return 'acc_mov'
    if op.startswith('v_cvt'):
        return 'cvt'
    if op.startswith('v_'):
        return 'valu'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith('global_') or op.startswith('buffer_') or op.startswith('scratch_') or op.startswith('flat_'):
        return 'vmem:' + op.split('_')[1]
    if op.startswith('s_waitcnt'):
        return 'waitcnt'
    if op.startswith('s_barrier'):
        return 'barrier'
    if op.startswith('s_nop'):
        return 'nop'
    if op.startswith('s_'):
        return 'salu'
    return 'other'


tot = Counter()
for name, ops in blocks:
    c = Counter(cls(o) for o in ops)
    tot.update(c)
    if c['mfma'] >= min_mfma:
        print('%-14s %5d instr: ' % (name, len(ops)) + ', '.join('%s %d' % kv for kv in sorted(c.items())))
        vops = Counter(o for o in ops if cls(o) in ('valu', 'cvt', 'acc_mov'))
        print('      valu: ' + ', '.join('%s %d' % kv for kv in vops.most_common(14)))
print('TOTAL: ' + ', '.join('%s %d' % kv for kv in sorted(tot.items())))
