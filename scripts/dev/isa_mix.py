"""Instruction mix of one kernel of a device-only assembly listing (hipcc -S --cuda-device-only): per basic block that
contains MFMAs, the counts of MFMA / VALU / LDS / VMEM / SALU / waitcnt instructions.  On gfx950 the f32 MFMA shares the
vector ALU: every VALU instruction next to the MFMAs costs about 4 cycles of MFMA time (scripts/dev/ub/mfma_valu.hip)."""
import collections
import re
import sys


def cls(op):
  if op.startswith('v_mfma'): return 'mfma'
  if op.startswith('v_'): return 'valu'
  if op.startswith('ds_'): return 'lds'
  if op.startswith(('global_', 'buffer_', 'scratch_', 'flat_')): return 'vmem'
  if op.startswith('s_waitcnt'): return 'wait'
  if op.startswith('s_'): return 'salu'
  return 'other'


def main(path, pattern, verbose=False):
  src = open(path).read()
  for m in re.finditer(r'^(_Z\w*%s\w*):[^\n]*\n(.*?)\n\s*s_endpgm' % pattern, src, re.S | re.M):
    name, body = m.group(1), m.group(2).split('\n')
    blocks, cur, label = [], [], 'entry'
    for l in body:
      if re.match(r'^\.LBB\d+_\d+:', l):
        blocks.append((label, cur)); label, cur = l.strip(), []
      else:
        cur.append(l.strip())
    blocks.append((label, cur))
    print(name)
    tot = collections.Counter()
    for label, b in blocks:
      c = collections.Counter()
      ops = collections.Counter()
      for l in b:
        if not l or l.startswith((';', '.')): continue
        op = l.split()[0]
        c[cls(op)] += 1
        if cls(op) == 'valu': ops[op] += 1
      tot += c
      if c['mfma'] > 0:
        print('  %-12s %s' % (label, dict(c)))
        if verbose: print('      valu:', dict(ops.most_common(12)))
    print('  total       ', dict(tot))


if __name__ == '__main__':
  main(sys.argv[1], sys.argv[2], len(sys.argv) > 3)
