"""Print the kernels of one training step (between two adam_kernel dispatches, or two input-stage dispatches) from a rocprofv3 kernel trace."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('adam_kernel')]
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
if len(idx) >= 2:
  a, b = idx[which - 1] + 1, idx[which] + 1
else:
  # the single-GPU step of round 6 ends with adam_segments_kernel (twice per step): a step = from one input-stage dispatch to the next
  first = next(r['Kernel_Name'] for r in rows if 'dynimg' in r['Kernel_Name'] or 'window' in r['Kernel_Name'])
  idx = [i for i, r in enumerate(rows) if r['Kernel_Name'] == first]
  which = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
  a, b = idx[which - 1], idx[which]
tot = 0
t0 = int(rows[a]['Start_Timestamp'])
agg = {}
prev_end = None
gaps = 0.0
for r in rows[a:b]:
  d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
  tot += d
  name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')
  g = (r.get('Grid_Size_X') or r.get('Grid_Size', '?'), r.get('Grid_Size_Y', ''), r.get('Grid_Size_Z', ''))
  gap = (int(r['Start_Timestamp']) - prev_end) / 1e3 if prev_end is not None else 0.0
  gaps += max(gap, 0.0)
  prev_end = int(r['End_Timestamp'])
  print('%9.1f us  @%8.1f  gap %5.1f  %-45s grid %s' % (d, (int(r['Start_Timestamp']) - t0) / 1e3, gap, name[:45], g))
  agg[name] = agg.get(name, 0) + d
print('sum of kernel time %.1f us; wall %.1f us; gaps between dispatches %.1f us' % (tot, (int(rows[b - 1]['End_Timestamp']) - t0) / 1e3, gaps))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]):
  print('%9.1f us %5.1f%%  %s' % (v, 100 * v / tot, k))
