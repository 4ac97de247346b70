"""Tail of one data-parallel step (from conv2's filter gradient to the last Adam launch) out of a rocprofv3 kernel trace of dp_trace.py."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('void dynimg_goal_onepass')]
a, b = starts[-3], starts[-2]
t0 = int(rows[a]['Start_Timestamp'])
show = len(sys.argv) > 2
prev = None
for r in rows[a:b]:
  name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')
  if 'halo_wgrad' in name:
    show = True
  if show:
    gap = (int(r['Start_Timestamp']) - prev) / 1e3 if prev is not None else 0.0
    print('%9.1f us  @%8.1f .. %8.1f  gap %5.1f  %s' % ((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, (int(r['Start_Timestamp']) - t0) / 1e3,
                                            (int(r['End_Timestamp']) - t0) / 1e3, gap, name[:70]))
  prev = int(r['End_Timestamp'])
print('step: %.1f us from the first kernel to the next step\'s first kernel' % ((int(rows[b]['Start_Timestamp']) - t0) / 1e3))
