"""Join PMC passes by dispatch order for the conv kernels of one step; print per-dispatch derived metrics."""
import csv, sys, collections, re
d = sys.argv[1]
def load(i):
  rows = list(csv.DictReader(open('%s/p%d_counter_collection.csv' % (d, i))))
  out = collections.OrderedDict()
  for r in rows:
    key = int(r['Dispatch_Id'])
    e = out.setdefault(key, {'name': re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', ''), 'grid': r.get('Grid_Size', '')})
    e[r['Counter_Name']] = float(r['Counter_Value'])
  return out
def durations(i):
  rows = list(csv.DictReader(open('%s/p%d_kernel_trace.csv' % (d, i))))
  return {int(r['Dispatch_Id']): (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows}
P = [load(i) for i in (1, 2, 3, 4)]
D = durations(1)
# dispatch ids should align between passes (same program); take last step: after the second-to-last adam
ids = list(P[0].keys())
adam = [i for i in ids if P[0][i]['name'].startswith('adam_kernel')]
if len(adam) >= 2:
  lo, hi = adam[-2], adam[-1]
else:      # round 6: the single-GPU step ends with adam_segments_kernel (twice per step): a step = from one input-stage dispatch to the next
  starts = [i for i in ids if 'dynimg' in P[0][i]['name'] or 'window' in P[0][i]['name']]
  first = P[0][starts[0]]['name']
  starts = [i for i in starts if P[0][i]['name'] == first]
  lo, hi = starts[-2] - 1, starts[-1] - 1
print('%-34s %8s %6s %6s %6s %6s %7s %8s %8s %6s %6s' % ('kernel', 'us', 'mfma%', 'wait%', 'winst%', 'valu%', 'ldsbc%', 'rdMB', 'wrMB', 'TB/s', 'clkGHz'))
for i in ids:
  if not (lo < i <= hi): continue
  a = P[0][i]
  if not (a['name'].startswith('conv') or a['name'].startswith('dynimg') or a['name'].startswith('heads')): continue
  us = D.get(i, 0)
  wc = a.get('SQ_WAVE_CYCLES', 1)
  busy = a.get('SQ_BUSY_CYCLES', 1)
  b = P[1].get(i, {}); c = P[2].get(i, {})
  rd = b.get('FETCH_SIZE', 0) * 1024 * 2 / 1e6   # KB units; x2 gfx950 correction for wide reads
  wr = c.get('WRITE_SIZE', 0) * 1024 / 1e6
  gui = b.get('GRBM_GUI_ACTIVE', 0)
  # MFMA busy cycles are per-SIMD cycles summed; utilisation vs (busy cycles * 4 SIMD * CUs)? use ratio to wave cycles*4 as rough
  mf = a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0)
  print('%-34s %8.1f %6.1f %6.1f %6.1f %6.1f %7.2f %8.1f %8.1f %6.2f %6.2f' % (
      a['name'][:34], us, 100 * mf / (us * 1e-6 * 2.4e9 * 1024) if us else 0, 100 * a.get('SQ_WAIT_ANY', 0) / wc, 100 * a.get('SQ_WAIT_INST_ANY', 0) / wc,
      100 * a.get('SQ_ACTIVE_INST_VALU', 0) / wc, 100 * a.get('SQ_LDS_BANK_CONFLICT', 0) / max(a.get('SQ_LDS_IDX_ACTIVE', 1), 1),
      rd, wr, (rd + wr) / us if us else 0,      # MB / us = TB/s
       gui / 8 / (us * 1e3) if us else 0))
