"""profiles/rNN/pmc_roofline_kernel.json from the PMC passes (scripts/dev/pmc.sh): HBM traffic per launch of the kernel
bench.py names in `roofline`, averaged over its launches in ONE step.  FETCH_SIZE is doubled (gfx950 reports half the
bytes of wide coalesced reads: MI355X_MICROARCH.md, HBM section); WRITE_SIZE as is; both in KiB units -> bytes.
usage: pmc_roofline.py <pmc dir> "<kernel name>" <out.json>"""
import csv, json, os, re, sys, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
d, kname, out = sys.argv[1], sys.argv[2], sys.argv[3]
def load(i):
  o = collections.OrderedDict()
  for r in csv.DictReader(open('%s/p%d_counter_collection.csv' % (d, i))):
    e = o.setdefault(int(r['Dispatch_Id']), {'name': re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')})
    e[r['Counter_Name']] = float(r['Counter_Value'])
  return o
F, W = load(2), load(3)
ids = list(F.keys())
adam = [i for i in ids if F[i]['name'].startswith('adam_kernel')]
if len(adam) >= 2:
  lo, hi = adam[-2], adam[-1]
else:      # round 6: the single-GPU step ends with adam_segments_kernel (twice per step): a step = from one input-stage dispatch to the next
  starts = [i for i in ids if 'dynimg' in F[i]['name'] or 'window' in F[i]['name']]
  first = F[starts[0]]['name']
  starts = [i for i in starts if F[i]['name'] == first]
  lo, hi = starts[-2] - 1, starts[-1] - 1
norm = lambda s: s.replace(' ', '')
sel = [i for i in ids if lo < i <= hi and norm(F[i]['name']) == norm(kname)]
assert sel, 'kernel %r not found in the last step' % kname
rd = [F[i].get('FETCH_SIZE', 0) * 1024 * 2 for i in sel]
wr = [W.get(i, {}).get('WRITE_SIZE', 0) * 1024 for i in sel]
rec = {'kernel': kname, 'launches_in_step': len(sel), 'fetch_bytes_x2': rd, 'write_bytes': wr,
       'traffic_bytes': int((sum(rd) + sum(wr)) / len(sel)),
       'csrc_sha16': __import__('bench').csrc_sha16(),      # the kernel sources these passes ran: bench.py calls the figure stale for any other
       'method': 'rocprofv3 --pmc, separate passes for FETCH_SIZE and WRITE_SIZE; FETCH_SIZE x 2 (gfx950 wide-read correction)'}
json.dump(rec, open(out, 'w'), indent=1)
print(rec)
