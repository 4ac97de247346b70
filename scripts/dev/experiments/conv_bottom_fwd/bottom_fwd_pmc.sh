#!/bin/bash
# SQ + FETCH/WRITE counters + effective clock of the encoder bottom forward: the one-launch kernel (conv_bottom_fwd.hip, this directory) and the
# two kernels it would replace (separate --pmc passes; GEECO_LIB selects a variant build).  out: gpurun_out/bfpmc/report.txt
R=$GRAFT_REPO_ROOT
export GEECO_DEV=1 GEECO_LIB=${GEECO_LIB:-libgeeco_hip_dev.so}
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/bfpmc${1:-}
rm -rf $out; mkdir -p $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE"; do
  i=$((i+1))
  BF_QUICK=1 timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out -o p$i -- python3 $R/scripts/dev/experiments/conv_bottom_fwd/bottom_fwd_bench.py > $out/log$i.txt 2>&1 || { echo "rocprof failed"; tail -5 $out/log$i.txt; exit 1; }
done
python3 - $out <<'PY' | tee $out/report.txt
import csv, sys, collections, re
d = sys.argv[1]
def load(i):
  out = collections.OrderedDict()
  for r in csv.DictReader(open('%s/p%d_counter_collection.csv' % (d, i))):
    e = out.setdefault(int(r['Dispatch_Id']), {'name': re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')})
    e[r['Counter_Name']] = float(r['Counter_Value'])
  return out
def dur(i):
  return {int(r['Dispatch_Id']): (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open('%s/p%d_kernel_trace.csv' % (d, i)))}
P = [load(i) for i in (1, 2, 3)]; D = [dur(i) for i in (1, 2, 3)]
last = collections.OrderedDict()
for i, e in P[0].items():
  if e['name'].startswith('conv'): last[e['name']] = i
print('%-44s %8s %6s %6s %6s %6s %7s %8s %8s %7s' % ('kernel (last dispatch)', 'us', 'mfma%', 'wait%', 'winst%', 'valu%', 'ldsbc%', 'rdMB', 'wrMB', 'clkGHz'))
for name, i in last.items():
  a = P[0][i]; us = D[0].get(i, 0); wc = a.get('SQ_WAVE_CYCLES', 1)
  us2 = D[1].get(i, us)
  print('%-44s %8.1f %6.1f %6.1f %6.1f %6.1f %7.2f %8.1f %8.1f %7.2f' % (name[:44], us, 100 * a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (us * 1e-6 * 2.4e9 * 1024) if us else 0,
    100 * a.get('SQ_WAIT_ANY', 0) / wc, 100 * a.get('SQ_WAIT_INST_ANY', 0) / wc, 100 * a.get('SQ_ACTIVE_INST_VALU', 0) / wc,
    100 * a.get('SQ_LDS_BANK_CONFLICT', 0) / max(a.get('SQ_LDS_IDX_ACTIVE', 1), 1), P[1].get(i, {}).get('FETCH_SIZE', 0) * 1024 * 2 / 1e6, P[2].get(i, {}).get('WRITE_SIZE', 0) * 1024 / 1e6,
    P[1].get(i, {}).get('GRBM_GUI_ACTIVE', 0) / 8 / (us2 * 1e3) if us2 else 0))
print('mfma% = SQ_VALU_MFMA_BUSY_CYCLES / (time x 2.4 GHz x 1024 SIMDs); clkGHz = GRBM_GUI_ACTIVE / 8 / time of the FETCH pass (reads high on sub-0.3 ms dispatches); rdMB = FETCH_SIZE x 2 (gfx950 wide-read correction)')
PY
