#!/bin/bash
# Same-box A/B of the product library against the development build (= round 4's library content: 22 more kernels, 232 KB more
# device code): graph-replayed step traces of both, the small launch-bound kernels side by side.  VERDICT r04 #6 asks whether
# the small kernels' time is a cold instruction cache: if it were, the leaner library would show it here.
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/strip_ab
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for v in product dev; do
  if [ $v = dev ]; then export GEECO_DEV=1 GEECO_LIB=libgeeco_hip_dev.so; else unset GEECO_DEV GEECO_LIB; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/$v$rep -o r1 -- python3 $R/bench.py --steps 40 --warmup 10 --skip-cpu --skip-layers --skip-other-configs --skip-input-pipeline --skip-inference --skip-dp-one-rank > $out/$v$rep.json 2> $out/$v$rep.err || { echo "$v failed"; tail -5 $out/$v$rep.err; exit 1; }
done
done
unset GEECO_DEV GEECO_LIB
cd $R
python3 - $out <<'PY' | tee $out/report.txt
import csv, glob, json, re, statistics, sys
out = sys.argv[1]
small = ['heads_loss_lds_kernel', 'lstm_gates_fwd_slabs_kernel', 'lstm_gates_bwd_kernel', 'lstm_step_bwd_kernel', 'lstm_step_bwd_finish_kernel',
         'gemm_f32_kernel', 'conv_splitk_epilogue_kernel', 'conv_splitk_epilogue_state_kernel', 'adam_prepare_kernel', 'adam_kernel',
         'wgrad_reduce_batch_kernel', 'conv_top_bwd_kernel', 'dynimg_norm_kernel']
res = {}
for v in ('product', 'dev'):
  for rep in (1, 2):
    f = glob.glob('%s/%s%d/**/*kernel_trace.csv' % (out, v, rep), recursive=True)[0]
    per = {}
    for r in csv.DictReader(open(f)):
      n = re.sub(r'[<(].*', '', r['Kernel_Name']).replace('void ', '')
      per.setdefault(n, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    d = json.loads(open('%s/%s%d.json' % (out, v, rep)).read().strip().splitlines()[-1])
    res[(v, rep)] = (per, d['ms_per_step'], d['step_ms']['median'])
print('%-36s %10s %10s %10s %10s' % ('kernel (median us over the run)', 'product#1', 'dev#1', 'product#2', 'dev#2'))
for k in small:
  print('%-36s ' % k + ' '.join('%10.2f' % statistics.median(res[(v, rep)][0].get(k, [float('nan')])) for rep in (1, 2) for v in ('product', 'dev')))
tot = lambda per: sum(statistics.median(t) * (len(t) / max(len(per['adam_kernel']), 1)) for t in per.values())
print('%-36s ' % 'sum of kernel medians per step' + ' '.join('%10.1f' % tot(res[(v, rep)][0]) for rep in (1, 2) for v in ('product', 'dev')))
print('%-36s ' % 'bench ms/step (under the profiler)' + ' '.join('%10.4f' % res[(v, rep)][1] for rep in (1, 2) for v in ('product', 'dev')))
PY
find $out -name "*kernel_trace.csv" -delete; find $out -name "*agent_info.csv" -delete
