#!/bin/bash
# PMC passes over a short eager bench run (separate passes: SQ set, FETCH_SIZE, WRITE_SIZE)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc -o p$i -- python3 $R/bench.py --steps 2 --warmup 1 --skip-cpu --skip-layers --skip-other-configs --skip-input-pipeline --skip-inference --skip-dp-one-rank --no-graph > $R/gpurun_out/pmc/p$i.log 2>&1
  echo "pass $i rc=$?"
done
ls $R/gpurun_out/pmc | head -20
