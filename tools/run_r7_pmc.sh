#!/bin/bash
# round 6: SQ counters of k_sweep_two with and without the measure-free periods
mkdir -p gpurun_out
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
P2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
for k in off on; do
  if [ $k = off ]; then export SBSIM_TWO_NO_SKIP=1; else unset SBSIM_TWO_NO_SKIP; fi
  echo "== measure-free periods $k" | tee -a gpurun_out/r7_pmc_two.txt
  tools/pmc_two_rows.sh "$P1" "$P2" 2>&1 | tee -a gpurun_out/r7_pmc_two.txt
done
