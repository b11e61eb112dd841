#!/bin/bash
# Counter passes over the in-tree LSTM projection GEMMs alone (tools/gemm_bench.py <rows> eager).  usage: tools/gemm_pmc.sh <tag> <rows>
set -e -o pipefail
tag=$1; rows=$2
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag
mkdir -p $out && cd /tmp && export TMPDIR=/tmp
declare -A G
G[wait]="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU"
G[inst]="SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
G[mem]="SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"
G[tcp]="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
G[tcc]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
G[ta]="TA_BUSY_avr TA_TA_BUSY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"
for g in wait inst mem tcp tcc ta; do
  if timeout -k 10 200 rocprofv3 --pmc ${G[$g]} --kernel-trace --output-format csv -d $out/p_$g -- python3 $R/tools/gemm_bench.py $rows eager > $out/pmc_$g.log 2>&1; then
    cp $(find $out/p_$g -name "*counter_collection.csv" | head -1) $out/pmc_$g.csv
  else
    echo "[pmc] group $g failed"; tail -5 $out/pmc_$g.log
  fi
  rm -rf $out/p_$g
done
python3 $R/tools/pmc_wait_states.py $out gemm
