# counters of ONE linear kernel form on ONE BEATs layer shape: bash tools/pmc_linear_pp.sh <tag> [form=pp] [shape=qkv]
tag=${1:-x}; form=${2:-pp}; shape=${3:-qkv}; export TMPDIR=/tmp; mkdir -p gpurun_out
i=0
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" \
            "SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
            "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_avr" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  SHAPE=$shape timeout 200 rocprofv3 --kernel-trace --pmc $pass -d gpurun_out/pmcpp_${tag}_$i -o p -- python tools/linear_bench.py $form > gpurun_out/pmcpp_${tag}_$i.log 2>&1
done
python tools/pmc_wait_summary.py gpurun_out/pmcpp_${tag}_*/p_results.db > gpurun_out/pmcpp_${tag}.md 2>&1
grep -E "linear|gemm|kernel \|" gpurun_out/pmcpp_${tag}.md | cut -c1-1500
rm -rf gpurun_out/pmcpp_${tag}_[0-9]
