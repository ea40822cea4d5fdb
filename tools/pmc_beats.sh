# stall-oriented PMC passes over the BEATs extractor (tools/beats_bench.py): bash tools/pmc_beats.sh <tag>
tag=${1:-x}; export TMPDIR=/tmp; mkdir -p gpurun_out
for pass in "a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "b SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_CMD_FIFO_FULL" "c SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES SQ_INSTS_VALU" "m SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"; do
  set -- $pass; name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc $@ -d gpurun_out/pmcbeats_${tag}_$name -o $name -- python tools/beats_bench.py > gpurun_out/pmcbeats_${tag}_$name.log 2>&1
done
python tools/pmc_wait_summary.py gpurun_out/pmcbeats_${tag}_a/a_results.db gpurun_out/pmcbeats_${tag}_b/b_results.db gpurun_out/pmcbeats_${tag}_c/c_results.db gpurun_out/pmcbeats_${tag}_m/m_results.db > gpurun_out/pmcbeats_wait_${tag}.md 2>&1
grep -E "attention|linear_big|gemm_bf16x3|posconv|kernel \|" gpurun_out/pmcbeats_wait_${tag}.md | cut -c1-600
rm -rf gpurun_out/pmcbeats_${tag}_[abcm]
