# stall-oriented PMC passes over the eager step (separate runs per counter group; --kernel-trace only, as gpurun requires)
tag=${1:-x}; export TMPDIR=/tmp; mkdir -p gpurun_out
for pass in "a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "b SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_CMD_FIFO_FULL" "c SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES"; do
  set -- $pass; name=$1; shift
  timeout 400 rocprofv3 --kernel-trace --pmc $@ -d gpurun_out/pmcw_${tag}_$name -o $name -- python bench.py --surface driver --steps 2 --warmup 1 --no-graph --no-cpu-baseline --prefetch off > gpurun_out/pmcw_${tag}_$name.log 2>&1
done
python tools/pmc_wait_summary.py gpurun_out/pmcw_${tag}_a/a_results.db gpurun_out/pmcw_${tag}_b/b_results.db gpurun_out/pmcw_${tag}_c/c_results.db > gpurun_out/pmcw_${tag}.md 2>&1
head -40 gpurun_out/pmcw_${tag}.md | cut -c1-150
