tag=${1:-x}; export TMPDIR=/tmp; mkdir -p gpurun_out
python tools/linear_bench.py | tee gpurun_out/linear_$tag.txt
for pass in "a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "b SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES" "c SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" "f FETCH_SIZE" "w WRITE_SIZE"; do
  set -- $pass; name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc $@ -d gpurun_out/pmclin_${tag}_$name -o $name -- python tools/linear_bench.py packed > gpurun_out/pmclin_${tag}_$name.log 2>&1
done
python tools/pmc_wait_summary.py gpurun_out/pmclin_${tag}_a/a_results.db gpurun_out/pmclin_${tag}_b/b_results.db gpurun_out/pmclin_${tag}_c/c_results.db gpurun_out/pmclin_${tag}_f/f_results.db gpurun_out/pmclin_${tag}_w/w_results.db > gpurun_out/pmclin_${tag}.md 2>&1
grep -E "linear|kernel \|" gpurun_out/pmclin_${tag}.md | cut -c1-500
