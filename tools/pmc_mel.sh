# stall-oriented PMC passes over the mel kernel alone (tools/probe_mel.py): bash tools/pmc_mel.sh <tag>
tag=${1:-x}; export TMPDIR=/tmp; mkdir -p gpurun_out
for pass in "a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "b SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_CMD_FIFO_FULL" "c SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES SQ_INSTS_VALU" "f FETCH_SIZE" "w WRITE_SIZE"; do
  set -- $pass; name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc $@ -d gpurun_out/pmcmel_${tag}_$name -o $name -- python tools/probe_mel.py 0 2 > gpurun_out/pmcmel_${tag}_$name.log 2>&1
done
python tools/pmc_wait_summary.py gpurun_out/pmcmel_${tag}_a/a_results.db gpurun_out/pmcmel_${tag}_b/b_results.db gpurun_out/pmcmel_${tag}_c/c_results.db gpurun_out/pmcmel_${tag}_f/f_results.db gpurun_out/pmcmel_${tag}_w/w_results.db > gpurun_out/pmcmel_${tag}.md 2>&1
grep -E "mel|kernel \|" gpurun_out/pmcmel_${tag}.md | cut -c1-400
