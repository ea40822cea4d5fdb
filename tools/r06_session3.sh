#!/bin/bash
# round 6, GPU session 3: the fix (swapped operand in src0) applied to the multi-frame reproducer; the product kernel; full suite; bench
export TMPDIR=/tmp; mkdir -p gpurun_out; O=gpurun_out/r06s3.log; : > $O
echo "== reproducer with the product's operand order (MEL_SAFE_OPSEL): persistent run 8 / run 8, beside gemm / storm" >> $O
for so in safe8p safe8; do for bes in gemm storm; do
  timeout 300 python tools/mel_repro/race.py tools/_melrepro_$so.so 1500 $bes 2>&1 | tail -1 | cut -c1-300 >> $O
done; done
echo "== control: the unfixed reproducer, same box" >> $O
timeout 300 python tools/mel_repro/race.py tools/_melrepro_run8.so 300 storm 2>&1 | tail -1 | cut -c1-300 >> $O
echo "== product kernel" >> $O
timeout 200 python tools/probe_mel.py 0 2 0 2>&1 | grep mel_wave >> $O
cat $O
bash tools/gpu_cycle.sh r06a tests bench
