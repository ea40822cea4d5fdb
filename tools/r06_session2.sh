#!/bin/bash
# round 6, GPU session 2: which instruction fails (MEL_CHECK), and is the one-frame kernel immune under a GEMM storm?
export TMPDIR=/tmp; mkdir -p gpurun_out; O=gpurun_out/r06s2.log; : > $O
timeout 300 python tools/mel_repro/race.py tools/_melrepro_chk8p.so 600 gemm gpurun_out/r06_mel_check_gemm.json 2>&1 | tail -12 | cut -c1-1200 >> $O
echo "== storm: reproducer (persistent run 8), reproducer run 8 one run per workgroup, run 4 (one frame per wave, loop form)" >> $O
timeout 300 python tools/mel_repro/race.py tools/_melrepro_run8p.so 400 storm 2>&1 | tail -1 | cut -c1-400 >> $O
timeout 300 python tools/mel_repro/race.py tools/_melrepro_run8.so 400 storm 2>&1 | tail -1 | cut -c1-400 >> $O
timeout 300 python tools/mel_repro/race.py tools/_melrepro_run4.so 1000 storm 2>&1 | tail -1 | cut -c1-400 >> $O
echo "== storm: product kernel, variants" >> $O
for lib in - tools/_libsed_wg.so; do
  timeout 400 python tools/mel_graph_race.py $lib 2000 storm 2>&1 | tail -1 | cut -c1-400 >> $O
done
timeout 300 python tools/mel_graph_race.py - 600 storm wg 2>&1 | tail -1 | cut -c1-400 >> $O
cat $O
