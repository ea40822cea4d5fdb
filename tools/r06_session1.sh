#!/bin/bash
# round 6, GPU session 1: the one-frame-per-wave mel kernel (variants, timing, 3000-replay co-run), the multi-frame reproducer with dumps
export TMPDIR=/tmp; mkdir -p gpurun_out; O=gpurun_out/r06s1.log; : > $O
for lib in "" tools/_libsed_wg.so tools/_libsed_w6.so tools/_libsed_w8.so tools/_libsed_w6g.so tools/_libsed_w12g.so; do
  echo "== probe lib=$lib" >> $O; SED_PROBE_LIB=$lib timeout 200 python tools/probe_mel.py 0 2 0 2>&1 | grep mel_wave >> $O
done
for lib in - tools/_libsed_wg.so tools/_libsed_w6g.so tools/_libsed_w8.so; do for bes in gemm tails; do
  timeout 300 python tools/mel_graph_race.py $lib 3000 $bes 2>&1 | tail -1 | cut -c1-300 >> $O
done; done
echo "== reproducer" >> $O
timeout 200 python tools/mel_repro/race.py tools/_melrepro_run8p.so 600 gemm 2>&1 | tail -1 | cut -c1-400 >> $O
timeout 300 python tools/mel_repro/race.py tools/_melrepro_dump8p.so 1000 gemm gpurun_out/r06_mel_dump_gemm.json 2>&1 | tail -8 | cut -c1-1500 >> $O
timeout 200 python tools/mel_repro/race.py tools/_melrepro_pad8p.so 1000 gemm 2>&1 | tail -1 | cut -c1-400 >> $O
timeout 200 python tools/mel_repro/race.py tools/_melrepro_run8p.so 600 gemm 2>&1 | tail -1 | cut -c1-400 >> $O
cat $O
