#!/bin/bash
# same-box A/B of bench.py argument sets (alternated twice): bash tools/ab_tuning.sh <tag> "<args A>" "<args B>" ["<args C>" ...]
tag=$1; shift
mkdir -p gpurun_out
for rep in 1 2; do
  i=0
  for a in "$@"; do
    i=$((i+1))
    timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline $a 2>gpurun_out/abt_${tag}_${i}_$rep.err | tail -1 > gpurun_out/abt_${tag}_${i}_$rep.json
    python -c "import json;d=json.load(open('gpurun_out/abt_${tag}_${i}_$rep.json'));print('[$i] $a', $rep, d['ms_per_step'])" 2>&1 | tail -1
  done
done
