#!/bin/bash
# One GPU-box cycle: parity tests, a bench line, a kernel trace.  Usage (from the repo root, via gpurun):
#   bash tools/gpu_cycle.sh <tag> [pytest-args...]
tag=${1:-x}; shift
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 400 python -m pytest tests -m gpu -x -q "$@" 2>&1 | tail -5
timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_$tag.json | cut -c1-330
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof_$tag -o $tag -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_$tag.log 2>&1
python tools/rocpd_summary.py gpurun_out/prof_$tag/${tag}_results.db 0.6 > gpurun_out/prof_$tag.md 2>&1
head -48 gpurun_out/prof_$tag.md | cut -c1-110
