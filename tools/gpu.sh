#!/bin/bash
# Build the gfx950 library (hipcc cross-compile, mtime-cached) and THEN hand the tree to gpurun: a stale libsed_hip.so travelling
# to the GPU box cost three diagnostic runs once.   usage: tools/gpu.sh <timeout_s> '<command>'
set -e
cd "$(dirname "$0")/.."
python -m desed_task_amd.build > /tmp/sed_build.log 2>&1 || { tail -30 /tmp/sed_build.log; exit 1; }
python -c "import sys; sys.path.insert(0,'tests/emu'); import build_emu; build_emu.build()" >/dev/null
t=$1; shift
exec /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
