#!/bin/bash
# bash tools/mel_repro/build.sh <name> [-D...]  ->  tools/_melrepro_<name>.so   (diagnostics only; see mel_wave_multiframe.hip)
set -e
cd "$(dirname "$0")/../.."
name=${1:-run8}; shift || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function -I desed_task_amd/csrc "$@" \
    tools/mel_repro/mel_wave_multiframe.hip -o tools/_melrepro_$name.so
echo tools/_melrepro_$name.so
