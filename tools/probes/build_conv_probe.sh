#!/bin/bash
# builds tools/probes/bin/conv_probe against the in-tree kernels object (run deepcut-cnn_amd/build.py first)
set -e
cd "$(dirname "$0")/../.."
mkdir -p tools/probes/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -c tools/probes/conv_probe.cpp -o tools/probes/bin/conv_probe.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 tools/probes/bin/conv_probe.o -x none deepcut-cnn_amd/lib/kernels.hip.o -o tools/probes/bin/conv_probe
echo tools/probes/bin/conv_probe
