#!/bin/bash
# builds tools/probes/bin/pair_probe (the bottleneck-pair kernel alone: no dependency on the library)
set -e
cd "$(dirname "$0")/../.."
mkdir -p tools/probes/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-inline-asm -c tools/probes/pair_kernel.hip -o tools/probes/bin/pair_kernel.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -c tools/probes/pair_probe.cpp -o tools/probes/bin/pair_probe.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 tools/probes/bin/pair_probe.o tools/probes/bin/pair_kernel.o -o tools/probes/bin/pair_probe
echo tools/probes/bin/pair_probe
