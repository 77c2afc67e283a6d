#!/bin/bash
# The library's HOST code (parsers, graph construction, InsertSplits, shape inference, lowering, filter packing, the C ABI) under
# AddressSanitizer: the host translation units are rebuilt instrumented (the kernels' object file is taken as built), and
# the CPU test suite plus the definition fuzzer run against that library.  No GPU involved.  usage: tools/asan_host_suite.sh [outdir]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-/tmp/deepcut_asan}
mkdir -p "$OUT"
python "$R/deepcut-cnn_amd/build.py" > /dev/null        # (the *.hip.o of the normal build)
HOST="formats.cpp hdf5_reader.cpp runtime.cpp net_init.cpp net_lower.cpp net_tune.cpp net_run.cpp net_image.cpp net_group.cpp streams.cpp multi_gpu.cpp c_api.cpp"
for f in $HOST; do
  /opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fsanitize=address -fno-gpu-sanitize -fno-omit-frame-pointer \
    -Wno-inline-asm -c "$R/deepcut-cnn_amd/csrc/$f" -o "$OUT/$f.o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address -shared-libasan -o "$OUT/libdeepcut_hip.so" \
  $(for f in $HOST; do echo "$OUT/$f.o"; done) "$R"/deepcut-cnn_amd/lib/*.hip.o   # (every kernel translation unit as built)
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
export DEEPCUT_HIP_LIB="$OUT/libdeepcut_hip.so" LD_PRELOAD="$RT"
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:protect_shadow_gap=0:max_allocation_size_mb=8192:allocator_may_return_null=1
cd "$R"
# (deselected: tests that link a binary of their own against the library, or build their own instrumented harness)
python -m pytest tests -q -m "not gpu" -p no:cacheprovider --deselect tests/test_abi_misuse.py --deselect tests/test_cxx_facade.py \
  --deselect tests/test_format_fuzz.py
for s in 11 12; do python tools/fuzz_net_definitions.py $s 2500; done
