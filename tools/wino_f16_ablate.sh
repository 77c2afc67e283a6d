#!/bin/bash
# Ablation builds of wino_h23 (csrc/wino_f16.hip, DC_WINO_ABL): the library linked with one variant of that translation unit each, run
# through the probe's stamps (results are WRONG by construction: timing only).  Build part runs anywhere; `run` on the GPU box.
#   bash tools/wino_f16_ablate.sh build           (CPU container: hipcc cross-compiles)
#   gpurun -- 'bash tools/wino_f16_ablate.sh run'
set -u
R=$(cd $(dirname $0)/.. && pwd); L=$R/deepcut-cnn_amd/lib; B=$R/tools/probes/bin; mkdir -p $B
case ${1:-build} in
build)
  for a in 1 2 3 4; do
    D="-DDC_WINO_ABL=$a"; [ $a = 4 ] && D="-DDC_WINO_LATE_STORE=1"   # 4: the staging store at the end of the step (a variant, correct results)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-inline-asm $D -c $R/deepcut-cnn_amd/csrc/wino_f16.hip -o $B/wino_abl$a.o || exit 1
    objs=$(ls $L/*.o | grep -v wino_f16)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/lib_wabl$a.so $objs $B/wino_abl$a.o 2>/dev/null || exit 1
  done; ls -la $B/lib_wabl*.so ;;
run)
  for a in ${ABLS:-0 1 2 3 4}; do
    if [ $a = 0 ]; then unset DEEPCUT_HIP_LIB; else export DEEPCUT_HIP_LIB=$B/lib_wabl$a.so; fi
    echo "ABL $a"; DC_DEBUG_TIMING=0 python $R/tools/wino_f16_probe.py --stamps --shapes ${SHAPES:-res4} 2>&1 | grep -A1 "launch 0" | grep mean | cut -c1-300
  done ;;
esac
