#!/bin/bash
# ON THE GPU BOX: the -m gpu suite only (arguments are passed to pytest).  gpurun --timeout 1500 -- 'bash tools/gpu_tests.sh r02b [pytest args]'
TAG=${1:-t}; shift
mkdir -p gpurun_out/$TAG
if [ $# -gt 0 ] && [ -e "$1" ]; then T="$1"; shift; else T=tests; fi
timeout 1300 python -m pytest $T -m gpu -q "$@" > gpurun_out/$TAG/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/$TAG/pytest.log
tail -15 gpurun_out/$TAG/pytest.log
