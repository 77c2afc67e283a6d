#!/bin/bash
# ONE script for everything that runs ON THE GPU BOX through gpurun (it replaces round 2's seventeen gpu_*.sh wrappers):
#
#   gpurun --timeout 1500 -- 'bash tools/gpu.sh <mode> <tag> [args...]'        output under gpurun_out/<tag>/
#
#   tests  <tag> [path] [pytest args]     the -m gpu suite (or one file / -k selection)
#   bench  <tag> [bench.py args]          one bench line, tiles tuned from scratch, per-launch table + per-shape summary
#                                         (f16 batch 8: `bench <tag> --dtype f16 --batch 8 --streams 2`)
#   check  <tag>                          tests + the default bench line + a one-forward-at-a-time per-launch table
#   probe  <tag> [conv_probe args]        tools/probes/conv_probe: ragged shapes first (short timeout), then the timed table
#   stamps <tag> <bench args> -- idx...   DC_DEBUG_TIMING phase stamps of the given launch indices
#   ab     <tag> VAR v0 v1 [bench args]   interleaved A/B of an environment switch, same tune cache, 3 repetitions
#   libs   <tag> name... [-- bench args]  interleaved A/B of library builds tools/probes/bin/lib_<name>.so vs the in-tree one
#   sweep  <tag> streams|coalesce|queues  forwards in flight / cross-request batching / hardware-queue sweeps
#   pmc    <tag> [bench args]             FETCH_SIZE and WRITE_SIZE passes (separate runs) -> HBM bytes per launch and per shape
#   tiles  <tag> set... [-- bench args]   tune-cache overrides under load: a set is `key-prefix=tile[,key-prefix=tile...]` (key as in the
#                                         cache file, e.g. h12512/256/2304=d128x256x64_w241_s3); `base` = the seeded cache as is
#
# TUNE=<file> seeds the tune cache (default: the newest profiles/r*_tune_cache.txt); TUNE=none tunes from scratch.
set -u
MODE=${1:?mode}; TAG=${2:?tag}; shift 2
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
QUIET="--no-cpu-baseline --no-f16-line --no-resnet101 --coalesce 0"
seed_cache() {
  local src=${TUNE:-$(ls profiles/r*_tune_cache.txt 2>/dev/null | sort | tail -1)}
  rm -f $OUT/tune_cache.txt
  [ "$src" != none ] && [ -n "$src" ] && [ -s "$src" ] && cp $src $OUT/tune_cache.txt
  export DC_TUNE_CACHE=$OUT/tune_cache.txt
}
line() {  # print value / one-at-a-time of a bench json: line <label> <file>
  python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    o = d.get("one_forward_at_a_time", {})
    x = d.get("cross_request_batching", {})
    print("%-30s value %8.1f (%4.0f TF/s)  one-at-a-time %8.1f (%4.0f TF/s)%s" % (sys.argv[1], d["value"], d.get("tflops", 0), o.get("value", 0), o.get("tflops", 0),
          "  coalesced %.1f" % x["value"] if x else ""))
except Exception as e:
    print(sys.argv[1], "failed:", e)
PY
}
case $MODE in
tests)
  if [ $# -gt 0 ] && [ -e "$1" ]; then T="$1"; shift; else T=tests; fi
  timeout 1300 python -m pytest $T -m gpu -q "$@" > $OUT/pytest.log 2>&1
  echo "pytest rc=$?" >> $OUT/pytest.log
  tail -15 $OUT/pytest.log ;;
bench)
  TUNE=${TUNE:-none} seed_cache
  timeout 500 python bench.py $QUIET --steps 10 --warmup 2 --breakdown $OUT/per_launch.txt "$@" > $OUT/bench.json 2> $OUT/bench.err
  line "bench $*" $OUT/bench.json
  python tools/breakdown.py $OUT/per_launch.txt > $OUT/per_shape_summary.txt; head -${LINES_OUT:-16} $OUT/per_shape_summary.txt ;;
check)
  export DC_TUNE_CACHE=$OUT/tune_cache.txt
  timeout 1100 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -5 $OUT/pytest.log
  rm -f $DC_TUNE_CACHE  # (the driver's bench starts without a cache: tiles tuned in the process, tune_in_flight with isolated timings)
  timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 1500 $OUT/bench.json
  timeout 300 python bench.py $QUIET --streams 1 --breakdown $OUT/per_launch.txt > $OUT/bench_s1.json 2>> $OUT/bench.err; tail -3 $OUT/bench.err ;;
probe)
  # correctness on the ragged shapes first, under a short timeout: a barrier bug must not hang the box
  timeout 60 tools/probes/bin/conv_probe --shapes tiny_3x3,tiny_d2,tiny_odd,tiny_w1,tiny_w3 --batch 3 --reps 5 "$@" > $OUT/tiny.txt 2>&1 || { echo "tiny run failed/hung"; tail -20 $OUT/tiny.txt; exit 1; }
  if grep -q WRONG $OUT/tiny.txt; then grep -B2 WRONG $OUT/tiny.txt | head -40; echo "WRONG results on the tiny shapes: not timing"; exit 1; fi
  timeout 300 tools/probes/bin/conv_probe "$@" > $OUT/probe.txt 2>&1
  grep -c WRONG $OUT/probe.txt; tail -5 $OUT/probe.txt ;;
stamps)
  seed_cache
  args=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  for idx in "$@"; do
    DC_DEBUG_TIMING=$idx timeout 200 python bench.py $QUIET --no-graph --streams 1 --steps 2 --warmup 1 "${args[@]}" 2>&1 >/dev/null | grep -A1 "dc timing" | tail -3
  done ;;
ab)
  seed_cache
  VAR=$1; V0=$2; V1=$3; shift 3
  for rep in 1 2 3; do for v in $V0 $V1; do
    env $VAR=$v timeout 300 python bench.py $QUIET --steps 100 --warmup 10 "$@" > $OUT/$VAR$v.json 2> $OUT/$VAR$v.err
    line "$VAR=$v" $OUT/$VAR$v.json
  done; done ;;
tiles)
  seed_cache; cp $OUT/tune_cache.txt $OUT/base_cache.txt
  sets=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do sets+=("$1"); shift; done; [ $# -gt 0 ] && shift
  for rep in 1 2; do for st in base "${sets[@]}"; do
    python - $OUT/base_cache.txt $OUT/tune_cache.txt "$st" <<'PY'
import sys
src, dst, st = sys.argv[1:4]
over = [] if st == "base" else [kv.split("=") for kv in st.split(",")]
out = []
for l in open(src):
    k = l.split()[0] if l.split() else ""
    for pre, tile in over:
        if k.startswith(pre):
            l = "%s %s\n" % (k, tile)
    out.append(l)
open(dst, "w").writelines(out)
PY
    timeout 300 python bench.py $QUIET --steps 30 --warmup 4 "$@" > $OUT/t.json 2> $OUT/t.err
    line "$st" $OUT/t.json
  done; done ;;
libs)
  names=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do names+=("$1"); shift; done; [ $# -gt 0 ] && shift
  for rep in 1 2; do for which in base "${names[@]}"; do
    if [ $which = base ]; then unset DEEPCUT_HIP_LIB; else export DEEPCUT_HIP_LIB=$PWD/tools/probes/bin/lib_$which.so; fi
    seed_cache
    timeout 300 python bench.py $QUIET --steps 100 --warmup 10 "$@" > $OUT/$which.json 2> $OUT/$which.err
    line "$which" $OUT/$which.json
  done; done ;;
sweep)
  seed_cache
  case ${1:-streams} in
  streams) for s in 1 2 3 4 6 8; do timeout 300 python bench.py $QUIET --steps 30 --warmup 3 --streams $s > $OUT/s$s.json 2> $OUT/s$s.err; line "streams $s" $OUT/s$s.json; done ;;
  coalesce) for s in 1 2 3; do for c in 2 3 4 6 8; do timeout 300 python bench.py --no-cpu-baseline --no-f16-line --coalesce $c --streams $s --steps 40 --warmup 3 > $OUT/s${s}c$c.json 2>/dev/null; line "executors $s coalesce $c" $OUT/s${s}c$c.json; done; done ;;
  queues) for q in 2 4 8; do for s in 3 4 6 8; do GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py $QUIET --steps 30 --warmup 3 --streams $s > $OUT/q${q}s$s.json 2>/dev/null; line "queues $q streams $s" $OUT/q${q}s$s.json; done; done ;;
  esac ;;
pmc)
  seed_cache
  cd /tmp && export TMPDIR=/tmp
  PMC_CMD="python $R/bench.py $QUIET --streams 1 --no-graph --steps 2 --warmup 1 $*"
  timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f -- $PMC_CMD > /dev/null 2> $OUT/pmc_fetch.err
  timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o w -- $PMC_CMD > /dev/null 2> $OUT/pmc_write.err
  cd $R
  F=$(find $OUT/pmc_fetch -name "*.db" | head -1); W=$(find $OUT/pmc_write -name "*.db" | head -1)
  python tools/pmc_hbm_traffic.py $F $W "$TAG: $*" > $OUT/pmc_hbm_traffic.json 2> $OUT/pmc.err; head -30 $OUT/pmc_hbm_traffic.json
  python tools/pmc_per_shape.py $F $W > $OUT/pmc_hbm_traffic_per_shape.txt 2>> $OUT/pmc.err; head -24 $OUT/pmc_hbm_traffic_per_shape.txt
  rm -rf $OUT/pmc_fetch $OUT/pmc_write ;;
*) echo "unknown mode $MODE"; exit 2 ;;
esac
