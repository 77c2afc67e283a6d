#!/bin/bash
# ON THE GPU BOX: tile overrides for the grouped float16 pyramid with N groups in flight.
#   bash tools/gpu_group_tiles.sh <tag> <inflight> 'set' ...      a set = 'regex=tile[;regex=tile...]' on the cache KEY; `base` = as tuned
# The first run tunes and writes the cache; every set rewrites matching lines and re-measures in a fresh process (interleaved twice).
set -u
TAG=$1; NF=$2; shift 2
R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export DC_TUNE_CACHE=$OUT/cache.txt
rm -f $DC_TUNE_CACHE
timeout 600 python tools/group_profile.py --out $OUT --no-members --inflight $NF 2>&1 | grep "^grouped\|in flight"
cp $DC_TUNE_CACHE $OUT/base_cache.txt
for rep in 1 2; do for st in base "$@"; do
  python - $OUT/base_cache.txt $DC_TUNE_CACHE "$st" <<'PY'
import re, sys
src, dst, st = sys.argv[1:4]
over = [] if st == "base" else [kv.split("=") for kv in st.split(";")]
out = []
for l in open(src):
    k = l.rsplit(" ", 1)[0]
    for pat, tile in over:
        if re.search(pat, k):
            l = "%s %s\n" % (k, tile)
    out.append(l)
open(dst, "w").writelines(out)
PY
  echo "== $st"
  timeout 600 python tools/group_profile.py --out $OUT/x --no-members --inflight $NF 2>&1 | grep "^grouped\|in flight"
done; done
