#!/bin/bash
# ON THE GPU BOX: SQ counters of the forced wino_h23 launch on the conv4_x 3x3 shape (batch 8, 544x736), three passes of eight counters
# (separate runs: --pmc with --kernel-trace only).  WAVE_CYCLES / WAIT_* / ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles
# (32 per v_mfma_f32_32x32x16_f16); LDS_BANK_CONFLICT / LDS_IDX_ACTIVE: extra / all LDS-array cycles.   gpurun -- 'bash tools/wino_f16_pmc.sh'
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
  n=$1; shift
  rm -rf /tmp/p_$n
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/p_$n -o l -- python $GRAFT_REPO_ROOT/tools/wino_f16_probe.py --stamps --shapes res4 > /dev/null 2>&1
  python - <<PY
import sqlite3,glob
c=sqlite3.connect(glob.glob("/tmp/p_$n/**/*.db",recursive=True)[0])
for r in c.execute("select counter_name, sum(value) from counters_collection where kernel_name like '%wino_h23%' group by counter_name"): print("%-32s %14.0f" % r)
PY
}
run a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU
run b SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_INSTS_LDS
run c SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_IFETCH GRBM_GUI_ACTIVE
