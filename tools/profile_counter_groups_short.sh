#!/bin/bash
# occupancy / wait / LDS-conflict / L2 counters for one workload of bench.py: bash tools/profile_counter_groups_short.sh <tag>   (BENCH_ARGS as in profile_counters.sh)
R=${GRAFT_REPO_ROOT:-$(pwd)}
bash $R/tools/profile_counters.sh "$1" \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
  "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
  "GRBM_GUI_ACTIVE GRBM_COUNT"
