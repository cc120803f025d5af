#!/bin/bash
# round-3 PMC look at the composite (one frame at a time, the driver's 20 poses) + the list of SQ counters of this box
export TMPDIR=/tmp
TAG=${1:-r03pmc}
(cd /tmp && rocprofv3 -L 2>/dev/null | grep -o -E "\bSQ_[A-Z0-9_]+" | sort -u > $OLDPWD/gpurun_out/${TAG}_sq_counters.txt)
scripts/gpu_pmc_quick.sh $TAG "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU GRBM_GUI_ACTIVE" "SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD"
