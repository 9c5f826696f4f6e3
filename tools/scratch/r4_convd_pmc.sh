#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="tools/ab_conv.py --batch 64 --layers 0 --kinds down --sets convd=1;convd=0 --rounds 1 --rep 3 --check 0"
run() { local name=$1; shift; timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/cdpmc_$name -o p -- python3 $ARGS > gpurun_out/cdpmc_$name.log 2>&1; }
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run l2 TCC_HIT_sum TCC_MISS_sum
run fetch FETCH_SIZE
python3 tools/pmc_layers.py gpurun_out/cdpmc_sq gpurun_out/cdpmc_lds gpurun_out/cdpmc_l2 gpurun_out/cdpmc_fetch --match conv --csv gpurun_out/r4_convd_pmc.csv > /dev/null
cat gpurun_out/r4_convd_pmc.csv | cut -c1-260
for d in sq lds l2 fetch; do rm -rf gpurun_out/cdpmc_$d; done
