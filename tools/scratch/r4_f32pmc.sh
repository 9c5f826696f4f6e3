#!/bin/bash
# PMC passes over the fp32 mode's iteration (eager launches): what bounds gemm_mfma32_kernel?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export RNAGAN_GRAPHS=0
ARGS="--precision fp32 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extras"
run() { local name=$1; shift; timeout 900 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/f32pmc_$name -o p -- python3 bench.py $ARGS > gpurun_out/f32pmc_$name.log 2>&1; }
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run l2 TCC_HIT_sum TCC_MISS_sum
run fetch FETCH_SIZE
python3 tools/pmc_layers.py gpurun_out/f32pmc_sq gpurun_out/f32pmc_lds gpurun_out/f32pmc_l2 gpurun_out/f32pmc_fetch --match mfma32 --csv gpurun_out/r4_f32pmc_layers.csv > /dev/null
cat gpurun_out/r4_f32pmc_layers.csv | cut -c1-260
for d in sq lds l2 fetch; do rm -rf gpurun_out/f32pmc_$d; done
