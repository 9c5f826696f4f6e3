#!/bin/bash
# PMC passes over the conv microbench (one pass per counter group: MI355X_MICROARCH.md "rocprofv3 PMC slots").
# usage: tools/pmc_conv.sh <out-prefix under gpurun_out> <ab_conv.py args...>
set -u
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS="$*"
run() {  # name, counters...
  local name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/${OUT}_$name -o p -- python3 tools/ab_conv.py --check 0 --rounds 1 --rep 2 $ARGS > gpurun_out/${OUT}_$name.log 2>&1
}
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
run l2 TCC_HIT_sum TCC_MISS_sum
python3 tools/pmc_layers.py gpurun_out/${OUT}_sq gpurun_out/${OUT}_lds gpurun_out/${OUT}_fetch gpurun_out/${OUT}_write gpurun_out/${OUT}_l2 --csv gpurun_out/${OUT}_summary.csv | grep -E "kernel|conv8|convp|gather_gemm|wgrad"
