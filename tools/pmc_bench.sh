#!/bin/bash
# PMC evidence for the benchmark at HEAD (batch 64, eager launches: counter collection stalls on graph-launched kernels).
# One pass per counter group (MI355X_MICROARCH.md "rocprofv3 PMC slots"); summaries are written under gpurun_out/<prefix>_*.
# usage: tools/pmc_bench.sh <prefix>
set -u
OUT=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export RNAGAN_GRAPHS=0
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extras"
run() {  # name, counters...
  local name=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/${OUT}_$name -o p -- python3 bench.py $ARGS > gpurun_out/${OUT}_$name.log 2>&1
}
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
run l2 TCC_HIT_sum TCC_MISS_sum
python3 tools/pmc_layers.py gpurun_out/${OUT}_sq gpurun_out/${OUT}_lds gpurun_out/${OUT}_fetch gpurun_out/${OUT}_write gpurun_out/${OUT}_l2 --csv gpurun_out/${OUT}_layers.csv > /dev/null
python3 tools/pmc_traffic.py gpurun_out/${OUT}_fetch gpurun_out/${OUT}_write > gpurun_out/${OUT}_traffic.json
grep -E "kernel,|conv8|convp|gather_gemm|wgrad" gpurun_out/${OUT}_layers.csv
python3 tools/pmc_family.py gpurun_out/${OUT}_layers.csv > gpurun_out/${OUT}_mfma_util.json
cat gpurun_out/${OUT}_traffic.json
for d in sq lds fetch write l2; do rm -rf gpurun_out/${OUT}_$d; done   # raw traces stay on the box (64 MiB copy-back limit)
