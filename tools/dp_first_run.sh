#!/bin/bash
# tools/dp_first_run.sh -- the FIRST run of the data-parallel path on real multi-GPU hardware (no 8-GPU node was
# available while this was built): what to run, in which order, and what each line must show.
#
#   bash tools/dp_first_run.sh [max_gpus=8] [steps=20] [warmup=5]
#
# 1. RCCL's own bus bandwidth for the step's messages (90 MB bf16 = the generator's / discriminator's gradient tail, and
#    the 4.7 MB factor all-gather), measured with torch.distributed on all ranks -- the number DESIGN's estimate assumed
#    (~300 GB/s bus bandwidth at 8 ranks).
# 2. bench.py --gpus {1,2,4,8}: one JSON line each -> gpurun_out/dp_first_run/scale_N.json; asserts config.ranks == N and
#    collective_backend == rccl; prints imgs/s and the weak-scaling efficiency against N = 1.
# 3. the four knobs at the largest N, one change at a time against the defaults:
#      RNAGAN_DP_OVERLAP=0        no overlap of a train_op's all-reduce with the next train_op's prefix
#      RNAGAN_DP_G0_FACTORS=0     generator layer 0's gradient on the wire as the 67 M-element product, not as factors
#      RNAGAN_SPLIT_BN_DP=1       fused split-K BatchNorm kernels under DP (default OFF with > 1 rank until this A/B says
#                                 otherwise: their in-launch rendezvous needs every workgroup resident beside RCCL's kernels;
#                                 bench.py raises if a rendezvous timed out -- rna_gan_amd.ops_hip.check_handoffs)
#      NCCL_MAX_NCHANNELS=8/16    fewer CUs for RCCL's kernels
# Nothing here needs the network or root.  Every bench.py call is the driver's own launch form (torchrun, 127.0.0.1).
set -u
MAXG=${1:-8}; STEPS=${2:-20}; WARM=${3:-5}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT"
OUT=gpurun_out/dp_first_run; mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
NDEV=$(python3 -c 'import torch; print(torch.cuda.device_count())')
echo "devices visible: $NDEV (asked for up to $MAXG)"
if [ "$NDEV" -lt 2 ]; then echo "needs >= 2 GPUs: only the N = 1 line will be produced"; fi
[ "$MAXG" -gt "$NDEV" ] && MAXG=$NDEV
PORT=29611
EXTRA=""; [ "${DP_FIRST_RUN_SELFTEST:-0}" = "1" ] && EXTRA="--no-roofline --no-ceilings"     # (the self-test checks that the lines run, not their fractions)

run_bench () {   # run_bench N tag [ENV=VAL ...]
  local n=$1 tag=$2; shift 2
  local f="$OUT/${tag}.json"
  if [ "$n" -eq 1 ]; then
    env "$@" python3 bench.py --gpus 1 --steps "$STEPS" --warmup "$WARM" --no-cpu-baseline --no-extras $EXTRA > "$f" 2> "$OUT/${tag}.err"
  else
    PORT=$((PORT + 1))
    env "$@" python3 -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port "$PORT" \
        bench.py --gpus "$n" --steps "$STEPS" --warmup "$WARM" --no-cpu-baseline --no-extras $EXTRA > "$f" 2> "$OUT/${tag}.err"
  fi
  local rc=$?
  if [ $rc -ne 0 ]; then echo "  $tag: bench.py exited $rc (see $OUT/${tag}.err)"; tail -3 "$OUT/${tag}.err"; return $rc; fi
  python3 - "$f" "$n" "$tag" <<'EOF'
import json, sys
rec = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
n, tag = int(sys.argv[2]), sys.argv[3]
cfg = rec["config"]
assert rec["n_gpus"] == n and cfg["ranks"] == n, ("rank count", rec["n_gpus"], cfg["ranks"], n)
if n > 1:
    assert str(cfg["collective_backend"]).startswith("rccl"), cfg["collective_backend"]
print("  %-28s N=%d  %9.1f imgs/s  %7.3f ms/iteration  roofline.frac=%s" % (
    tag, n, rec["value"], rec["ms_per_step"], (rec.get("roofline") or {}).get("frac")))
EOF
}

# DP_FIRST_RUN_SELFTEST=1: the kit checks ITSELF on a one-GPU box -- every knob line below runs through the data-parallel route
# on a one-rank RCCL group (RNAGAN_FORCE_DP=1), so a knob that no longer exists or a route that no longer runs fails here
# (tests/test_dp_first_run_gpu.py) and not on the first multi-GPU node.  The numbers of such a run mean nothing.
SELFTEST=${DP_FIRST_RUN_SELFTEST:-0}
knob () {   # knob tag [ENV=VAL ...]
  local tag=$1; shift
  if [ "$SELFTEST" = "1" ]; then
    PORT=$((PORT + 1))
    run_bench 1 "$tag" RNAGAN_FORCE_DP=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT "$@"
  else
    run_bench "$MAXG" "$tag" "$@"
  fi
}

echo "== 1. RCCL bus bandwidth for the step's messages =="
if [ "$MAXG" -ge 2 ]; then
  PORT=$((PORT + 1))
  python3 -m torch.distributed.run --nnodes=1 --nproc-per-node "$MAXG" --master-addr 127.0.0.1 --master-port "$PORT" \
      tools/rccl_busbw.py 2> "$OUT/busbw.err" | tee "$OUT/busbw.txt"
fi

echo "== 2. weak scaling, defaults =="
for n in 1 2 4 8; do
  [ "$n" -le "$MAXG" ] && run_bench "$n" "scale_$n"
done
python3 - "$OUT" <<'EOF'
import json, os, sys
d = sys.argv[1]
vals = {}
for n in (1, 2, 4, 8):
    p = os.path.join(d, "scale_%d.json" % n)
    if os.path.exists(p) and os.path.getsize(p):
        vals[n] = json.loads(open(p).read().strip().splitlines()[-1])["value"]
if 1 in vals:
    for n, v in sorted(vals.items()):
        print("  N=%d: %.1f imgs/s, efficiency vs N=1: %.3f" % (n, v, v / (n * vals[1])))
EOF

if [ "$MAXG" -ge 2 ] || [ "$SELFTEST" = "1" ]; then
  echo "== 3. knobs at N=$MAXG (one change each; compare with scale_$MAXG above) =="
  knob "knob_overlap0"        RNAGAN_DP_OVERLAP=0
  knob "knob_g0factors0"      RNAGAN_DP_G0_FACTORS=0
  knob "knob_splitbn_dp1"     RNAGAN_SPLIT_BN_DP=1
  knob "knob_nchannels8"      NCCL_MAX_NCHANNELS=8
  knob "knob_nchannels16"     NCCL_MAX_NCHANNELS=16
  knob "knob_prefix_bwd1"     RNAGAN_DP_PREFIX_BWD=1      # round 3's prefix (whole backward of the real half): longer cover
  # route "whole": the single process's train_op bodies (double batch, shared generator pass), collectives NOT hidden -- the A/B
  # that prices the prefixes: hide the all-reduces, or keep the 0.5-0.6 ms per iteration the prefix route gives up
  knob "knob_route_whole"     RNAGAN_DP_ROUTE=whole
  knob "knob_route_whole_splitbn" RNAGAN_DP_ROUTE=whole RNAGAN_SPLIT_BN_DP=1
  # BASELINE configs[3]: the fp16 build.  Its all-reduce is fp32 by default (loss-scaled weight gradients overflow an fp16 wire:
  # rna_gan_amd/dist.py F16_WIRE); the second line prices what the 16-bit wire would save (CHECK the losses for NaN there)
  knob "fp16_wire_fp32"       RNAGAN_BENCH_PRECISION=fp16
  knob "fp16_wire_f16"        RNAGAN_BENCH_PRECISION=fp16 RNAGAN_DP_F16_WIRE=1
  echo "== one-rank overhead of the DP route (RNAGAN_FORCE_DP=1 on one GPU vs the single-process path) =="
  # (the DP route at one rank needs a process group of one: RANK / WORLD_SIZE / MASTER_* make dist.init_from_env create it)
  PORT=$((PORT + 1))
  run_bench 1 "force_dp_1rank"             RNAGAN_FORCE_DP=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT
  PORT=$((PORT + 1))
  run_bench 1 "force_dp_1rank_prefix1"     RNAGAN_FORCE_DP=1 RNAGAN_DP_PREFIX_BWD=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT
  PORT=$((PORT + 1))
  run_bench 1 "force_dp_1rank_whole"       RNAGAN_FORCE_DP=1 RNAGAN_DP_ROUTE=whole RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT
fi
echo "records under $OUT/"
