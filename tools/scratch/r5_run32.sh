#!/bin/bash
# last check at HEAD: smoke(), the driver's default bench command (timed), the multi-rank launcher on one GPU (gloo)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_32; mkdir -p $O
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.log | cut -c1-250
SECONDS=0
timeout 900 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$? in ${SECONDS}s"; tail -1 $O/bench_default.json | cut -c1-600
