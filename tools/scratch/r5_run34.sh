#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_34; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1; tail -5 $O/pytest.log | cut -c1-250
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.log | cut -c1-250
timeout 900 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r5_34/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['config']['extras']['fp32_step']['ms_per_step'], d['config']['extras']['fp32_step'].get('f32mma1'), d['config']['extras']['fp32_step'].get('roofline_fp32',{}).get('frac'))
PY
