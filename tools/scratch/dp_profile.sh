#!/bin/bash
# kernel statistics of the benchmark on the data-parallel code path with one rank (RCCL group of 1) vs the plain path
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29513
rm -rf gpurun_out/dp_prof gpurun_out/sp_prof
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 RNAGAN_FORCE_DP=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/dp_prof -- python3 bench.py --no-cpu-baseline --no-roofline --steps 40 > gpurun_out/dp_prof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sp_prof -- python3 bench.py --no-cpu-baseline --no-roofline --steps 40 > gpurun_out/sp_prof.log 2>&1
python3 - <<'PY'
import csv, glob
def load(d):
    f = glob.glob(d + '/**/*kernel_stats.csv', recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    it = [int(r['Calls']) for r in rows if 'adam_dev_kernel' in r['Name'] or 'adam' in r['Name'].lower() and 'hyper' not in r['Name']]
    return rows
dp, sp = load('gpurun_out/dp_prof'), load('gpurun_out/sp_prof')
def per(rows):
    n = sum(int(r['Calls']) for r in rows if 'adam_hyper_kernel' in r['Name']) / 3.0
    return {r['Name'].replace('(anonymous namespace)::', '')[:90]: (int(r['TotalDurationNs']) / n / 1e3, int(r['Calls']) / n) for r in rows}, n
a, na = per(dp); b, nb = per(sp)
print('iters', na, nb, 'kernel us/iter dp %.0f  single %.0f' % (sum(v[0] for v in a.values()), sum(v[0] for v in b.values())))
keys = sorted(set(a) | set(b), key=lambda k: -abs(a.get(k, (0, 0))[0] - b.get(k, (0, 0))[0]))
for k in keys[:22]:
    print('%+8.1f us  dp %7.1f (%5.1f x)  single %7.1f (%5.1f x)  %s' % (a.get(k, (0, 0))[0] - b.get(k, (0, 0))[0], *a.get(k, (0, 0)), *b.get(k, (0, 0)), k))
PY
grep -o '"ms_per_step": [0-9.]*' gpurun_out/dp_prof.log gpurun_out/sp_prof.log
