#!/bin/bash
# the whole GPU suite (no -x: every failure is listed), then smoke()
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_tests; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; tail -15 $O/pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -5 $O/smoke.log
