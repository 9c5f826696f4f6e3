#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_ops_gpu.py -q -x -k "convd_plane" 2>&1 | tail -3
python3 tools/ab_conv.py --batch 64 --layers 0 --kinds down --sets "convd=0;convd=1" --rounds 5 2>&1 | grep "^L1"
python3 tools/ab_conv.py --batch 128 --layers 0 --kinds down --sets "convd=0;convd=1" --rounds 3 2>&1 | grep "^L1"
