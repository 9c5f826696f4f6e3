#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_ops_gpu.py -q -x -k "conv_down_up_wgrad or f32_matrix or g0_and_head or linear" 2>&1 | tail -5
python3 -m pytest tests/test_train_gpu.py tests/test_engine_gpu.py tests/test_vae_gpu.py -q -x -k "fp32 or float32 or reference_trainops or public_functional or tight or l2 or vae" 2>&1 | tail -3
sed -i 's/r4_fp32_prof8/r4_fp32_prof9/g' tools/scratch/r4_f32prof.sh
bash tools/scratch/r4_f32prof.sh
