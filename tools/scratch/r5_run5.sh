#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_5; mkdir -p $O
timeout 300 python tools/scratch/dbg_convp64.py > $O/dbg.log 2>&1; grep -c "mismatch 0 of" $O/dbg.log; grep "mismatch" $O/dbg.log
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -k "convp or conv_down_up or conv_layers or patch or mask or conv8 or split" > $O/pytest_convp.log 2>&1
echo "rc=$?" >> $O/pytest_convp.log; tail -3 $O/pytest_convp.log
bash tools/scratch/ab_step.sh rawvec c8mask > $O/ab_libs.log 2>&1; cat $O/ab_libs.log
cp tools/scratch/lib_c8mask.so rna_gan_amd/librnagan_hip.so
timeout 1200 python -m pytest tests/test_dp2_gpu.py -x -q -m gpu -k "world4 or world8" -s > $O/pytest_dp.log 2>&1
echo "rc=$?" >> $O/pytest_dp.log; grep -E "passed|failed|rc=|wire" $O/pytest_dp.log | tail -5
timeout 600 python tools/ab_conv.py --layers 1,2 --kinds up --rounds 3 --sets "conv8=5" > $O/ab_conv_mask.log 2>&1; tail -8 $O/ab_conv_mask.log
