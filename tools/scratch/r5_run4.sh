#!/bin/bash
# round 5, call 4: convp's 64-byte-contiguous stores (parity, WRITE_SIZE, step A/B), the fp32 mode's step time, new tests
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_4; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -k "convp or conv_down_up or conv_layers or patch or mask" > $O/pytest_convp.log 2>&1
echo "rc=$?" >> $O/pytest_convp.log; tail -4 $O/pytest_convp.log
for v in rawvec convp64; do
  cp tools/scratch/lib_$v.so rna_gan_amd/librnagan_hip.so
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w_$v -o p -- python3 tools/ab_conv.py --check 0 --rounds 1 --rep 2 --layers 0 --kinds up --sets "convp=1" > $O/pmc_w_$v.log 2>&1
  python3 tools/pmc_layers.py $O/pmc_w_$v | grep -E "kernel|convp" > $O/pmc_w_$v.txt; echo "== $v"; cat $O/pmc_w_$v.txt
  rm -rf $O/pmc_w_$v
done
timeout 300 python3 tools/ab_conv.py --layers 0 --kinds up --sets "convp=1" --rounds 3 > $O/ab_conv_convp64.log 2>&1; tail -5 $O/ab_conv_convp64.log
cp tools/scratch/lib_rawvec.so rna_gan_amd/librnagan_hip.so
timeout 300 python3 tools/ab_conv.py --layers 0 --kinds up --sets "convp=1" --rounds 3 > $O/ab_conv_base.log 2>&1; tail -5 $O/ab_conv_base.log
bash tools/scratch/ab_step.sh rawvec convp64 > $O/ab_libs.log 2>&1; cat $O/ab_libs.log
cp tools/scratch/lib_convp64.so rna_gan_amd/librnagan_hip.so
# fp32 mode: step time with a long warm-up, then its kernel statistics
timeout 600 python3 bench.py --precision fp32 --steps 5 --warmup 12 --no-cpu-baseline --no-roofline 2> $O/fp32.err | tail -1 > $O/fp32.json; grep -o '"ms_per_step": [0-9.]*' $O/fp32.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fp32prof -- python3 bench.py --precision fp32 --steps 5 --warmup 12 --no-cpu-baseline --no-roofline > /dev/null 2> $O/fp32prof.err
python3 tools/prof_summary.py $O/fp32prof 45 > $O/fp32prof.txt 2>&1; head -50 $O/fp32prof.txt | cut -c1-170
rm -rf $O/fp32prof
timeout 1500 python -m pytest tests/test_dp2_gpu.py tests/test_train_gpu.py tests/test_wire_error_cpu.py -x -q -m "gpu or not gpu" -k "world4 or world8 or inference_fp8 or wire or ring" -s > $O/pytest_new.log 2>&1
echo "rc=$?" >> $O/pytest_new.log; grep -E "passed|failed|rc=|wire|fp8 |reference size" $O/pytest_new.log | tail -12
