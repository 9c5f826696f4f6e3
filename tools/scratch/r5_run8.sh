#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_8; mkdir -p $O
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py tests/test_train_gpu.py -x -q -m gpu -s > $O/pytest.log 2>&1
echo "rc=$?" >> $O/pytest.log; grep -E "nsplit|passed|failed|rc=|seeds x 3|reference size" $O/pytest.log | tail -14
# wgrad8n rotation: PMC fetch + hit rate, two builds
for v in slabadam w8nrot; do
  cp tools/scratch/lib_$v.so rna_gan_amd/librnagan_hip.so
  for c in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $c | cut -c1-5)
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${v}_$tag -o p -- python3 tools/ab_conv.py --check 0 --rounds 1 --rep 2 --layers 0 --kinds down --wgrad --sets "wgrad8n=1" > $O/pmc_${v}_$tag.log 2>&1
  done
  python3 tools/pmc_layers.py $O/pmc_${v}_FETCH $O/pmc_${v}_TCC_H | grep -E "kernel|wgrad" > $O/pmc_$v.txt; echo "== $v"; cat $O/pmc_$v.txt
  rm -rf $O/pmc_${v}_FETCH $O/pmc_${v}_TCC_H
done
cp tools/scratch/lib_w8nrot.so rna_gan_amd/librnagan_hip.so
timeout 600 python3 tools/ab_conv.py --layers 0 --kinds down --wgrad --sets "wgrad8n=1" --rounds 3 > $O/ab_w8n_new.log 2>&1; tail -4 $O/ab_w8n_new.log
cp tools/scratch/lib_slabadam.so rna_gan_amd/librnagan_hip.so
timeout 600 python3 tools/ab_conv.py --layers 0 --kinds down --wgrad --sets "wgrad8n=1" --rounds 3 > $O/ab_w8n_old.log 2>&1; tail -4 $O/ab_w8n_old.log
cp tools/scratch/lib_w8nrot.so rna_gan_amd/librnagan_hip.so
# the whole package: round-4 tree vs the working tree, interleaved
timeout 1500 python tools/ab_trees.py --trees r4=tools/scratch/_r4tree,head=. --rounds 5 --steps 40 --json $O/ab_trees.json > $O/ab_trees.log 2>&1; tail -8 $O/ab_trees.log
