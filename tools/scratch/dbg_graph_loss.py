"""Per-iteration losses of bench.hip_workload's one_step(), graphs on / off (RNAGAN_GRAPHS): are the loss tensors a replayed
train_op returns current?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
args = bench.parse_args(["--batch", os.environ.get("DBG_BATCH", "64")])
dev = torch.device("cuda:0")
one_step, flush, N, info = bench.hip_workload(args, 0, 1, dev)
for it in range(int(os.environ.get("DBG_ITERS", "12"))):
    ls = one_step()
    torch.cuda.synchronize()
    print(it, ["%.6f" % float(l.item()) for l in ls], [l.data_ptr() for l in ls], flush=True)
