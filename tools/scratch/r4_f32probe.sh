#!/bin/bash
# what bounds gemm_mfma32_kernel?  probe builds (results wrong by construction): no MFMAs / no global loads in the k-loop / neither
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cp rna_gan_amd/librnagan_hip.so /tmp/lib_keep.so
cat > /tmp/f32micro.py <<'PY'
import sys, os, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from rna_gan_amd.ops_hip import HipOps
from rna_gan_amd.engine import ConvW
ops = HipOps(torch.float32, "cuda:0")
N = 64
def timeit(fn, rep=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rep): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / rep * 1e3
for l in (1, 2, 4):
    c, s = 64 << l, 128 >> l
    I, O, hs = c, 2 * c, s
    w = torch.randn(O, 4, 4, I, device="cuda") * 0.02
    cw = ConvW(w, None, torch.zeros_like(w), None, "OHWI")
    x = torch.randn(N, hs, hs, I, device="cuda"); g = torch.randn(N, hs // 2, hs // 2, O, device="cuda")
    fl = 2.0 * N * (hs // 2) ** 2 * O * I * 16
    for kind, fn in (("down", lambda: ops.conv_down(x, cw)), ("up", lambda: ops.conv_up(g, cw)), ("wgrad", lambda: ops.conv_wgrad(g, x, cw, False))):
        us = timeit(fn)
        print("L%d %-5s %8.1f us %6.1f TF" % (l + 1, kind, us, fl / us / 1e6), flush=True)
PY
for v in f32base f32nomfma f32noload f32neither; do
  cp tools/scratch/lib_$v.so rna_gan_amd/librnagan_hip.so
  echo "== $v"; python3 /tmp/f32micro.py 2>&1 | grep -E "^L"
done
cp /tmp/lib_keep.so rna_gan_amd/librnagan_hip.so
