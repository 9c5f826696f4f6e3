import sys, os, tempfile, pathlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import test_dp_gpu as t
p = pathlib.Path(tempfile.mkdtemp())
a = t._run(p, 0); b = t._run(p, 1)
for i, (x, y) in enumerate(zip(a["losses"], b["losses"])):
    print(i, "%.5f %.5f %.2e" % (x, y, abs(x - y) / (abs(x) + 1)))
for k in ("g", "d"):
    print(k, float((a[k] - b[k]).norm() / a[k].norm()))
