import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "revisit-bpr_amd"); sys.path.insert(0, "tests")
from test_gpu_api import build, batches, OPTS, _run_loop
from revisit_bpr.models.bpr import set_backend
U, I, d, B = 300, 200, 64, 64
reg = {"user": 0.0016, "item": 0.0001, "neg": 0.00375}
data = batches(U, I, B, 8, seed=5)
for opt_name in ("adam", "rmsprop"):
    res = {}
    for backend in ("hip", "torch"):
        set_backend(backend)
        m = build(U, I, d, reg, True, seed=11)
        o = OPTS[opt_name](m.parameters())
        snaps = []
        for k in range(8):
            _run_loop(m, o, data[k:k+1])
            snaps.append({n: v.detach().clone() for n, v in m.state_dict().items()})
        res[backend] = snaps
        set_backend("hip")
    for k in range(8):
        for n in res["hip"][k]:
            e = (res["hip"][k][n] - res["torch"][k][n]).abs()
            print(opt_name, "step", k + 1, n.split(".")[-2], "max %.2e  #>2e-5 %d / %d" % (e.max().item(), (e > 2e-5).sum().item(), e.numel()))
