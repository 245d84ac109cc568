"""Tile-width / K-split sweep of ss_linear_bf16_train over the products of one PPO update pass (batch 53248, the reference MLP), against
torch's bf16 matmul on the same shapes.  Prints per product the time of every tile width and the best; sums per pass."""
import ctypes as C, os, subprocess, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smplsim_amd._lib import lib
M = int(os.environ.get("ROWS", "53248"))
dims = [320, 2048, 1536, 1024, 1024, 512, 512]
ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
bf = dict(dtype=torch.bfloat16, device="cuda")
shapes = []
for i in range(6):
    shapes.append(("fwd%d" % (i + 1), "fwd", M, dims[i + 1], dims[i]))
for i in range(5, 0, -1):
    shapes.append(("dX%d" % (i + 1), "dx", M, dims[i], dims[i + 1]))
for i in range(6):
    shapes.append(("dW%d" % (i + 1), "dw", dims[i + 1], dims[i] + 64, M))
mode = os.environ.get("MODE", "sweep")
tot_best, tot_torch, flops = 0.0, 0.0, 0.0
for name, kind, m, n, k in shapes:
    x = torch.randn(m, k, device="cuda").to(torch.bfloat16); w = torch.randn(n, k, device="cuda").to(torch.bfloat16)
    res = {}
    for bn in (64, 128, 192, 256):
        if bn > n: continue
        os.environ["SS_MLP_TRAIN_BN_LIVE"] = str(bn)
        if kind == "dw":
            y = torch.zeros(m, n, device="cuda")
            f = lambda: lib().ss_linear_bf16_train(ptr(x), ptr(w), None, None, ptr(y), None, None, m, n, k, n, 0, 0, 1, st)
        else:
            y = torch.empty(m, n, **bf); yt = torch.empty(n, m, **bf); g = torch.empty(m, n, **bf)
            if kind == "fwd":
                f = lambda: lib().ss_linear_bf16_train(ptr(x), ptr(w), None, None, ptr(y), ptr(yt), ptr(g), m, n, k, n, m, 1, 0, st)
            else:
                f = lambda: lib().ss_linear_bf16_train(ptr(x), ptr(w), None, ptr(g), ptr(y), ptr(yt), None, m, n, k, n, m, 0, 0, st)
        res[bn] = timeit(f)
    tt = timeit(lambda: torch.matmul(x, w.t()))
    b = min(res, key=res.get)
    gf = 2.0 * m * n * k / 1e9
    flops += gf; tot_best += res[b]; tot_torch += tt
    print(f"{name:5s} [{m} x {n}, K {k}] {gf:6.0f} GFLOP  " + "  ".join(f"bn{c}: {t:7.1f} us" for c, t in res.items()) + f"   best bn{b} {gf / res[b] * 1e3:6.0f} TF/s   torch bf16 matmul {tt:7.1f} us {gf / tt * 1e3:6.0f} TF/s")
print(f"one pass (6 fwd + 5 dX + 6 dW): {flops / 1e3:.2f} TFLOP; best tiles {tot_best / 1e3:.2f} ms = {flops / tot_best:.0f} TF/s; torch matmul alone {tot_torch / 1e3:.2f} ms = {flops / tot_torch:.0f} TF/s")
