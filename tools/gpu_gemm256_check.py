"""The 256 x 256 kernel of ss_linear_bf16_train (SS_MLP_TRAIN_256=1) against the 128-row kernel (=0) and torch.matmul (hipBLASLt) on every product of one
PPO update pass (ROWS rows, the reference MLP), uniform random [-1, 1) operands.  Interleaved rounds, median.  Columns: g256 / k128 = the product with
the outputs the pass needs (forward: result + transposed result + derivative; dX: multiplying operand, result + transposed; dW: fp32 accumulate),
*_plain = one bf16 output only (what torch.matmul computes)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from smplsim_amd._lib import lib
M = int(os.environ.get("ROWS", "53248"))
dims = [320, 2048, 1536, 1024, 1024, 512, 512]
ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
bf = dict(dtype=torch.bfloat16, device="cuda")
def once(fn, reps=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
shapes = []
for i in range(6): shapes.append(("fwd%d" % (i + 1), "fwd", M, dims[i + 1], dims[i]))
for i in range(5, 0, -1): shapes.append(("dX%d" % (i + 1), "dx", M, dims[i], dims[i + 1]))
for i in range(6): shapes.append(("dW%d" % (i + 1), "dw", dims[i + 1], dims[i] + 64, M))
tot = {"g256": 0.0, "k128": 0.0, "torch": 0.0}; flops = 0.0
for name, kind, m, n, k in shapes:
    x = (torch.rand(m, k, device="cuda") * 2 - 1).to(torch.bfloat16); w = (torch.rand(n, k, device="cuda") * 2 - 1).to(torch.bfloat16)
    if kind == "dw":
        y = torch.zeros(m, n, device="cuda")
        f = lambda: lib().ss_linear_bf16_train(ptr(x), ptr(w), None, None, ptr(y), None, None, m, n, k, n, 0, 0, 1, st)
    else:
        y = torch.empty(m, n, **bf); yt = torch.empty(n, m, **bf); g = torch.empty(m, n, **bf)
        if kind == "fwd":
            f = lambda: lib().ss_linear_bf16_train(ptr(x), ptr(w), None, None, ptr(y), ptr(yt), ptr(g), m, n, k, n, m, 1, 0, st)
        else:
            f = lambda: lib().ss_linear_bf16_train(ptr(x), ptr(w), None, ptr(g), ptr(y), ptr(yt), None, m, n, k, n, m, 0, 0, st)
    plain = lambda: lib().ss_linear_bf16_train(ptr(x), ptr(w), None, None, ptr(y), None, None, m, n, k, n, 0, 0, 1 if kind == "dw" else 0, st)
    def run(which, fn):
        if which == "torch": return once(lambda: torch.matmul(x, w.t()))
        os.environ["SS_MLP_TRAIN_256"] = "1" if which.startswith("g256") else "0"
        return once(fn)
    ok256 = k % 128 == 0
    variants = (["g256", "g256_plain"] if ok256 else []) + ["k128", "k128_plain", "torch"]
    t = {v: [] for v in variants}
    pick = lambda v: plain if v.endswith("plain") else f
    for v in variants: run(v, pick(v))   # warm-up
    for rnd in range(5):
        for v in variants: t[v].append(run(v, pick(v)))
    med = {v: float(np.median(t[v])) for v in variants}
    err = ""
    if ok256:                                                      # the 256-tile kernel's result of this product against fp32 on the same operands
        os.environ["SS_MLP_TRAIN_256"] = "1"
        ref = x[:2048].float() @ w.float().t()
        if kind == "dw":
            y.zero_(); f(); torch.cuda.synchronize(); got = y[:2048]
        else:
            plain(); torch.cuda.synchronize(); got = y[:2048].float()
        e_ = float((got - ref).abs().max() / ref.abs().max())
        err = f"  max err / max |ref| {e_:.1e}" + ("  <-- WRONG RESULT: the timings of this line mean nothing" if not e_ < 2e-2 else "")
    gf = 2.0 * m * n * k / 1e9
    flops += gf
    tot["g256"] += med.get("g256", med["k128"]); tot["k128"] += med["k128"]; tot["torch"] += med["torch"]
    print(f"{name:5s} [{m} x {n}, K {k}] {gf:6.0f} GFLOP  " + "  ".join(f"{v}: {med[v]:7.1f} us {gf / med[v] * 1e3:5.0f} TF/s" for v in variants) + err, flush=True)
print(f"one pass: {flops / 1e3:.2f} TFLOP; 256-tile kernel {tot['g256'] / 1e3:.2f} ms = {flops / tot['g256'] * 1e3:.0f} TF/s; 128-row kernel {tot['k128'] / 1e3:.2f} ms = {flops / tot['k128'] * 1e3:.0f}; torch.matmul alone {tot['torch'] / 1e3:.2f} ms = {flops / tot['torch'] * 1e3:.0f}")
