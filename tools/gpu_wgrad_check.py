"""ss_wgrad_bf16 (weight gradient from dZ and h as they lie: contraction over rows, fragments by ds_read_b64_tr_b16) against the fp32 product, and timed against
the transposed-operand path it replaces (ss_linear_bf16_train's accumulating form on dZ^T, h^T) and torch (dz.t() @ h), on the weight gradients of one PPO update pass."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from smplsim_amd._lib import lib
M = int(os.environ.get("ROWS", "53248"))
dims = [384, 2048, 1536, 1024, 1024, 512, 512]
ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def once(fn, reps=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
tot = {"tn": 0.0, "old": 0.0, "torch": 0.0}
for i in range(6):
    n_out, n_in = dims[i + 1], dims[i]
    g = torch.Generator(device="cuda").manual_seed(i)
    dz = (torch.rand(M, n_out, device="cuda", generator=g) * 2 - 1).to(torch.bfloat16); h = (torch.rand(M, n_in, device="cuda", generator=g) * 2 - 1).to(torch.bfloat16)
    dzt, ht = dz.t().contiguous(), torch.zeros(n_in + 64, M, dtype=torch.bfloat16, device="cuda")
    ht[:n_in] = h.t()
    dw = torch.zeros(n_out, n_in, device="cuda"); dw_old = torch.zeros(n_out, n_in + 64, device="cuda")
    f_tn = lambda: lib().ss_wgrad_bf16(ptr(dz), ptr(h), ptr(dw), M, n_out, n_in, n_out, n_in, n_in, st)
    f_old = lambda: lib().ss_linear_bf16_train(ptr(dzt), ptr(ht), None, None, ptr(dw_old), None, None, n_out, n_in + 64, M, n_in + 64, 0, 0, 1, st)
    f_torch = lambda: torch.matmul(dz.t(), h)
    assert f_tn() == 0
    torch.cuda.synchronize()
    ref = dz[:, :256].float().t() @ h.float()
    err = float((dw[:256] - ref).abs().max() / ref.abs().max())
    t = {"tn": [], "old": [], "torch": []}
    for k, f in (("tn", f_tn), ("old", f_old), ("torch", f_torch)): once(f)
    for rnd in range(5):
        for k, f in (("tn", f_tn), ("old", f_old), ("torch", f_torch)): t[k].append(once(f))
    med = {k: float(np.median(v)) for k, v in t.items()}
    for k in tot: tot[k] += med[k]
    gf = 2.0 * M * n_out * n_in / 1e9
    print(f"dW{i + 1} [{n_out} x {n_in}, K {M}] {gf:5.0f} GFLOP  untransposed {med['tn']:7.1f} us {gf / med['tn'] * 1e3:5.0f} TF/s   transposed operands {med['old']:7.1f} us   torch dz.t() @ h {med['torch']:7.1f} us"
          f"   max err / max |ref| {err:.1e}" + ("  <-- WRONG" if not err < 1e-3 else ""), flush=True)
print(f"six weight gradients: untransposed {tot['tn']:.0f} us, transposed operands {tot['old']:.0f} us, torch {tot['torch']:.0f} us")
