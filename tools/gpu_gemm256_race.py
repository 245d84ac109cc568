"""Race screen of the 256 x 256 GEMM kernel (csrc/ss_gemm256.h: copies in flight across barriers, a read one phase after the wait that retires it): the same
launch REPS times on the update's shapes while a second stream streams 1 GB copies through HBM (the copies' latency moves), every result compared bit for bit
with the first launch's and once with the fp32 product.  A copy that lands after its reader would show up as a launch that differs."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smplsim_amd._lib import lib
os.environ["SS_MLP_TRAIN_256"] = "1"
REPS = int(os.environ.get("REPS", "100"))
ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
bf = dict(dtype=torch.bfloat16, device="cuda")
side = torch.cuda.Stream()
hog_a = torch.empty(1 << 29, **bf); hog_b = torch.empty_like(hog_a)
bad = 0
for name, kind, m, n, k in (("forward 53248 x 1536, K 2048", "fwd", 53248, 1536, 2048), ("forward 53248 x 512, K 512", "fwd", 53248, 512, 512),
                            ("dX 53248 x 1024, K 512", "dx", 53248, 1024, 512), ("dX 53248 x 2048, K 1536", "dx", 53248, 2048, 1536),
                            ("ragged 5000 x 1000, K 384", "fwd", 5000, 1000, 384)):
    g = torch.Generator(device="cuda").manual_seed(m + n + k)
    x = (torch.rand(m, k, device="cuda", generator=g) * 2 - 1).to(torch.bfloat16); w = (torch.rand(n, k, device="cuda", generator=g) * 2 - 1).to(torch.bfloat16) / k ** 0.5
    mul = (torch.rand(m, n, device="cuda", generator=g) + 0.5).to(torch.bfloat16)
    ldt = (m + 7) // 8 * 8
    first = None
    for rep in range(REPS):
        with torch.cuda.stream(side):
            hog_b.copy_(hog_a)
        y = torch.zeros(m, n, **bf); yt = torch.zeros(n, ldt, **bf); d = torch.zeros(m, n, **bf)
        if kind == "fwd":
            rc = lib().ss_linear_bf16_train(ptr(x), ptr(w), None, None, ptr(y), ptr(yt), ptr(d), m, n, k, n, ldt, 1, 0, st)
        else:
            rc = lib().ss_linear_bf16_train(ptr(x), ptr(w), None, ptr(mul), ptr(y), ptr(yt), None, m, n, k, n, ldt, 0, 0, st)
        assert rc == 0
        torch.cuda.synchronize()
        if first is None:
            first = (y.clone(), yt.clone(), d.clone())
            z = x[:4096].float() @ w.float().t()
            ref = torch.nn.functional.silu(z) if kind == "fwd" else z * mul[:4096].float()
            err = float((y[:4096].float() - ref).abs().max() / ref.abs().max())
            assert err < 2e-2, err
            assert torch.equal(yt[:, :m], y.t())
        elif not (torch.equal(y, first[0]) and torch.equal(yt, first[1]) and torch.equal(d, first[2])):
            bad += 1
    print(f"{name}: {REPS} launches under a concurrent 1 GB copy stream, first vs fp32 reference {err:.1e}, launches that differ from the first: {bad}", flush=True)
print("race screen", "CLEAN" if bad == 0 else f"FAILED ({bad})")
