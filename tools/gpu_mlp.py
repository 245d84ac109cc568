#!/usr/bin/env python3
"""Policy-inference microbenchmark: the fused bf16 MFMA path (learning/fast_policy.py) against the torch policy in fp32 and
under bf16 autocast, 4096 x 289 observations through the reference's 2048-1536-1024-1024-512-512 MLP."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from smplsim_amd.learning.fast_policy import FusedPolicyInference
from smplsim_amd.learning.networks import PolicyGaussian

M = int(os.environ.get("NENV", "4096"))
pol = PolicyGaussian(289, 69).cuda().eval()
obs = torch.randn(M, 289, device="cuda")
fast = FusedPolicyInference(pol)
flop = 2.0 * M * sum(l.in_features * l.out_features for l in fast.layers)
gen = torch.Generator(device="cuda"); gen.manual_seed(0)

def timeit(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n

with torch.no_grad():
    t32 = timeit(lambda: pol.select_action(obs.clamp(-5, 5), generator=gen))
    with torch.autocast("cuda", dtype=torch.bfloat16):
        t16 = timeit(lambda: pol.select_action(obs.clamp(-5, 5), generator=gen).float())
    tf = timeit(lambda: fast.select_action(obs, generator=gen))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    import ctypes as C
    from smplsim_amd._lib import lib
    per = []
    x = torch.randn(M, 2048, device="cuda").to(torch.bfloat16)
    for l, w, b in zip(fast.layers, fast.w, fast.b):
        K, N = w.shape[1], w.shape[0]
        y = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
        f = lambda: lib().ss_linear_bf16(C.c_void_p(x.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(y.data_ptr()), M, N, K, N, 1, 0, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        t = timeit(f, 100)
        per.append(f"{K}x{N}: {t*1e6:.1f} us = {2.0*M*N*K/t/1e12:.0f} TF")
print(f"policy forward, {M} envs, {flop/1e9:.1f} GFLOP: torch fp32 {t32*1e6:.0f} us | torch bf16 autocast {t16*1e6:.0f} us | fused MFMA {tf*1e6:.0f} us ({flop/tf/1e12:.0f} TFLOP/s incl. launches)")
print("layers (launch + kernel, back to back):", "; ".join(per))
