#!/usr/bin/env python3
"""One layer of the policy MLP (4096 x 2048 -> 1536, bf16 MFMA kernel) launched repeatedly: target of the rocprofv3 PMC passes."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from smplsim_amd._lib import lib
M, K, N = 4096, int(os.environ.get("GK", 2048)), int(os.environ.get("GN", 1536))
x = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = torch.randn(N, K, device="cuda").to(torch.bfloat16) * 0.02
b = torch.zeros(N, device="cuda"); y = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
p = lambda t: C.c_void_p(t.data_ptr())
for _ in range(int(os.environ.get("REPS", 30))):
    lib().ss_linear_bf16(p(x), p(w), p(b), p(y), M, N, K, N, 1, 0, C.c_void_p(torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
