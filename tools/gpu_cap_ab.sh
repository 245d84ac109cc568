cd $GRAFT_REPO_ROOT
for v in base cap12; do
if [ $v = base ]; then L=smplsim_amd/libsmplsim_hip.so; else L=smplsim_amd/variants/libsmplsim_hip_$v.so; fi
SMPLSIM_HIP_LIB=$L python - <<PY
import sys, time, os, torch
sys.path.insert(0, os.getcwd())
from smplsim_amd.batch import SMPLSimVecEnv, _check, _ptr
from smplsim_amd._lib import lib
N = 4096
env = SMPLSimVecEnv(N, autoreset=True, seed=1234, self_collision=True)
g = torch.Generator(device=env.device); g.manual_seed(1234)
env.reset()
for _ in range(20): env.step(torch.rand(N, 69, generator=g, device=env.device) * 2 - 1)
tr = torch.zeros(N, dtype=torch.int32, device=env.device); _check(lib().ss_debug_self_truncation(env.handle, _ptr(tr)))
torch.cuda.synchronize(); t0 = time.perf_counter(); mx = 0
for _ in range(60):
    env.step(torch.rand(N, 69, generator=g, device=env.device) * 2 - 1); mx = max(mx, int(env.self_contacts.max()))
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 60
print("$v", f"{N/dt:,.0f} env-steps/s {1e3*dt:.3f} ms/step  truncated mj_step frac {tr.sum().item()/(N*60*15):.4f} max kept {mx}", env.launch_info())
PY
done
