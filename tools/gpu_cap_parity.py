"""Parity against the oracle (100 Newton iterations, float64) on the benchmark distribution versus the kernel's
Newton-iteration cap per mj_step: worst and high-percentile relative state errors over a sample of env-steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from helpers import oracle_model
from oracle import oracle as O
from smplsim_amd.batch import SMPLSimVecEnv
om = oracle_model()
npy = lambda t: t.detach().cpu().numpy()
for cap in tuple(int(c) for c in os.environ.get("CAPS", "8,6,5,4,3").split(",")):
    env = SMPLSimVecEnv(512, autoreset=True, seed=11, newton_iters=cap)
    g = torch.Generator(device=env.device); g.manual_seed(11)
    env.reset()
    rs = np.random.default_rng(0)
    errs = []
    for t in range(60):
        act = torch.rand(512, 69, generator=g, device=env.device) * 2 - 1
        pick = rs.choice(512, 10, replace=False) if t >= 10 and t % 2 == 1 else []
        pre = {k: npy(getattr(env, k)).copy() for k in ("qpos", "qvel", "qpos_prev", "qvel_prev", "qacc_warm", "nwarn")} if len(pick) else None
        env.step(act)
        if not len(pick): continue
        torch.cuda.synchronize()
        pq, pv, nw, a_np, te, tu, it = npy(env.qpos), npy(env.qvel), npy(env.nwarn), npy(act), npy(env.terminated), npy(env.truncated), npy(env.solver_iters)
        for i in pick:
            if nw[i] != pre["nwarn"][i] or te[i] or tu[i]: continue
            d = O.OracleData(om)
            d.qpos = pre["qpos_prev"][i]; d.qvel = pre["qvel_prev"][i]; d.forward()
            d.qpos = pre["qpos"][i]; d.qvel = pre["qvel"][i]; d.warm = pre["qacc_warm"][i]
            for s_ in range(15):
                d.ctrl = d.spd_torque(a_np[i]); d.step()
            sc = max(1.0, np.abs(d.qvel).max())
            errs.append((np.abs(pq[i] - d.qpos).max() / sc, np.abs(pv[i] - d.qvel).max() / sc, it[i]))
    e = np.array(errs)
    print(f"cap {cap}: n={len(e)} rel dqpos p50 {np.median(e[:,0]):.1e} p99 {np.quantile(e[:,0],.99):.1e} max {e[:,0].max():.1e} | rel dqvel p50 {np.median(e[:,1]):.1e} p99 {np.quantile(e[:,1],.99):.1e} max {e[:,1].max():.1e} | iters max {e[:,2].max():.0f}")
