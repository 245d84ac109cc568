#!/usr/bin/env python3
"""On-device PPO on the batched stepper: the training loop of the reference's `python smpl_sim/run.py env=speed`
 with sampler, GAE and update on the GPU.  Prints one JSON line per epoch and a timing summary
(env-steps/s of the sampler alone = stepper + policy inference, and of the whole epoch including the update).

    python tools/train_ppo.py --task HumanoidSpeed --envs 4096 --epochs 5
    python tools/train_ppo.py --task HumanoidIm --envs 2048 --epochs 30      # motion imitation (synthetic clips unless --motion-file)
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from smplsim_amd.agents.ppo import AgentPPO, PPOConfig
from smplsim_amd.batch import SMPLSimVecEnv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--task", default="HumanoidSpeed")
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--min-batch-size", type=int, default=51200)
    ap.add_argument("--hidden", default="2048,1536,1024,1024,512,512")
    ap.add_argument("--opt-epochs", type=int, default=10)
    ap.add_argument("--amp-bf16", action="store_true", help="bf16 autocast for the update passes (not the reference numerics)")
    ap.add_argument("--mfma-inference", action="store_true", help="sampler: policy forward by the library's fused bf16 MFMA kernels")
    ap.add_argument("--mfma-update", action="store_true", help="update: forward and backward passes of both networks on the library's own GEMM (learning/fused_train.py)")
    ap.add_argument("--save", default="")
    ap.add_argument("--log-every", type=int, default=1)
    ap.add_argument("--motion-file", default="", help="HumanoidIm: AMASS-style pickle ({key: {pose_aa, trans, fps}}); default = synthetic clips")
    args = ap.parse_args()
    if args.task == "HumanoidIm":
        from smplsim_amd.batch import ShardModel
        from smplsim_amd.imitation import SMPLSimImitationVecEnv
        from smplsim_amd.motion_lib import MotionLibSMPL, Skeleton
        model = ShardModel(device=0)
        if args.motion_file:
            clips = args.motion_file
        else:
            sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            from bench import synthetic_clips
            clips = synthetic_clips(256, 300, 77)
        ml = MotionLibSMPL(clips, Skeleton.from_model_const(model.mc), device=0)
        ml.load_motions()
        env = SMPLSimImitationVecEnv(args.envs, ml, model=model, seed=0)
    else:
        env = SMPLSimVecEnv(args.envs, task=args.task, autoreset=True, seed=0)
    cfg = PPOConfig(hidden=tuple(int(x) for x in args.hidden.split(",")), min_batch_size=args.min_batch_size, opt_num_epochs=args.opt_epochs, amp_bf16=args.amp_bf16, mfma_inference=args.mfma_inference, mfma_update=args.mfma_update)
    agent = AgentPPO(env, cfg, seed=0)
    ts, tu, n = 0.0, 0.0, 0
    for ep in range(args.epochs):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        batch = agent.sample()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        info = agent.update_params(batch)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        T, N = batch["rewards"].shape
        if ep > 0:                                           # epoch 0 carries allocator / library warm-up
            ts += t1 - t0; tu += t2 - t1; n += T * N
        if ep % args.log_every and ep != args.epochs - 1:
            continue
        ep_len = float(T * N / max(1.0, float((1.0 - batch["not_done"]).sum())))
        print(json.dumps({"epoch": ep, "samples": T * N, "sample_s": round(t1 - t0, 4), "update_s": round(t2 - t1, 4),
                          "mean_episode_len": round(ep_len, 1),
                          **{k: round(float(v), 5) for k, v in info.items()}}))
    if n:
        print(json.dumps({"summary": "on-device PPO", "task": args.task, "envs": args.envs, "mlp": args.hidden,
                          "sampler_env_steps_per_s": round(n / ts), "epoch_env_steps_per_s": round(n / (ts + tu)),
                          "sample_fraction": round(ts / (ts + tu), 3)}))
    if args.task == "HumanoidIm":                               # tracking quality of the mean action, every clip from its first frame
        agent.policy_net.eval()
        print(json.dumps({"eval": "compute_metrics_lite over all clips (mm)", **{k: round(v, 3) if isinstance(v, float) else v for k, v in
                          env.evaluate(lambda o: agent._prep_actions(agent.policy_net.select_action(agent._prep_obs(o), True))).items()}}))
        print(json.dumps({"eval": "PD clip replay (no policy)", **{k: round(v, 3) if isinstance(v, float) else v for k, v in env.evaluate().items()}}))
    if args.save:
        torch.save(agent.get_full_state_weights(), args.save)


if __name__ == "__main__":
    main()
