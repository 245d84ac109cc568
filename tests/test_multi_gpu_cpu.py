"""The N > 1 path on CPU: world_size-2 gloo processes, each stepping an emulator-backed shard of the env batch.

What runs on the GPUs as one process per device over RCCL (bench.py --gpus N under torch.distributed.run) runs here as
two processes over gloo, with the package's library handle / device / stream lookup patched onto the wavefront emulator
(the same patches as the emu_backend fixture, applied inside the worker processes) and bench.py's torch.cuda plumbing
(bench.Gpu) replaced by host stand-ins.  Covered: shard ranges and seeds reproduce the single-process batch bit for bit;
the whole-job throughput arithmetic; the 8-GPU forms of BASELINE configs 3 (getup / Fall) and 5 (imitation) and the headline.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_PATCH = r'''
import os, sys, time, json
ROOT = %r
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from smplsim_amd import _lib, batch, shard
from wave_emu import emu
_lib._LIB = emu.lib()                                   # the emulator build of the same C ABI, host tensors
batch._shard_device = lambda index: torch.device("cpu")
batch._launch_stream = lambda device: None
'''

_SHARDS = _PATCH + r'''
rank, local_rank, world = shard.rank_info()
dist = shard.init_process_group("gloo")
TOTAL, STEPS = 10, 3
lo, hi = shard.shard_range(TOTAL, world, rank)
rs = np.random.default_rng(42)                          # the job's action / target streams, indexed by GLOBAL env id
acts = torch.tensor(rs.uniform(-1, 1, (STEPS, TOTAL, 69)), dtype=torch.float32)
trand = torch.tensor(rs.uniform(size=(STEPS + 1, TOTAL, 4)), dtype=torch.float32)

def rollout(lo, hi, seed):
    env = batch.SMPLSimVecEnv(hi - lo, task="HumanoidSpeed", seed=seed, autoreset=False)
    env.reset(task_rand=trand[STEPS, lo:hi].contiguous())
    for t in range(STEPS):
        obs, rew, term, trunc, _ = env.step(acts[t, lo:hi].contiguous(), task_rand=trand[t, lo:hi].contiguous())
    return torch.cat([env.qpos, env.qvel, obs, rew[:, None]], dim=1)

mine = rollout(lo, hi, shard.shard_seed(7, rank))
per = -(-TOTAL // world)
pad = torch.zeros(per, mine.shape[1]); pad[: hi - lo] = mine
parts = [torch.zeros_like(pad) for _ in range(world)]
dist.all_gather(parts, pad)                              # test-side gather only: the data path itself has no collective
n_total = shard.sum_over_ranks(dist, world, hi - lo)
if rank == 0:
    whole = rollout(0, TOTAL, 7)                         # the same job as ONE shard
    got = torch.cat([parts[r][: shard.shard_range(TOTAL, world, r)[1] - shard.shard_range(TOTAL, world, r)[0]] for r in range(world)])
    print(json.dumps({"equal": bool(torch.equal(got, whole)), "units": n_total, "rows": int(got.shape[0]),
                      "moved": float((whole[:, :76] - whole[:1, :76]).abs().max())}))
dist.destroy_process_group()
'''

_BENCH = _PATCH + r'''
import bench

class Ev:
    def record(self): self.t = time.perf_counter()
    def elapsed_time(self, other): return 1e3 * (other.t - self.t)

class HostGpu:                                          # bench.Gpu with host stand-ins (gloo instead of RCCL)
    backend = "gloo"
    device = staticmethod(lambda index: torch.device("cpu"))
    device_count = staticmethod(lambda: int(os.environ.get("SS_TEST_HOST_DEVICES", "2")))
    sync = staticmethod(lambda: None)
    event = staticmethod(lambda: Ev())
    stream_ptr = staticmethod(lambda dev: None)

bench.Gpu = HostGpu
bench.main(sys.argv[1:])
'''


def _run_world(tmp_path, script, args=(), world=2, timeout=600):
    path = tmp_path / "worker.py"
    path.write_text(script % ROOT)
    port = 29700 + (os.getpid() * 7 + len(args)) % 250
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(path), *args], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=timeout) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-3000:] for o in outs]
    return json.loads(outs[0][0].strip().splitlines()[-1])


def test_two_ranks_stepping_shards_reproduce_the_single_process_batch(tmp_path):
    """Each gloo rank steps its shard_range of a 10-env speed-task job (5 + 5 envs, shard seeds 7 and 8) on the emulator; the
    gathered state / observation / reward rows equal the single-process 10-env run bit for bit."""
    res = _run_world(tmp_path, _SHARDS)
    assert res["equal"] and res["units"] == 10 and res["rows"] == 10 and res["moved"] > 1e-3


@pytest.mark.parametrize("workload,extra", [("smpl", []), ("getup", []), ("imitation", ["--clips", "4", "--clip-frames", "40"])])
def test_bench_main_under_world_size_2(tmp_path, workload, extra):
    """bench.py's own main() as two gloo ranks on the emulator: the 8-GPU forms of BASELINE configs 2, 3 and 5 (independent
    shards, barrier + max-over-ranks timing, whole-job value printed by rank 0 only)."""
    n, steps = 3, 2
    res = _run_world(tmp_path, _BENCH, ["--gpus", "2", "--workload", workload, "--envs-per-gpu", str(n), "--steps", str(steps),
                                        "--warmup", "1", "--no-cpu-baseline", *extra])
    assert res["n_gpus"] == 2 and res["scaling"] == "weak" and res["steps"] == steps and res["unit"] == "env-steps/s"
    assert abs(res["value"] - 2 * n * steps / (res["ms_per_step"] * 1e-3 * steps)) < 1e-6 * res["value"]
    assert res["config"]["envs_per_gpu"] == n and res["config"]["obs_finite"] and "cpu_baseline" not in res
    assert res["roofline"]["bound"] == "hbm" and res["roofline"]["achieved"] > 0


def test_bench_main_strong_scaling_switch(tmp_path):
    """--envs-total: the metric's "4096-env rollout at 1/2/4/8 MI355X" read literally — the same envs split over the ranks
    ("scaling": "strong"); the line carries the bound of that reading (config.scaling_note)."""
    res = _run_world(tmp_path, _BENCH, ["--gpus", "2", "--envs-total", "6", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    assert res["scaling"] == "strong" and res["config"]["envs_per_gpu"] == 3 and res["config"]["envs_total"] == 6
    assert abs(res["value"] - 6 * 2 / (res["ms_per_step"] * 1e-3 * 2)) < 1e-6 * res["value"]
    assert "1.25x" in res["config"]["scaling_note"] and "--envs-total" in res["config"]["scaling_note"]


def _run_plain(tmp_path, args, extra_env=None, timeout=900):
    """`python worker.py --gpus N ...` with NO rank variables in the environment and no process fan-out on the test's side: the ranks
    are whatever bench.main starts by itself."""
    path = tmp_path / "worker.py"
    path.write_text(_BENCH % ROOT)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(OMP_NUM_THREADS="2", **(extra_env or {}))
    return subprocess.run([sys.executable, str(path), *args], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)


def test_plain_gpus_flag_spawns_ranks(tmp_path):
    """`bench.py --gpus 2` as the driver calls it (no torchrun around it): main() starts the two ranks itself (spawn_ranks ->
    torch.distributed.run on 127.0.0.1), each pins its slice of the host cores, rank 0 prints ONE line with n_gpus = 2 and the
    whole-job value.  The reference's fan-out is one call as well (examples/benchmark.py:78-81)."""
    n, steps = 3, 2
    pr = _run_plain(tmp_path, ["--gpus", "2", "--envs-per-gpu", str(n), "--steps", str(steps), "--warmup", "1", "--no-cpu-baseline",
                               "--no-reference-contact-set"])
    assert pr.returncode == 0, pr.stderr[-3000:]
    lines = [l for l in pr.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, pr.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["envs_total"] == 2 * n and res["scaling"] == "weak"
    assert abs(res["value"] - 2 * n * steps / (res["ms_per_step"] * 1e-3 * steps)) < 1e-6 * res["value"]
    if len(os.sched_getaffinity(0)) >= 2:
        assert res["config"]["host_cores_pinned_per_rank"] == len(os.sched_getaffinity(0)) // 2


def test_plain_gpus_flag_refuses_more_ranks_than_devices(tmp_path):
    """Fewer devices than --gpus: exit status != 0 and no result line (never a line that says n_gpus 1 for a --gpus 8 command);
    likewise a launcher whose WORLD_SIZE is not the --gpus asked for."""
    pr = _run_plain(tmp_path, ["--gpus", "4", "--envs-per-gpu", "3", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                    {"SS_TEST_HOST_DEVICES": "2"}, timeout=120)
    assert pr.returncode != 0 and "only 2 device(s) visible" in pr.stderr and not any(l.startswith("{") for l in pr.stdout.splitlines())
    pr = _run_plain(tmp_path, ["--gpus", "4", "--envs-per-gpu", "3", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                    {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"}, timeout=120)
    assert pr.returncode != 0 and "WORLD_SIZE=1" in pr.stderr and not any(l.startswith("{") for l in pr.stdout.splitlines())


def test_ranks_pin_disjoint_slices_of_the_host_cores():
    """shard.pin_host_threads: every rank of a node keeps its own slice of the cores the process may run on (bench.py calls it before the
    process group starts); a single rank, or fewer cores than ranks, leaves the affinity alone."""
    from smplsim_amd import shard
    if not hasattr(os, "sched_getaffinity"):
        pytest.skip("no affinity call on this platform")
    before = sorted(os.sched_getaffinity(0))
    try:
        assert shard.pin_host_threads(0, 1) is None and sorted(os.sched_getaffinity(0)) == before
        assert shard.pin_host_threads(0, len(before) + 1) is None
        if len(before) >= 2:
            a = shard.pin_host_threads(0, 2); os.sched_setaffinity(0, before)
            b = shard.pin_host_threads(1, 2)
            assert a and b and not set(a) & set(b) and set(a) | set(b) <= set(before) and sorted(os.sched_getaffinity(0)) == sorted(b)
    finally:
        os.sched_setaffinity(0, before)
