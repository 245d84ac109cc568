"""RCCL smoke on one GPU: the exact calls bench.py makes for N>1 (init, barrier, max all-reduce), world size 1."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
import torch
from smplsim_amd import shard
dist = shard.init_process_group("nccl", 0)
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.barrier(); torch.cuda.synchronize()
t = torch.tensor([3.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
print("nccl ok", float(t.item()))
dist.destroy_process_group()
