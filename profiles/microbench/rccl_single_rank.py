"""RCCL sanity on a one-GPU box: the collectives bench.py / apss.dist use, with a world of one rank (the most the box can
host); catches API misuse (device_id=, group creation, tensor dtypes), not multi-GPU behaviour."""
import os

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
t = torch.tensor([1.0, 2.0, 3.0], dtype=torch.float64, device=dev)
dist.all_reduce(t)
n = torch.tensor([5], dtype=torch.int64, device=dev)
out = [torch.zeros_like(n)]
dist.all_gather(out, n)
g = dist.new_group([0])
p = torch.ones(7, dtype=torch.float32, device=dev)
dist.all_reduce(p, op=dist.ReduceOp.SUM, group=g)
m = torch.tensor([0.5], dtype=torch.float64, device=dev)
dist.all_reduce(m, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
print("rccl single-rank ok", t.tolist(), out[0].item(), p.sum().item(), m.item())
dist.destroy_process_group()
