"""RCCL API smoke at world size 1 (the calls bench.py makes at N > 1): init with device_id, barrier, float64
all_gather_into_tensor, all_reduce(MAX), destroy.  torchrun --nproc-per-node 1 scripts/nccl_smoke.py"""
import os
import torch
import torch.distributed as dist
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
lr = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(lr)
dev = torch.device("cuda", lr)
dist.init_process_group("nccl", device_id=dev)
w = dist.get_world_size()
loc = torch.arange(40, dtype=torch.float64, device=dev).reshape(10, 4)
out = torch.empty((w * 10, 4), dtype=torch.float64, device=dev)
dist.barrier()
dist.all_gather_into_tensor(out, loc)
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
torch.cuda.synchronize()
assert torch.equal(out[:10], loc) and t.item() == 1.5
print("rccl smoke ok: world", w, "backend", dist.get_backend())
dist.destroy_process_group()
