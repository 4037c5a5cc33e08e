"""GPU-box rehearsal of sharding.AsyncGather with a 1-rank RCCL group (the only NCCL world a 1-GPU box allows)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from mrs_optic_flow_amd import sharding
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
ag = sharding.AsyncGather((8, 4, 2), torch.float64, dev, 8)
for i in range(5):
    buf = ag.slot(); buf.fill_(float(i)); ag.submit().done()
ag.drain(); torch.cuda.synchronize()
print("async gather ok", [float(f[0, 0, 0]) for f in ag.full])
x = torch.arange(6, dtype=torch.float64, device=dev).reshape(3, 2)
print("sync gather ok", sharding.gather_results(x, 3).tolist())
dist.barrier(); dist.destroy_process_group()
