"""By hand on a GPU box:  python tests/manual/rccl_world1.py
RCCL (torch.distributed backend "nccl") initialises on this image with a device id, and shard.reduce_max / reduce_sum -- the only
collectives bench.py issues for N > 1 -- work on a float64 device scalar.  World size 1: the one-GPU box cannot host two RCCL ranks."""
import os, sys
sys.path.insert(0, os.getcwd())
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import __graft_entry__ as entry
fa = entry.load_package()
from flash_attention_cuda_c_amd import shard
print("max", shard.reduce_max(1.25), "sum", shard.reduce_sum(2.5))
dist.barrier(); dist.destroy_process_group(); print("nccl world-1 ok")
