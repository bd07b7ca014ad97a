"""Times sharding.gather_results between the ranks of a torch.distributed run (gloo when
NMPC_BENCH_REHEARSE=1 lets the ranks share a GPU, RCCL otherwise)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from nmpc_amd import sharding

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
rehearse = os.environ.get("NMPC_BENCH_REHEARSE") == "1"
local = int(os.environ["LOCAL_RANK"]) % torch.cuda.device_count() if rehearse else int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
dist.init_process_group("gloo" if rehearse else "nccl", rank=rank, world_size=world)
B = 4096
u = torch.rand((B, 3), dtype=torch.float64, device=dev); st = torch.zeros(B, dtype=torch.int32, device=dev)
it = torch.full((B,), rank, dtype=torch.int32, device=dev)
for _ in range(2):
    sharding.gather_results(u, st, it, counts=[B] * world)
torch.cuda.synchronize(); dist.barrier()
t0 = time.perf_counter()
for _ in range(10):
    r = sharding.gather_results(u, st, it, counts=[B] * world)
torch.cuda.synchronize()
if rank == 0:
    print("gather of %d x %d records: %.3f ms per call; iterations column ok: %s"
          % (world, B, (time.perf_counter() - t0) / 10 * 1e3, bool((r[2].view(world, B)[1] == 1).all())))
dist.destroy_process_group()
