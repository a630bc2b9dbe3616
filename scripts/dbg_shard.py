import sys, os, time
sys.path.insert(0, os.getcwd())
import bench, torch
print("start", file=sys.stderr, flush=True)
try:
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:29611", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    print("pg ok", file=sys.stderr, flush=True)
    t = torch.ones(4, device="cuda", dtype=torch.float64); dist.all_reduce(t); torch.cuda.synchronize(); print("allreduce ok", t, file=sys.stderr, flush=True)
    bench.WORKLOADS["shard1m"]["P"] = 40000
    r = bench.shard_leg(0, 1, 0, dist, torch, steps=2, warmup=1)
    print("leg", r, file=sys.stderr, flush=True)
except BaseException as e:
    import traceback; traceback.print_exc()
print("end", file=sys.stderr, flush=True)
