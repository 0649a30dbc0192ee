#!/usr/bin/env python3
"""Does an RCCL collective survive hipGraph capture here (torch.cuda.CUDAGraph), and what does a replayed all-gather cost?"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29618", RANK="0", WORLD_SIZE="1")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
part = torch.randn((1, 977, 3), dtype=torch.float64, device=dev)
out = torch.zeros((1, 1, 977, 3), dtype=torch.float64, device=dev)
acc = torch.zeros_like(part)
K = 50
dist.all_gather_into_tensor(out.view(1, 977, 3), part)       # communicator up before capture
torch.cuda.synchronize()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        part.add_(1.0)
        dist.all_gather_into_tensor(out.view(1, 977, 3), part)
        acc.add_(out[0])
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        for _ in range(K):
            part.add_(1.0)
            dist.all_gather_into_tensor(out.view(1, 977, 3), part)
            acc.add_(out[0])
    ok = True
except Exception as exc:
    ok = False
    print(json.dumps({"captured": False, "error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}))
if ok:
    before = acc.clone(); p0 = part.clone()
    g.replay(); torch.cuda.synchronize()
    want = before + sum((p0 + i + 1) for i in range(K))
    correct = bool(torch.allclose(acc, want))
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / (20 * K) * 1e6
    print(json.dumps({"captured": True, "replay_correct": correct, "us_per_iteration_add_gather_add": round(us, 2)}))
dist.destroy_process_group()
