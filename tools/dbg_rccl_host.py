"""Does a 1-rank RCCL all_reduce block the host while the GPU has a backlog?  (debugging aid)"""
import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
a = torch.randn(8192, 8192, device=dev)
g = torch.randn(32 * 1024 * 1024, device=dev)
side = torch.cuda.Stream()
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        b = a @ a                       # ~7 ms each in fp32 -> ~140 ms backlog
    t1 = time.perf_counter()
    ev = torch.cuda.Event(); ev.record()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        w = dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=True)
    t2 = time.perf_counter()
    w.wait()
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print(f"enqueue matmuls {1e3*(t1-t0):.2f} ms | all_reduce call {1e3*(t2-t1):.2f} ms | wait() {1e3*(t3-t2):.2f} ms | drain {1e3*(t4-t3):.2f} ms")
dist.destroy_process_group()
