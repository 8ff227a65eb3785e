import os, torch, torch.distributed as dist
rank=int(os.environ["RANK"]); world=int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
x=torch.full((1000,), float(rank+1), device="cuda")
dist.all_reduce(x); print(rank, "all_reduce", x[0].item())
y=torch.empty(2000, device="cuda")
try:
    dist.all_gather_into_tensor(y, torch.full((1000,), float(rank), device="cuda")); print(rank, "ag_into", y[0].item(), y[1500].item())
except Exception as e:
    print(rank, "ag_into failed", repr(e)[:200])
try:
    outs=[torch.empty(3, device="cuda") for _ in range(world)]
    dist.all_gather(outs, torch.full((3,), float(rank), device="cuda")); print(rank, "all_gather", [o[0].item() for o in outs])
except Exception as e:
    print(rank, "all_gather failed", repr(e)[:200])
try:
    a=torch.arange(4, device="cuda", dtype=torch.float32)+10*rank; b=torch.empty(4, device="cuda")
    dist.all_to_all_single(b, a); print(rank, "a2a", b.tolist())
except Exception as e:
    print(rank, "a2a failed", repr(e)[:200])
dist.barrier(); dist.destroy_process_group()
