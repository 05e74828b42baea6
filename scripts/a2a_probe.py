import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
for n in (50_000_000, 200_000_000, 275_000_000, 300_000_000):
    a = torch.arange(n, dtype=torch.int32, device="cuda")
    b = torch.zeros(n + 1000, dtype=torch.int32, device="cuda")
    dist.all_to_all_single(b[:n], a[:n], [n], [n])
    torch.cuda.synchronize()
    bad = int((b[:n] != a).sum().item())
    print("n=%d (%.2f GB): mismatching elements %d, first bad %s" % (n, n * 4 / 1e9, bad, int((b[:n] != a).nonzero()[0]) if bad else None), flush=True)
dist.destroy_process_group()
