"""Where a batch's time goes on the multi-GPU path (run under torchrun; world size 1 works: the exchange is then a device copy)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import numpy as np
rank = int(os.environ.get("RANK", 0)); world = int(os.environ.get("WORLD_SIZE", 1)); local = int(os.environ.get("LOCAL_RANK", 0))
torch.cuda.set_device(local)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
import bfc_amd
from bfc_amd import gen
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import mg_protocol as bdist
rs = gen.ReadSet(seed=2 + rank, G=4_600_000, cov=100)
seq, qual, off = rs.reads()
s_seq, s_qual = bfc_amd.to_stream(seq, off), bfc_amd.to_stream(qual, off)
stride, br = rs.L + 1, 786432
g = bfc_amd.GpuCounter(31, 33, device=local, max_batch_pos=br * stride, rank=rank, n_ranks=world)
eng = bdist.GpuEngine(g)
d_seq = g.dev_alloc(len(s_seq)); d_qual = g.dev_alloc(len(s_qual)); g.h2d(d_seq, s_seq); g.h2d(d_qual, s_qual)
acc = dict(scatter=0.0, exchange=0.0, process=0.0, total=0.0)
for step in range(4):
    g.reset()
    t_step = time.perf_counter()
    for r0 in range(0, rs.n_reads, br):
        r1 = min(rs.n_reads, r0 + br)
        t0 = time.perf_counter(); counts = eng.scatter(d_seq + r0 * stride, d_qual + r0 * stride, (r1 - r0) * stride)
        t1 = time.perf_counter(); seg = bdist.exchange(eng, counts)
        t2 = time.perf_counter(); eng.process(seg)
        t3 = time.perf_counter()
        if step:
            acc["scatter"] += t1 - t0; acc["exchange"] += t2 - t1; acc["process"] += t3 - t2
    g.sync()
    if step:
        acc["total"] += time.perf_counter() - t_step
if rank == 0:
    print({k_: round(v / 3 * 1e3, 3) for k_, v in acc.items()}, "ms per step (host wall per phase; stage B of a batch runs under the next batch's scatter + exchange)")
dist.destroy_process_group()
