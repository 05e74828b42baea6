"""Create / use / destroy every kind of context many times: device memory (hipMemGetInfo) and host RSS must come back."""
import os, sys, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # only for mem_get_info
import bfc_amd
from bfc_amd import gen

rs = gen.ReadSet(seed=5, G=200_000, cov=20)
seq, qual, off = rs.reads()
s_seq, s_qual = bfc_amd.to_stream(seq, off), bfc_amd.to_stream(qual, off)
soff = off + np.arange(rs.n_reads + 1, dtype=np.uint64)


def rss_mb():
    return int(open("/proc/self/statm").read().split()[1]) * 4096 // (1 << 20)


def one(i):
    g = bfc_amd.GpuCounter(31, 28, max_batch_pos=len(s_seq) + 64, track_order=bool(i & 1))
    g.count_host(s_seq, s_qual)
    tab = g.export_table()
    kc = bfc_amd.GpuKcov(g, max_pos=len(s_seq) + 64) if hasattr(bfc_amd, "GpuKcov") else None
    if kc is not None:
        kc.close()
    tab.close(); g.close()
    g = bfc_amd.GpuCounter(31, 28, filter_mode=1, max_batch_pos=len(s_seq) + 64)
    g.count_host(s_seq, s_qual)
    bf = g.export_bloom(1, resident=bool(i & 2))
    g.close()
    tr = bfc_amd.GpuTrimmer(31, bf, max_pos=len(s_seq) + 64, max_reads=rs.n_reads)
    tr.trim(s_seq, soff, 0.9)
    tr.close(); bf.close()


one(0); one(3)
torch.cuda.synchronize()
free0, _ = torch.cuda.mem_get_info(); rss0 = rss_mb()
for i in range(40):
    one(i)
torch.cuda.synchronize()
free1, _ = torch.cuda.mem_get_info(); rss1 = rss_mb()
print("device memory free: %.1f -> %.1f MiB; host RSS %d -> %d MiB after 40 more rounds" % (free0 / 2**20, free1 / 2**20, rss0, rss1))
assert free0 - free1 < 64 << 20, "device memory leak"
assert rss1 - rss0 < 256, "host memory leak"
print("leak check ok")
