"""CPU stand-in for the two GPU stages of the multi-GPU path, built on the oracle (tests only): lets the REAL exchange code
(bfc_amd.dist.exchange / count_batch over torch.distributed) run under gloo with world_size 2 on a box without GPUs, and
pins the owner-computes protocol: ownership by bloom block, rank-major file order, disjoint tables."""
import ctypes as C

import numpy as np
import torch

import oracle


class CpuEngine:
    rec_words = 3  # y0, y1, (rank-major file index << 1 | is_high)

    def __init__(self, rank, world, k, bf_shift, n_hashes=4, l_pre=20, nb1=8, cap=1 << 20):
        self.rank, self.world, self.k, self.b, self.nh, self.nb1 = rank, world, k, bf_shift, n_hashes, nb1
        self.L = oracle.lib()
        self.bf = self.L.orc_bf_new(bf_shift, n_hashes)     # full-size filter; only the owned blocks are ever touched
        self.ch = self.L.orc_ch_new(k, l_pre)
        self.send = torch.zeros(cap * 3, dtype=torch.int64)
        self.recv = torch.zeros(cap * 3, dtype=torch.int64)
        self.n_recv = 0
        self.ordinal = 0
        self.n_seen = 0

    def bucket_of(self, hash_):
        blk = int(hash_) & ((1 << (self.b - 9)) - 1)
        return blk * self.nb1 >> (self.b - 9)               # top bits of the block id -> contiguous bucket ranges per owner

    def scatter(self, share, _unused, _n):
        seq, qual, off = share
        tmp = oracle.Counter(self.k, 12)                    # throw-away state: only the per-k-mer trace is used
        tr = tmp.count(seq, qual, off, trace=True)
        tmp.close()
        n = len(tr)
        rec = np.zeros((n, 3), dtype=np.uint64)
        bkt = np.zeros(n, dtype=np.int64)
        for i in range(n):
            idx = (self.rank << 40) | (self.ordinal + i)
            rec[i] = (tr[i, 1], tr[i, 2], (idx << 1) | (int(tr[i, 3]) & 1))
            bkt[i] = self.bucket_of(tr[i, 0])
        self.ordinal += n
        order = np.argsort(bkt, kind="stable")
        self.send[:n * 3] = torch.from_numpy(rec[order].reshape(-1).view(np.int64))
        return np.bincount(bkt, minlength=self.nb1).astype(np.uint32)

    def process(self, seg_cnt):
        n = int(seg_cnt.sum())
        rec = self.recv[:n * 3].numpy().view(np.uint64).reshape(n, 3)
        for i in np.argsort(rec[:, 2], kind="stable"):      # rank-major file order
            y = (C.c_uint64 * 2)(int(rec[i, 0]), int(rec[i, 1]))
            h = self.L.orc_hash_from_y(self.k, y)
            assert self.bucket_of(h) * self.world // self.nb1 == self.rank, "record routed to the wrong owner"
            if self.L.orc_bf_insert(self.bf, h) == self.nh:
                self.n_seen += 1
                self.L.orc_ch_insert(self.ch, y, int(rec[i, 2]) & 1)

    def result(self):
        n = 1 << (self.b - 3)
        bits = np.ctypeslib.as_array(C.cast(self.L.orc_bf_bits(self.bf), C.POINTER(C.c_uint8)), shape=(n,)).copy()
        l_pre = self.L.orc_ch_lpre(self.ch)
        sizes = np.zeros(1 << l_pre, dtype=np.uint32)
        cnt = self.L.orc_ch_export(self.ch, sizes.ctypes.data_as(C.POINTER(C.c_uint32)), None)
        slots = np.zeros(cnt, dtype=np.uint64)
        self.L.orc_ch_export(self.ch, sizes.ctypes.data_as(C.POINTER(C.c_uint32)), slots.ctypes.data_as(C.POINTER(C.c_uint64)))
        return bits, sizes, slots, self.n_seen
