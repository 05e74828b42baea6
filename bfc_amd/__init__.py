"""bfc_amd -- MI355X-native k-mer counting for BFC (count.c + bbf.c + htab.c of lh3/bfc r181).

The product is ``libbfc_gpu.so`` (C ABI: include/bfc_gpu.h); this package is its thin Python
mirror of the reference's count-phase interface.  Nothing here computes k-mers on the CPU.
"""
from .api import (BfcGpuError, GpuCounter, GpuGroup, GpuKcov, GpuTrimmer, HostBloom, HostTable, bfc_count, bfc_opt_by_size, bfc_opt_init, pack_planes, to_stream)  # noqa: F401
from ._lib import BfcOpt  # noqa: F401
