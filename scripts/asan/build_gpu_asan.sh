#!/bin/bash
# The drop-in binary (reference main() + reader + corrector, unmodified, over this library) with every HOST translation unit
# under AddressSanitizer; device code is built as usual (-fno-gpu-sanitize).  Needs /root/reference; output in oracle/_ref/.
#   bash scripts/asan/build_gpu_asan.sh && gpurun -- 'bash scripts/asan/run_gpu_asan.sh'
set -e
cd "$(dirname "$0")/../.."
REF=${REF:-/root/reference}
CL=/opt/rocm/lib/llvm/bin/clang
O=oracle/_ref/asan; mkdir -p $O
SAN="-fsanitize=address -fno-omit-frame-pointer -g -O1"
for f in kthread utils bseq bfc; do $CL $SAN -w -I$REF -c -o $O/$f.o $REF/$f.c; done
$CL $SAN -w -I$REF -Dbfc_correct=bfc_correct_cpu -c -o $O/correct_cpu.o $REF/correct.c
for f in bfc_host bfc_count bfc_trim; do $CL $SAN -std=gnu99 -Iinclude -Ibfc_amd/csrc -c -o $O/$f.o bfc_amd/csrc/$f.c; done
for f in bfcg_kernels bfcg_ctx; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fsanitize=address -fno-omit-frame-pointer -fno-gpu-sanitize -Wno-pass-failed -std=c++17 -Wno-unused-value -Iinclude -Ibfc_amd/csrc -c -o $O/$f.o bfc_amd/csrc/$f.hip; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fsanitize=address -fno-gpu-sanitize -rdynamic -o oracle/_ref/bfc-dropin-asan $O/*.o -lm -lz -lpthread
echo built oracle/_ref/bfc-dropin-asan
