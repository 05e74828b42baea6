// Achievable HBM bandwidth on the box (SURVEY 8d asks for a probe next to the 8 TB/s spec figure): device properties, hipMemcpyDtoD,
// and read-only / write-only / copy kernels over 8 GiB with 16-byte accesses.
// Build: hipcc --offload-arch=gfx950 -O3 -o build/hbm_probe scripts/probes/hbm_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_read(const uint4 *__restrict__ a, uint64_t n, uint4 *sink)
{
	uint4 acc = make_uint4(0, 0, 0, 0);
	for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) { uint4 v = a[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
	if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) *sink = acc;
}
__global__ __launch_bounds__(256) void k_write(uint4 *__restrict__ a, uint64_t n)
{
	for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) a[i] = make_uint4((uint32_t)i, 1, 2, 3);
}
__global__ __launch_bounds__(256) void k_copy(const uint4 *__restrict__ a, uint4 *__restrict__ b, uint64_t n)
{
	for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) b[i] = a[i];
}

template <typename F> static double timed(F f)
{
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	f(); CK(hipDeviceSynchronize());
	CK(hipEventRecord(e0, 0)); f(); f(); f(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
	float ms; CK(hipEventElapsedTime(&ms, e0, e1));
	return ms / 3;
}

int main()
{
	hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
	printf("%s: %d CUs, memory clock %d kHz x bus %d bit => %.0f GB/s (x2 for DDR), %.1f GiB, L2 %d MiB\n", pr.name, pr.multiProcessorCount, pr.memoryClockRate,
	       pr.memoryBusWidth, (double)pr.memoryClockRate * 1e3 * pr.memoryBusWidth / 8 / 1e9, pr.totalGlobalMem / 1073741824.0, pr.l2CacheSize >> 20);
	const uint64_t bytes = 8ull << 30, n = bytes / 16;
	uint4 *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
	const int grid = 256 * 32;
	double ms;
	ms = timed([&] { CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0)); });
	printf("hipMemcpyDtoD 8 GiB          : %7.3f ms = %6.0f GB/s read + %6.0f GB/s written = %6.0f GB/s total\n", ms, bytes / ms / 1e6, bytes / ms / 1e6, 2 * bytes / ms / 1e6);
	ms = timed([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n); });
	printf("copy kernel (16 B per lane)  : %7.3f ms = %6.0f GB/s total\n", ms, 2 * bytes / ms / 1e6);
	ms = timed([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, n, b); });
	printf("read-only kernel             : %7.3f ms = %6.0f GB/s\n", ms, bytes / ms / 1e6);
	ms = timed([&] { hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, b, n); });
	printf("write-only kernel            : %7.3f ms = %6.0f GB/s\n", ms, bytes / ms / 1e6);
	return 0;
}
