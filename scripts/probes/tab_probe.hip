// Microbenchmark behind DESIGN.md section 7 item 1: what do 8-byte table upserts (load + atomicCAS) cost on MI355X as a function of
// where they land?  (a) anywhere in a table of 2^T slots, (b) inside the issuing workgroup's own segment of 2^S slots.
// Build: hipcc --offload-arch=gfx950 -O3 -o build/tab_probe scripts/probes/tab_probe.hip ; run: build/tab_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }

// mode 0: random slot in the whole table; mode 1: random slot in the segment of this workgroup (seg_shift slots); per thread n_per upserts
template <int MODE, bool ATOMIC_LOAD>
__global__ __launch_bounds__(256) void k_upsert(unsigned long long *tab, int tab_shift, int seg_shift, int n_per, uint64_t seed)
{
	const uint64_t gid = blockIdx.x * 256ull + threadIdx.x;
	const uint64_t n_seg = 1ull << (tab_shift - seg_shift);
	for (int i = 0; i < n_per; ++i) {
		uint64_t h = mix(gid * 977 + i + seed);
		uint64_t pos;
		if (MODE == 0) pos = h & ((1ull << tab_shift) - 1);
		else pos = ((blockIdx.x % n_seg) << seg_shift) | (h & ((1ull << seg_shift) - 1));
		unsigned long long cur = ATOMIC_LOAD ? __hip_atomic_load(&tab[pos], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : tab[pos];
		for (;;) {
			unsigned long long nv = cur + 1;
			unsigned long long old = atomicCAS(&tab[pos], cur, nv);
			if (old == cur) break;
			cur = old;
		}
	}
}

// unit costs: OP 0 = plain load only (result folded into a dummy store that never happens), 1 = non-returning atomicAdd only,
// 2 = returning atomicAdd only, 3 = atomicCAS only (expected 0 -> mostly fails after the first pass), 4 = plain load + non-returning atomicAdd
template <int OP>
__global__ __launch_bounds__(256) void k_unit(unsigned long long *tab, int tab_shift, int n_per, uint64_t seed, unsigned long long *sink)
{
	const uint64_t gid = blockIdx.x * 256ull + threadIdx.x;
	unsigned long long acc = 0;
	for (int i = 0; i < n_per; ++i) {
		uint64_t pos = mix(gid * 977 + i + seed) & ((1ull << tab_shift) - 1);
		if (OP == 0) acc += tab[pos];
		else if (OP == 1) __hip_atomic_fetch_add(&tab[pos], 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		else if (OP == 2) acc += atomicAdd(&tab[pos], 1ULL);
		else if (OP == 3) acc += atomicCAS(&tab[pos], 0ULL, 1ULL);
		else { unsigned long long v = tab[pos]; if (v != 0xdeadbeefULL) __hip_atomic_fetch_add(&tab[pos], 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
	}
	if (acc == 0x123456789ULL) *sink = acc;
}
template <int OP>
static void unit(const char *name, unsigned long long *tab, int tab_shift, int blocks, int n_per)
{
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	hipLaunchKernelGGL((k_unit<OP>), dim3(blocks), dim3(256), 0, 0, tab, tab_shift, n_per, 7, tab);
	CK(hipDeviceSynchronize());
	CK(hipEventRecord(e0, 0));
	hipLaunchKernelGGL((k_unit<OP>), dim3(blocks), dim3(256), 0, 0, tab, tab_shift, n_per, 99, tab);
	CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
	float ms; CK(hipEventElapsedTime(&ms, e0, e1));
	double n = (double)blocks * 256 * n_per;
	printf("%-58s table 2^%d slots: %8.3f ms for %.0f M ops = %6.1f G ops/s\n", name, tab_shift, ms, n / 1e6, n / ms / 1e6);
}

template <int MODE, bool AL>
static void run(const char *name, unsigned long long *tab, int tab_shift, int seg_shift, int blocks, int n_per)
{
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	hipLaunchKernelGGL((k_upsert<MODE, AL>), dim3(blocks), dim3(256), 0, 0, tab, tab_shift, seg_shift, n_per, 1);
	CK(hipDeviceSynchronize());
	CK(hipEventRecord(e0, 0));
	hipLaunchKernelGGL((k_upsert<MODE, AL>), dim3(blocks), dim3(256), 0, 0, tab, tab_shift, seg_shift, n_per, 12345);
	CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
	float ms; CK(hipEventElapsedTime(&ms, e0, e1));
	double n = (double)blocks * 256 * n_per;
	printf("%-58s table 2^%d slots, segment 2^%d: %8.3f ms for %.0f M upserts = %6.1f G upserts/s\n", name, tab_shift, seg_shift, ms, n / 1e6, n / ms / 1e6);
}

int main()
{
	const int big = 30, small = 25;
	unsigned long long *tab; CK(hipMalloc(&tab, 8ull << big)); CK(hipMemset(tab, 0, 8ull << big));
	const int blocks = 262144, n_per = 1; // 67 M upserts, 256 per workgroup (one aggregated hand-over of config c3)
	run<0, true>("random, atomic load + CAS", tab, big, 12, blocks, n_per);
	run<0, false>("random, plain load + CAS", tab, big, 12, blocks, n_per);
	run<0, true>("random, atomic load + CAS", tab, small, 12, blocks, n_per);
	run<1, true>("own segment (32 KB), atomic load + CAS", tab, big, 12, blocks, n_per);
	run<1, false>("own segment (32 KB), plain load + CAS", tab, big, 12, blocks, n_per);
	run<1, true>("own segment (8 KB), atomic load + CAS", tab, big, 10, blocks, n_per);
	run<1, true>("own segment (32 KB), 4 upserts per thread", tab, big, 12, blocks, 4);
	run<0, true>("random, 4 upserts per thread", tab, big, 12, blocks, 4);
	for (int t = 30; t >= 25; t -= 5) {
		unit<0>("plain 8-byte load only", tab, t, blocks, 4);
		unit<1>("non-returning atomicAdd only", tab, t, blocks, 4);
		unit<2>("returning atomicAdd only", tab, t, blocks, 4);
		unit<3>("atomicCAS only", tab, t, blocks, 4);
		unit<4>("plain load + non-returning atomicAdd", tab, t, blocks, 4);
	}
	return 0;
}
