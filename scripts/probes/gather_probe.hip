// What can random 64-byte gathers into a 16 GiB array reach on MI355X?  The ceiling beside k_query's rate (bfcg_kernels.hip: one 64-byte bloom block per
// query, bbf.c:47-63; config c5's trim pass: 22 G queries/s = 1.4 TB/s = 0.18 of the HBM peak SURVEY 8d prices a bloom query at).  VERDICT r4 item 5(iii).
//   coop4:  four adjacent lanes fetch one block's four 16-byte quarters with one instruction (k_query's access), U gathers in flight per lane
//   lane16: every lane fetches 16 bytes of its OWN random block (the request-rate ceiling: 4x the requests for the same bytes)
//   coop8:  eight lanes fetch a 128-byte line (what a query would cost if a block were a line)
// Build: hipcc --offload-arch=gfx950 -O3 -o build/gather_probe scripts/probes/gather_probe.hip ; run: build/gather_probe [log2 bytes, default 34]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }

// G lanes share one gather of G x 16 bytes; every lane keeps U gathers in flight; n_per rounds
template <int G, int U>
__global__ __launch_bounds__(256) void k_gather(const uint4 *__restrict__ buf, int log2_units /* 16-byte units */, int n_per, uint64_t seed, unsigned int *sink)
{
	const uint64_t gid = blockIdx.x * 256ull + threadIdx.x, grp = gid / G, member = gid % G;
	const uint64_t n_obj = (1ull << log2_units) / G; // objects of G x 16 bytes
	uint32_t acc = 0;
	for (int i = 0; i < n_per; ++i) {
		uint4 v[U];
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const uint64_t obj = mix((grp * (uint64_t)n_per + i) * U + u + seed) & (n_obj - 1); // (a power of two: a 64-bit division here would be the probe's own bottleneck)
			v[u] = buf[obj * G + member];
		}
#pragma unroll
		for (int u = 0; u < U; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
	}
	if (acc == 0x12345678u) *sink = acc;
}

template <int G, int U>
static void run(const char *name, const uint4 *buf, int log2_units, unsigned int *sink)
{
	const int blocks = 256 * 64, n_per = 64 / U;
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	hipLaunchKernelGGL((k_gather<G, U>), dim3(blocks), dim3(256), 0, 0, buf, log2_units, n_per, 1, sink);
	CK(hipDeviceSynchronize());
	float best = 1e30f;
	for (int rep = 0; rep < 3; ++rep) {
		CK(hipEventRecord(e0, 0));
		hipLaunchKernelGGL((k_gather<G, U>), dim3(blocks), dim3(256), 0, 0, buf, log2_units, n_per, 1000 + rep, sink);
		CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
		float ms; CK(hipEventElapsedTime(&ms, e0, e1));
		if (ms < best) best = ms;
	}
	const double n = (double)blocks * 256 / G * n_per * U;
	printf("%-44s %3d B per gather, %d in flight per lane: %8.3f ms for %6.0f M gathers = %6.1f G gathers/s = %5.2f TB/s\n", name, G * 16, U, best, n / 1e6, n / best / 1e6, n * G * 16 / best / 1e9);
}

int main(int argc, char **argv)
{
	const int log2_bytes = argc > 1 ? atoi(argv[1]) : 34;
	uint4 *buf; unsigned int *sink;
	CK(hipMalloc(&buf, 1ull << log2_bytes)); CK(hipMemset(buf, 0, 1ull << log2_bytes)); CK(hipMalloc(&sink, 4));
	printf("random gathers into %.0f GiB\n", (double)(1ull << log2_bytes) / (1ull << 30));
	const int lu = log2_bytes - 4;
	run<4, 1>("coop4 (k_query's access)", buf, lu, sink);
	run<4, 2>("coop4", buf, lu, sink);
	run<4, 4>("coop4", buf, lu, sink);
	run<4, 8>("coop4", buf, lu, sink);
	run<4, 16>("coop4", buf, lu, sink);
	run<1, 4>("lane16 (one 16-byte piece per lane)", buf, lu, sink);
	run<1, 16>("lane16", buf, lu, sink);
	run<8, 4>("coop8 (128-byte lines)", buf, lu, sink);
	run<8, 8>("coop8", buf, lu, sink);
	run<2, 8>("coop2 (32-byte sectors)", buf, lu, sink);
	return 0;
}
