// valu_rate.hip -- issue cost of the VALU instructions K1 is made of, on gfx950: a wave runs N dependent-free copies of one instruction
// in a loop; cycles per instruction per wave from s_memtime (100 MHz on this part? no: s_memtime counts at a fixed clock, so the probe
// reports ns per wave-instruction from wall time of a grid that fills every SIMD with W waves).
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

#define KERNEL(name, body)                                                                                   \
	__global__ __launch_bounds__(256) void name(uint64_t *out, int iters, uint32_t s0)                        \
	{                                                                                                         \
		uint32_t a = threadIdx.x + s0, b = a * 3 + 1, c = b ^ 0x55, d = c + 7, e = d * 5, f = e ^ a, g = f + 11, h = g ^ b; \
		uint64_t A = ((uint64_t)a << 32) | b, B = ((uint64_t)c << 32) | d, C2 = ((uint64_t)e << 32) | f, D = ((uint64_t)g << 32) | h; \
		for (int i = 0; i < iters; ++i) { REP64(body) }                                                       \
		out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + e + f + g + h + A + B + C2 + D;                  \
	}

// four independent chains per body so that dependency latency does not limit issue
KERNEL(k_add_u32,   asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
KERNEL(k_and_b32,   asm volatile("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
KERNEL(k_alignbit,  asm volatile("v_alignbit_b32 %0, %0, %4, 7\n v_alignbit_b32 %1, %1, %4, 7\n v_alignbit_b32 %2, %2, %4, 7\n v_alignbit_b32 %3, %3, %4, 7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
KERNEL(k_lshl_add_u32, asm volatile("v_lshl_add_u32 %0, %0, 3, %4\n v_lshl_add_u32 %1, %1, 3, %4\n v_lshl_add_u32 %2, %2, 3, %4\n v_lshl_add_u32 %3, %3, 3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
KERNEL(k_mul_lo_u32, asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
KERNEL(k_bfrev,     asm volatile("v_bfrev_b32 %0, %0\n v_bfrev_b32 %1, %1\n v_bfrev_b32 %2, %2\n v_bfrev_b32 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
KERNEL(k_cndmask,   asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e) : "vcc");)
KERNEL(k_cndmask_sgpr, asm volatile("v_cndmask_b32_e64 %0, %0, %4, s[20:21]\n v_cndmask_b32_e64 %1, %1, %4, s[20:21]\n v_cndmask_b32_e64 %2, %2, %4, s[20:21]\n v_cndmask_b32_e64 %3, %3, %4, s[20:21]" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e) : "s20", "s21");)
KERNEL(k_cmp_cndmask, asm volatile("v_cmp_lt_u32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %4, vcc\n v_cmp_lt_u32 vcc, %2, %4\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e) : "vcc");)
KERNEL(k_cmp_only, asm volatile("v_cmp_lt_u32 vcc, %0, %4\n v_cmp_lt_u32 vcc, %1, %4\n v_cmp_lt_u32 vcc, %2, %4\n v_cmp_lt_u32 vcc, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e) : "vcc");)
KERNEL(k_cndmask_dst, asm volatile("v_cndmask_b32 %0, %1, %4, vcc\n v_cndmask_b32 %1, %2, %4, vcc\n v_cndmask_b32 %2, %3, %4, vcc\n v_cndmask_b32 %3, %0, %4, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e) : "vcc");)
KERNEL(k_bfi, asm volatile("v_bfi_b32 %0, %4, %0, %1\n v_bfi_b32 %1, %4, %1, %2\n v_bfi_b32 %2, %4, %2, %3\n v_bfi_b32 %3, %4, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
KERNEL(k_addco_addc, asm volatile("v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_add_co_u32 %2, vcc, %2, %4\n v_addc_co_u32 %3, vcc, %3, %4, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e) : "vcc");)
KERNEL(k_lshl_add_u64, asm volatile("v_lshl_add_u64 %0, %0, 3, %4\n v_lshl_add_u64 %1, %1, 3, %4\n v_lshl_add_u64 %2, %2, 3, %4\n v_lshl_add_u64 %3, %3, 3, %4" : "+v"(A), "+v"(B), "+v"(C2), "+v"(D) : "v"(A));)
KERNEL(k_mad_u64_u32, asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3" : "+v"(A), "+v"(B), "+v"(C2), "+v"(D) : "v"(e), "v"(f) : "vcc");)
KERNEL(k_lshrrev_b64, asm volatile("v_lshrrev_b64 %0, 3, %0\n v_lshrrev_b64 %1, 3, %1\n v_lshrrev_b64 %2, 3, %2\n v_lshrrev_b64 %3, 3, %3" : "+v"(A), "+v"(B), "+v"(C2), "+v"(D));)
KERNEL(k_lshlrev_b64, asm volatile("v_lshlrev_b64 %0, 3, %0\n v_lshlrev_b64 %1, 3, %1\n v_lshlrev_b64 %2, 3, %2\n v_lshlrev_b64 %3, 3, %3" : "+v"(A), "+v"(B), "+v"(C2), "+v"(D));)
KERNEL(k_xor3,      asm volatile("v_bitop3_b32 %0, %0, %4, %1 bitop3:0x96\n v_bitop3_b32 %1, %1, %4, %2 bitop3:0x96\n v_bitop3_b32 %2, %2, %4, %3 bitop3:0x96\n v_bitop3_b32 %3, %3, %4, %0 bitop3:0x96" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
KERNEL(k_and_or,    asm volatile("v_and_or_b32 %0, %0, %4, %1\n v_and_or_b32 %1, %1, %4, %2\n v_and_or_b32 %2, %2, %4, %3\n v_and_or_b32 %3, %3, %4, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
KERNEL(k_mov_b64,   asm volatile("v_mov_b64 %0, %1\n v_mov_b64 %1, %2\n v_mov_b64 %2, %3\n v_mov_b64 %3, %0" : "+v"(A), "+v"(B), "+v"(C2), "+v"(D));)
KERNEL(k_pk_add_u16, asm volatile("v_pk_add_u16 %0, %0, %4\n v_pk_add_u16 %1, %1, %4\n v_pk_add_u16 %2, %2, %4\n v_pk_add_u16 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
KERNEL(k_mul_hi_u32, asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
KERNEL(k_mad_u32_u24, asm volatile("v_mad_u32_u24 %0, %0, %4, %1\n v_mad_u32_u24 %1, %1, %4, %2\n v_mad_u32_u24 %2, %2, %4, %3\n v_mad_u32_u24 %3, %3, %4, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
KERNEL(k_bfe_u32,   asm volatile("v_bfe_u32 %0, %0, 3, 9\n v_bfe_u32 %1, %1, 3, 9\n v_bfe_u32 %2, %2, 3, 9\n v_bfe_u32 %3, %3, 3, 9" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
KERNEL(k_salu_mix,  asm volatile("v_add_u32 %0, %0, %4\n s_add_u32 s20, s20, 1\n v_add_u32 %1, %1, %4\n s_add_u32 s21, s21, 1\n v_add_u32 %2, %2, %4\n s_add_u32 s22, s22, 1\n v_add_u32 %3, %3, %4\n s_add_u32 s23, s23, 1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e) : "s20", "s21", "s22", "s23", "scc");)

typedef void (*kfn)(uint64_t *, int, uint32_t);
struct ent { const char *name; kfn fn; int per_body; };

int main()
{
	ent tab[] = {{"v_add_u32", k_add_u32, 4}, {"v_and_b32", k_and_b32, 4}, {"v_alignbit_b32", k_alignbit, 4}, {"v_lshl_add_u32", k_lshl_add_u32, 4},
	             {"v_mul_lo_u32", k_mul_lo_u32, 4}, {"v_mul_hi_u32", k_mul_hi_u32, 4}, {"v_mad_u32_u24", k_mad_u32_u24, 4}, {"v_bfrev_b32", k_bfrev, 4}, {"v_bfe_u32", k_bfe_u32, 4},
	             {"v_cndmask_b32 (vcc, uninitialised)", k_cndmask, 4}, {"v_cndmask_b32_e64 (sgpr pair)", k_cndmask_sgpr, 4}, {"v_cmp + v_cndmask pairs (count both)", k_cmp_cndmask, 4}, {"v_cmp_lt_u32 -> vcc", k_cmp_only, 4}, {"v_cndmask_b32 dst != src", k_cndmask_dst, 4}, {"v_bfi_b32", k_bfi, 4}, {"v_bitop3_b32 (xor3)", k_xor3, 4}, {"v_and_or_b32", k_and_or, 4}, {"v_pk_add_u16", k_pk_add_u16, 4},
	             {"v_add_co+v_addc (pair=2)", k_addco_addc, 4}, {"v_lshl_add_u64", k_lshl_add_u64, 4}, {"v_mad_u64_u32", k_mad_u64_u32, 4},
	             {"v_lshrrev_b64", k_lshrrev_b64, 4}, {"v_lshlrev_b64", k_lshlrev_b64, 4}, {"v_mov_b64", k_mov_b64, 4}, {"v_add_u32 + s_add_u32 interleaved (VALU count)", k_salu_mix, 4}};
	hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
	const int cus = p.multiProcessorCount;
	const double clk = p.clockRate * 1e3; // Hz
	printf("# %s, %d CUs, %.0f MHz; waves per SIMD given per column\n", p.gcnArchName, cus, clk / 1e6);
	uint64_t *out; hipMalloc(&out, (size_t)cus * 64 * 256 * 8);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	const int iters = 2000;
	fflush(stdout);
	printf("%-50s %10s %10s %10s   (cycles per wave-instruction per SIMD; 1.0 = one wave64 instruction issued per cycle... full rate on a SIMD16 would be 4.0)\n", "instruction", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD");
	for (auto &t : tab) {
		printf("%-50s", t.name);
		for (int wps : {1, 2, 4}) {
			const int blocks = cus * wps; // 256 threads = 4 waves = one per SIMD of a CU
			hipLaunchKernelGGL(t.fn, dim3(blocks), dim3(256), 0, 0, out, 10, 1u);
			hipDeviceSynchronize();
			hipEventRecord(e0);
			hipLaunchKernelGGL(t.fn, dim3(blocks), dim3(256), 0, 0, out, iters, 1u);
			hipEventRecord(e1); hipEventSynchronize(e1);
			float ms; hipEventElapsedTime(&ms, e0, e1);
			const double inst_per_simd = (double)iters * 64 * t.per_body * wps;
			printf(" %10.3f", ms * 1e-3 * clk / inst_per_simd);
		}
		printf("\n"); fflush(stdout);
	}
	return 0;
}
