// shift64_hazard.hip -- is a variable 64-bit vector shift (v_lshlrev_b64 with the shift amount in a VGPR)
// safe on this GPU when (A) its shift-amount register is overwritten right after it, (B) its result is
// consumed right after it, with several waves sharing a SIMD?  Counts wrong results.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/shift64_hazard tools/shift64_hazard.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void probe(unsigned long long *bad, int iters) {
  unsigned s = (threadIdx.x * 7u + blockIdx.x * 13u) & 63u;
  unsigned long long wrong = 0;
  for (int it = 0; it < iters; it++) {
    unsigned amt = s, lo, hi;
    const unsigned want_lo = s < 32u ? 1u << s : 0u, want_hi = s >= 32u ? 1u << (s - 32u) : 0u;
    unsigned long long x;
    if (MODE == 0) { /* A: overwrite the shift amount one, two, three instructions later */
      asm volatile("v_lshlrev_b64 %0, %1, 1\n\tv_mov_b32 %1, 0" : "=&v"(x), "+v"(amt));
    } else if (MODE == 1) {
      asm volatile("v_lshlrev_b64 %0, %1, 1\n\tv_add_u32 %1, 17, %1\n\tv_mov_b32 %1, 0" : "=&v"(x), "+v"(amt));
    } else if (MODE == 2) { /* B: the compiler's own code, result consumed at once */
      x = 1ull << amt;
    } else if (MODE == 3) { /* C: the sequence of the failing loop: shift, compare on the amount, select on the high half, amount overwritten */
      unsigned sel, xl, xh;
      asm volatile("v_lshlrev_b64 v[40:41], %0, 1\n\tv_cmp_gt_u32 vcc, 64, %0\n\ts_add_i32 s0, s0, 0\n\tv_add_u32 %1, 0xc0, %0\n\t"
                   "v_cndmask_b32 %0, 0, v41, vcc\n\tv_mov_b32 %2, v40\n\tv_mov_b32 %3, v41"
                   : "+v"(amt), "=&v"(sel), "=&v"(xl), "=&v"(xh) : : "vcc", "s0", "v40", "v41");
      (void)sel;
      x = ((unsigned long long)xh << 32) | xl;
    }
    if (MODE == 4) { /* E: a 32-bit VALU op reads v60 / v61, the 64-bit shift right behind it overwrites v[60:61] */
      const unsigned a = s * 2654435761u + 12345u, b = ~a * 40503u, y = s * 97u + 1u;
      unsigned o1, o2, rl, rh;
      asm volatile("v_mov_b32 v60, %4\n\tv_mov_b32 v61, %5\n\ts_nop 4\n\t"
                   "v_or_b32 %0, v61, %6\n\tv_or_b32 %1, v60, %6\n\tv_lshlrev_b64 v[60:61], %7, 1\n\t"
                   "s_nop 7\n\tv_mov_b32 %2, v60\n\tv_mov_b32 %3, v61"
                   : "=&v"(o1), "=&v"(o2), "=&v"(rl), "=&v"(rh) : "v"(a), "v"(b), "v"(y), "v"(amt) : "v60", "v61");
      wrong += (o1 != (b | y)) | (o2 != (a | y)) | (rl != want_lo) | (rh != want_hi);
      s = (s * 5u + 3u) & 63u;
      continue;
    }
    lo = (unsigned)x; hi = (unsigned)(x >> 32);
    wrong += (lo != want_lo) | (hi != want_hi);
    s = (s * 5u + 3u + amt) & 63u;
  }
  if (wrong) atomicAdd(bad, wrong);
}

int main() {
  unsigned long long *d_bad, h_bad;
  CHECK(hipMalloc(&d_bad, 8));
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int iters = 20000;
  for (int wgs_per_cu = 1; wgs_per_cu <= 8; wgs_per_cu *= 8) {
    for (int mode = 0; mode < 5; mode++) {
      CHECK(hipMemset(d_bad, 0, 8));
      const int grid = prop.multiProcessorCount * wgs_per_cu;
      switch (mode) {
      case 0: hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(256), 0, 0, d_bad, iters); break;
      case 1: hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(256), 0, 0, d_bad, iters); break;
      case 2: hipLaunchKernelGGL(probe<2>, dim3(grid), dim3(256), 0, 0, d_bad, iters); break;
      case 3: hipLaunchKernelGGL(probe<3>, dim3(grid), dim3(256), 0, 0, d_bad, iters); break;
      default: hipLaunchKernelGGL(probe<4>, dim3(grid), dim3(256), 0, 0, d_bad, iters); break;
      }
      CHECK(hipDeviceSynchronize());
      CHECK(hipMemcpy(&h_bad, d_bad, 8, hipMemcpyDeviceToHost));
      printf("waves/SIMD %d mode %d: %llu wrong of %llu\n", wgs_per_cu, mode, h_bad, (unsigned long long)grid * 256 * iters);
    }
  }
  return 0;
}
