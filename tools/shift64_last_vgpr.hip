// shift64_last_vgpr.hip -- minimal reproducer for the kernel-4 fault of DESIGN.md 3.4.
//
// What the ISA-level bisection of the failing build found (tools/k4_fault_isa_variants.py, tools/k4_fault_repro.md):
// the only wrong instruction instance was `v_lshlrev_b64 v[44:45], v47, 1` in a kernel that owns v0..v47 -- the
// 32-bit SHIFT AMOUNT sits in the LAST vector register of the wave's allocation.  Every wrong set bit of the
// failing build is `1 << (v0 & 63)`: the value the hardware substitutes for an out-of-range source register (v0).
// Reading: the operand fetch of the 64-bit shift treats src0 as a register pair v[47:48]; v48 is past the
// allocation; depending on timing the fetch is flagged out of range and v0 is used instead.
//
// probe<MODE>: the kernel owns exactly 48 registers (pinned by a clobber of v47).  v0 is set to 33 so that a
// substituted amount shows up as 1 << 33.
//   MODE 0: v_lshlrev_b64 v[44:45], v47, 1     amount in the last register           (the failing shape)
//   MODE 1: v_lshlrev_b64 v[44:45], v46, 1     amount in the last but one            (control)
//   MODE 2: v_lshlrev_b32 v44, v47, 1          32-bit shift, amount in the last one  (control)
//   MODE 3: v_lshrrev_b64 v[44:45], v47, v[42:43]   the right shift, amount in the last register
//   MODE 4: v_ashrrev_i64 v[44:45], v47, v[42:43]   the arithmetic one
//   MODE 5: MODE 0 with `s_nop 1` in front of the shift (the amount is written two instructions earlier otherwise)
//   MODE 6: v_mad_u64_u32 v[44:45], s[40:41], v47, v46, v[42:43]   32-bit factor in the last register (src0)
//   MODE 7: v_mad_u64_u32 v[44:45], s[40:41], v46, v47, v[42:43]   ... as src1
//   MODE 8: v_lshlrev_b64 v[52:53], v55, 1 in a kernel that owns 56 registers
//   MODE 9: v_lshlrev_b64 v[42:43], v44, 1 in a kernel whose highest register is v44 (45 used, 48 owned)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/shift64_last_vgpr tools/shift64_last_vgpr.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct report { unsigned long long wrong, substituted_v0; };

template <int MODE>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_num_vgpr(24))) void probe(report *rep, int iters) {
  extern __shared__ unsigned char tab[];
  for (int i = threadIdx.x; i < 12288; i += blockDim.x) tab[i] = (unsigned char)(i * 7);
  __syncthreads();
  unsigned s = (threadIdx.x * 7u + blockIdx.x * 13u) & 63u;
  unsigned long long wrong = 0, subst = 0;
  for (int it = 0; it < iters; it++) {
    s = (s * 5u + 3u + tab[(s * 191u + it) % 12288u]) & 63u; /* keeps the LDS busy like the real loop */
    unsigned lo, hi;
    const unsigned long long src = 0x8000000000000001ull * (it | 1u);
    unsigned long long want, with_v0;
    if (MODE == 6 || MODE == 7) {
      want = src + (unsigned long long)s * s; with_v0 = src + 33ull * s;
    } else if (MODE == 0 || MODE == 5 || MODE == 1 || MODE == 8 || MODE == 9) {
      want = 1ull << s; with_v0 = 1ull << 33;
    } else if (MODE == 2) {
      want = (unsigned long long)(1u << (s & 31u)); with_v0 = 1ull << 1;
    } else if (MODE == 3) {
      want = src >> s; with_v0 = src >> 33;
    } else {
      want = (unsigned long long)((long long)src >> s); with_v0 = (unsigned long long)((long long)src >> 33);
    }
#define COMMON_IN "v"(s), "v"((unsigned)src), "v"((unsigned)(src >> 32))
#define COMMON_CLOB "v0", "v42", "v43", "v44", "v45", "v46", "v47"
    if (MODE == 0)
      asm volatile("v_mov_b32 v0, 33\n\tv_mov_b32 v47, %2\n\tv_mov_b32 v46, %2\n\tv_lshlrev_b64 v[44:45], v47, 1\n\t"
                   "v_mov_b32 %0, v44\n\tv_mov_b32 %1, v45" : "=&v"(lo), "=&v"(hi) : COMMON_IN : COMMON_CLOB);
    if (MODE == 1)
      asm volatile("v_mov_b32 v0, 33\n\tv_mov_b32 v47, %2\n\tv_mov_b32 v46, %2\n\tv_lshlrev_b64 v[44:45], v46, 1\n\t"
                   "v_mov_b32 %0, v44\n\tv_mov_b32 %1, v45" : "=&v"(lo), "=&v"(hi) : COMMON_IN : COMMON_CLOB);
    if (MODE == 2)
      asm volatile("v_mov_b32 v0, 33\n\tv_mov_b32 v47, %2\n\tv_mov_b32 v46, %2\n\tv_lshlrev_b32 v44, v47, 1\n\t"
                   "v_mov_b32 %0, v44\n\tv_mov_b32 %1, 0" : "=&v"(lo), "=&v"(hi) : COMMON_IN : COMMON_CLOB);
    if (MODE == 3)
      asm volatile("v_mov_b32 v0, 33\n\tv_mov_b32 v42, %3\n\tv_mov_b32 v43, %4\n\tv_mov_b32 v47, %2\n\t"
                   "v_lshrrev_b64 v[44:45], v47, v[42:43]\n\t"
                   "v_mov_b32 %0, v44\n\tv_mov_b32 %1, v45" : "=&v"(lo), "=&v"(hi) : COMMON_IN : COMMON_CLOB);
    if (MODE == 4)
      asm volatile("v_mov_b32 v0, 33\n\tv_mov_b32 v42, %3\n\tv_mov_b32 v43, %4\n\tv_mov_b32 v47, %2\n\t"
                   "v_ashrrev_i64 v[44:45], v47, v[42:43]\n\t"
                   "v_mov_b32 %0, v44\n\tv_mov_b32 %1, v45" : "=&v"(lo), "=&v"(hi) : COMMON_IN : COMMON_CLOB);
    if (MODE == 5)
      asm volatile("v_mov_b32 v0, 33\n\tv_mov_b32 v47, %2\n\tv_mov_b32 v46, %2\n\ts_nop 1\n\tv_lshlrev_b64 v[44:45], v47, 1\n\t"
                   "v_mov_b32 %0, v44\n\tv_mov_b32 %1, v45" : "=&v"(lo), "=&v"(hi) : COMMON_IN : COMMON_CLOB);
    if (MODE == 6)
      asm volatile("v_mov_b32 v0, 33\n\tv_mov_b32 v42, %3\n\tv_mov_b32 v43, %4\n\tv_mov_b32 v47, %2\n\tv_mov_b32 v46, %2\n\t"
                   "v_mad_u64_u32 v[44:45], s[40:41], v47, v46, v[42:43]\n\t"
                   "v_mov_b32 %0, v44\n\tv_mov_b32 %1, v45" : "=&v"(lo), "=&v"(hi) : COMMON_IN : COMMON_CLOB, "s40", "s41");
    if (MODE == 7)
      asm volatile("v_mov_b32 v0, 33\n\tv_mov_b32 v42, %3\n\tv_mov_b32 v43, %4\n\tv_mov_b32 v47, %2\n\tv_mov_b32 v46, %2\n\t"
                   "v_mad_u64_u32 v[44:45], s[40:41], v46, v47, v[42:43]\n\t"
                   "v_mov_b32 %0, v44\n\tv_mov_b32 %1, v45" : "=&v"(lo), "=&v"(hi) : COMMON_IN : COMMON_CLOB, "s40", "s41");
    if (MODE == 8)
      asm volatile("v_mov_b32 v0, 33\n\tv_mov_b32 v55, %2\n\tv_lshlrev_b64 v[52:53], v55, 1\n\t"
                   "v_mov_b32 %0, v52\n\tv_mov_b32 %1, v53" : "=&v"(lo), "=&v"(hi) : COMMON_IN : "v0", "v52", "v53", "v55");
    if (MODE == 9)
      asm volatile("v_mov_b32 v0, 33\n\tv_mov_b32 v44, %2\n\tv_lshlrev_b64 v[42:43], v44, 1\n\t"
                   "v_mov_b32 %0, v42\n\tv_mov_b32 %1, v43" : "=&v"(lo), "=&v"(hi) : COMMON_IN : "v0", "v42", "v43", "v44");
    const unsigned long long got = ((unsigned long long)hi << 32) | lo;
    wrong += got != want;
    subst += got != want && got == with_v0;
  }
  if (wrong) { atomicAdd(&rep->wrong, wrong); atomicAdd(&rep->substituted_v0, subst); }
}

template <int MODE>
static void run(const char *what, report *d_rep, int cus, int block, int per_cu, int iters) {
  report h;
  CHECK(hipMemset(d_rep, 0, sizeof(report)));
  hipLaunchKernelGGL(probe<MODE>, dim3(cus * per_cu), dim3(block), 12288, 0, d_rep, iters);
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(&h, d_rep, sizeof(report), hipMemcpyDeviceToHost));
  printf("%-62s %2d waves/SIMD: %llu wrong of %llu, %llu of them = the result for amount v0\n", what,
         block / 256 * (per_cu > 2 ? 2 : per_cu), h.wrong, (unsigned long long)cus * per_cu * block * iters, h.substituted_v0);
}

int main() {
  report *d_rep;
  CHECK(hipMalloc(&d_rep, sizeof(report)));
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount, iters = 20000;
  for (int pass = 0; pass < 2; pass++) {
    const int block = pass ? 1024 : 256, per_cu = pass == 0 ? 1 : 2;
    run<0>("v_lshlrev_b64, amount in the LAST register of 48", d_rep, cus, block, per_cu, iters);
    run<5>("the same behind s_nop 1", d_rep, cus, block, per_cu, iters);
    run<1>("v_lshlrev_b64, amount in the last but one (control)", d_rep, cus, block, per_cu, iters);
    run<2>("v_lshlrev_b32, amount in the last register (control)", d_rep, cus, block, per_cu, iters);
    run<3>("v_lshrrev_b64, amount in the last register", d_rep, cus, block, per_cu, iters);
    run<4>("v_ashrrev_i64, amount in the last register", d_rep, cus, block, per_cu, iters);
    run<6>("v_mad_u64_u32, 32-bit src0 in the last register", d_rep, cus, block, per_cu, iters);
    run<7>("v_mad_u64_u32, 32-bit src1 in the last register", d_rep, cus, block, per_cu, iters);
    run<8>("v_lshlrev_b64, amount in v55 of a 56-register kernel", d_rep, cus, block, per_cu, iters);
    run<9>("v_lshlrev_b64, amount in v44, highest used; 48 owned", d_rep, cus, block, per_cu, iters);
  }
  return 0;
}
