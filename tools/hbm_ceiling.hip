// hbm_ceiling.hip -- what a plain streaming kernel reaches on this GPU with the traffic shape of the
// propagation kernels: per node one 512-byte row of each of two input arrays is read and one row of
// each of two output arrays is written (queens-64: state row + forbidden-set row).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/hbm_ceiling tools/hbm_ceiling.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <typename T, int ROWS_PER_WAVE_ITER, int NT>
__global__ __launch_bounds__(512) void stream2nt(const T *__restrict__ a, const T *__restrict__ b, T *__restrict__ c,
                                                 T *__restrict__ d, long long rows, int row_elems) {
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long long waves = (long long)gridDim.x * (blockDim.x >> 6);
  for (long long r = wave * ROWS_PER_WAVE_ITER; r < rows; r += waves * ROWS_PER_WAVE_ITER) {
    T x[ROWS_PER_WAVE_ITER], y[ROWS_PER_WAVE_ITER];
#pragma unroll
    for (int k = 0; k < ROWS_PER_WAVE_ITER; k++) {
      const long long rr = r + k < rows ? r + k : rows - 1;
      x[k] = (NT & 1) ? __builtin_nontemporal_load(&a[rr * row_elems + lane]) : a[rr * row_elems + lane];
      y[k] = (NT & 1) ? __builtin_nontemporal_load(&b[rr * row_elems + lane]) : b[rr * row_elems + lane];
    }
#pragma unroll
    for (int k = 0; k < ROWS_PER_WAVE_ITER; k++) {
      if (r + k < rows) {
        if (NT & 2) {
          __builtin_nontemporal_store(x[k], &c[(r + k) * row_elems + lane]);
          __builtin_nontemporal_store(y[k], &d[(r + k) * row_elems + lane]);
        } else {
          c[(r + k) * row_elems + lane] = x[k];
          d[(r + k) * row_elems + lane] = y[k];
        }
      }
    }
  }
}

template <typename T, int ROWS_PER_WAVE_ITER>
__global__ __launch_bounds__(512) void stream2(const T *__restrict__ a, const T *__restrict__ b, T *__restrict__ c,
                                               T *__restrict__ d, long long rows, int row_elems) {
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long long waves = (long long)gridDim.x * (blockDim.x >> 6);
  for (long long r = wave * ROWS_PER_WAVE_ITER; r < rows; r += waves * ROWS_PER_WAVE_ITER) {
    T x[ROWS_PER_WAVE_ITER], y[ROWS_PER_WAVE_ITER];
#pragma unroll
    for (int k = 0; k < ROWS_PER_WAVE_ITER; k++) {
      const long long rr = r + k < rows ? r + k : rows - 1;
      x[k] = a[rr * row_elems + lane];
      y[k] = b[rr * row_elems + lane];
    }
#pragma unroll
    for (int k = 0; k < ROWS_PER_WAVE_ITER; k++) {
      if (r + k < rows) {
        c[(r + k) * row_elems + lane] = x[k];
        d[(r + k) * row_elems + lane] = y[k];
      }
    }
  }
}

template <typename F>
static float time_ms(F f, int reps) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 3; i++) f();
  CHECK(hipDeviceSynchronize());
  float best = 1e30f, sum = 0;
  for (int i = 0; i < reps; i++) {
    CHECK(hipEventRecord(e0));
    f();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    sum += ms; if (ms < best) best = ms;
  }
  printf("    avg %.4f ms  best %.4f ms", sum / reps, best);
  return sum / reps;
}

int main(int argc, char **argv) {
  const long long rows = argc > 1 ? atoll(argv[1]) : (1 << 18);
  const size_t row_bytes = 512;
  const size_t bytes = rows * row_bytes;
  void *a, *b, *c, *d;
  CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes)); CHECK(hipMalloc(&c, bytes)); CHECK(hipMalloc(&d, bytes));
  CHECK(hipMemset(a, 1, bytes)); CHECK(hipMemset(b, 2, bytes));
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const double total = 4.0 * bytes;
  printf("rows %lld, %zu B per row per array, traffic %.1f MB per launch, %d CUs\n", rows, row_bytes, total / 1e6, cus);
  for (int wg_per_cu = 1; wg_per_cu <= 8; wg_per_cu *= 2) {
    const int grid = cus * wg_per_cu;
    printf("persistent grid %5d x 512, 8 B/lane, 1 row/iter:", grid);
    float ms = time_ms([&] { hipLaunchKernelGGL((stream2<unsigned long long, 1>), dim3(grid), dim3(512), 0, 0, (const unsigned long long *)a, (const unsigned long long *)b, (unsigned long long *)c, (unsigned long long *)d, rows, 64); }, 20);
    printf("  -> %.0f GB/s\n", total / ms / 1e6);
    printf("persistent grid %5d x 512, 8 B/lane, 2 rows/iter:", grid);
    ms = time_ms([&] { hipLaunchKernelGGL((stream2<unsigned long long, 2>), dim3(grid), dim3(512), 0, 0, (const unsigned long long *)a, (const unsigned long long *)b, (unsigned long long *)c, (unsigned long long *)d, rows, 64); }, 20);
    printf("  -> %.0f GB/s\n", total / ms / 1e6);
    printf("persistent grid %5d x 512, 8 B/lane, 4 rows/iter:", grid);
    ms = time_ms([&] { hipLaunchKernelGGL((stream2<unsigned long long, 4>), dim3(grid), dim3(512), 0, 0, (const unsigned long long *)a, (const unsigned long long *)b, (unsigned long long *)c, (unsigned long long *)d, rows, 64); }, 20);
    printf("  -> %.0f GB/s\n", total / ms / 1e6);
  }
  {
    const int grid = (int)((rows + 7) / 8);
    printf("one row per wave, grid %d x 512, 8 B/lane:", grid);
    float ms = time_ms([&] { hipLaunchKernelGGL((stream2<unsigned long long, 1>), dim3(grid), dim3(512), 0, 0, (const unsigned long long *)a, (const unsigned long long *)b, (unsigned long long *)c, (unsigned long long *)d, rows, 64); }, 20);
    printf("  -> %.0f GB/s\n", total / ms / 1e6);
  }
  for (int nt = 1; nt <= 3; nt++) {
    const int grid = cus * 2;
    printf("persistent grid %5d x 512, 2 rows/iter, nontemporal %s%s:", grid, (nt & 1) ? "loads " : "", (nt & 2) ? "stores" : "");
    float ms = 0;
    if (nt == 1) ms = time_ms([&] { hipLaunchKernelGGL((stream2nt<unsigned long long, 2, 1>), dim3(grid), dim3(512), 0, 0, (const unsigned long long *)a, (const unsigned long long *)b, (unsigned long long *)c, (unsigned long long *)d, rows, 64); }, 20);
    if (nt == 2) ms = time_ms([&] { hipLaunchKernelGGL((stream2nt<unsigned long long, 2, 2>), dim3(grid), dim3(512), 0, 0, (const unsigned long long *)a, (const unsigned long long *)b, (unsigned long long *)c, (unsigned long long *)d, rows, 64); }, 20);
    if (nt == 3) ms = time_ms([&] { hipLaunchKernelGGL((stream2nt<unsigned long long, 2, 3>), dim3(grid), dim3(512), 0, 0, (const unsigned long long *)a, (const unsigned long long *)b, (unsigned long long *)c, (unsigned long long *)d, rows, 64); }, 20);
    printf("  -> %.0f GB/s\n", total / ms / 1e6);
  }
  {
    printf("hipMemcpyAsync D2D of two arrays:");
    float ms = time_ms([&] { (void)hipMemcpyAsync(c, a, bytes, hipMemcpyDeviceToDevice, 0); (void)hipMemcpyAsync(d, b, bytes, hipMemcpyDeviceToDevice, 0); }, 20);
    printf("  -> %.0f GB/s\n", total / ms / 1e6);
  }
  return 0;
}
