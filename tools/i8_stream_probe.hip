// i8_stream_probe.hip — standalone diagnostic (not part of the product library): what a bare int8 MFMA stream sustains on this chip
// for the two gfx950 shapes, on random bytes (the chip lowers its clock under matrix load and the clock it holds depends on the
// shape, MI355X_MICROARCH.md "DVFS give-back" (7)).  Operands in registers, accumulators per wave as in the sliced GEMM tile.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/i8_stream_probe tools/i8_stream_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
typedef int i4v __attribute__((ext_vector_type(4)));
typedef int i16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ i4v rnd4(unsigned& s) {
  i4v r;
  for (int k = 0; k < 4; ++k) { s = s * 1664525u + 1013904223u; r[k] = (int)(s ^ (s >> 13)); }
  return r;
}

// NACC independent accumulators of 32x32 (16 regs each)
template <int NACC>
__global__ __launch_bounds__(256) void k_s32(int* out, int iters, int zero) {
  unsigned s = (threadIdx.x + 1) * 2654435761u + blockIdx.x;
  i4v a[4], b[4];
  for (int k = 0; k < 4; ++k) { a[k] = rnd4(s); b[k] = rnd4(s); if (zero) { a[k] = (i4v){0,0,0,0}; b[k] = a[k]; } }
  i16v acc[NACC];
  for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[n & 3], b[(n >> 2) & 3], acc[n], 0, 0, 0);
  }
  int t = 0;
  for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) t += acc[n][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}
// NACC independent accumulators of 16x16 (4 regs each)
template <int NACC>
__global__ __launch_bounds__(256) void k_s16(int* out, int iters, int zero) {
  unsigned s = (threadIdx.x + 1) * 2654435761u + blockIdx.x;
  i4v a[4], b[4];
  for (int k = 0; k < 4; ++k) { a[k] = rnd4(s); b[k] = rnd4(s); if (zero) { a[k] = (i4v){0,0,0,0}; b[k] = a[k]; } }
  i4v acc[NACC];
  for (int n = 0; n < NACC; ++n) acc[n] = (i4v){0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[n & 3], b[(n >> 2) & 3], acc[n], 0, 0, 0);
  }
  int t = 0;
  for (int n = 0; n < NACC; ++n) for (int r = 0; r < 4; ++r) t += acc[n][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

template <class F>
static double timeit(F launch, int reps) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  int* d; CK(hipMalloc(&d, sizeof(int) * cus * 8 * 256));
  const int iters = 4000;
  for (int zero = 0; zero < 2; ++zero)
    for (int wps = 1; wps <= 2; ++wps) {  // waves per SIMD
      const int blocks = cus * wps;       // 256 threads = 4 waves = one per SIMD
      {
        double ms = timeit([&]() { hipLaunchKernelGGL((k_s32<12>), dim3(blocks), dim3(256), 0, 0, d, iters, zero); }, 5);
        double ops = 2.0 * 32768.0 * 12 * iters * blocks * 4;
        printf("STREAM 32x32x32 i8  %s  %d wave/SIMD, 12 acc: %.3f ms  %.2f POP/s\n", zero ? "zeros " : "random", wps, ms, ops / ms * 1e-12);
      }
      {
        double ms = timeit([&]() { hipLaunchKernelGGL((k_s16<48>), dim3(blocks), dim3(256), 0, 0, d, iters, zero); }, 5);
        double ops = 2.0 * 16384.0 * 48 * iters * blocks * 4;
        printf("STREAM 16x16x64 i8  %s  %d wave/SIMD, 48 acc: %.3f ms  %.2f POP/s\n", zero ? "zeros " : "random", wps, ms, ops / ms * 1e-12);
      }
    }
  return 0;
}
