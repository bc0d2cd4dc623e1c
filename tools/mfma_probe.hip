// mfma_probe.hip — standalone diagnostic (not part of the product library).
//   1. prints the lane -> (row, col) map of v_mfma_f64_16x16x4_f64 operands/results
//   2. measures fp64 MFMA and fp64 VALU FMA throughput (roofline denominators, measured on the box)
//   3. measures HBM stream-read bandwidth
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/mfma_probe tools/mfma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

// D = A(16x4) * B(4x16); A given row-major [16][4], B row-major [4][16]; assumed operand map:
// lane l holds A[l&15][l>>4], B[l>>4][l&15]; result reg r of lane l is written to out[l*4+r]
__global__ void k_layout(const double* A, const double* B, double* out) {
  int l = threadIdx.x;
  double a = A[(l & 15) * 4 + (l >> 4)];
  double b = B[(l >> 4) * 16 + (l & 15)];
  d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; r++) out[l * 4 + r] = c[r];
}

__global__ void k_mfma_rate(double* out, int iters) {
  double a = threadIdx.x * 1e-3 + 1.0, b = 1.0 - threadIdx.x * 1e-3;
  d4 c0 = {0,0,0,0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
  for (int i = 0; i < iters; i++) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c4, 0, 0, 0);
    c5 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c5, 0, 0, 0);
    c6 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c6, 0, 0, 0);
    c7 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c7, 0, 0, 0);
  }
  d4 s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

__global__ void k_valu_rate(double* out, int iters) {
  double x = threadIdx.x * 1e-9, m = 1.0000001, a = 1e-9;
  double c0 = x, c1 = x + 1, c2 = x + 2, c3 = x + 3, c4 = x + 4, c5 = x + 5, c6 = x + 6, c7 = x + 7;
  for (int i = 0; i < iters; i++) {
    c0 = fma(c0, m, a); c1 = fma(c1, m, a); c2 = fma(c2, m, a); c3 = fma(c3, m, a);
    c4 = fma(c4, m, a); c5 = fma(c5, m, a); c6 = fma(c6, m, a); c7 = fma(c7, m, a);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
}

// both pipes at once: even waves MFMA, odd waves VALU (do the fp64 pipes overlap?)
__global__ void k_both_rate(double* out, int iters) {
  int wave = threadIdx.x >> 6;
  double r;
  if (wave & 1) {
    double x = threadIdx.x * 1e-9, m = 1.0000001, a = 1e-9;
    double c0 = x, c1 = x + 1, c2 = x + 2, c3 = x + 3, c4 = x + 4, c5 = x + 5, c6 = x + 6, c7 = x + 7;
    for (int i = 0; i < iters * 16; i++) {  // 16*8 FMA = 128 VALU instr = same flops as 8 MFMA
      c0 = fma(c0, m, a); c1 = fma(c1, m, a); c2 = fma(c2, m, a); c3 = fma(c3, m, a);
      c4 = fma(c4, m, a); c5 = fma(c5, m, a); c6 = fma(c6, m, a); c7 = fma(c7, m, a);
    }
    r = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
  } else {
    double a = threadIdx.x * 1e-3 + 1.0, b = 1.0 - threadIdx.x * 1e-3;
    d4 c0 = {0,0,0,0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
    for (int i = 0; i < iters; i++) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
      c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c4, 0, 0, 0);
      c5 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c5, 0, 0, 0);
      c6 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c6, 0, 0, 0);
      c7 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c7, 0, 0, 0);
    }
    d4 s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    r = s[0] + s[1] + s[2] + s[3];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

__global__ void k_stream(const double2* __restrict__ in, double* out, size_t n2) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  double s = 0;
  for (; i < n2; i += stride) { double2 v = in[i]; s += v.x + v.y; }
  if (s == 123.456) out[0] = s;
}

static float time_kernel(void (*launch)(void*), void* arg, int reps) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch(arg); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; i++) launch(arg);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s  CUs=%d  clock=%d MHz  arch=%s\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1000, prop.gcnArchName);
  // ---- 1. layout -------------------------------------------------------------------------
  std::vector<double> A(64, 0), B(64, 0), out(256);
  double *dA, *dB, *dO; CK(hipMalloc(&dA, 512)); CK(hipMalloc(&dB, 512)); CK(hipMalloc(&dO, 2048));
  // A[i][k] = i+1 for k==0 ; B[k][j] = (j+1)*100 for k==0  => D[i][j] = (i+1)*(j+1)*100, asymmetric in scale
  for (int i = 0; i < 16; i++) A[i * 4 + 0] = i + 1;
  for (int j = 0; j < 16; j++) B[0 * 16 + j] = (j + 1) * 100.0;
  CK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dO); CK(hipDeviceSynchronize());
  CK(hipMemcpy(out.data(), dO, 2048, hipMemcpyDeviceToHost));
  int ok_f64map = 1, ok_f32map = 1;
  for (int l = 0; l < 64; l++)
    for (int r = 0; r < 4; r++) {
      double v = out[l * 4 + r];
      int i = (int)(v / 100.0 + 0.5);  // = (row+1)*(col+1)
      int col = l & 15;
      int row_f64 = (l >> 4) + 4 * r, row_f32 = 4 * (l >> 4) + r;
      if (i != (row_f64 + 1) * (col + 1)) ok_f64map = 0;
      if (i != (row_f32 + 1) * (col + 1)) ok_f32map = 0;
    }
  printf("LAYOUT f64map(row=(l>>4)+4r,col=l&15)=%d  f32map(row=4(l>>4)+r)=%d\n", ok_f64map, ok_f32map);
  printf("LAYOUT lane0: %g %g %g %g | lane16: %g %g %g %g | lane1: %g %g %g %g\n", out[0], out[1], out[2], out[3], out[64], out[65], out[66], out[67], out[4], out[5], out[6], out[7]);
  // k-dimension check: A[i][k]=1 only for k=2,i=3 ; B[k][j]=1 only for k=2,j=5 -> D[3][5]=1
  std::fill(A.begin(), A.end(), 0); std::fill(B.begin(), B.end(), 0);
  A[3 * 4 + 2] = 1; B[2 * 16 + 5] = 7;
  CK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dO); CK(hipDeviceSynchronize());
  CK(hipMemcpy(out.data(), dO, 2048, hipMemcpyDeviceToHost));
  for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) if (out[l * 4 + r] != 0) printf("LAYOUT kcheck: nonzero %g at lane %d reg %d (expect lane 5+16*3=53 reg 0 under f64map)\n", out[l * 4 + r], l, r);
  // ---- 2. rates ------------------------------------------------------------------------------
  const int blocks = prop.multiProcessorCount * 8, threads = 256, iters = 2000;
  double* dR; CK(hipMalloc(&dR, sizeof(double) * blocks * threads));
  struct Arg { double* p; int it; int blocks; int threads; } arg = {dR, iters, blocks, threads};
  float ms = time_kernel([](void* a) { Arg* g = (Arg*)a; hipLaunchKernelGGL(k_mfma_rate, dim3(g->blocks), dim3(g->threads), 0, 0, g->p, g->it); }, &arg, 5);
  double waves = (double)blocks * threads / 64;
  double flops = waves * iters * 8.0 * 2048.0;
  printf("RATE mfma_f64_16x16x4: %.3f ms  %.2f TFLOP/s  (%.1f cycles/MFMA/SIMD at %d MHz)\n", ms, flops / ms * 1e-9, (ms * 1e-3 * prop.clockRate * 1e3) / (waves * iters * 8.0 / (prop.multiProcessorCount * 4)), prop.clockRate / 1000);
  ms = time_kernel([](void* a) { Arg* g = (Arg*)a; hipLaunchKernelGGL(k_valu_rate, dim3(g->blocks), dim3(g->threads), 0, 0, g->p, g->it * 16); }, &arg, 5);
  flops = waves * iters * 16.0 * 8.0 * 128.0;
  printf("RATE v_fma_f64: %.3f ms  %.2f TFLOP/s\n", ms, flops / ms * 1e-9);
  ms = time_kernel([](void* a) { Arg* g = (Arg*)a; hipLaunchKernelGGL(k_both_rate, dim3(g->blocks), dim3(g->threads), 0, 0, g->p, g->it); }, &arg, 5);
  flops = waves * iters * 8.0 * 2048.0;  // half the waves do MFMA, half VALU, same flops each
  printf("RATE mixed (half waves MFMA, half VALU): %.3f ms  %.2f TFLOP/s combined\n", ms, flops / ms * 1e-9);
  // ---- 3. HBM stream ---------------------------------------------------------------------------
  size_t bytes = (size_t)4 << 30;
  double2* dS; CK(hipMalloc(&dS, bytes)); CK(hipMemset(dS, 0, bytes));
  struct SArg { double2* p; double* o; size_t n; } sarg = {dS, dR, bytes / 16};
  ms = time_kernel([](void* a) { SArg* g = (SArg*)a; hipLaunchKernelGGL(k_stream, dim3(256 * 16), dim3(256), 0, 0, g->p, g->o, g->n); }, &sarg, 5);
  printf("RATE hbm stream read (4 GiB): %.3f ms  %.2f TB/s\n", ms, bytes / ms * 1e-9);
  return 0;
}
