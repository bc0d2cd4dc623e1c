// i8_gemm_probe.hip — standalone diagnostic (not part of the product library).
// Sliced-integer GEMM used by the int8 metric assembly experiment: C[c][p] = sum_k V[c][k] Z[p][k] where V and Z are given as
// S signed-byte slices each (most significant first); slice products i+j < S are accumulated exactly in int32 on
// v_mfma_i32_32x32x32_i8, one accumulator set per weight g = i+j, and combined in fp64 at the end.
// Checks the kernel against a CPU loop on a small case with asymmetric data, then times it at the config-3 shape
// (8192 chains x 2080 column pairs x 10000 data rows).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++20 -mllvm -amdgpu-mfma-vgpr-form=1 -o tools/i8_gemm_probe tools/i8_gemm_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

#include "../riemannhamiltonianmontecarlo_amd/csrc/metric_i8.hip.h"

template <int S, int WN, int TN, int PIN = 1>
static double run_case(int nC, int NP, int K, bool check, int reps) {
  constexpr int BM = 128, BN = 32 * TN * WN;
  const int nCp = (nC + BM - 1) / BM * BM, NPp = (NP + BN - 1) / BN * BN, nks = (K + 31) / 32;
  const size_t szV = (size_t)S * nks * nCp * 32, szZ = (size_t)S * nks * NPp * 32;
  std::vector<int8_t> hV(szV), hZ(szZ);
  uint32_t st = 12345u;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return (int8_t)(st >> 24); };
  for (auto& x : hV) x = rnd();
  for (auto& x : hZ) x = rnd();
  int8_t *dV, *dZ; double* dC;
  CK(hipMalloc(&dV, szV)); CK(hipMalloc(&dZ, szZ)); CK(hipMalloc(&dC, sizeof(double) * (size_t)nCp * NPp));
  CK(hipMemcpy(dV, hV.data(), szV, hipMemcpyHostToDevice)); CK(hipMemcpy(dZ, hZ.data(), szZ, hipMemcpyHostToDevice));
  const int nCB = nCp / BM, nPB = NPp / BN;
  const int grid = (nCB + 7) / 8 * 8 * nPB;
  constexpr int lds = i8_lds_bytes<S, WN, TN>();
  CK(hipFuncSetAttribute((const void*)k_gemm_i8_probe<S, WN, TN, PIN>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  auto launch = [&]() { hipLaunchKernelGGL((k_gemm_i8_probe<S, WN, TN, PIN>), dim3(grid), dim3(128 * WN), lds, 0, dV, dZ, nCp, NPp, nks, nC, NP, dC); };
  launch(); CK(hipDeviceSynchronize());
  double maxerr = 0;
  if (check) {
    std::vector<double> hC((size_t)nCp * NPp);
    CK(hipMemcpy(hC.data(), dC, sizeof(double) * hC.size(), hipMemcpyDeviceToHost));
    for (int c = 0; c < nC; c += 7)
      for (int p = 0; p < NP; p += 5) {
        double ref = 0;
        for (int g = S - 1; g >= 0; --g) {
          long long acc = 0;
          for (int i = 0; i <= g; ++i) {
            const int j = g - i;
            for (int ks = 0; ks < nks; ++ks)
              for (int k = 0; k < 32; ++k)
                acc += (long long)hV[(((size_t)i * nks + ks) * nCp + c) * 32 + k] * hZ[(((size_t)j * nks + ks) * NPp + p) * 32 + k];
          }
          ref += (double)acc * ldexp(1.0, -8 * g);
        }
        maxerr = fmax(maxerr, fabs(ref - hC[(size_t)c * NPp + p]));
      }
    printf("check S=%d WN=%d TN=%d nC=%d NP=%d K=%d: max abs err %.3g\n", S, WN, TN, nC, NP, K, maxerr);
  }
  double ms = 0;
  if (reps > 0) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) launch();
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float f; CK(hipEventElapsedTime(&f, e0, e1)); ms = f / reps;
    const double P = S * (S + 1) / 2.0;
    const double ops = 2.0 * P * nC * (double)NP * K;  // unpadded problem
    printf("time S=%d WN=%d TN=%d PIN=%d nC=%d NP=%d(+pad %d) K=%d: %.3f ms  %.2f POPS int8 (%d blocks)\n", S, WN, TN, PIN, nC, NP, NPp, K, ms, ops / ms * 1e-12, grid);
  }
  CK(hipFree(dV)); CK(hipFree(dZ)); CK(hipFree(dC));
  return ms;
}

int main(int argc, char** argv) {
  if (argc > 1) {  // single timed case for counter collection
    const int S = atoi(argv[1]);
    if (S == 1) {  // small-batch shapes: one chain block, full k range and one 21-stage piece of it
      run_case<6, 4, 1, 1>(128, 2080, 10000, false, 10);
      run_case<6, 4, 1, 1>(128, 2080, 672, false, 10);
      run_case<6, 4, 1, 1>(1920, 2080, 672, false, 10);
      return 0;
    }
    if (S == 60) {  // the library configuration alone (built with -DI8_ABLATE=n for the ablations: see metric_i8.hip.h)
      run_case<6, 4, 1, 1>(8192, 2080, 10000, false, 8);
      run_case<5, 4, 1, 1>(8192, 2080, 10000, false, 8);
      return 0;
    }
    if (S == 61) {  // without the pinned issue order (sched_barrier)
      run_case<6, 4, 1, 0>(8192, 2080, 10000, false, 8);
      run_case<6, 4, 1, 1>(300, 300, 1000, true, 0);
      return 0;
    }
    if (S == 70) {  // where the time of a tile goes: whole rounds vs a ragged last one, fixed cost per tile (k sweep at the leverage shape),
                    // and the 128 x 32 two-wave tile that would take the 32 left-over pairs of the assembly (with and without a k split)
      run_case<6, 4, 1, 1>(8192, 2080, 10000, false, 8);
      run_case<6, 4, 1, 1>(8192, 2048, 10000, false, 8);
      run_case<6, 4, 1, 1>(8192, 10000, 2080, false, 8);
      run_case<6, 4, 1, 1>(8192, 10000, 4160, false, 4);
      run_case<6, 4, 1, 1>(8192, 10000, 1040, false, 8);
      run_case<6, 1, 1, 0>(300, 32, 1000, true, 0);
      run_case<6, 1, 1, 0>(8192, 32, 10000, false, 8);
      run_case<6, 1, 1, 0>(32768, 32, 2500, false, 8);
      run_case<6, 1, 1, 0>(65536, 32, 1250, false, 8);
      return 0;
    }
    if (S == 22) {  // 4 waves, 64x64 wave tile, one wave per SIMD (512 registers): a third fewer LDS fragment reads per MFMA
      run_case<6, 2, 2, 1>(300, 300, 1000, true, 0);
      run_case<6, 2, 2, 1>(8192, 2080, 10000, false, 5);
      run_case<6, 2, 2, 0>(8192, 2080, 10000, false, 5);
      run_case<6, 4, 1, 1>(8192, 2080, 10000, false, 5);
      run_case<5, 2, 2, 1>(8192, 2080, 10000, false, 5);
      run_case<5, 4, 1, 1>(8192, 2080, 10000, false, 5);
      return 0;
    }
    if (S == 44) {  // 4 slices: the 64 x 64 wave tile (4 waves, 256 accumulators: build WITHOUT -amdgpu-mfma-vgpr-form) against the library's 64 x 32
      run_case<4, 2, 2, 1>(300, 300, 1000, true, 0);
      run_case<4, 2, 2, 1>(8192, 2048, 10000, false, 8);
      run_case<4, 4, 1, 1>(8192, 2048, 10000, false, 8);
      run_case<4, 2, 2, 1>(8192, 10000, 2080, false, 8);
      run_case<4, 4, 1, 1>(8192, 10000, 2080, false, 8);
      return 0;
    }
    if (S == 42) {  // 4 slices: two 4-wave workgroups per CU (128 x 64 tiles, 72 KB of LDS each) against one 8-wave workgroup (128 x 128)
      run_case<4, 2, 1, 1>(300, 300, 1000, true, 0);
      run_case<4, 2, 1, 1>(8192, 2048, 10000, false, 8);
      run_case<4, 4, 1, 1>(8192, 2048, 10000, false, 8);
      run_case<4, 2, 1, 0>(8192, 2048, 10000, false, 8);
      return 0;
    }
    if (S == 5) run_case<5, 4, 1>(8192, 2080, 10000, false, 3);
    if (S == 6) run_case<6, 2, 1>(8192, 2080, 10000, false, 3);
    return 0;
  }
  run_case<6, 4, 1, 1>(300, 300, 1000, true, 0);
  run_case<6, 4, 1, 1>(8192, 2080, 10000, false, 5);
  run_case<6, 2, 1, 0>(8192, 2080, 10000, false, 5);
  run_case<5, 4, 1, 1>(8192, 2080, 10000, false, 5);
  return 0;
}
