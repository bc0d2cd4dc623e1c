// metric_i8.hip.h — metric assembly on the int8 matrix cores (RMHMC_FLAG_INT8_METRIC), rmhmc.py:57,119,137.
//
// G_ab = sum_n v_n x_na x_nb is a plain GEMM once the data are written as Z[n][(a,b)] = x_na x_nb (a <= b, fixed for all chains):
//     C[chain][pair] = sum_n V[chain][n] Z[pair][n]
// Both operands are cut into S signed-byte slices of a fixed-point number (most significant first; exact two's-complement
// digits, see split_digits) and every slice product with i + j < S is accumulated EXACTLY in int32 by
// v_mfma_i32_32x32x32_i8 (32 cycles for 32768 MACs: ~50x the fp64 matrix rate).  One accumulator set per weight g = i + j,
// combined in fp64 in the epilogue.  The truncation error is 2^-(8S-2) of max|v| max|z| per term: S = 5 gives a norm-wise
// error of 2e-12 on G at config 3, S = 6 1e-14 (the level of fp64 summation itself); see DESIGN.md and tools/i8_sweep.py.
//
// Operand layout in HBM ("stage major": the tile one workgroup needs for one k-stage of 32 data rows is contiguous):
//     Vs[S][nks][nCp][32]  int8     chains,       nCp = chains rounded up to 128
//     Zs[S][nks][NPp][32]  int8     column pairs, NPp = D(D+1)/2 rounded up to the tile width
// Workgroup = 4 waves (2 x 2), wave tile 64 chains x 32*TN pairs, three LDS buffers, 32-byte rows with a one-bit swizzle (i8_lds_off).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef int i4v __attribute__((ext_vector_type(4)));
typedef int i16v __attribute__((ext_vector_type(16)));

#define I8_BM 128
#ifndef I8_EXP
#define I8_EXP 0  // diagnostic builds of tools/i8_gemm_probe only: 1 no global loads, 2 no fragment reads, 4 no barrier (wrong results)
#endif
#define I8_ROWB 32  // LDS bytes per tile row per slice; the two 16-byte halves of rows 8..15 (mod 16) are swapped so that
                    // every ds_read_b128 lane group {0-3,12-15,20-27} / {4-11,16-19,28-31} hits 16 distinct 16-byte slots
__device__ __forceinline__ int i8_lds_off(int row, int half) { return row * I8_ROWB + ((half ^ ((row >> 3) & 1)) << 4); }

// XCD-aware tile order.  Workgroups b, b+8, b+16, ... run on the same XCD (round-robin dispatch) and about 32 consecutive ones
// of them are resident at a time, so XCD x gets the chain blocks x, x+8, ... and walks its (chain block, pair block) grid in
// super-tiles of I8_GC x I8_GP tiles: the 32 resident workgroups then share 4 V tiles and 8 Z tiles per stage through that
// XCD's L2 instead of fetching 19 distinct ones.
#define I8_GC 4
#define I8_GP 8
__device__ __forceinline__ bool i8_tile_of_block(int b, int nCB, int nPB, int& cb, int& pb) {
  const int xcd = b & 7, j = b >> 3;
  const int cbs = (nCB - xcd + 7) >> 3;  // chain blocks of this XCD
  if (j >= cbs * nPB) return false;
  const int cg = j / (I8_GC * nPB);
  const int gc = min(I8_GC, cbs - cg * I8_GC);
  const int r = j - cg * I8_GC * nPB;
  const int pg = r / (gc * I8_GP);
  const int gp = min(I8_GP, nPB - pg * I8_GP);
  const int r2 = r - pg * gc * I8_GP;
  cb = xcd + 8 * (cg * I8_GC + r2 / gp);
  pb = pg * I8_GP + r2 % gp;
  return true;
}

// S slices, workgroup of 2 x WN waves, wave tile 64 chains x 32*TN pairs: workgroup tile 128 x (32*TN*WN).
template <int S, int WN, int TN, class Epilogue>
__device__ __forceinline__ void gemm_i8_tile(const int8_t* __restrict__ Vs, const int8_t* __restrict__ Zs, int nCp, int NPp, int nks,
                                             int cb, int pb, Epilogue&& epi) {
  constexpr int BM = I8_BM, BN = 32 * TN * WN, ROWS = BM + BN, NT = 128 * WN;
  constexpr int STAGE = S * ROWS * I8_ROWB;       // bytes of one LDS buffer
  constexpr int NU = (2 * ROWS + NT - 1) / NT;    // 16-byte units a thread stages per slice
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  // staging: unit u = t + k NT of the combined tile (rows 0..BM-1 = chains, then the pair rows); row u>>1, half u&1
  const size_t strideV = (size_t)nks * nCp * 32, strideZ = (size_t)nks * NPp * 32;
  const int8_t* gsrc[NU];
  size_t gstep[NU], gslice[NU];
  int ldst[NU];
  bool on[NU];
#pragma unroll
  for (int k = 0; k < NU; ++k) {
    const int u = t + k * NT;
    on[k] = u < 2 * ROWS;
    const bool isA = u < 2 * BM;
    gsrc[k] = isA ? Vs + ((size_t)cb * BM) * 32 + (size_t)u * 16 : Zs + ((size_t)pb * BN) * 32 + (size_t)(u - 2 * BM) * 16;
    gstep[k] = isA ? (size_t)nCp * 32 : (size_t)NPp * 32;
    gslice[k] = isA ? strideV : strideZ;
    ldst[k] = i8_lds_off(u >> 1, u & 1);
  }
  i4v rg[S][NU];
  auto gload = [&](int ks) {
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
      for (int k = 0; k < NU; ++k)
        if (on[k]) rg[s][k] = *(const i4v*)(gsrc[k] + s * gslice[k] + (size_t)ks * gstep[k]);
  };
  auto lstore = [&](int buf) {
    unsigned char* base = lds + buf * STAGE;
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
      for (int k = 0; k < NU; ++k)
        if (on[k]) *(i4v*)(base + s * ROWS * I8_ROWB + ldst[k]) = rg[s][k];
  };
  i16v acc[S][2][TN];
#pragma unroll
  for (int g = 0; g < S; ++g)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[g][a][b][r] = 0;
  // (rows 32 apart share the swizzle bit, so the second tile of a wave is a constant 32 rows further)
  const int fragA = i8_lds_off(wm * 64 + (lane & 31), lane >> 5);
  const int fragB = i8_lds_off(BM + wn * 32 * TN + (lane & 31), lane >> 5);
  auto readA = [&](const unsigned char* base, int i, i4v (&fa)[2]) {
#pragma unroll
    for (int a = 0; a < 2; ++a) fa[a] = *(const i4v*)(base + i * ROWS * I8_ROWB + fragA + a * 32 * I8_ROWB);
  };
  auto readB = [&](const unsigned char* base, int j, i4v (&fb)[TN]) {
#pragma unroll
    for (int b = 0; b < TN; ++b) fb[b] = *(const i4v*)(base + j * ROWS * I8_ROWB + fragB + b * 32 * I8_ROWB);
  };

  // Three LDS buffers: stage ks+2 is written while stage ks is multiplied, so stage ks+1 is already visible (made so by the
  // previous barrier) and its fragments are fetched into each register as soon as the last MFMA that reads the register has been
  // issued.  Slice products in the order (a_0; b_{S-1} .. b_0), (a_1; b_{S-2} .. b_0), ...: b_j is dead after group S-1-j, and the next
  // stage asks for b_{S-1} first, so no fragment is waited for except in the prologue.
  gload(0);
  lstore(0);
  if (nks > 1) { gload(1); lstore(1); }
  __syncthreads();
  i4v fb[S][TN], fa[2][2];
#pragma unroll
  for (int j = 0; j < S; ++j) readB(lds, j, fb[j]);
  readA(lds, 0, fa[0]);
  int cur = 0;  // buffer of stage ks
  for (int ks = 0; ks < nks; ++ks) {
    const int nxt = cur == 2 ? 0 : cur + 1, wr = nxt == 2 ? 0 : nxt + 1;
    const bool more = ks + 1 < nks;
    if (!(I8_EXP & 1) && ks + 2 < nks) gload(ks + 2);
    const unsigned char* bc = lds + cur * STAGE;
    const unsigned char* bn = lds + nxt * STAGE;
#pragma unroll
    for (int i = 0; i < S; ++i) {
      if (I8_EXP & 2) {
      } else if (i + 1 < S) readA(bc, i + 1, fa[(i + 1) & 1]);
      else if (more && (S & 1) == 0) readA(bn, 0, fa[0]);  // even S: fa[0] is free during the last group
#pragma unroll
      for (int j = S - 1 - i; j >= 0; --j)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b)
            acc[i + j][a][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[i & 1][a], fb[j][b], acc[i + j][a][b], 0, 0, 0);
      if (!(I8_EXP & 2) && more) readB(bn, S - 1 - i, fb[S - 1 - i]);
      if (!(I8_EXP & 2) && i + 1 == S && more && (S & 1) == 1) readA(bn, 0, fa[0]);  // odd S: the last group itself reads fa[0]
    }
    if (!(I8_EXP & 1) && ks + 2 < nks) lstore(wr);
    if (!(I8_EXP & 4)) __syncthreads();
    cur = nxt;
  }
  // epilogue: combine the weights, smallest first.  C/D map of the 32x32 forms: col = lane&31, row = (r&3) + 8(r>>2) + 4(lane>>5)
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        double val = 0.0;
#pragma unroll
        for (int g = S - 1; g >= 0; --g) val = val * 0.00390625 + (double)acc[g][a][b][r];  // Horner in 2^-8
        const int row = cb * BM + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int col = pb * BN + wn * 32 * TN + b * 32 + (lane & 31);
        epi(row, col, val);
      }
}

// ---------------------------------------------------------------------------------------------
// The same tile with LDS-DMA staging (global_load_lds_dwordx4): no staging registers, no ds_write, and the loads of stage ks+2
// are in flight while stages ks and ks+1 are multiplied (two stages of latency tolerance instead of one; the register-staged
// loop above stalls on the Infinity-Cache / HBM latency of its single-stage prefetch).  One wave instruction writes 64 x 16
// contiguous LDS bytes, so the swizzle of i8_lds_off goes on the per-lane SOURCE address.  Counted s_waitcnt vmcnt + raw
// s_barrier as the CDNA guide prescribes: a stage is read one barrier after the wait that retired its loads.
// ---------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt immediate");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int S, int WN, int TN, class Epilogue>
__device__ __forceinline__ void gemm_i8_tile_glds(const int8_t* __restrict__ Vs, const int8_t* __restrict__ Zs, int nCp, int NPp, int nks,
                                                  int cb, int pb, Epilogue&& epi) {
  constexpr int BM = I8_BM, BN = 32 * TN * WN, ROWS = BM + BN, NT = 128 * WN;
  constexpr int STAGE = S * ROWS * I8_ROWB;
  constexpr int NU = (2 * ROWS + NT - 1) / NT;
  static_assert((2 * ROWS) % 64 == 0, "whole waves of 16-byte units");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const size_t strideV = (size_t)nks * nCp * 32, strideZ = (size_t)nks * NPp * 32;
  const int8_t* gsrc[NU];
  size_t gstep[NU], gslice[NU];
  int lbase[NU];  // wave-uniform LDS byte offset of the wave's 64 units
  bool on[NU];
  int n_on = 0;
#pragma unroll
  for (int k = 0; k < NU; ++k) {
    const int u = t + k * NT;
    on[k] = (u & ~63) < 2 * ROWS;  // wave-uniform
    n_on += on[k] ? 1 : 0;
    const bool isA = u < 2 * BM;
    const int row = u >> 1, half = (u & 1) ^ ((row >> 3) & 1);  // the unit stored at LDS slot u holds this logical half
    const int lrow = isA ? row : row - BM;
    gsrc[k] = (isA ? Vs + ((size_t)cb * BM) * 32 : Zs + ((size_t)pb * BN) * 32) + (size_t)lrow * 32 + half * 16;
    gstep[k] = isA ? (size_t)nCp * 32 : (size_t)NPp * 32;
    gslice[k] = isA ? strideV : strideZ;
    lbase[k] = __builtin_amdgcn_readfirstlane((u & ~63) * 16);
  }
  auto gl = [&](int ks, int buf) {
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
      for (int k = 0; k < NU; ++k)
        if (on[k])
          __builtin_amdgcn_global_load_lds((gbl_ptr_t)(gsrc[k] + s * gslice[k] + (size_t)ks * gstep[k]),
                                           (lds_ptr_t)(lds + buf * STAGE + s * ROWS * I8_ROWB + lbase[k]), 16, 0, 0);
  };
  // all but the newest stage's loads of this wave have landed (the waves of a partly filled last unit row issue fewer)
  auto retire_older = [&]() {
    if (NU == 1 || n_on == NU) wait_vmcnt<S * NU>();
    else wait_vmcnt<S*(NU - 1)>();
  };
  i16v acc[S][2][TN];
#pragma unroll
  for (int g = 0; g < S; ++g)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[g][a][b][r] = 0;
  const int fragA = i8_lds_off(wm * 64 + (lane & 31), lane >> 5);
  const int fragB = i8_lds_off(BM + wn * 32 * TN + (lane & 31), lane >> 5);

  gl(0, 0);
  if (nks > 1) { gl(1, 1); retire_older(); } else wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  int cur = 0;
  for (int ks = 0; ks < nks; ++ks) {
    const int nxt = cur == 2 ? 0 : cur + 1, wr = nxt == 2 ? 0 : nxt + 1;
    if (ks + 2 < nks) gl(ks + 2, wr);
    const unsigned char* bc = lds + cur * STAGE;
    i4v fb[S][TN], fa[2][2];
    // issue order pinned (sched_barrier) so that the fragments of product group i+1 are in flight while group i is multiplied
    // and the compiler's counted lgkmcnt waits retire only what the next MFMA needs
#pragma unroll
    for (int a = 0; a < 2; ++a) fa[0][a] = *(const i4v*)(bc + fragA + a * 32 * I8_ROWB);
#pragma unroll
    for (int j = S - 1; j >= 0; --j)
#pragma unroll
      for (int b = 0; b < TN; ++b) fb[j][b] = *(const i4v*)(bc + j * ROWS * I8_ROWB + fragB + b * 32 * I8_ROWB);
#pragma unroll
    for (int i = 0; i < S; ++i) {
      if (i + 1 < S) {
#pragma unroll
        for (int a = 0; a < 2; ++a) fa[(i + 1) & 1][a] = *(const i4v*)(bc + (i + 1) * ROWS * I8_ROWB + fragA + a * 32 * I8_ROWB);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = S - 1 - i; j >= 0; --j) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b)
            acc[i + j][a][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[i & 1][a], fb[j][b], acc[i + j][a][b], 0, 0, 0);
        if (i == 0) __builtin_amdgcn_sched_barrier(0);  // first group: start on b_{S-1} while b_{S-2}.. are still landing
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (ks + 2 < nks) retire_older(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    cur = nxt;
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        double val = 0.0;
#pragma unroll
        for (int g = S - 1; g >= 0; --g) val = val * 0.00390625 + (double)acc[g][a][b][r];
        const int row = cb * BM + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int col = pb * BN + wn * 32 * TN + b * 32 + (lane & 31);
        epi(row, col, val);
      }
}

// ---------------------------------------------------------------------------------------------
// Ping-pong form (8 waves): waves w and w+4 share a SIMD; while one of them issues its S(S+1)/2 * TM * TN MFMAs back to back
// from registers, the other one fetches its fragments of the next stage from LDS, then they swap (two barriers per stage).  The
// matrix pipe of every SIMD always has exactly one wave feeding it.  LDS-DMA staging three stages ahead into three buffers:
// a buffer is free as soon as the second group has read its fragments (barrier 1), and the loads of stage ks+3 issued then have
// two full stages to land.
// Workgroup tile (32 TM WM) x (32 TN WN) with WM WN = 8; BM = 32 TM WM must equal I8_BM.
// ---------------------------------------------------------------------------------------------
template <int S, int WM, int WN, int TM, int TN, class Epilogue>
__device__ __forceinline__ void gemm_i8_tile_pp(const int8_t* __restrict__ Vs, const int8_t* __restrict__ Zs, int nCp, int NPp, int nks,
                                                int cb, int pb, Epilogue&& epi) {
  static_assert(WM * WN == 8 && 32 * TM * WM == I8_BM, "8 waves, 128 chain rows");
  constexpr int BM = I8_BM, BN = 32 * TN * WN, ROWS = BM + BN, NT = 512;
  constexpr int STAGE = S * ROWS * I8_ROWB;
  constexpr int NU = (2 * ROWS + NT - 1) / NT;
  static_assert((2 * ROWS) % 64 == 0, "whole waves of 16-byte units");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int grp = wave >> 2;                 // waves w and w+4 sit on the same SIMD
  const int wm = wave % WM, wn = wave / WM;  // any bijection onto the WM x WN wave grid
  const size_t strideV = (size_t)nks * nCp * 32, strideZ = (size_t)nks * NPp * 32;
  const int8_t* gsrc[NU];
  size_t gstep[NU], gslice[NU];
  int lbase[NU];
  bool on[NU];
  int n_on = 0;
#pragma unroll
  for (int k = 0; k < NU; ++k) {
    const int u = t + k * NT;
    on[k] = (u & ~63) < 2 * ROWS;
    n_on += on[k] ? 1 : 0;
    const bool isA = u < 2 * BM;
    const int row = u >> 1, half = (u & 1) ^ ((row >> 3) & 1);
    const int lrow = isA ? row : row - BM;
    gsrc[k] = (isA ? Vs + ((size_t)cb * BM) * 32 : Zs + ((size_t)pb * BN) * 32) + (size_t)lrow * 32 + half * 16;
    gstep[k] = isA ? (size_t)nCp * 32 : (size_t)NPp * 32;
    gslice[k] = isA ? strideV : strideZ;
    lbase[k] = __builtin_amdgcn_readfirstlane((u & ~63) * 16);
  }
  auto gl = [&](int ks, int buf) {
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
      for (int k = 0; k < NU; ++k)
        if (on[k])
          __builtin_amdgcn_global_load_lds((gbl_ptr_t)(gsrc[k] + s * gslice[k] + (size_t)ks * gstep[k]),
                                           (lds_ptr_t)(lds + buf * STAGE + s * ROWS * I8_ROWB + lbase[k]), 16, 0, 0);
  };
  // leave at most `stages` of this wave's newest stage loads in flight
  auto retire = [&](int stages) {
    const bool full = (NU == 1 || n_on == NU);
    if (stages >= 2) { if (full) wait_vmcnt<2 * S * NU>(); else wait_vmcnt<2 * S*(NU - 1)>(); }
    else if (stages == 1) { if (full) wait_vmcnt<S * NU>(); else wait_vmcnt<S*(NU - 1)>(); }
    else wait_vmcnt<0>();
  };
  i16v acc[S][TM][TN];
#pragma unroll
  for (int g = 0; g < S; ++g)
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[g][a][b][r] = 0;
  const int fragA = i8_lds_off(wm * 32 * TM + (lane & 31), lane >> 5);
  const int fragB = i8_lds_off(BM + wn * 32 * TN + (lane & 31), lane >> 5);
  i4v fa[S][TM], fb[S][TN];
  auto read_frags = [&](int buf) {
    const unsigned char* bc = lds + buf * STAGE;
#pragma unroll
    for (int i = 0; i < S; ++i) {
#pragma unroll
      for (int a = 0; a < TM; ++a) fa[i][a] = *(const i4v*)(bc + i * ROWS * I8_ROWB + fragA + a * 32 * I8_ROWB);
#pragma unroll
      for (int b = 0; b < TN; ++b) fb[i][b] = *(const i4v*)(bc + i * ROWS * I8_ROWB + fragB + b * 32 * I8_ROWB);
    }
  };
  auto multiply = [&]() {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < S; ++i)
#pragma unroll
      for (int j = 0; j < S - i; ++j)
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b)
            acc[i + j][a][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[i][a], fb[j][b], acc[i + j][a][b], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };

  gl(0, 0);
  if (nks > 1) gl(1, 1);
  if (nks > 2) gl(2, 2);
  retire(nks > 2 ? 2 : nks > 1 ? 1 : 0);  // stage 0 landed
  __builtin_amdgcn_s_barrier();
  if (grp == 0 || (I8_EXP & 8)) read_frags(0);
  int cur = 0;  // buffer of stage ks
  for (int ks = 0; ks < nks; ++ks) {
    const int nxt = cur == 2 ? 0 : cur + 1;
    // phase 1: group 0 multiplies stage ks, group 1 fetches its fragments of stage ks
    if (I8_EXP & 8) {
    } else if (grp == 0) multiply(); else read_frags(cur);
    retire(ks + 2 < nks ? 1 : 0);                     // stage ks+1 landed (this wave's share)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // group 1 is done with buffer `cur`
    __builtin_amdgcn_s_barrier();
    // phase 2: group 1 multiplies stage ks, group 0 fetches stage ks+1; buffer `cur` is refilled with stage ks+3
    if (!(I8_EXP & 16) && ks + 3 < nks) gl(ks + 3, cur);
    if (I8_EXP & 8) {
    } else if (grp == 1) multiply(); else if (ks + 1 < nks) read_frags(nxt);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    cur = nxt;
  }
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        double val = 0.0;
#pragma unroll
        for (int g = S - 1; g >= 0; --g) val = val * 0.00390625 + (double)acc[g][a][b][r];
        const int row = cb * BM + (wm * TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int col = pb * BN + (wn * TN + b) * 32 + (lane & 31);
        epi(row, col, val);
      }
}

template <int S, int BN>
constexpr int i8_pp_lds_bytes() { return 3 * S * (I8_BM + BN) * I8_ROWB; }

template <int S, int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(512) void k_gemm_i8_pp_probe(const int8_t* __restrict__ Vs, const int8_t* __restrict__ Zs, int nCp, int NPp,
                                                          int nks, double* __restrict__ C) {
  int cb, pb;
  if (!i8_tile_of_block(blockIdx.x, nCp / I8_BM, NPp / (32 * TN * WN), cb, pb)) return;
  gemm_i8_tile_pp<S, WM, WN, TM, TN>(Vs, Zs, nCp, NPp, nks, cb, pb, [&](int row, int col, double val) { C[(size_t)row * NPp + col] = val; });
}

template <int S, int WN, int TN>
constexpr int i8_lds_bytes() { return 3 * S * (I8_BM + 32 * TN * WN) * I8_ROWB; }

// probe / unit-test form: C[row][col] = sum_g acc_g 2^(-8g)
template <int S, int WN, int TN, int OCC, int GLDS>
__global__ __launch_bounds__(128 * WN) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void k_gemm_i8_probe(const int8_t* __restrict__ Vs, const int8_t* __restrict__ Zs, int nCp, int NPp, int nks,
                                                       double* __restrict__ C) {
  int cb, pb;
  if (!i8_tile_of_block(blockIdx.x, nCp / I8_BM, NPp / (32 * TN * WN), cb, pb)) return;
  auto epi = [&](int row, int col, double val) { C[(size_t)row * NPp + col] = val; };
  if (GLDS) gemm_i8_tile_glds<S, WN, TN>(Vs, Zs, nCp, NPp, nks, cb, pb, epi);
  else gemm_i8_tile<S, WN, TN>(Vs, Zs, nCp, NPp, nks, cb, pb, epi);
}

// ---------------------------------------------------------------------------------------------
// operand preparation
// ---------------------------------------------------------------------------------------------
// exact balanced base-256 digits of a two's-complement integer, least significant first: N = sum_k d_k 256^k, d_k in [-128, 127]
template <int S>
__device__ __forceinline__ void split_digits(long long N, int (&d)[S]) {
#pragma unroll
  for (int k = 0; k < S; ++k) {
    const int b = (int)(signed char)(N & 0xFF);
    d[k] = b;
    N = (N - b) >> 8;
  }
}

struct I8Pairs {
  const short* pa;      // pair p = (pa[p], pb[p]), pa <= pb < D; p >= NP is padding
  const short* pb;
  const double* scale;  // 2^(e_p - 14): C[c][p] * scale = sum_n v_n z_np
  int NP, NPp;
};

// e_p: smallest exponent with max_n |x_na x_nb| < 2^e_p.  One workgroup per pair.
__global__ __launch_bounds__(256) void k_zmax(const double* __restrict__ Xt, int M, int Mp, I8Pairs pr, int* __restrict__ ze,
                                              double* __restrict__ scale) {
  __shared__ double red[256];
  const int p = blockIdx.x;
  double m = 0.0;
  if (p < pr.NP) {
    const double* xa = Xt + (size_t)pr.pa[p] * Mp;
    const double* xb = Xt + (size_t)pr.pb[p] * Mp;
    for (int n = threadIdx.x; n < M; n += 256) m = fmax(m, fabs(xa[n] * xb[n]));
  }
  red[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    int e = 0;
    if (red[0] > 0.0 && red[0] < 1e300) (void)frexp(red[0], &e);  // max = m 2^e, m in [0.5, 1)
    ze[p] = e;
    scale[p] = ldexp(1.0, e - 14);
  }
}

// Zs[s][ks][p][k] = digit s (most significant first) of rint(x_na x_nb 2^(8S-2-e_p)), data row n = 32 ks + k.  A thread cuts 4 rows.
template <int S>
__global__ __launch_bounds__(256) void k_zsplit(const double* __restrict__ Xt, int M, int Mp, I8Pairs pr, const int* __restrict__ ze,
                                                int nks, int8_t* __restrict__ Zs) {
  const int p = blockIdx.y;
  const int q = blockIdx.x * 256 + threadIdx.x;  // group of 4 data rows
  if (q * 4 >= nks * 32) return;
  int w[S];
#pragma unroll
  for (int s = 0; s < S; ++s) w[s] = 0;
  if (p < pr.NP) {
    const double* xa = Xt + (size_t)pr.pa[p] * Mp;
    const double* xb = Xt + (size_t)pr.pb[p] * Mp;
    const int sh = 8 * S - 2 - ze[p];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int n = 4 * q + k;
      const double z = n < M ? xa[n] * xb[n] : 0.0;
      int d[S];
      split_digits<S>((long long)rint(ldexp(z, sh)), d);
#pragma unroll
      for (int s = 0; s < S; ++s) w[s] |= (d[S - 1 - s] & 0xFF) << (8 * k);
    }
  }
  const int ks = q >> 3, k4 = q & 7;
#pragma unroll
  for (int s = 0; s < S; ++s) *(int*)(Zs + (((size_t)s * nks + ks) * pr.NPp + p) * 32 + 4 * k4) = w[s];
}

// Vs[s][ks][c][k] = digit s of rint(v 2^(8S)), v in [0, 1/4].  One workgroup per chain; a non-finite v (diverged chain) raises
// vbad[c], which turns the chain's G into NaN in the epilogue exactly as it would be in floating point.
template <int S>
__global__ __launch_bounds__(256) void k_vsplit(const double* __restrict__ vrow, int Mp, int n_chains, const int* __restrict__ phase, int nks,
                                                int nCp, int8_t* __restrict__ Vs, int* __restrict__ vbad) {
  const int c = blockIdx.x;
  if (phase[c] != 1) return;
  const double* v = vrow + (size_t)c * Mp;
  int bad = 0;
  for (int q = threadIdx.x; q < nks * 8; q += 256) {
    int w[S];
#pragma unroll
    for (int s = 0; s < S; ++s) w[s] = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int n = 4 * q + k;
      double x = n < Mp ? v[n] : 0.0;
      if (!(x >= 0.0 && x <= 0.25)) { bad = 1; x = 0.0; }
      int d[S];
      split_digits<S>((long long)rint(ldexp(x, 8 * S)), d);
#pragma unroll
      for (int s = 0; s < S; ++s) w[s] |= (d[S - 1 - s] & 0xFF) << (8 * k);
    }
    const int ks = q >> 3, k4 = q & 7;
#pragma unroll
    for (int s = 0; s < S; ++s) *(int*)(Vs + (((size_t)s * nks + ks) * nCp + c) * 32 + 4 * k4) = w[s];
  }
  bad = __syncthreads_or(bad);
  if (threadIdx.x == 0) vbad[c] = bad;
}

// the assembly proper: G[c] = sym(C[c][:] * scale) + I/alpha, natural row-major DP x DP like k_assemble
template <int S, int WN, int TN>
__global__ __launch_bounds__(128 * WN) void k_assemble_i8(const int8_t* __restrict__ Vs, const int8_t* __restrict__ Zs, int nCp, int nks,
                                                          I8Pairs pr, int n_chains, const int* __restrict__ phase,
                                                          const int* __restrict__ vbad, int DP, double inv_alpha, double* __restrict__ Gq) {
  int cb, pb;
  if (!i8_tile_of_block(blockIdx.x, nCp / I8_BM, pr.NPp / (32 * TN * WN), cb, pb)) return;
  gemm_i8_tile<S, WN, TN>(Vs, Zs, nCp, pr.NPp, nks, cb, pb, [&](int c, int p, double val) {
    if (c >= n_chains || p >= pr.NP) return;
    if (phase[c] != 1) return;
    const int a = pr.pa[p], b = pr.pb[p];
    double g = val * pr.scale[p];
    if (a == b) g += inv_alpha;
    if (vbad[c]) g = __builtin_nan("");
    double* G = Gq + (size_t)c * DP * DP;
    G[a * DP + b] = g;
    G[b * DP + a] = g;
  });
}
