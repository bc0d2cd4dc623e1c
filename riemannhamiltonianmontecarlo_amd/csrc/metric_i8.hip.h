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
// Error bound (what tests/test_gpu_int8_stress.py asserts and what RMHMC_FLAG_INT8_CERTIFY checks in rmhmc_set_data).
//   Fixed-point grids: v in [0, 1/4] is rounded to a multiple of 2^-8S (absolute grid: |dv| <= 2^-(8S+1); refined to 2^-(8S+vexp_c) for a
//   chain whose v is provably below 2^-(2+vexp_c) on every data row, see VSlice in kernels.hip.h: the bounds below then hold with an
//   extra factor 2^-vexp_c, i.e. relative to that chain's own largest v);
//   z_n,ab = x_na x_nb to a multiple of 2^(e_ab-8S+2), 2^e_ab > max_n |z_n,ab| (one exponent per column pair: |dz| <= 2^(e_ab-8S+1)).
//   The integer GEMM is exact for the slice products it keeps (i + j < S); the dropped ones are at most
//   (S-1) 2^14 256^(S-2) grid units per term.  Per data row n, with v <= 2^-2 and |z| < 2^e_ab:
//       |d(v z)|  <=  v |dz| + |z| |dv| + dropped  <=  2^(e_ab-8S-1) + 2^(e_ab-8S-1) + (S-1) 2^(e_ab-8S)  =  S 2^(e_ab-8S)
//   hence, for ANY chain state,
//       |dG_ab|  <=  S M 2^(e_ab-8S)            (worst case; rounding errors are sign-random, the typical error is ~sqrt(M) smaller)
//   The bound scales with the columns like G_ab itself (e_ab follows the scale of x_a x_b), so badly scaled columns and an all-ones
//   intercept cost nothing.  What does cost: a data row far larger than the others raises e_ab for its pairs and coarsens the grid of
//   every other row.  The certificate  max_ab S M 2^(e_ab-8S) / sqrt(G0_aa G0_bb),  G0 = X'X/4 + I/alpha (the metric at w = 0),  is
//   evaluated by rmhmc_set_data: 3e-12 for N(0,1) data at M = 10^4, S = 6; 1.4e-8 with one row 1000x the rest, which
//   RMHMC_FLAG_INT8_CERTIFY sends to the fp64 kernels (tolerance 1e-9).  At a state whose mean curvature vbar = mean(v) is below 1/4
//   the data part of G shrinks and the relative bound grows by 1/(4 vbar), until the prior floor: G >= I/alpha always, so
//   |dG_ab| alpha <= S M 2^(e_ab-8S) alpha (7e-7 worst case at config 3) bounds any state; a chain saturated on every row (all
//   |x_n.w| large: needs an intercept-like column) gets the refined grid above and stays at the accuracy of an unsaturated one.
//   Leverage pass: the columns are equilibrated by exact powers of two first (x~_a = x_a 2^-cexp[a], |x~| < 1; Q~_ab = G^-1_ab
//   2^(cexp[a]+cexp[b]), the inverse of the equilibrated metric), because x' G^-1 x is a sum of O(1) terms whose two factors vary by
//   the SQUARE of the column scales in opposite directions.  Then Q~ carries the CHAIN's exponent and Z~ the DATA ROW's, so h_n is
//   accurate relative to max|Q~| max_a|x~_na|^2 for every (chain, row): |dh_n| <= S NP 2^(eq_c + ez_n + 4 - 8S), NP = D(D+1)/2.
//   Delta assembly (I8Delta below; S = 6, the evaluation that ends a leapfrog step): G = G(last position iterate) + the integer GEMM of
//   the DIFFERENCE of the two v grids.  The grids and their rounding are those of a full assembly (the difference of two grid values is
//   exact); only the dropped slice products (i + j >= S) enter twice, once with the digits of the base matrix and once with those of
//   the difference: |dG_ab| <= (2S - 1) M 2^(e_ab-8S), which is what the certificate uses when the delta assembly is on.
//   Inner iterates: at S = 6 the assemblies whose G only steers a fixed-point iterate (position iterates before the last, launch_assemble in
//   rmhmc_hip.hip) use the S - 1 most significant slices of the same operands, i.e. the bounds above with S - 1 for THOSE matrices only; every
//   G that enters a Hamiltonian, a leverage, rmhmc_metric or the last iterate has the full S (RMHMC_FLAG_INT8_INNER_FULL: all of them).
//
// Operand layout in HBM ("stage major": the tile one workgroup needs for one k-stage of 32 data rows is contiguous):
//     Vs[S][nks][nCp][32]  int8     chains,       nCp = chains rounded up to 128
//     Zs[S][nks][NPp][32]  int8     column pairs, NPp = D(D+1)/2 rounded up to the tile width
// Workgroup = 2 x WN waves, wave tile 64 chains x 32*TN pairs, three LDS buffers filled by LDS-DMA, 32-byte rows with a one-bit
// swizzle (i8_lds_off).  A slicing with S digits contains the one with S - 1 as its leading planes (balanced digits), which is what the
// inner iterates and the S - 1 instantiations of the kernels use.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>

typedef int i4v __attribute__((ext_vector_type(4)));
typedef int i16v __attribute__((ext_vector_type(16)));

#define I8_BM 128
#define I8_ROWB 32  // LDS bytes per tile row per slice; the two 16-byte halves of rows 8..15 (mod 16) are swapped so that
                    // every ds_read_b128 lane group {0-3,12-15,20-27} / {4-11,16-19,28-31} hits 16 distinct 16-byte slots
__device__ __forceinline__ int i8_lds_off(int row, int half) { return row * I8_ROWB + ((half ^ ((row >> 3) & 1)) << 4); }

// XCD-aware tile order.  Workgroups b, b+8, b+16, ... run on the same XCD (round-robin dispatch) and about 32 consecutive ones
// of them are resident at a time, so XCD x gets the chain blocks x, x+8, ... and walks its (chain block, pair block) grid in
// super-tiles of I8_GC x I8_GP tiles: the 32 resident workgroups then share 4 V tiles and 8 Z tiles per stage through that
// XCD's L2 instead of fetching 19 distinct ones.
#ifndef I8_GC
#define I8_GC 4
#define I8_GP 8
#endif
#ifndef I8_DSPREAD
#define I8_DSPREAD 1   // the LDS-DMA loads of stage ks+2 are issued one slice per product group instead of all at the top of the stage, where
                       // they collided with every wave's burst of fragment reads: 3.07-3.14 -> 3.01-3.06 ms per launch at config 3, S = 6
                       // (two slices per long group, after the group's MFMAs, or one per MFMA pair of the first group: no better;
                       // product groups in reverse order: worse; profiles/r02_i8_tile_ablation.txt).  0: all at the top (round 1)
#endif
__device__ __forceinline__ bool i8_tile_of_block(int b, int nCB, int nPB, int& cb, int& pb) {
  if (nCB < 8) {  // fewer chain blocks than XCDs: the grouping below would leave whole XCDs idle, so the tiles are simply dealt round
    if (b >= nCB * nPB) return false;
    cb = b % nCB;
    pb = b / nCB;
    return true;
  }
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

// ---------------------------------------------------------------------------------------------
// One workgroup tile: S slices, 2 x WN waves, wave tile 64 chains x 32*TN pairs, workgroup tile 128 x (32*TN*WN).
// Staging by LDS-DMA (global_load_lds_dwordx4): no staging registers, no ds_write, and the loads of stage ks+2 are in flight
// while stages ks and ks+1 are multiplied (a register-staged loop with a one-stage prefetch stalled on the Infinity-Cache /
// HBM latency; an 8-wave ping-pong schedule measured the same as this one: the loop runs at ~80 % of what a bare MFMA stream
// reaches, and that stream is power limited to ~2.85 POP/s on random bytes).  One wave instruction writes
// 64 x 16 contiguous LDS bytes, so the swizzle of i8_lds_off goes on the per-lane SOURCE address.  Counted s_waitcnt vmcnt +
// raw s_barrier as the CDNA guide prescribes: a stage is read one barrier after the wait that retired its loads.
// rows_real / cols_real: extent of the unpadded problem inside this tile; waves whose whole wave tile is padding skip the
// MFMAs (they still stage and synchronise), which halves the cost of a mostly empty last pair block.
// ---------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt immediate");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Hand-placed fragment reads and waits (I8_ASMFRAG, the TN = 1 forms).  The compiler closes the fragment reads of a stage with one
// s_waitcnt lgkmcnt(0) before the first MFMA - ten ds_read_b128 (the next group's A fragments included) where that MFMA needs three -
// and with eight waves starting a stage together after the barrier the whole workgroup's 80 KB of reads are serialised in front of the
// first matrix instruction.  Here the reads are inline assembly (the compiler does not track them) and every MFMA pair is preceded by
// the counted wait for exactly its operands; ds_read results return in issue order.  The waits name their fragments as in/out
// operands so that the MFMAs using them cannot be scheduled above the wait.
template <int OFF>
__device__ __forceinline__ i4v lds_read_b128(unsigned addr) {
  i4v r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
  return r;
}
template <int N>
__device__ __forceinline__ void wait_lgkm(i4v& a, i4v& b, i4v& c) {
  asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_lgkm(i4v& a, i4v& b, i4v& c, i4v& d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_lgkm(i4v& a, i4v& b) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}
#ifndef I8_ASMFRAG
#define I8_ASMFRAG 1
#endif

// I8_ABLATE (tools/i8_gemm_probe.hip only; the library never defines it): bit 0 no LDS-DMA loads, bit 1 no fragment reads (the
// fragments of the first stage are kept), bit 2 no barriers, bit 3 LDS-DMA loads for the first four stages of a tile only (the products
// then run on real bytes without a feed).  Timing ablations: the results are meaningless.
#ifndef I8_ABLATE
#define I8_ABLATE 0
#endif
// sum_g acc_g 2^(-8g) in fp64 (Horner from the least significant weight): the ONE place where the integer sums become a double, so
// that every route to a G entry (one launch, k pieces summed as integers, the left-over pair blocks) rounds alike
template <int S, class F>
__device__ __forceinline__ double i8_combine(F&& acc_of) {
  double val = 0.0;
#pragma unroll
  for (int g = S - 1; g >= 0; --g) val = val * 0.00390625 + (double)acc_of(g);
  return val;
}

// LDS stage buffers of a tile: three (loads two stages ahead), four for the short stages of S <= 4 in the forms whose loads are issued
// unconditionally (ALLON below).  A stage of S = 4 is 20 MFMAs per wave, ~0.65 us: two of them do not cover the latency of an LDS-DMA
// load that misses L2 (MFMA busy 61 % against 72-76 % for S = 5, 6 on the same tile); three stages ahead = 128 KB of the 160.
#ifndef I8_NBUF4
#define I8_NBUF4 1
#endif
template <int S, int WN, int TN>
constexpr int i8_nbuf() {
  constexpr int ROWS = I8_BM + 32 * TN * WN, NT = 128 * WN, NU = (2 * ROWS + NT - 1) / NT;
  return (I8_NBUF4 && S <= 4 && 2 * ROWS == NU * NT && 4 * S * ROWS * 32 <= 140 * 1024) ? 4 : 3;
}

// RAW: the epilogue gets the int32 accumulators themselves, epi(row, col, g, acc_g), instead of their fp64 combination.
template <int S, int WN, int TN, int PIN, bool RAW = false, class Epilogue>
__device__ __forceinline__ void gemm_i8_tile(const int8_t* __restrict__ Vs, const int8_t* __restrict__ Zs, int nCp, int NPp, int nks_total, int nks,
                                             int cb, int pb, int rows_real, int cols_real, Epilogue&& epi) {
  constexpr int BM = I8_BM, BN = 32 * TN * WN, ROWS = BM + BN, NT = 128 * WN;
  constexpr int STAGE = S * ROWS * I8_ROWB;
  constexpr int NU = (2 * ROWS + NT - 1) / NT;
  static_assert((2 * ROWS) % 64 == 0, "whole waves of 16-byte units");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  // nks stages starting at the given pointers; nks_total = stages per slice plane (a launch may cover part of the k range)
  const size_t strideV = (size_t)nks_total * nCp * 32, strideZ = (size_t)nks_total * NPp * 32;
  // A wave's 64 units of a stage lie all in the A (chain) part or all in the B part, so the operand base, the stage step and the slice
  // stride are wave uniform (SGPRs) and only the unit's offset inside the tile is per lane: the loads take the scalar-base form,
  // one 32-bit VGPR of address per unit instead of 64-bit pointer arithmetic in the loop.
  const int8_t* gbase[NU];
  size_t gstep[NU], gslice[NU];
  unsigned goff[NU];
  int lbase[NU];  // wave-uniform LDS byte offset of the wave's 64 units
  bool on[NU];
  int n_on = 0;
#pragma unroll
  for (int k = 0; k < NU; ++k) {
    const int u = t + k * NT;
    on[k] = (u & ~63) < 2 * ROWS;  // wave-uniform
    n_on += on[k] ? 1 : 0;
    const bool isA = __builtin_amdgcn_readfirstlane((int)((u & ~63) < 2 * BM)) != 0;  // wave-uniform
    const int row = u >> 1, half = (u & 1) ^ ((row >> 3) & 1);  // the unit stored at LDS slot u holds this logical half
    const int lrow = isA ? row : row - BM;
    gbase[k] = isA ? Vs + ((size_t)cb * BM) * 32 : Zs + ((size_t)pb * BN) * 32;
    goff[k] = (unsigned)(lrow * 32 + half * 16);
    gstep[k] = isA ? (size_t)nCp * 32 : (size_t)NPp * 32;
    gslice[k] = isA ? strideV : strideZ;
    lbase[k] = __builtin_amdgcn_readfirstlane((u & ~63) * 16);
  }
  // ALLON: every thread owns exactly one 16-byte unit of the stage (the 8-wave form: 2 ROWS = 512 units = 512 threads).  Then the loads
  // are issued UNCONDITIONALLY - in the last two stages of the tile, which have nothing left to prefetch, the last stage is fetched again
  // into a buffer nobody reads any more - so that a stage of the main loop is one basic block: with the conditional issue the
  // compiler closed every product group with s_waitcnt lgkmcnt(0), i.e. the first MFMA of a stage waited for all ten fragment
  // reads instead of the two it needs.
#ifndef I8_ALLON
#define I8_ALLON 1
#endif
  constexpr bool ALLON = I8_ALLON && (2 * ROWS == NU * NT);
  constexpr int NBUF = I8_ALLON ? i8_nbuf<S, WN, TN>() : 3, PD = NBUF - 1;  // ring of stage buffers, prefetch distance
  static_assert(NBUF == 3 || ALLON, "the deeper ring relies on the unconditional issue");
  auto gl1 = [&](int ks, int buf, int s) {  // slice s of stage ks
    if (I8_ABLATE & 1) return;
    if ((I8_ABLATE & 8) && ks >= 4) return;  // (bit 3: only the first stages of a tile are fetched: real bytes in LDS, no feed afterwards)
#pragma unroll
    for (int k = 0; k < NU; ++k)
      if (ALLON || on[k])
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(gbase[k] + (s * gslice[k] + (size_t)ks * gstep[k]) + goff[k]),
                                         (lds_ptr_t)(lds + buf * STAGE + s * ROWS * I8_ROWB + lbase[k]), 16, 0, 0);
  };
  auto gl = [&](int ks, int buf) {
#pragma unroll
    for (int s = 0; s < S; ++s) gl1(ks, buf, s);
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

  // wave-uniform.  Only in the 8-wave form: with one wave per SIMD nothing is gained, and a branch around the MFMAs makes the
  // compiler shuttle the AGPR accumulators through VGPR copies (972 v_accvgpr moves in the S = 6 loop, 1.4x slower)
  const bool work = WN != 4 || (wm * 64 < rows_real && wn * 32 * TN < cols_real);
  if constexpr (NBUF == 3) {
    gl(0, 0);
    if (nks > 1) { gl(1, 1); retire_older(); } else wait_vmcnt<0>();
  } else {  // (ALLON: stages past the end are the last stage again, into buffers nobody reads)
#pragma unroll
    for (int d = 0; d < PD; ++d) gl(min(d, nks - 1), d);
    wait_vmcnt<(PD - 1) * S * NU>();
  }
  __builtin_amdgcn_s_barrier();
  int cur = 0;
  i4v fb[S][TN], fa[2][2];
  for (int ks = 0; ks < nks; ++ks) {
    // wr: the buffer of stage ks - 1, free since the last barrier.  (Written as additions and compares: `cur == 0 ? NBUF - 1 : cur - 1` made
    // the compiler compute it on the VALU and read it back - v_sub_co_u32 / v_readfirstlane in front of every stage's m0 set-up, +5 % on the
    // S = 5 launches of config 5.)  The four-buffer ring (S <= 4) keeps the other form: with it the workgroups of an XCD stay in step and
    // every operand tile is fetched once per super-tile - rocprofv3 --pmc FETCH_SIZE 1 043 396 KiB on every launch at config 3 against
    // 2 059 220 - 2 108 123 with the additions, 1.414 against 1.444 ms per launch (same box, tools/fetch_ab.sh).
    const int nxt = cur + 1 >= NBUF ? 0 : cur + 1;
    int wr;
    if constexpr (NBUF == 4) {
      wr = cur == 0 ? NBUF - 1 : cur - 1;
    } else {
      wr = cur + (NBUF - 1);
      if (wr >= NBUF) wr -= NBUF;
    }
    const bool spread = I8_DSPREAD && (WN != 4 || work);  // (waves that skip their MFMAs issue their loads at the top)
    const int kpre = ALLON ? min(ks + PD, nks - 1) : ks + PD;  // stage to prefetch (ALLON: clamped, always issued)
    if (!spread && (ALLON || ks + PD < nks)) gl(kpre, wr);
    const unsigned char* bc = lds + ((I8_ABLATE & 2) ? 0 : cur) * STAGE;
    if (WN != 4 || work) {
    const bool rd = !(I8_ABLATE & 2) || ks == 0;
    if (I8_ASMFRAG && TN <= 2 && 2 + S * TN + 2 <= 15 && !(I8_ABLATE & 2)) {   // (lgkmcnt is a 4-bit counter)
      // read order: A_0 (2), B_{S-1} .. B_0 (TN each), A_1 (2); group i >= 1 issues A_{i+1} first.  Outstanding reads before the MFMAs
      // of (0, j): the ones issued after B_j, i.e. TN j + 2; before group i >= 1: the two of A_{i+1} (none for the last group).
      const unsigned aA = (unsigned)(size_t)(lds_ptr_t)(lds) + cur * STAGE + fragA, aB = (unsigned)(size_t)(lds_ptr_t)(lds) + cur * STAGE + fragB;
      fa[0][0] = lds_read_b128<0>(aA);
      fa[0][1] = lds_read_b128<32 * I8_ROWB>(aA);
      [&]<int... J>(std::integer_sequence<int, J...>) {
        ([&] {
          fb[S - 1 - J][0] = lds_read_b128<(S - 1 - J) * ROWS * I8_ROWB>(aB);
          if constexpr (TN == 2) fb[S - 1 - J][TN - 1] = lds_read_b128<(S - 1 - J) * ROWS * I8_ROWB + 32 * I8_ROWB>(aB);
        }(), ...);
      }(std::make_integer_sequence<int, S>{});
      [&]<int... I>(std::integer_sequence<int, I...>) {
        ([&] {
          constexpr int i = I;
          if constexpr (i + 1 < S) {
            fa[(i + 1) & 1][0] = lds_read_b128<(i + 1) * ROWS * I8_ROWB>(aA);
            fa[(i + 1) & 1][1] = lds_read_b128<(i + 1) * ROWS * I8_ROWB + 32 * I8_ROWB>(aA);
          }
          if (spread && (ALLON || ks + PD < nks)) gl1(kpre, wr, i);
          if constexpr (i > 0) wait_lgkm<(i + 1 < S) ? 2 : 0>(fa[i & 1][0], fa[i & 1][1]);
          [&]<int... JJ>(std::integer_sequence<int, JJ...>) {
            ([&] {
              constexpr int j = S - 1 - i - JJ;
              if constexpr (i == 0) {
                if constexpr (TN == 2) wait_lgkm<TN * j + 2>(fa[0][0], fa[0][1], fb[j][0], fb[j][TN - 1]);
                else wait_lgkm<j + 2>(fa[0][0], fa[0][1], fb[j][0]);
              }
#pragma unroll
              for (int b = 0; b < TN; ++b) {
                acc[i + j][0][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[i & 1][0], fb[j][b], acc[i + j][0][b], 0, 0, 0);
                acc[i + j][1][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[i & 1][1], fb[j][b], acc[i + j][1][b], 0, 0, 0);
              }
            }(), ...);
          }(std::make_integer_sequence<int, S - i>{});
        }(), ...);
      }(std::make_integer_sequence<int, S>{});
    } else {
    // issue order pinned (sched_barrier) so that the fragments of product group i+1 are in flight while group i is multiplied
    if (rd) {
#pragma unroll
    for (int a = 0; a < 2; ++a) fa[0][a] = *(const i4v*)(bc + fragA + a * 32 * I8_ROWB);
#pragma unroll
    for (int j = S - 1; j >= 0; --j)
#pragma unroll
      for (int b = 0; b < TN; ++b) fb[j][b] = *(const i4v*)(bc + j * ROWS * I8_ROWB + fragB + b * 32 * I8_ROWB);
    }
#pragma unroll
    for (int i = 0; i < S; ++i) {
      if (i + 1 < S && rd) {
#pragma unroll
        for (int a = 0; a < 2; ++a) fa[(i + 1) & 1][a] = *(const i4v*)(bc + (i + 1) * ROWS * I8_ROWB + fragA + a * 32 * I8_ROWB);
      }
      if (spread && (ALLON || ks + PD < nks)) gl1(kpre, wr, i);
      if (PIN) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = S - 1 - i; j >= 0; --j) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b)
            acc[i + j][a][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[i & 1][a], fb[j][b], acc[i + j][a][b], 0, 0, 0);
        if (PIN && i == 0) __builtin_amdgcn_sched_barrier(0);  // first group: start on b_{S-1} while b_{S-2}.. are still landing
      }
      if (PIN) __builtin_amdgcn_sched_barrier(0);
    }
    }
    }
    if constexpr (NBUF == 3) { if (ALLON || ks + 2 < nks) retire_older(); else wait_vmcnt<0>(); }
    else wait_vmcnt<(PD - 1) * S * NU>();  // stage ks + 1 has landed; the PD - 1 newer ones may be in flight
    if (!(I8_ABLATE & 4)) __builtin_amdgcn_s_barrier();
    cur = nxt;
  }
  if (ALLON) wait_vmcnt<0>();  // (the re-fetches of the last two stages: no LDS-DMA write may outlive the workgroup)
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = cb * BM + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int col = pb * BN + wn * 32 * TN + b * 32 + (lane & 31);
        if constexpr (RAW) {
#pragma unroll
          for (int g = 0; g < S; ++g) epi(row, col, g, acc[g][a][b][r]);
        } else {
          epi(row, col, i8_combine<S>([&](int g) { return acc[g][a][b][r]; }));
        }
      }
}

template <int S, int WN, int TN>
constexpr int i8_lds_bytes() { return i8_nbuf<S, WN, TN>() * S * (I8_BM + 32 * TN * WN) * I8_ROWB; }

// probe / unit-test form (tools/i8_gemm_probe.hip): C[row][col] = sum_g acc_g 2^(-8g)
template <int S, int WN, int TN, int PIN>
__global__ __launch_bounds__(128 * WN) __attribute__((amdgpu_waves_per_eu(WN >= 2 ? WN / 2 : 1, WN >= 4 ? WN / 2 : 2))) void k_gemm_i8_probe(const int8_t* __restrict__ Vs, const int8_t* __restrict__ Zs, int nCp, int NPp, int nks,
                                                            int nC, int NP, double* __restrict__ C) {
  int cb, pb;
  if (!i8_tile_of_block(blockIdx.x, nCp / I8_BM, NPp / (32 * TN * WN), cb, pb)) return;
  gemm_i8_tile<S, WN, TN, PIN>(Vs, Zs, nCp, NPp, nks, nks, cb, pb, nC - cb * I8_BM, NP - pb * 32 * TN * WN,
                          [&](int row, int col, double val) {
#ifdef I8_PROBE_NOSTORE  // (timing only: the epilogue's conversions stay, its stores go)
                            if (val == 1.2345e300)
#endif
                            C[(size_t)row * NPp + col] = val; });
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
  const short* pa;      // pair p = (pa[p], pb[p]), pb <= pa < D, row-major over the lower triangle; p >= NP is padding
  const short* pb;
  const double* scale;  // 2^(e_p - 14): C[c][p] * scale = sum_n v_n z_np
  int NP, NPp;
  const int* cexp;      // leverage pass: column a is used as x_a 2^-cexp[a] (|.| < 1) and G^-1_ab as G^-1_ab 2^(cexp[a]+cexp[b]): exact
                        // power-of-two equilibration, so that x' G^-1 x is cut into bytes in units in which every column counts alike
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

// Vs[s][ks][c][k] = digit s of rint(v 2^(8S)), v in [0, 1/4].  One workgroup per 8 chains, a thread cuts 4 data rows of one
// chain: per stage the 8 chains' bytes are 256 contiguous bytes of every slice plane.  A non-finite v (diverged chain) raises
// vbad[c], which turns the chain's G into NaN in the epilogue exactly as it would be in floating point.  Also writes the identity
// padding of G.
// DELTA (large-D path, see I8Delta below): the planes hold N_old of the previous position iterate; they get the digits of N_new - N_old,
// vexp is only read (the chain's exponent goes to vd.vexp_d, a changed one marks the chain in vd.rebase and its planes get N_new
// itself), max |N_new - N_old| of the launch goes to *vd.dmax.
struct VDelta {
  int* vexp_d;
  int* rebase;
  unsigned long long* dmax;
  int force_rebase;
};
template <int S, bool DELTA = false>
__global__ __launch_bounds__(256) void k_vsplit(const double* __restrict__ vrow, int Mp, int n_chains, const int* __restrict__ phase, int nks,
                                                int nCp, int8_t* __restrict__ Vs, int* __restrict__ vbad, int D, int DP, double inv_alpha,
                                                double* __restrict__ Gq, int* __restrict__ vexp, VDelta vd = VDelta{}) {
  __shared__ int sbad[8];
  __shared__ unsigned long long smax[8];
  __shared__ unsigned long long sdmax;
  if (DELTA && threadIdx.x == 0) sdmax = 0ull;
  const int t = threadIdx.x;
  const int cl = (t >> 3) & 7, k4 = t & 7, ksl = t >> 6;
  const int c = blockIdx.x * 8 + cl;
  const bool live = c < n_chains && phase[min(c, n_chains - 1)] == 1;
  if (t < 8) { sbad[t] = 0; smax[t] = 0ull; }
  __syncthreads();
  // the chain's own exponent for the v grid (see VSlice in kernels.hip.h; here v is at hand, so the exact maximum is used):
  // max_n v_n < 2^(-2 - vsh)
  if (live) {
    const double* v = vrow + (size_t)c * Mp;
    double mx = 0.0;
    for (int ks = ksl; ks < nks; ks += 4) {
      const int n0 = 32 * ks + 4 * k4;
      const double2 v01 = *(const double2*)(v + n0), v23 = *(const double2*)(v + n0 + 2);
      const double m4 = fmax(fmax(v01.x, v01.y), fmax(v23.x, v23.y));
      if (m4 > mx && m4 <= 0.25) mx = m4;  // (non-finite / out-of-range values are flagged below, not scaled for)
    }
    atomicMax(&smax[cl], (unsigned long long)__double_as_longlong(mx));  // v >= 0: the bit patterns order like the values
  }
  __syncthreads();
  int vsh = 0;
  {
    const double mx = __longlong_as_double((long long)smax[cl]);
    int e = 0;
    if (mx > 0.0) { (void)frexp(mx, &e); vsh = min(900, max(0, -2 - e)); }
    if constexpr (DELTA) {
      if (live && k4 == 0 && ksl == 0) { vd.vexp_d[c] = vsh; vd.rebase[c] = (vsh != vexp[c] || vd.force_rebase) ? 1 : 0; }
    } else {
      if (live && k4 == 0 && ksl == 0) vexp[c] = vsh;
    }
  }
  bool rebase = false;
  double dmx = 0.0;
  if constexpr (DELTA) rebase = live && (vsh != vexp[c] || vd.force_rebase);
  int bad = 0;
  if (live) {
    const double* v = vrow + (size_t)c * Mp;
    for (int ks = ksl; ks < nks; ks += 4) {
      const int n0 = 32 * ks + 4 * k4;  // n0 + 3 < 32 nks <= Mp
      const double2 v01 = *(const double2*)(v + n0), v23 = *(const double2*)(v + n0 + 2);
      const double x4[4] = {v01.x, v01.y, v23.x, v23.y};
      int w[S];
#pragma unroll
      for (int s = 0; s < S; ++s) w[s] = 0;
      long long No[4] = {0, 0, 0, 0};
      if constexpr (DELTA) {
        if (!rebase) {
#pragma unroll
          for (int s = 0; s < S; ++s) {  // most significant plane first: N = sum_s digit_s 256^(S-1-s)
            const int wo = *(const int*)(Vs + (((size_t)s * nks + ks) * nCp + c) * 32 + 4 * k4);
#pragma unroll
            for (int k = 0; k < 4; ++k) No[k] = No[k] * 256 + (long long)(signed char)((wo >> (8 * k)) & 0xFF);
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        double x = x4[k];
        if (!(x >= 0.0 && x <= 0.25)) { bad = 1; x = 0.0; }
        int d[S];
        long long N = (long long)rint(ldexp(x, 8 * S + vsh));
        if constexpr (DELTA) {
          N -= No[k];
          dmx = fmax(dmx, fabs((double)N));
        }
        split_digits<S>(N, d);
#pragma unroll
        for (int s = 0; s < S; ++s) w[s] |= (d[S - 1 - s] & 0xFF) << (8 * k);
      }
#pragma unroll
      for (int s = 0; s < S; ++s) *(int*)(Vs + (((size_t)s * nks + ks) * nCp + c) * 32 + 4 * k4) = w[s];
    }
  }
  if (bad) atomicOr(&sbad[cl], 1);
  if (DELTA && dmx > 0.0) atomicMax(&sdmax, (unsigned long long)__double_as_longlong(dmx));
  __syncthreads();
  if (DELTA && t == 0 && sdmax) atomicMax(vd.dmax, sdmax);
  if (t < 8 && blockIdx.x * 8 + t < n_chains && phase[blockIdx.x * 8 + t] == 1) vbad[blockIdx.x * 8 + t] = sbad[t];
  // padding rows D..DP-1 of G (lower triangle): the GEMM writes pairs below D only, and the blocked factorisation of the large-D
  // path runs over all DP rows and uses Gq as a workspace in between, so the identity padding is renewed with every assembly
  if (D < DP)
    for (int q = 0; q < 8; ++q) {
      const int cq = blockIdx.x * 8 + q;
      if (cq >= n_chains || phase[cq] != 1) continue;
      double* G = Gq + (size_t)cq * DP * DP;
      for (int e = t; e < (DP - D) * DP; e += 256) {
        const int r = D + e / DP, cc = e % DP;
        if (cc <= r) G[(size_t)r * DP + cc] = (cc == r) ? inv_alpha : 0.0;
      }
    }
}

// Delta assembly (the evaluation at the end of a leapfrog step, launch_assemble_i8_delta): the planes hold the digits of
// N_new - N_old, G(w_old) is in Gq, and G(w_new) = G(w_old) + sum_n (N_new - N_old)_n z_n 2^-(8S + vexp) is summed from as many of the
// LEAST significant planes as max |N_new - N_old| needs: the instantiations S' = 6, 5, 4 are all launched on the V planes S - S' ..
// S - 1 (and the Z planes 0 .. S' - 1: the same products i + j < S as the full assembly, minus those with a zero V plane), each
// reads the maximum the row pass left in *dsel and all but the one whose S' matches return at once.  cscale = 2^-8(S - S').
// The same for the second position iterate of a step (an inner iterate: its G only steers the next one and is summed from the five
// most significant planes): G(w_2) = G(w_1) + the planes t .. 4 of N_2 - N_1, S' = 5 (t = 0) when the maximum needs all six digits,
// S' = 4 (t = 1) otherwise - which is what 8192 chains at stationarity always get (max |v_2 - v_1| = 2^-11.3 +- 0.35 against 2^-9).
struct I8Delta {
  const unsigned long long* dsel;  // null: ordinary assembly
  const int* rebase;               // chains whose planes hold N_new itself (the exponent of their grid has changed): G is overwritten
  double cscale;
  int need_lo, need_hi;            // this launch does the work when the digits the maximum needs lie in [need_lo, need_hi]
  const double* Gbase;             // the matrix that is added to (null: Gq itself; the large-D path factors Gq in place and keeps a copy)
  __device__ __forceinline__ bool skip() const;
};
// S' balanced digits hold |N| <= 127 (256^S' - 1) / 255 = 0.498 256^S'
__device__ __forceinline__ int i8_delta_slices(unsigned long long bits) {
  const double mx = __longlong_as_double((long long)bits);
  if (mx <= 0.49 * 4294967296.0) return 4;
  if (mx <= 0.49 * 1099511627776.0) return 5;
  return 6;
}

__device__ __forceinline__ bool I8Delta::skip() const {
  if (!dsel) return false;
  const int need = i8_delta_slices(*dsel);
  return need < need_lo || need > need_hi;
}

// the assembly proper: lower triangle of G[c] = C[c][:] * scale + I/alpha, natural row-major DP x DP like k_assemble (the
// Cholesky kernels never read above the diagonal; rmhmc_metric mirrors it on the host for G_out)
// (waves per SIMD stated explicitly: with 4 waves per workgroup the compiler otherwise budgets 256 registers and shuttles
// accumulators through AGPR copies)
template <int S, int WN, int TN>
__device__ __forceinline__ void assemble_i8_body(const int8_t* __restrict__ Vs, const int8_t* __restrict__ Zs, int nCp, int nks_total,
                                                 int ks0, int nk, int accumulate, const I8Pairs& pr, int n_chains, const int* __restrict__ phase,
                                                 const int* __restrict__ vbad, int DP, double inv_alpha, double* __restrict__ Gq,
                                                 size_t plane_stride, const int* __restrict__ vexp, int npb, const I8Delta& dl) {
  int cb, pb;  // npb: pair blocks of this launch (all of them, or the full ones when k_assemble_i8_tail takes the ragged rest)
  if (!i8_tile_of_block(blockIdx.x, nCp / I8_BM, npb, cb, pb)) return;
  // data rows [32 ks0, 32 (ks0 + nk)): long data sets are summed in several launches so that the int32 accumulators cannot overflow.
  // gridDim.y > 1 (small batches, too few tiles to fill the chip): the k range is cut into gridDim.y pieces, piece y writes plane y of
  // Gq (plane_stride apart) and k_sum_planes adds the planes up in a fixed order.
  const int per = (nk + (int)gridDim.y - 1) / (int)gridDim.y;
  const int kb = ks0 + (int)blockIdx.y * per, kn = min(per, ks0 + nk - kb);
  double* __restrict__ Gout = Gq + (size_t)blockIdx.y * plane_stride;
  const bool first = blockIdx.y == 0;
  // What the epilogue needs per chain and per pair of the tile goes to LDS up front, one element per thread and all loads of the
  // workgroup in flight together: read inside the epilogue they were dependent loads (phase -> branch -> scale, ...) that each of
  // the lane's 32 outputs waited for in turn, with nothing else resident on the CU to hide them.
  constexpr int BN = 32 * TN * WN;
  __shared__ double s_cmul[I8_BM];   // chain: 2^-vexp (the chain's v grid is 2^-(8S + vexp)), NaN for a flagged chain
  __shared__ double s_pscale[BN];    // pair: scale, with the sign bit set on the diagonal pairs (scale > 0)
  __shared__ int s_coff[I8_BM];      // chain: 1 when the chain is written
  __shared__ int s_poff[BN];         // pair: a DP + b, -1 for padding
  for (int i = threadIdx.x; i < I8_BM + BN; i += 128 * WN) {
    if (i < I8_BM) {
      const int c = cb * I8_BM + i;
      const bool ok = c < n_chains && phase[min(c, n_chains - 1)] == 1;
      s_coff[i] = ok ? ((dl.rebase && dl.rebase[c]) ? 3 : 1) : 0;  // (3: never added to what Gq holds)
      s_cmul[i] = ok ? (vbad[c] ? __builtin_nan("") : ldexp(dl.cscale, -vexp[c])) : 0.0;
    } else {
      const int j = i - I8_BM, p = pb * BN + j;
      const bool ok = p < pr.NP;
      const int a = ok ? pr.pa[p] : 0, b = ok ? pr.pb[p] : 0;
      s_poff[j] = ok ? a * DP + b : -1;
      s_pscale[j] = ok ? (a == b ? -pr.scale[p] : pr.scale[p]) : 0.0;
    }
  }
  __syncthreads();
  gemm_i8_tile<S, WN, TN, (WN == 4)>(Vs + (size_t)kb * nCp * 32, Zs + (size_t)kb * pr.NPp * 32, nCp, pr.NPp, nks_total, kn, cb, pb,
                                     n_chains - cb * I8_BM, pr.NP - pb * 32 * TN * WN, [&](int c, int p, double val) {
    const int ci = c - cb * I8_BM, pj = p - pb * BN;
    const int off = s_poff[pj];
    const double ps = s_pscale[pj];
    const bool ok = s_coff[ci] != 0 && off >= 0;
    // lower triangle only (pairs run along its rows, contiguous in p): all the factor kernels read
    double* gp = Gout + (size_t)min(c, n_chains - 1) * DP * DP + max(off, 0);
    double g = (val * fabs(ps)) * s_cmul[ci];
    // accumulate bit 0: a later piece of a long data set; bit 1: delta assembly (all chains but the re-based ones)
    if (accumulate & 1) g += *gp;
    else if ((accumulate & 2) && s_coff[ci] != 3) g += dl.Gbase ? dl.Gbase[(size_t)min(c, n_chains - 1) * DP * DP + max(off, 0)] : *gp;
    else if (ps < 0.0 && first) g += inv_alpha;
    if (ok) *gp = g;
  });
}
template <int S, int WN, int TN>
__global__ __launch_bounds__(128 * WN) __attribute__((amdgpu_waves_per_eu(WN / 2, WN / 2))) void k_assemble_i8(const int8_t* __restrict__ Vs, const int8_t* __restrict__ Zs, int nCp, int nks_total,
                                                          int ks0, int nk, int accumulate, I8Pairs pr, int n_chains, const int* __restrict__ phase,
                                                          const int* __restrict__ vbad, int DP, double inv_alpha, double* __restrict__ Gq,
                                                          size_t plane_stride, const int* __restrict__ vexp, int npb, I8Delta dl) {
  if (dl.skip()) return;
  assemble_i8_body<S, WN, TN>(Vs, Zs, nCp, nks_total, ks0, nk, accumulate, pr, n_chains, phase, vbad, DP, inv_alpha, Gq, plane_stride, vexp, npb, dl);
}
// Delta assembly, ONE launch for whatever the difference needs (it used to be three - the 6-, 5- and 4-slice instantiations, two of which
// looked at *dsel and returned: ~4.6 us each, twelve of them per leapfrog step with the tail kernels): the kernel reads the maximum the row
// pass has left and runs the tile of that many slices on the planes seff - S' .. seff - 1 (seff = 6: every digit counts; seff = 5: an inner
// iterate on five-slice accuracy, which needs 5 slices where the difference asks for 6 and 4 otherwise).  Same tile code, same planes,
// same scale as the separate launches: bit-identical G.
__device__ __forceinline__ int i8_delta_pick(const I8Delta& dl, int seff) {
  const int need = i8_delta_slices(*dl.dsel);
  return seff == 6 ? need : (need == 6 ? 5 : 4);
}
template <int WN, int TN>
__global__ __launch_bounds__(128 * WN) __attribute__((amdgpu_waves_per_eu(WN / 2, WN / 2))) void k_assemble_i8_sel(const int8_t* __restrict__ Vs, size_t vplane, int seff, const int8_t* __restrict__ Zs, int nCp, int nks_total,
                                                          int ks0, int nk, int accumulate, I8Pairs pr, int n_chains, const int* __restrict__ phase,
                                                          const int* __restrict__ vbad, int DP, double inv_alpha, double* __restrict__ Gq,
                                                          size_t plane_stride, const int* __restrict__ vexp, int npb, I8Delta dl) {
  const int Sp = i8_delta_pick(dl, seff);
  dl.cscale = __builtin_ldexp(1.0, -8 * (seff - Sp));
  Vs += (size_t)(seff - Sp) * vplane;
  if (Sp == 6) assemble_i8_body<6, WN, TN>(Vs, Zs, nCp, nks_total, ks0, nk, accumulate, pr, n_chains, phase, vbad, DP, inv_alpha, Gq, plane_stride, vexp, npb, dl);
  else if (Sp == 5) assemble_i8_body<5, WN, TN>(Vs, Zs, nCp, nks_total, ks0, nk, accumulate, pr, n_chains, phase, vbad, DP, inv_alpha, Gq, plane_stride, vexp, npb, dl);
  else assemble_i8_body<4, WN, TN>(Vs, Zs, nCp, nks_total, ks0, nk, accumulate, pr, n_chains, phase, vbad, DP, inv_alpha, Gq, plane_stride, vexp, npb, dl);
}

// The pairs beyond the last FULL block of 32 WN pairs (D = 64: 2080 = 16 x 128 + 32) as tiles of their own.  In the main launch
// they made a 17th pair block per chain block whose workgroups, three wave columns idle, still took 0.4 of a full tile's time, after
// the 1024 full tiles had filled the 256 CUs exactly four times over: 2.91 ms against 2.71 ms for 2048 pairs (tools/i8_gemm_probe 70).
// Here: 128 chains x 32 pairs per two-wave workgroup, the k range in gridDim.y pieces so that the launch fills the chip (the pass is
// bound by reading V once, 0.10 ms at config 3), int32 accumulators to Tq[piece][g][chain][32 ntail]; k_assemble_i8_tailsum adds
// the pieces AS INTEGERS and converts once, with the main epilogue's own expression: the G entries are bit-identical to the ones
// the 17th pair block produced.
template <int S>
__device__ __forceinline__ void assemble_i8_tail_body(
    const int8_t* __restrict__ Vs, const int8_t* __restrict__ Zs, int nCp, int NPp, int NP, int nks_total, int ks0, int nk, int n_chains, int pb32_0,
    int ntail, int* __restrict__ Tq) {
  const int nCB = nCp / I8_BM;
  const int cb = blockIdx.x % nCB, tb = blockIdx.x / nCB;
  const int per = (nk + (int)gridDim.y - 1) / (int)gridDim.y;
  const int kb = ks0 + (int)blockIdx.y * per, kn = min(per, ks0 + nk - kb);
  int* __restrict__ T = Tq + (size_t)blockIdx.y * S * nCp * 32 * ntail;
  const int pb32 = pb32_0 + tb;
  if (kn <= 0) {  // (a piece past the end of the range: its slots must still read as zero)
    for (int i = threadIdx.x; i < S * I8_BM * 32; i += 128) {
      const int g = i / (I8_BM * 32), r = i % (I8_BM * 32);
      T[((size_t)g * nCp + cb * I8_BM + (r >> 5)) * 32 * ntail + tb * 32 + (r & 31)] = 0;
    }
    return;
  }
  gemm_i8_tile<S, 1, 1, 0, true>(Vs + (size_t)kb * nCp * 32, Zs + (size_t)kb * NPp * 32, nCp, NPp, nks_total, kn, cb, pb32, n_chains - cb * I8_BM,
                                 NP - pb32 * 32, [&](int c, int p, int g, int a) {
    T[((size_t)g * nCp + c) * 32 * ntail + (p - pb32_0 * 32)] = a;  // (c < nCp, p inside the tail: padding rows / pairs hold zeros)
  });
}
template <int S>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(1, 2))) void k_assemble_i8_tail(
    const int8_t* __restrict__ Vs, const int8_t* __restrict__ Zs, int nCp, int NPp, int NP, int nks_total, int ks0, int nk, int n_chains, int pb32_0,
    int ntail, int* __restrict__ Tq, I8Delta dl) {
  if (dl.skip()) return;
  assemble_i8_tail_body<S>(Vs, Zs, nCp, NPp, NP, nks_total, ks0, nk, n_chains, pb32_0, ntail, Tq);
}
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(1, 2))) void k_assemble_i8_tail_sel(
    const int8_t* __restrict__ Vs, size_t vplane, int seff, const int8_t* __restrict__ Zs, int nCp, int NPp, int NP, int nks_total, int ks0, int nk,
    int n_chains, int pb32_0, int ntail, int* __restrict__ Tq, I8Delta dl) {
  const int Sp = i8_delta_pick(dl, seff);
  Vs += (size_t)(seff - Sp) * vplane;
  if (Sp == 6) assemble_i8_tail_body<6>(Vs, Zs, nCp, NPp, NP, nks_total, ks0, nk, n_chains, pb32_0, ntail, Tq);
  else if (Sp == 5) assemble_i8_tail_body<5>(Vs, Zs, nCp, NPp, NP, nks_total, ks0, nk, n_chains, pb32_0, ntail, Tq);
  else assemble_i8_tail_body<4>(Vs, Zs, nCp, NPp, NP, nks_total, ks0, nk, n_chains, pb32_0, ntail, Tq);
}
template <int S>
__device__ __forceinline__ void assemble_i8_tailsum_body(const int* __restrict__ Tq, int pieces, int nCp, int ntail, int pb32_0, int accumulate,
                                                         const I8Pairs& pr, int n_chains, const int* __restrict__ phase, const int* __restrict__ vbad,
                                                         int DP, double inv_alpha, double* __restrict__ Gq, const int* __restrict__ vexp, const I8Delta& dl) {
  const int W = 32 * ntail;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int c = (int)(i / W), j = (int)(i % W);
  const int p = pb32_0 * 32 + j;
  if (c >= n_chains || p >= pr.NP || phase[c] != 1) return;
  int acc[S];
#pragma unroll
  for (int g = 0; g < S; ++g) acc[g] = 0;
  for (int y = 0; y < pieces; ++y)
#pragma unroll
    for (int g = 0; g < S; ++g) acc[g] += Tq[(((size_t)y * S + g) * nCp + c) * W + j];
  const double val = i8_combine<S>([&](int g) { return acc[g]; });
  const int a = pr.pa[p], b = pr.pb[p];
  double* gp = Gq + (size_t)c * DP * DP + a * DP + b;
  double gv = (val * pr.scale[p]) * (vbad[c] ? __builtin_nan("") : ldexp(dl.cscale, -vexp[c]));
  if (accumulate & 1) gv += *gp;
  else if ((accumulate & 2) && !(dl.rebase && dl.rebase[c])) gv += dl.Gbase ? dl.Gbase[(size_t)c * DP * DP + a * DP + b] : *gp;
  else if (a == b) gv += inv_alpha;
  *gp = gv;
}
template <int S>
__global__ __launch_bounds__(256) void k_assemble_i8_tailsum(const int* __restrict__ Tq, int pieces, int nCp, int ntail, int pb32_0, int accumulate,
                                                             I8Pairs pr, int n_chains, const int* __restrict__ phase, const int* __restrict__ vbad,
                                                             int DP, double inv_alpha, double* __restrict__ Gq, const int* __restrict__ vexp, I8Delta dl) {
  if (dl.skip()) return;
  assemble_i8_tailsum_body<S>(Tq, pieces, nCp, ntail, pb32_0, accumulate, pr, n_chains, phase, vbad, DP, inv_alpha, Gq, vexp, dl);
}
__global__ __launch_bounds__(256) void k_assemble_i8_tailsum_sel(int seff, const int* __restrict__ Tq, int pieces, int nCp, int ntail, int pb32_0, int accumulate,
                                                                 I8Pairs pr, int n_chains, const int* __restrict__ phase, const int* __restrict__ vbad,
                                                                 int DP, double inv_alpha, double* __restrict__ Gq, const int* __restrict__ vexp, I8Delta dl) {
  const int Sp = i8_delta_pick(dl, seff);
  dl.cscale = __builtin_ldexp(1.0, -8 * (seff - Sp));
  if (Sp == 6) assemble_i8_tailsum_body<6>(Tq, pieces, nCp, ntail, pb32_0, accumulate, pr, n_chains, phase, vbad, DP, inv_alpha, Gq, vexp, dl);
  else if (Sp == 5) assemble_i8_tailsum_body<5>(Tq, pieces, nCp, ntail, pb32_0, accumulate, pr, n_chains, phase, vbad, DP, inv_alpha, Gq, vexp, dl);
  else assemble_i8_tailsum_body<4>(Tq, pieces, nCp, ntail, pb32_0, accumulate, pr, n_chains, phase, vbad, DP, inv_alpha, Gq, vexp, dl);
}

// dst[i] = sum over planes of src[plane][i], in plane order (deterministic)
__global__ __launch_bounds__(256) void k_sum_planes(double* __restrict__ dst, const double* __restrict__ src, int planes, size_t plane_stride, size_t count) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  double s = src[i];
  for (int q = 1; q < planes; ++q) s += src[(size_t)q * plane_stride + i];
  dst[i] = s;
}

// ---------------------------------------------------------------------------------------------
// Leverage pass on the same machinery (rmhmc.py:64-77,142-156 through the identity tr_d = sum_n c_n h_n x_nd):
//     h_n = x_n' G^-1 x_n = sum_p Q[chain][p] Z[n][p],   Q_p = G^-1_ab (a = b) or 2 G^-1_ab (a > b)
// is the transposed GEMM: contraction over the column pairs, one output per (chain, data row).  Q is cut per chain with the
// chain's own exponent, Z per data row with the row's exponent (the contraction index cannot carry a scale), so the fixed
// operand is a second sliced copy of x_a x_b:
//     Qs[S][nkp][nCp][32]   Zt[S][nkp][NRp][32]     nkp = ceil(NP/32) stages of 32 pairs, NRp = data rows rounded up to the tile
// The epilogue multiplies by c_n and writes R[chain][n] = c_n h_n; k_trvec contracts R with X on the fp64 matrix cores.
// ---------------------------------------------------------------------------------------------
// e'_n: smallest exponent with max_p |x~_na x~_nb| = (max_a |x~_na|)^2 < 2^e'_n, x~_na = x_na 2^-cexp[a]; zscale[n] = 2^e'_n
__global__ __launch_bounds__(256) void k_zrowmax(const double* __restrict__ Xr, int M, int D, int DP, int NRp, const int* __restrict__ cexp,
                                                 int* __restrict__ ze, double* __restrict__ zscale) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= NRp) return;
  double m = 0.0;
  if (n < M)
    for (int d = 0; d < D; ++d) m = fmax(m, fabs(ldexp(Xr[(size_t)n * DP + d], -cexp[d])));
  m *= m;
  int e = 0;
  if (m > 0.0 && m < 1e300) (void)frexp(m, &e);
  ze[n] = e;
  zscale[n] = ldexp(1.0, e);
}

template <int S>
__global__ __launch_bounds__(256) void k_ztsplit(const double* __restrict__ Xr, int M, int DP, I8Pairs pr, const int* __restrict__ ze,
                                                 int nkp, int NRp, int8_t* __restrict__ Zt) {
  const int n = blockIdx.x * 32 + (threadIdx.x >> 3);  // 8 threads per data row and stage, 4 pairs each
  const int kp = blockIdx.y, k4 = threadIdx.x & 7;
  if (n >= NRp) return;
  int w[S];
#pragma unroll
  for (int s = 0; s < S; ++s) w[s] = 0;
  if (n < M) {
    const int sh = 8 * S - 2 - ze[n];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int p = 32 * kp + 4 * k4 + k;
      double z = 0.0;
      if (p < pr.NP) {
        const int a = pr.pa[p], b = pr.pb[p];
        z = ldexp(Xr[(size_t)n * DP + a] * Xr[(size_t)n * DP + b], -(pr.cexp[a] + pr.cexp[b]));
      }
      int d[S];
      split_digits<S>((long long)rint(ldexp(z, sh)), d);
#pragma unroll
      for (int s = 0; s < S; ++s) w[s] |= (d[S - 1 - s] & 0xFF) << (8 * k);
    }
  }
#pragma unroll
  for (int s = 0; s < S; ++s) *(int*)(Zt + (((size_t)s * nkp + kp) * NRp + n) * 32 + 4 * k4) = w[s];
}

// Q slices of one chain per workgroup.  qscale[c] = 2^(e_c - 12) (so that R = sum_g acc_g 2^-8g * qscale * zscale), NaN when
// G^-1 is not finite (failed factorisation): the trace term, the momentum and the Hamiltonian then become NaN => rejected.
template <int S>
__global__ __launch_bounds__(256) void k_qsplit(const double* __restrict__ Ginv, int DP, I8Pairs pr, const int* __restrict__ phase, int nkp,
                                                int nCp, int8_t* __restrict__ Qs, double* __restrict__ qscale) {
  // One pass over the chain's G^-1: every thread keeps the (at most QMAX x 4) equilibrated entries of its quads in registers between the
  // maximum and the slicing (it used to gather them twice, with a nine-barrier tree reduction in between: 167 us per launch at config 3,
  // 83 % of the wave cycles in s_waitcnt).  More than QMAX x 256 quads (D > 64 never comes here with that many): the second gather again.
  constexpr int QMAX = 3;
  __shared__ double red[4];
  const int c = blockIdx.x, t = threadIdx.x;
  if (phase[c] != 1) return;
  const double* __restrict__ Gi = Ginv + (size_t)c * DP * DP;
  const int nq = nkp * 8;
  double qv[QMAX][4];
  double m = 0.0;
  bool bad = false;
  // entry p of the equilibrated inverse, off-diagonal entries doubled (they count twice in x' G^-1 x); mag: its size before the doubling
  auto entry = [&](int p, double& mag) {
    mag = 0.0;
    if (p >= pr.NP) return 0.0;
    const int a = pr.pa[p], b = pr.pb[p];
    const double q = ldexp(Gi[a * DP + b], pr.cexp[a] + pr.cexp[b]);
    mag = fabs(q);
    return a == b ? q : 2.0 * q;
  };
#pragma unroll
  for (int i = 0; i < QMAX; ++i)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int q4 = t + 256 * i;
      double mag = 0.0;
      qv[i][k] = q4 < nq ? entry(4 * q4 + k, mag) : 0.0;
      bad |= !(mag < 1e300);
      m = fmax(m, mag);
    }
  for (int q4 = t + 256 * QMAX; q4 < nq; q4 += 256)
    for (int k = 0; k < 4; ++k) {
      double mag;
      (void)entry(4 * q4 + k, mag);
      bad |= !(mag < 1e300);
      m = fmax(m, mag);
    }
  if (bad) m = __builtin_inf();
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) m = fmax(m, __shfl_xor(m, o, 64));
  if ((t & 63) == 0) red[t >> 6] = m;
  __syncthreads();
  const double mx = 2.0 * fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));  // off-diagonal entries count twice (as before: the same scale, the same digits)
  const bool ok = mx < 1e300;
  int e = 0;
  if (ok && mx > 0.0) (void)frexp(mx, &e);
  if (t == 0) qscale[c] = ok ? ldexp(1.0, e - 12) : __builtin_nan("");
  const int shf = 8 * S - 2 - e;
  auto put = [&](int q4, const double (&q)[4]) {
    int w[S];
#pragma unroll
    for (int s = 0; s < S; ++s) w[s] = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int d[S];
      split_digits<S>((long long)rint(ldexp(ok ? q[k] : 0.0, shf)), d);
#pragma unroll
      for (int s = 0; s < S; ++s) w[s] |= (d[S - 1 - s] & 0xFF) << (8 * k);
    }
    const int kp = q4 >> 3, k4 = q4 & 7;
#pragma unroll
    for (int s = 0; s < S; ++s) *(int*)(Qs + (((size_t)s * nkp + kp) * nCp + c) * 32 + 4 * k4) = w[s];
  };
#pragma unroll
  for (int i = 0; i < QMAX; ++i)
    if (t + 256 * i < nq) put(t + 256 * i, qv[i]);
  for (int q4 = t + 256 * QMAX; q4 < nq; q4 += 256) {
    double mag;
    const double q[4] = {entry(4 * q4, mag), entry(4 * q4 + 1, mag), entry(4 * q4 + 2, mag), entry(4 * q4 + 3, mag)};
    put(q4, q);
  }
}

// R[c][n] = c_n (x_n' G^-1 x_n)   (mulc = 0: h_n alone, the large-D trace kernel multiplies by c_n itself)
template <int S, int WN, int TN>
__global__ __launch_bounds__(128 * WN) __attribute__((amdgpu_waves_per_eu(WN / 2, WN / 2))) void k_leverage_i8(
    const int8_t* __restrict__ Qs, const int8_t* __restrict__ Zt, int nCp, int NRp, int nkp_total, int kp0, int nk, int accumulate, int mulc,
    int n_chains, int Mp, const int* __restrict__ phase, const double* __restrict__ qscale, const double* __restrict__ zscale,
    const double* __restrict__ crow, double* __restrict__ R, size_t plane_stride) {
  int cb, rb;
  if (!i8_tile_of_block(blockIdx.x, nCp / I8_BM, NRp / (32 * TN * WN), cb, rb)) return;
  const int per = (nk + (int)gridDim.y - 1) / (int)gridDim.y;  // k-split planes as in k_assemble_i8
  const int kb = kp0 + (int)blockIdx.y * per, kn = min(per, kp0 + nk - kb);
  double* __restrict__ Rout = R + (size_t)blockIdx.y * plane_stride;
  // per-chain and per-row factors of the tile to LDS up front (see k_assemble_i8)
  constexpr int BN = 32 * TN * WN;
  __shared__ double s_q[I8_BM], s_z[BN];
  __shared__ int s_ok[I8_BM];
  for (int i = threadIdx.x; i < I8_BM + BN; i += 128 * WN) {
    if (i < I8_BM) {
      const int c = cb * I8_BM + i;
      const bool ok = c < n_chains && phase[min(c, n_chains - 1)] == 1;
      s_ok[i] = ok ? 1 : 0;
      s_q[i] = ok ? qscale[c] : 0.0;
    } else {
      const int n = rb * BN + (i - I8_BM);
      s_z[i - I8_BM] = n < NRp ? zscale[n] : 0.0;
    }
  }
  __syncthreads();
  gemm_i8_tile<S, WN, TN, (WN == 4)>(Qs + (size_t)kb * nCp * 32, Zt + (size_t)kb * NRp * 32, nCp, NRp, nkp_total, kn, cb, rb,
                                     n_chains - cb * I8_BM, Mp - rb * 32 * TN * WN, [&](int c, int n, double val) {
    const int ci = c - cb * I8_BM;
    const bool ok = s_ok[ci] != 0 && n < Mp;
    const size_t o = (size_t)min(c, n_chains - 1) * Mp + min(n, Mp - 1);  // (always a valid address: the loads below are unconditional)
    double r = val * s_q[ci] * s_z[n - rb * BN];
    if (mulc) r *= crow[o];
    if (accumulate) r += Rout[o];
    if (ok) Rout[o] = r;
  });
}
