// kernels.hip.h — gfx950 device code of the RMHMC hot path (included by rmhmc_hip.hip).
//
// Algorithm: matrix-free generalised leapfrog for Bayesian logistic regression,
// following emilemathieu/RiemannHamiltonianMonteCarlo code/rmhmc.py:37-191.
// One chain per wavefront in every per-chain kernel; X is shared by all chains.
//
// Data layout in HBM (all float64):
//   Xr  [Mp][DP]   design matrix, row-major, zero padded (Mp = M rounded up to 64, DP = D up to 16*NB)
//   Xt  [DP][Mp]   its transpose (lane = data row kernels read it coalesced)
//   per chain c: vectors [c][DP], matrices [c][DP][DP], row vectors [c][Mp]
// Padded entries of every vector/matrix are kept at exactly 0.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef double d4 __attribute__((ext_vector_type(4)));

#define RM_LD 66            // LDS leading dimension of the per-chain DxD matrix (even: 16-byte aligned rows for ds_read_b128, conflict free)
#define RM_PI2 6.283185307179586476925286766559
// Packed form of the same LDS image (PK = true in the routines below): only the lower BLOCK triangle of 16 x 16 blocks is kept - the
// Cholesky / solve / inverse routines never touch a block above the diagonal -, row i of block row I = i >> 4 holding its 16 (I + 1)
// columns with leading dimension 16 I + 18 (16-byte aligned rows).  21.5 KB instead of 33.8 KB per matrix: seven chains per CU instead
// of four for the latency-bound per-chain kernels (factor + solve + inverse 1.76 -> 1.52 ms per step at config 3, same box).  Measured
// against it on the same box: leading dimensions 34 / 34 / 66 / 66 (every row in 16-byte slot i mod 16 as with RM_LD: a quarter of the
// bank conflicts, 25.6 KB, six chains per CU) took 1.94 ms - slower than the unpacked image; the smaller image wins despite its
// conflicts.  Reads past a row's end (lanes that run a loop whose result they discard) stay inside the allocation thanks to the 64
// doubles of slack.
#define RM_PK_DOUBLES (2688 + 64)
template <bool PK>
__device__ __forceinline__ int rm_row(int i) {
  if (!PK) return i * RM_LD;
  const int I = i >> 4;
  return (128 * I + 160) * I + (i & 15) * (16 * I + 18);
}
template <bool PK>
__device__ __forceinline__ int rm_len(int i) { return PK ? 16 * ((i >> 4) + 1) : 64; }  // columns of row i that exist

struct DevData {
  const double* Xr;
  const double* Xt;
  const double* t;
  int M, Mp, D, DP, nblk;   // nblk = Mp/64
  double inv_alpha;
  double log_prior_const;   // -0.5*log(2*pi*alpha)
};

// point record: everything the sampler needs at a position w (rmhmc.py:50-77 / :134-156)
struct Rec {
  double *w, *grad, *tr, *L, *Ginv, *ljl, *hld;
};

struct Chains {
  Rec cur, trj;
  double *p, *p0, *Hcur, *Hprop, *tau;
  int *steps_left, *phase, *status, *nsteps_last;
  int* cstale;  // 1: the chain's c tiles (ctile) are not those of trj.w - set when a proposal is rejected (trj falls back to cur), cleared by
                // the row pass of the next evaluation; k_mompass<.., 3> recomputes the tiles of a wave that holds such a chain
  // chains that have just rejected a proposal (k_iter_end appends, k_crestore at the start of the next step recomputes their c tiles
  // and empties the list); null where no c tiles are kept
  int *stale_list, *stale_count;
  // int8 metric path, generic kernels: flags / maxima that the row passes raise with atomics and the assemblies read are cleared by the
  // per-chain kernel that follows every assembly (k_factor_solve / k_factor_full) instead of by a memset node in front of every row pass
  int* i8_vbad;
  unsigned long long* i8_dmax;
  long long *iter, *accepted, *steps_done;
  // scratch
  double *wq, *uq, *PM, *u0, *q, *last, *Gq, *rv0, *rv2, *ljl_part, *qpart, *gpart;
  int n;
};

// ---------------------------------------------------------------------------------------------
// wave helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}
// sum over the 16 lanes that share lane>>4
__device__ __forceinline__ double row16_sum(double x) {
#pragma unroll
  for (int o = 8; o >= 1; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}
// sum over the 4 lanes that share lane&15
__device__ __forceinline__ double col4_sum(double x) {
  x += __shfl_xor(x, 16, 64);
  x += __shfl_xor(x, 32, 64);
  return x;
}

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 counter RNG — same stream specification as oracle/rmhmc_oracle.c:
// key = seed, counter = (chain lo, chain hi, iteration, block); blocks 0..ceil(D/2)-1 -> momentum
// normals (Box-Muller), block 0x40000000 -> (u_len, u_acc), block 0x40000001 -> g_dir.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
    uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
    uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}
__device__ __forceinline__ double u53(uint32_t a, uint32_t b) {
  return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6) + 0.5) * (1.0 / 9007199254740992.0);
}
__device__ __forceinline__ void rng_block(uint64_t seed, uint64_t chain, uint32_t iter, uint32_t block, double& U0, double& U1) {
  uint32_t c[4] = {(uint32_t)chain, (uint32_t)(chain >> 32), iter, block};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  U0 = u53(c[0], c[1]);
  U1 = u53(c[2], c[3]);
}

// ---------------------------------------------------------------------------------------------
// K1  row pass on the matrix cores, 16 chains per wavefront: F = X W (16 data rows x 16 chains per
// MFMA tile, K = D), then the per-(row, chain) scalars the other kernels consume.  Replaces the f/p/v
// blocks of rmhmc.py:51-53,99-100,116-118,134-136,166-168 and the c = v(1-2p) factor of :67,:148.
//   RP_V : out0 = v_n = p(1-p),  p = 1/(1+e^-f),  f = x_n.wq                          (rmhmc.py:116-118)
//   RP_F : out0 = v_n, out2 = c_n, partial sums over the row split of the likelihood gradient
//          X'(t - e^f/(1+e^f)) (rmhmc.py:100,140; a fourth small GEMM whose B operand is the accumulator
//          layout of F, as in k_mompass) and of the log joint sum_n f t - log(1+e^f)   (rmhmc.py:167-168)
// p = 1 / (1 + e^-f) in the naive form, which saturates exactly where the reference does; the log joint and gradient terms follow from
// p (softplus_sigmoid).
// Row assignment: 32-row blocks of two interleaved 16-row MFMA tiles (see the loop below); a lane holds eight consecutive data rows
// of its chain.  k_mompass and k_trvec use the same assignment (the c tiles of k_rowpass<RP_F> are read by k_mompass<.., 2>).
// ---------------------------------------------------------------------------------------------
typedef double d2 __attribute__((ext_vector_type(2)));
// Buffer loads (MUBUF, raw): address = descriptor base + an SGPR offset + a 32-bit lane offset, so the row passes keep ONE loop-invariant
// lane offset per operand stream and advance a scalar - no 64-bit vector address arithmetic in the loops (the global_load form cost a
// v_lshl_add_u64 per load: ~50 per 64 MFMAs in k_mompass).  No bounds (num_records 2^32 - 1): every offset stays below 4 GB of its base
// (rmhmc_create checks Mp * DP * 8; per-wave bases for the c tiles and R).
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
typedef unsigned int u2v __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t brsrc;
__device__ __forceinline__ brsrc buf_rsrc(const void* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, -1, 0x00020000); }
__device__ __forceinline__ d2 buf_d2(brsrc r, unsigned voff, unsigned soff) { return __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0)); }
__device__ __forceinline__ double buf_d1(brsrc r, unsigned voff, unsigned soff) { return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0)); }
__device__ __forceinline__ d4 buf_d4(brsrc r, unsigned voff, unsigned soff) {
  const d2 a = buf_d2(r, voff, soff), b = buf_d2(r, voff + 16u, soff);
  return (d4){a.x, a.y, b.x, b.y};
}
template <int N>
__device__ __forceinline__ void buf_dn(brsrc r, unsigned voff, unsigned soff, double (&x)[N]) {  // N consecutive doubles
#pragma unroll
  for (int i = 0; i + 1 < N; i += 2) { const d2 a = buf_d2(r, voff + 8u * i, soff); x[i] = a.x; x[i + 1] = a.y; }
  if (N & 1) x[N - 1] = buf_d1(r, voff + 8u * (N - 1), soff);
}
// Q[j] (j < S): byte r = balanced base-256 digit j (least significant first) of N_r = rint(v_r 2^(8S + vexp)), r = 0..3.
template <int S>
__device__ __forceinline__ void slice_digits(const d4& vv, double vscale, double vmagic, int vsh, int& bad, unsigned (&Q)[S]) {
  if constexpr (S <= 6) {
    // Fast path (at most 48 bits).  N = sum_j d_j 256^j with balanced digits d_j in [-128, 127]  <=>  N + B, B = sum_j<S 128 256^j, has the
    // UNSIGNED bytes d_j + 128.  One fma puts N + B into the mantissa of a double in [2^52, 2^53) (round to nearest even at ulp 1: the same
    // rounding as rint), so the digits are the mantissa bytes xor 0x80; a 4 x 6 byte transpose (v_perm_b32) gives one dword per slice.
    unsigned L[4], H[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double x = vv[r];
      if (!(x >= 0.0 && x <= 0.25)) { bad = 1; x = 0.0; }
      const double z = fma(x, vscale, vmagic);
      L[r] = (unsigned)__double2loint(z);
      H[r] = (unsigned)__double2hiint(z);
    }
    const unsigned a = __builtin_amdgcn_perm(L[1], L[0], 0x05010400u), b2 = __builtin_amdgcn_perm(L[1], L[0], 0x07030602u);
    const unsigned c = __builtin_amdgcn_perm(L[3], L[2], 0x05010400u), d2_ = __builtin_amdgcn_perm(L[3], L[2], 0x07030602u);
    unsigned Qf[6];
    Qf[0] = __builtin_amdgcn_perm(c, a, 0x05040100u) ^ 0x80808080u;
    Qf[1] = __builtin_amdgcn_perm(c, a, 0x07060302u) ^ 0x80808080u;
    Qf[2] = __builtin_amdgcn_perm(d2_, b2, 0x05040100u) ^ 0x80808080u;
    Qf[3] = __builtin_amdgcn_perm(d2_, b2, 0x07060302u) ^ 0x80808080u;
    const unsigned ah = __builtin_amdgcn_perm(H[1], H[0], 0x05010400u), ch = __builtin_amdgcn_perm(H[3], H[2], 0x05010400u);
    Qf[4] = __builtin_amdgcn_perm(ch, ah, 0x05040100u) ^ 0x80808080u;
    Qf[5] = __builtin_amdgcn_perm(ch, ah, 0x07060302u) ^ 0x80808080u;
#pragma unroll
    for (int j = 0; j < S; ++j) Q[j] = Qf[j];
  } else {
#pragma unroll
    for (int j = 0; j < S; ++j) Q[j] = 0u;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double x = vv[r];
      if (!(x >= 0.0 && x <= 0.25)) { bad = 1; x = 0.0; }
      // N = rint(x 2^(8S)) < 2^(8S-2) as two 32-bit words split at bit 24 (all in exact fp64 / int32 arithmetic: no 64-bit
      // integer emulation): three balanced digits from the low word, the carry and up to four more from the high word
      const double y = rint(ldexp(x, 8 * S + vsh));
      const double yh = floor(y * 5.9604644775390625e-08);  // 2^-24
      int lo = (int)(y - yh * 16777216.0), hi = (int)yh;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int bt = (int)(signed char)(lo & 0xFF);
        Q[j] |= (unsigned)(bt & 0xFF) << (8 * r);
        lo = (lo - bt) >> 8;
      }
      hi += lo;  // carry out of the low word
#pragma unroll
      for (int j = 3; j < S; ++j) {
        const int bt = (int)(signed char)(hi & 0xFF);
        Q[j] |= (unsigned)(bt & 0xFF) << (8 * r);
        hi = (hi - bt) >> 8;
      }
    }
  }
}

// The two halves of the fast path of slice_digits (S = 6) on their own, for the delta slices of k_rowpass<.., DELTA>:
// digits of z = 2^52 + B + N (four values) and back.  Qf[j] byte r = mantissa byte j of z_r, xor 0x80.
__device__ __forceinline__ void z_to_digits6(const double (&z)[4], unsigned (&Qf)[6]) {
  unsigned L[4], H[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { L[r] = (unsigned)__double2loint(z[r]); H[r] = (unsigned)__double2hiint(z[r]); }
  const unsigned a = __builtin_amdgcn_perm(L[1], L[0], 0x05010400u), b2 = __builtin_amdgcn_perm(L[1], L[0], 0x07030602u);
  const unsigned c = __builtin_amdgcn_perm(L[3], L[2], 0x05010400u), d2_ = __builtin_amdgcn_perm(L[3], L[2], 0x07030602u);
  Qf[0] = __builtin_amdgcn_perm(c, a, 0x05040100u) ^ 0x80808080u;
  Qf[1] = __builtin_amdgcn_perm(c, a, 0x07060302u) ^ 0x80808080u;
  Qf[2] = __builtin_amdgcn_perm(d2_, b2, 0x05040100u) ^ 0x80808080u;
  Qf[3] = __builtin_amdgcn_perm(d2_, b2, 0x07060302u) ^ 0x80808080u;
  const unsigned ah = __builtin_amdgcn_perm(H[1], H[0], 0x05010400u), ch = __builtin_amdgcn_perm(H[3], H[2], 0x05010400u);
  Qf[4] = __builtin_amdgcn_perm(ch, ah, 0x05040100u) ^ 0x80808080u;
  Qf[5] = __builtin_amdgcn_perm(ch, ah, 0x07060302u) ^ 0x80808080u;
}
// (the 4 x 4 byte transpose is its own inverse; the upper half of the high word is that of 2^52: N + B < 2^48)
__device__ __forceinline__ void digits6_to_z(const unsigned (&Qd)[6], double (&z)[4]) {
  unsigned Q[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) Q[j] = Qd[j] ^ 0x80808080u;
  const unsigned a = __builtin_amdgcn_perm(Q[1], Q[0], 0x05010400u), b2 = __builtin_amdgcn_perm(Q[1], Q[0], 0x07030602u);
  const unsigned c = __builtin_amdgcn_perm(Q[3], Q[2], 0x05010400u), d2_ = __builtin_amdgcn_perm(Q[3], Q[2], 0x07030602u);
  unsigned L[4], H[4];
  L[0] = __builtin_amdgcn_perm(c, a, 0x05040100u);
  L[1] = __builtin_amdgcn_perm(c, a, 0x07060302u);
  L[2] = __builtin_amdgcn_perm(d2_, b2, 0x05040100u);
  L[3] = __builtin_amdgcn_perm(d2_, b2, 0x07060302u);
  const unsigned ah = __builtin_amdgcn_perm(Q[5], Q[4], 0x05010400u), bh = __builtin_amdgcn_perm(Q[5], Q[4], 0x07030602u);
  H[0] = __builtin_amdgcn_perm(0x43300000u, ah, 0x07060100u);
  H[1] = __builtin_amdgcn_perm(0x43300000u, ah, 0x07060302u);
  H[2] = __builtin_amdgcn_perm(0x43300000u, bh, 0x07060100u);
  H[3] = __builtin_amdgcn_perm(0x43300000u, bh, 0x07060302u);
#pragma unroll
  for (int r = 0; r < 4; ++r) z[r] = __hiloint2double((int)H[r], (int)L[r]);
}

// p = 1 / (1 + e^-f), v = p (1 - p), c = v (1 - 2 p) for the four values of an accumulator tile (rmhmc.py:52-53,67,117-118,135-136,148), written
// as four dependency chains side by side.  The element-wise part of the row passes runs on the fp64 pipe the matrix instructions use, two
// wavefronts per SIMD, so what counts is the instruction count and that a chain's ~8-cycle latencies are filled by the other three:
//   e^-f  = 2^n e^r, n = rint(-f log2 e), r = -f - n ln 2 in two pieces, e^r by its Taylor polynomial to r^13 (|r| <= 0.3466: truncation
//           4e-18), ldexp: 18 instructions; overflow / underflow / NaN fall out of ldexp and the arithmetic (e^-f = inf for f < -709.78,
//           as np.exp gives the reference);
//   1 / d = rcp + two Newton steps (d = 1 + e^-f >= 1; an infinite d is clamped to 2^1000, i.e. p = 2^-1000 where the reference has 0).
// Within one ulp of the library exp and the IEEE quotient it replaces (39 -> 34 instructions per value, no branches).  EVERY generic-path
// kernel that needs p, v or c at a position goes through this function (k_rowpass, k_mompass, k_crestore), so that a c tile is the same
// bits whoever computed it (tests: test_first_momentum_pass_reuses_c_tiles_bit_identically).
__device__ __forceinline__ void sigmoid4(const d4& F, d4& p, d4& v, d4& c) {
  double dn[4], t[4], q[4], x[4], nn[4];
  // (|f| beyond 2^40 - a diverged chain - is clamped: the range reduction below is exact up to |n| < 2^53, and e^-f is 0 or inf long
  //  before; fmin / fmax drop a NaN, so f - f, which is NaN for a non-finite f and 0 otherwise, carries it into d)
#pragma unroll
  for (int r = 0; r < 4; ++r) { x[r] = fmin(fmax(-F[r], -0x1p40), 0x1p40); nn[r] = F[r] - F[r]; }
#pragma unroll
  for (int r = 0; r < 4; ++r) dn[r] = __builtin_rint(x[r] * 0x1.71547652b82fep+0);
#pragma unroll
  for (int r = 0; r < 4; ++r) t[r] = fma(-dn[r], 0x1.62e42fefa39efp-1, x[r]);
#pragma unroll
  for (int r = 0; r < 4; ++r) t[r] = fma(-dn[r], 0x1.abc9e3b39803fp-56, t[r]);
  constexpr double ck[12] = {1.0 / 6227020800.0, 1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0, 1.0 / 40320.0,
                             1.0 / 5040.0,       1.0 / 720.0,       1.0 / 120.0,      1.0 / 24.0,      1.0 / 6.0,      0.5};
#pragma unroll
  for (int r = 0; r < 4; ++r) q[r] = ck[0];
#pragma unroll
  for (int k = 1; k < 12; ++k)
#pragma unroll
    for (int r = 0; r < 4; ++r) q[r] = fma(q[r], t[r], ck[k]);
#pragma unroll
  for (int r = 0; r < 4; ++r) q[r] = fma(q[r], t[r], 1.0);
#pragma unroll
  for (int r = 0; r < 4; ++r) q[r] = fma(q[r], t[r], 1.0);
#pragma unroll
  for (int r = 0; r < 4; ++r) q[r] = ldexp(q[r], (int)dn[r]);            // e^-f
#pragma unroll
  for (int r = 0; r < 4; ++r) q[r] = q[r] > 0x1p1000 ? 0x1p1000 : q[r];   // (a NaN stays a NaN)
  double d[4], rc[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) d[r] = (1.0 + q[r]) + nn[r];
#pragma unroll
  for (int r = 0; r < 4; ++r) rc[r] = __builtin_amdgcn_rcp(d[r]);
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int r = 0; r < 4; ++r) rc[r] = fma(fma(-d[r], rc[r], 1.0), rc[r], rc[r]);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    p[r] = rc[r];
    const double om = 1.0 - rc[r];
    v[r] = rc[r] * om;
    c[r] = v[r] * fma(-2.0, rc[r], 1.0);
  }
}

// log(1 + e^f) and e^f / (1 + e^f) of the log joint and its gradient (rmhmc.py:100,140,167-168) from p = 1 / (1 + e^-f), which the row
// pass has at hand, instead of a second exp, an ocml log and a second divide (148 -> 77 fp64 instructions per data row in k_rowpass<RP_F>,
// whose time is the fp64 VALU's):  e^f / (1 + e^f) = p,  log(1 + e^f) = max(f, 0) - log(y) with y = p (f >= 0) or 1 - p (f < 0), y in
// [1/2, 1]; log(y) = k ln 2 + 2 atanh((m - 1) / (m + 1)) for m = y or 2 y in [sqrt(1/2), sqrt(2)) (|s| <= 0.1716: 11 odd terms to 1e-17).
// Same error class as the naive forms (absolute 1e-16 on the softplus, 2e-16 on the sigmoid), and the same overflow behaviour: where
// e^f overflows (f > 709.78) the reference gets log(inf) = inf and inf / inf = NaN, so do we.
__device__ __forceinline__ void softplus_sigmoid(double f, double p, double& sp, double& sg) {
  const double y = f >= 0.0 ? p : 1.0 - p;
  const bool lo = y < 0.70710678118654752;
  const double m = lo ? 2.0 * y : y;
  const double d = m + 1.0;  // in [1.70, 2.42): plain Newton reciprocal, no scaling
  double rc = __builtin_amdgcn_rcp(d);
  rc = fma(fma(-d, rc, 1.0), rc, rc);
  rc = fma(fma(-d, rc, 1.0), rc, rc);
  const double sn = m - 1.0;  // exact
  double sq = sn * rc;
  sq = fma(fma(-d, sq, sn), rc, sq);  // s = sn / d, correctly rounded up to the last half ulp
  const double s2 = sq * sq;
  double pl = 2.0 / 21.0;
  pl = fma(pl, s2, 2.0 / 19.0); pl = fma(pl, s2, 2.0 / 17.0); pl = fma(pl, s2, 2.0 / 15.0); pl = fma(pl, s2, 2.0 / 13.0);
  pl = fma(pl, s2, 2.0 / 11.0); pl = fma(pl, s2, 2.0 / 9.0); pl = fma(pl, s2, 2.0 / 7.0); pl = fma(pl, s2, 2.0 / 5.0);
  pl = fma(pl, s2, 2.0 / 3.0); pl = fma(pl, s2, 2.0);
  const double logy = fma(sq, pl, lo ? -0.69314718055994531 : 0.0);
  sp = fmax(f, 0.0) - logy;
  sg = p;
  if (f > 709.782712893384) { sp = __builtin_inf(); sg = __builtin_nan(""); }
  if (f != f) sp = f;  // (fmax drops a NaN)
}

enum { RP_V = 0, RP_F = 1, RP_G = 2 };  // RP_G: as RP_F without the v / c row vectors (plain HMC)
__device__ __forceinline__ int rm_perm16(int i) { return 4 * (i & 3) + (i >> 2); }

// int8 metric path (metric_i8.hip.h): the row pass emits v already cut into signed-byte slices, Vs[S][nks][nCp][32], instead of the
// fp64 row vector.  A lane holds four consecutive rows of a 16-row block for its chain (see above), i.e. one dword of every slice plane.
struct VSlice {
  int8_t* Vs;
  int* vbad;   // raised when a v is not in [0, 1/4] (non-finite: diverged chain); cleared by the host before the pass
  int nks, nCp, S;
  // Per-chain exponent of the v grid.  The grid 2^-8S is absolute; a chain saturated on EVERY data row (all |x_n.w| large, which needs an
  // intercept-like column bounded away from 0) would keep only its last byte.  From the column statistics cmin[d] = min_n |x_nd|,
  // cmax[d] = max_n |x_nd|:  |x_n.w| >= max_d |w_d| (cmin_d + cmax_d) - sum_d |w_d| cmax_d =: f_lo for every row, and v <= e^-|f|, so
  // v 2^vexp with vexp = floor(f_lo log2 e) - 3 (>= 0) still fits the grid; the assembly's epilogue divides by 2^vexp.
  int* vexp;
  const double *cmin, *cmax;
  // k_rowpass<.., DELTA> (the evaluation at the end of a leapfrog step): the planes hold N_old = rint(v_old 2^(8S + vexp)) of the last
  // position iterate, whose G is still in Gq.  The pass leaves the slices of N_new - N_old in their place, the chain's exponent in
  // vexp_d (vexp itself is only read) and max |N_new - N_old| over the group in dmax; a chain whose exponent has changed is marked in
  // rebase and gets the slices of N_new itself (its G is overwritten, not added to).  See launch_assemble_i8_delta.
  int* vexp_d = nullptr;
  int* rebase = nullptr;
  unsigned long long* dmax = nullptr;
  int force_rebase = 0;  // (tests)
};
// I8S: 0 = fp64 row vectors, else the number of byte slices (compile time: the slicing code is straight-line)
// (the slicing RP_F variant takes 190 VGPRs = 2 waves per SIMD; asked for 3 / 4 waves it spills: rowpass 2.51 -> 2.99 / 3.57 ms per step)
// CN: RP_F writes c in natural layout to out2 (off on the int8 path when c tiles are kept: its consumers take the tiles)
template <int NB, int MODE, int I8S = 0, bool CN = true, bool DELTA = false>
__global__ __launch_bounds__(256, 2) void k_rowpass(DevData dd, int n_chains, int nsplit, const int* __restrict__ phase,
                                                 const double* __restrict__ wq, double* __restrict__ out0,
                                                 double* __restrict__ out2, double* __restrict__ gpart,
                                                 double* __restrict__ ljl_part, VSlice vs = VSlice{}, d4* __restrict__ ctile = nullptr,
                                                 int* __restrict__ cstale = nullptr) {
  constexpr int DP = 16 * NB;
  constexpr int KK = DP / 4;
  constexpr bool I8 = I8S > 0;
  const int lane = threadIdx.x & 63;
  const int c0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
  if (c0 >= n_chains) return;
  const int split = blockIdx.y;
  const int rr = lane >> 4, ci = lane & 15;
  const int cj = min(c0 + ci, n_chains - 1);
  const bool live = (c0 + ci < n_chains) && (phase[cj] == 1);
  double Wb[KK];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) Wb[kk] = wq[(size_t)cj * DP + 4 * kk + rr];
  int vsh = 0;  // extra binary digits of the chain's v grid (see VSlice)
  if (I8 && MODE != RP_G) {
    double s1 = 0.0, m1 = 0.0;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      const double aw = fabs(Wb[kk]), cx = vs.cmax[4 * kk + rr];
      s1 = fma(aw, cx, s1);
      m1 = fmax(m1, aw * (vs.cmin[4 * kk + rr] + cx));
    }
    s1 = col4_sum(s1);
    m1 = fmax(m1, __shfl_xor(m1, 16, 64));
    m1 = fmax(m1, __shfl_xor(m1, 32, 64));
    const double flo = m1 - s1;
    if (flo > 4.0 && flo < 1e300) vsh = (int)fmin(900.0, floor(flo * 1.4426950408889634) - 3.0);
    if constexpr (DELTA) {
      if (live && rr == 0 && split == 0) { vs.vexp_d[cj] = vsh; vs.rebase[cj] = (vsh != vs.vexp[cj] || vs.force_rebase) ? 1 : 0; }
    } else {
      if (live && rr == 0 && split == 0) vs.vexp[cj] = vsh;
    }
  }
  bool rebase = false;
  double dmx = 0.0;  // DELTA: max |N_new - N_old| of the lane
  if constexpr (DELTA) rebase = vsh != vs.vexp[cj] || vs.force_rebase;
  double vscale = 1.0, vmagic = 4503599627370496.0;  // 2^(8S + vexp) and 2^52 + B (fast slicing, S <= 6)
  if (I8 && MODE != RP_G) {
    vscale = ldexp(1.0, 8 * I8S + vsh);
    double bsum = 0.0;
#pragma unroll
    for (int j = 0; j < I8S && j < 6; ++j) bsum += ldexp(128.0, 8 * j);
    vmagic += bsum;
  }
  // 32-row blocks = two 16-row MFMA tiles "A" and "B" whose rows interleave: lane (rr, ci) holds, for chain ci, data rows
  // n0 + 8 rr + 2 r (tile A, accumulator register r) and n0 + 8 rr + 2 r + 1 (tile B), i.e. EIGHT consecutive rows.  MFMA row i of
  // tile A is data row n0 + 2 rm_perm16(i), of tile B the next one, so ONE 16-byte load per lane brings the F operands of both tiles
  // (16 lanes: 256 contiguous bytes); a slice plane gets 8 bytes per lane = the chain's whole 32-byte stage row from four lanes, 512
  // contiguous bytes per store instruction; row vectors in natural layout move as 64 contiguous bytes per lane.  (With 16-row tiles and
  // 4-byte slice stores the row pass was bound by its memory instructions: without the stores 466 -> 299 us, without the loads -> 327.)
  const int nb16 = dd.Mp / 16, nb32 = dd.Mp / 32;
  const int per = (nb32 + nsplit - 1) / nsplit;
  const int B0 = split * per, B1 = min(nb32, B0 + per);
  // (wave-uniform base + one 32-bit per-lane byte offset: the loads take the scalar-base form, no 64-bit address registers per operand)
  const unsigned xt_off = (unsigned)(rr * dd.Mp + 2 * rm_perm16(ci)) * 8u;  // F operands: X[n0 + 2 perm(ci) + {0,1}][4kk+rr]
  const unsigned xr_off = (unsigned)(8 * rr * DP + NB * ci) * 8u;           // gradient operands: X[n0 + 8rr + k][NB ci + I]
  const double* __restrict__ xt_p = dd.Xt + (size_t)rr * dd.Mp + 2 * rm_perm16(ci);
  const brsrc rXt = buf_rsrc(dd.Xt);
  double lj = 0.0;
  int bad = 0;
  d4 Gr[NB];
#pragma unroll
  for (int I = 0; I < NB; ++I) Gr[I] = (d4){0.0, 0.0, 0.0, 0.0};
  // One register set for the F operands: the next block's are requested as soon as this block's F products have consumed them, and
  // land behind the element-wise work and the gradient products.
  d2 A[KK];
  auto load_a = [&](int B, int k0, int k1) {
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
      if (kk >= k0 && kk < k1) {
        // (buffer loads where they pay: the plain v pass 333 -> 321 us; the delta pass 439 -> 465 and RP_F 761 -> 781 with them, same box)
        if (MODE == RP_V && !DELTA) A[kk] = buf_d2(rXt, xt_off, (unsigned)(4 * kk * dd.Mp + B * 32) * 8u);
        else if (MODE == RP_V) A[kk] = *(const d2*)(xt_p + (size_t)(4 * kk) * dd.Mp + B * 32);  // (registers to spare; the 16 scalar bases would spill SGPRs)
        else A[kk] = *(const d2*)((const char*)(dd.Xt + (size_t)(4 * kk) * dd.Mp + (size_t)B * 32) + xt_off);
      }
  };
  // (the gradient operands stay global loads: as buffer loads they cost RP_F sixteen more spilled registers)
  auto xrow4 = [&](int n, double (&x)[NB]) {  // X[n + 8rr + ..][NB ci .. NB ci + NB - 1]
#pragma unroll
    for (int I = 0; I < NB; ++I) x[I] = *(const double*)((const char*)(dd.Xr + (size_t)n * DP + I) + xr_off);
  };
  // element-wise part of one tile: F -> p, v, c (sigmoid4) and, for RP_F / RP_G, the terms of the log joint and of its gradient
  // (rmhmc.py:100,140,167-168); rows nl + 2 r + h.  log(1 + e^f) = max(f, 0) - log y with y = p (f >= 0) or 1 - p (f < 0) in [1/2, 1], so
  // the lane's share of the log joint is  sum (f t - max(f, 0)) + log(prod y):  one multiplication per row and one log per LJ_FLUSH
  // blocks (the product of 8 LJ_FLUSH factors >= 1/2 cannot underflow) instead of a logarithm per row.  Padded rows (x = 0, t = 0) have
  // f = 0: they add nothing to the sum and a factor 1/2 each to the product, which is taken out again at the end (npad ln 2).  Where
  // e^f overflows (f > 709.78) the reference gets log(1 + e^f) = inf and e^f / (1 + e^f) = NaN: the largest f of the lane's rows is
  // kept and the lane's log-joint and gradient partials are set to -inf / NaN after the loop.
  constexpr int LJ_FLUSH = 64;
  double ly = 1.0, fhi = 0.0;
  auto elem = [&](const d4& F, const d4& tq, d4& vv, d4& cc, d4& rn) {
    d4 pp;
    sigmoid4(F, pp, vv, cc);
    if (MODE != RP_V) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double f = F[r], tn = tq[r];
        const double y = f >= 0.0 ? pp[r] : 1.0 - pp[r];
        ly *= y;
        lj += fma(f, tn, -fmax(f, 0.0));  // (fmax drops a NaN, f t keeps it)
        fhi = fmax(fhi, f);
        rn[r] = tn - pp[r];
      }
    }
  };
  // RP_F / RP_G: w lives in LDS (the lane's own KK values, written and read by the same lane: no barrier), not in 32 registers - with
  // them the kernel needs ~290 registers and spills.  (Round 2 re-read w from global memory through volatile loads, which the compiler
  // serialised: sixteen dependent cache round trips in front of the products of every block, the largest single cost of the pass.)
  __shared__ double s_w[MODE == RP_V ? 1 : 4 * KK * 64];
  double* const my_w = s_w + (MODE == RP_V ? 0 : ((threadIdx.x >> 6) * KK) * 64 + lane);
  if constexpr (MODE != RP_V) {
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) my_w[kk * 64] = Wb[kk];
  }
  if (B0 < B1) load_a(B0, 0, KK);
  for (int Bc = B0; Bc < B1; Bc += LJ_FLUSH) {
  const int Be = min(B1, Bc + LJ_FLUSH);
  for (int B = Bc; B < Be; ++B) {
    const int n0 = 32 * B, nl = n0 + 8 * rr;  // the lane's eight data rows nl .. nl+7
    // DELTA, v pass: the stored digits of this block (six 8-byte loads per lane) are requested in front of the F products instead of where the
    // slicing code needs them (the gradient variant has no registers to hold them that long)
    constexpr bool PREQ = DELTA && I8 && MODE == RP_V;
    uint2 oq[PREQ ? 6 : 1];
    if constexpr (PREQ) {
      const int8_t* vq = vs.Vs + ((size_t)min(B, vs.nks - 1) * vs.nCp + cj) * 32 + 8 * rr;
      const size_t plane = (size_t)vs.nks * vs.nCp * 32;
#pragma unroll
      for (int j = 0; j < 6; ++j) oq[j] = *(const uint2*)(vq + (size_t)(5 - j) * plane);
      __builtin_amdgcn_sched_barrier(0);
    }
    // gradient operands of a tile are requested just before its element-wise work and land behind it; the two tiles are worked off
    // one after the other to keep the live registers of the exp / softplus code low
    double xa[4][NB], xb[4][NB];
    d4 FA = (d4){0.0, 0.0, 0.0, 0.0}, FB = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      double wk;
      if constexpr (MODE != RP_V) wk = my_w[kk * 64];
      else wk = Wb[kk];
      FA = __builtin_amdgcn_mfma_f64_16x16x4f64(A[kk].x, wk, FA, 0, 0, 0);
      FB = __builtin_amdgcn_mfma_f64_16x16x4f64(A[kk].y, wk, FB, 0, 0, 0);
      // delta v pass: the operand pair is re-loaded with the next block's right behind the two products that read it (one load per MFMA pair
      // instead of sixteen in a row after the products; the last block re-loads its own): 437 -> 405 us with the digit prefetch above.  The
      // plain v pass (three wavefronts per SIMD) measured 1 % slower that way and keeps the burst; RP_F / RP_G: in halves behind the tiles' work
      if (MODE == RP_V && DELTA) {
        load_a(B + 1 < B1 ? B + 1 : B, kk, kk + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (MODE == RP_V && !DELTA && B + 1 < B1) load_a(B + 1, 0, KK);
    d4 tA = (d4){0.0, 0.0, 0.0, 0.0}, tB = tA;
    if (MODE != RP_V) {
#pragma unroll
      for (int r = 0; r < 4; ++r) xrow4(n0 + 2 * r, xa[r]);
      const d4 t0 = *(const d4*)(dd.t + nl), t1 = *(const d4*)(dd.t + nl + 4);
      tA = (d4){t0[0], t0[2], t1[0], t1[2]};
      tB = (d4){t0[1], t0[3], t1[1], t1[3]};
    }
    __builtin_amdgcn_sched_barrier(0);
    d4 vA, cA, rnA = (d4){0.0, 0.0, 0.0, 0.0}, vB, cB, rnB = rnA;
    elem(FA, tA, vA, cA, rnA);
    if (MODE != RP_V) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int I = 0; I < NB; ++I) Gr[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[r][I], rnA[r], Gr[I], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (MODE != RP_V) {
#pragma unroll
      for (int r = 0; r < 4; ++r) xrow4(n0 + 2 * r + 1, xb[r]);
      if (B + 1 < B1) load_a(B + 1, 0, KK / 2);
      __builtin_amdgcn_sched_barrier(0);
    }
    elem(FB, tB, vB, cB, rnB);
    if (MODE != RP_V) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int I = 0; I < NB; ++I) Gr[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(xb[r][I], rnB[r], Gr[I], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (B + 1 < B1) load_a(B + 1, KK / 2, KK);
    }
    if (live) {  // row vectors in natural layout: 64 contiguous bytes per lane
      const size_t o = (size_t)cj * dd.Mp + nl;
      if (MODE != RP_G && !I8) {
        *(d4*)(out0 + o) = (d4){vA[0], vB[0], vA[1], vB[1]};
        *(d4*)(out0 + o + 4) = (d4){vA[2], vB[2], vA[3], vB[3]};
      }
      if (MODE == RP_F) {
        if constexpr (CN) {
          *(d4*)(out2 + o) = (d4){cA[0], cB[0], cA[1], cB[1]};
          *(d4*)(out2 + o + 4) = (d4){cA[2], cB[2], cA[3], cB[3]};
        }
        // c in the accumulator layout, for k_mompass<NB, 2> / k_trvec at the same position (same chain group / block / lane / tile mapping)
        if (ctile) {
          ctile[((size_t)(c0 >> 4) * nb16 + 2 * B) * 64 + lane] = cA;
          ctile[((size_t)(c0 >> 4) * nb16 + 2 * B + 1) * 64 + lane] = cB;
        }
      }
    }
    if constexpr (I8 && MODE != RP_G) {
      // QA[j] / QB[j]: byte r = digit j (least significant first) of rint(v 2^(8S + vexp)) for the tile's element r; interleaved
      // they are the 8 bytes of slice plane S-1-j for rows nl .. nl+7.  Only live chains and stages inside the planes are stored.
      unsigned QA[I8S], QB[I8S];
      if constexpr (DELTA) {
        static_assert(!DELTA || I8S == 6, "delta slices: six planes");
        // z = 2^52 + B + N for the new values, the same for the stored ones from their digits; the difference is N_new - N_old exactly
        double zA[4], zB[4], oA[4], oB[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          double xa_ = vA[r], xb_ = vB[r];
          if (!(xa_ >= 0.0 && xa_ <= 0.25)) { bad = 1; xa_ = 0.0; }
          if (!(xb_ >= 0.0 && xb_ <= 0.25)) { bad = 1; xb_ = 0.0; }
          zA[r] = fma(xa_, vscale, vmagic);
          zB[r] = fma(xb_, vscale, vmagic);
          oA[r] = vmagic;
          oB[r] = vmagic;
        }
        if (live && B < vs.nks && !rebase) {
          const int8_t* vp = vs.Vs + ((size_t)B * vs.nCp + cj) * 32 + 8 * rr;
          const size_t plane = (size_t)vs.nks * vs.nCp * 32;
          unsigned OA[6], OB[6];
#pragma unroll
          for (int j = 0; j < 6; ++j) {
            const uint2 q = PREQ ? oq[j] : *(const uint2*)(vp + (size_t)(5 - j) * plane);
            OA[j] = __builtin_amdgcn_perm(q.y, q.x, 0x06040200u);
            OB[j] = __builtin_amdgcn_perm(q.y, q.x, 0x07050301u);
          }
          digits6_to_z(OA, oA);
          digits6_to_z(OB, oB);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double da = zA[r] - oA[r], db = zB[r] - oB[r];
          if (live && B < vs.nks) dmx = fmax(dmx, fmax(fabs(da), fabs(db)));  // (blocks past the planes: padding rows, never stored)
          zA[r] = da + vmagic;
          zB[r] = db + vmagic;
        }
        unsigned QA6[6], QB6[6];
        z_to_digits6(zA, QA6);
        z_to_digits6(zB, QB6);
#pragma unroll
        for (int j = 0; j < I8S; ++j) { QA[j] = QA6[j]; QB[j] = QB6[j]; }
      } else {
        slice_digits<I8S>(vA, vscale, vmagic, vsh, bad, QA);
        slice_digits<I8S>(vB, vscale, vmagic, vsh, bad, QB);
      }
      if (live && B < vs.nks) {
        int8_t* vp = vs.Vs + ((size_t)B * vs.nCp + cj) * 32 + 8 * rr;
        const size_t plane = (size_t)vs.nks * vs.nCp * 32;
#pragma unroll
        for (int j = 0; j < I8S; ++j) {
          uint2 q;
          q.x = __builtin_amdgcn_perm(QB[j], QA[j], 0x05010400u);  // A0 B0 A1 B1
          q.y = __builtin_amdgcn_perm(QB[j], QA[j], 0x07030602u);  // A2 B2 A3 B3
          *(uint2*)(vp + (size_t)(I8S - 1 - j) * plane) = q;
        }
      }
    }
  }
  if (MODE != RP_V) { lj += log(ly); ly = 1.0; }
  }
  if (I8 && MODE != RP_G && bad && live) atomicOr(&vs.vbad[cj], 1);
  if constexpr (DELTA) {
    if (!live) dmx = 0.0;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) dmx = fmax(dmx, __shfl_xor(dmx, o, 64));
    if (lane == 0 && dmx > 0.0) atomicMax(vs.dmax, (unsigned long long)__double_as_longlong(dmx));  // (>= 0: the bit patterns order like the values)
  }
  if (MODE == RP_F && ctile && cstale && live && rr == 0 && split == 0) cstale[cj] = 0;  // (the chain's c tiles are those of this w now)
  if (MODE != RP_V) {
    {  // the lane's padded rows (n >= M) among nl .. nl + 7 of its blocks: a factor 1/2 each in the products
      int npad = 0;
      for (int B = max(B0, (dd.M - 8 * rr - 7) / 32 - 1); B < B1; ++B)
        npad += max(0, min(8, 32 * B + 8 * rr + 8 - dd.M));
      lj = fma((double)npad, 0.69314718055994531, lj);
    }
    // (f > 709.78 on one of the chain's rows in this split: as the reference's overflowing e^f leaves them)
    const double fh = fmax(fmax(fhi, __shfl_xor(fhi, 16, 64)), fmax(__shfl_xor(fhi, 32, 64), __shfl_xor(fhi, 48, 64)));
    const bool ov = fh > 709.782712893384;
    if (ov) {
      lj = -__builtin_inf();
#pragma unroll
      for (int I = 0; I < NB; ++I) Gr[I] = (d4){__builtin_nan(""), __builtin_nan(""), __builtin_nan(""), __builtin_nan("")};
    }
    lj = col4_sum(lj);
    if (live && rr == 0) ljl_part[(size_t)cj * nsplit + split] = lj;
    if (live) {
      double* __restrict__ out = gpart + ((size_t)split * n_chains + c0 + ci) * DP;
#pragma unroll
      for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int d = NB * (rr + 4 * r) + I;
          if (d < dd.D) out[d] = Gr[I][r];
        }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// K2  metric assembly  G = X' diag(v) X + I/alpha  on the fp64 matrix cores (rmhmc.py:57,119,137).
// One chain per wavefront.  v_mfma_f64_16x16x4_f64: A[i][k], B[k][j] one f64 per lane with
// i/j = lane&15, k = lane>>4; D[(lane>>4)+4r][lane&15] in accumulator register r.
// Column permutation: tile index I in [0,NB) and in-tile index i map to column NB*i + I, so a
// lane loads NB *contiguous* doubles of its data row (a wave reads 4 rows = 4*DP*8 contiguous
// bytes per step) and the same registers serve as A (scaled by v_n) and B operands.
// Only tiles I<=J are computed (G is symmetric): NB(NB+1)/2 MFMAs per 4 data rows.
// ---------------------------------------------------------------------------------------------
// gridDim.y > 1 (small batches: fewer wavefronts than SIMDs, each walking all M rows): the rows are cut into gridDim.y ranges, range
// y writes plane y (plane_stride apart) and k_sum_planes adds the planes in a fixed order.
template <int NB>
__global__ __launch_bounds__(256) void k_assemble(DevData dd, int n_chains, const int* __restrict__ phase,
                                                  const double* __restrict__ vrow, double* __restrict__ Gq, size_t plane_stride) {
  constexpr int DP = 16 * NB;
  constexpr int NT = NB * (NB + 1) / 2;
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= n_chains) return;
  if (phase[c] != 1) return;
  const int rr = lane >> 4, ci = lane & 15;
  const double* __restrict__ xp = dd.Xr + (size_t)rr * DP + NB * ci;
  const double* __restrict__ vp = vrow + (size_t)c * dd.Mp + rr;
  d4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
  // Software pipeline over 16-row groups (4 MFMA k-chunks of 4 data rows): the loads of the next group are
  // issued before the 4*NT MFMAs of the current one.  Mp is a multiple of 64, so groups come in pairs.
  double xbA[4][NB], vA[4], xbB[4][NB], vB[4];
  auto load_group = [&](double (&xb)[4][NB], double (&vv)[4], int n1) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int I = 0; I < NB; ++I) xb[q][I] = xp[(size_t)(n1 + 4 * q) * DP + I];
      vv[q] = vp[n1 + 4 * q];
    }
  };
  auto compute_group = [&](const double (&xb)[4][NB], const double (&vv)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      double xa[NB];
#pragma unroll
      for (int I = 0; I < NB; ++I) xa[I] = vv[q] * xb[q][I];
      int t = 0;
#pragma unroll
      for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = I; J < NB; ++J) {
          acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[I], xb[q][J], acc[t], 0, 0, 0);
          ++t;
        }
    }
  };
  const int ng = dd.Mp / 32, per = (ng + (int)gridDim.y - 1) / (int)gridDim.y;
  const int nbeg = 32 * min(ng, (int)blockIdx.y * per), nend = 32 * min(ng, ((int)blockIdx.y + 1) * per);
  if (nbeg < nend) load_group(xbA, vA, nbeg);
  for (int n1 = nbeg; n1 < nend; n1 += 32) {
    load_group(xbB, vB, n1 + 16);
    compute_group(xbA, vA);
    if (n1 + 32 < nend) load_group(xbA, vA, n1 + 32);
    compute_group(xbB, vB);
  }
  // epilogue: scatter the permuted tiles into the natural row-major DPxDP matrix (+ I/alpha)
  double* __restrict__ G = Gq + (size_t)blockIdx.y * plane_stride + (size_t)c * DP * DP;
  int t = 0;
#pragma unroll
  for (int I = 0; I < NB; ++I)
#pragma unroll
    for (int J = I; J < NB; ++J) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = NB * (rr + 4 * r) + I;
        const int col = NB * ci + J;
        double val = acc[t][r];
        if (row == col && blockIdx.y == 0) val += dd.inv_alpha;
        // diagonal tiles hold both triangles (rounded differently): keep the lower one and mirror it,
        // so that G is exactly symmetric
        if (I != J || row >= col) {
          G[row * DP + col] = val;
          G[col * DP + row] = val;
        }
      }
      ++t;
    }
}

// ---------------------------------------------------------------------------------------------
// K2'  EXPERIMENT (RMHMC_FLAG_FP32_METRIC): the same assembly on the fp32 matrix cores, v_mfma_f32_16x16x4_f32 with f32
// operands and accumulators (twice the fp64 rate).  Same operand maps; the f32 result map is row = 4*(lane>>4)+r.
// colA/colB: first column of the 64-wide (or DP-wide) column block of the A and B operands; tiles I<=J only when they
// coincide.  Used for the precision sweep of DESIGN.md, never by default.
// ---------------------------------------------------------------------------------------------
typedef float f4 __attribute__((ext_vector_type(4)));
template <int NB>
__global__ __launch_bounds__(256) void k_assemble_f32(DevData dd, int n_chains, const int* __restrict__ phase,
                                                      const double* __restrict__ vrow, double* __restrict__ Gq, int nbk) {
  const int DP = dd.DP;
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= n_chains) return;
  if (phase[c] != 1) return;
  int bA = 0, bB = 0;
  if (nbk > 1) {  // blockIdx.y = bB(bB+1)/2 + bA
    while ((bB + 1) * (bB + 2) / 2 <= (int)blockIdx.y) ++bB;
    bA = blockIdx.y - bB * (bB + 1) / 2;
  }
  const bool diag = (bA == bB);
  const int WB = 16 * NB;  // column block width
  const int rr = lane >> 4, ci = lane & 15;
  const double* __restrict__ xpA = dd.Xr + (size_t)rr * DP + WB * bA + NB * ci;
  const double* __restrict__ xpB = dd.Xr + (size_t)rr * DP + WB * bB + NB * ci;
  const double* __restrict__ vp = vrow + (size_t)c * dd.Mp + rr;
  f4 acc[NB][NB];
#pragma unroll
  for (int I = 0; I < NB; ++I)
#pragma unroll
    for (int J = 0; J < NB; ++J) acc[I][J] = (f4){0.f, 0.f, 0.f, 0.f};
  for (int n1 = 0; n1 < dd.Mp; n1 += 16)
#pragma unroll
  for (int n0 = n1; n0 < n1 + 16; n0 += 4) {
    float xa[NB], xb[NB];
    const float vv = (float)vp[n0];
#pragma unroll
    for (int I = 0; I < NB; ++I) {
      xa[I] = vv * (float)xpA[(size_t)n0 * DP + I];
      xb[I] = (float)xpB[(size_t)n0 * DP + I];
    }
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
      for (int J = 0; J < NB; ++J)
        if (!diag || J >= I) acc[I][J] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[I], xb[J], acc[I][J], 0, 0, 0);
  }
  double* __restrict__ G = Gq + (size_t)c * DP * DP;
#pragma unroll
  for (int I = 0; I < NB; ++I)
#pragma unroll
    for (int J = 0; J < NB; ++J) {
      if (diag && J < I) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = WB * bA + NB * (4 * rr + r) + I;
        const int col = WB * bB + NB * ci + J;
        double val = (double)acc[I][J][r];
        if (row == col) val += dd.inv_alpha;
        if (!diag || I != J || row >= col) {
          G[(size_t)row * DP + col] = val;
          G[(size_t)col * DP + row] = val;
        }
      }
    }
}

// ---------------------------------------------------------------------------------------------
// K3b  fused quadratic-term pass on the matrix cores, 16 chains per wavefront:
//        q_c[d] = sum_n c_n(w_c) (x_n.u_c)^2 x_nd  = u' dG/dw_d u      (rmhmc.py:104-107,158-161)
// as three small GEMMs per 16 data rows:  F = X W, S = X U  (16 rows x 16 chains, K = D) and
// Q += X' R with R = c(F) S^2.  The accumulator layout of F/S (tile A: register r <-> data row n0 + 8(lane>>4) + 2r,
// tile B: the next row; column lane&15 = chain) is exactly the B-operand layout of the third product for the
// row quadruple r of that tile, so R feeds the matrix core without any lane movement.  X is read once per 16 chains.
// Rows are split over blockIdx.y; partial sums go to qpart[split][chain][d] (summed, in fixed order,
// by the momentum-update kernel: deterministic, no atomics).
// ---------------------------------------------------------------------------------------------
// c_n = v(1-2p) depends on the position only, and the fp64 matrix pipe is shared with the fp64 VALU (the exp / divide of c cost half
// as much again as the pass's 48 MFMAs per tile), so c is computed once per position and kept in the accumulator layout of this
// kernel ("tile native": ctile[(chain group * nb16 + row block) * 64 + lane] = the lane's four values, 2 KB contiguous per tile):
//   CM 0  c from F = X W as above
//   CM 1  the same, and the tile's values are stored                (first momentum fixed-point iteration of a step, rmhmc.py:102-110,
//         for the wavefronts of k_mompass<.., 3> that hold a chain whose tiles are stale)
//   CM 2  c loaded: no F product, no exp                             (the other K-1 iterations - same w -, and the pass of the point
//         evaluation rmhmc.py:158-161, whose row pass k_rowpass<RP_F> has just stored c for the same w in the same layout)
//   TRV   (CM 2, the pass of the point evaluation on the int8 path): the trace term's product tr_d = sum_n c_n h_n x_nd (rmhmc.py:148-156 through
//         the leverages h the int8 GEMM has left in R) rides along - the same X' R form over the same rows and the same c tiles, so the
//         operands and the tiles are read once for both products (it used to be a kernel of its own, k_trvec: 1.3 GB of R and c tiles per
//         launch, 0.36 ms).  u then lives in LDS to make room for the second accumulator set.
template <int NB, int CM, bool TRV = false>
__device__ __forceinline__ void mompass_body(const DevData& dd, int n_chains, int nsplit, const double* __restrict__ wq,
                                             const double* __restrict__ uq, double* __restrict__ qpart, d4* __restrict__ ctile,
                                             const double* __restrict__ R = nullptr, double* __restrict__ trpart = nullptr) {
  static_assert(!TRV || CM == 2, "the trace product rides on the pass that loads its c tiles");
  constexpr int DP = 16 * NB;
  constexpr int KK = DP / 4;
  const int lane = threadIdx.x & 63;
  const int c0 = (blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))) * 16;
  if (c0 >= n_chains) return;
  const int split = blockIdx.y;
  const int rr = lane >> 4, ci = lane & 15;
  const int cj = min(c0 + ci, n_chains - 1);
  double Wb[CM == 2 ? 1 : KK], Ub[KK];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) {
    if (CM != 2) Wb[kk] = wq[(size_t)cj * DP + 4 * kk + rr];
    Ub[kk] = uq[(size_t)cj * DP + 4 * kk + rr];
  }
  __shared__ double s_u[TRV ? 4 * KK * 64 : 1];  // (TRV: the lane's own u values, written and read by the same lane: no barrier)
  double* const my_u = s_u + (TRV ? ((threadIdx.x >> 6) * KK) * 64 + lane : 0);
  if constexpr (TRV) {
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) my_u[kk * 64] = Ub[kk];
  }
  const brsrc rXt = buf_rsrc(dd.Xt), rXr = buf_rsrc(dd.Xr), rR = buf_rsrc(TRV ? R + (size_t)c0 * dd.Mp : nullptr);
  const unsigned r_off = (unsigned)((cj - c0) * dd.Mp + 8 * rr) * 8u;  // the lane's eight h values of a block: 64 contiguous bytes
  d4 T[TRV ? NB : 1];
  if constexpr (TRV) {
#pragma unroll
    for (int I = 0; I < NB; ++I) T[I] = (d4){0.0, 0.0, 0.0, 0.0};
  }
  // 32-row blocks of two interleaved 16-row tiles A / B, exactly as in k_rowpass (whose c tiles CM = 2 reads): lane (rr, ci) holds data
  // rows n0 + 8 rr + 2 r (A) and n0 + 8 rr + 2 r + 1 (B); one 16-byte load per lane brings the F / S operands of both tiles.
  const int nb16 = dd.Mp / 16, nb32 = dd.Mp / 32;
  const int per = (nb32 + nsplit - 1) / nsplit;
  const int B0 = split * per, B1 = min(nb32, B0 + per);
  d4* __restrict__ ct = ctile + (size_t)(c0 >> 4) * nb16 * 64 + lane;
  const brsrc rC = buf_rsrc(ctile + (size_t)(c0 >> 4) * nb16 * 64);
  const unsigned c_off = (unsigned)lane * 32u;
  auto load_c = [&](int t) { return buf_d4(rC, c_off, (unsigned)t * 2048u); };  // tile t of this chain group
  d4 Q[NB];
#pragma unroll
  for (int I = 0; I < NB; ++I) Q[I] = (d4){0.0, 0.0, 0.0, 0.0};
  const unsigned xt_off = (unsigned)(rr * dd.Mp + 2 * rm_perm16(ci)) * 8u;  // A of F,S: X[n0 + 2 perm(ci) + {0,1}][4kk+rr]  (scalar base +
  const unsigned xr_off = (unsigned)(8 * rr * DP + NB * ci) * 8u;           // A of Q  : X[n0 + 8rr + k][NB*ci+I]            32-bit lane offset)
  // One register set for the F / S operands: the next block's (and its c tiles) are requested as soon as this block's F / S products
  // have consumed them and land behind the 32 MFMAs of the Q products.  (Left to itself the compiler keeps two operand registers in
  // flight and puts an s_waitcnt vmcnt(1) in front of every MFMA: 59 % MFMA busy.)
  d2 A[KK];
  d4 ccA = (d4){0.0, 0.0, 0.0, 0.0}, ccB = ccA;
  auto load_a = [&](int B, int k0, int k1) {
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
      if (kk >= k0 && kk < k1) A[kk] = buf_d2(rXt, xt_off, (unsigned)(4 * kk * dd.Mp + B * 32) * 8u);
  };
  auto xrow4 = [&](int n, double (&x)[NB]) { buf_dn<NB>(rXr, xr_off, (unsigned)(n * DP) * 8u, x); };  // X[n + 8rr + ..][NB ci .. NB ci + NB - 1]
  if constexpr (CM == 2) {
    // CM 2 (no element-wise work but the products c S^2): every operand register is re-loaded with the NEXT block's value right behind the
    // matrix instruction that read it last - one or two loads after every two to four MFMAs, each with a whole block (64 MFMAs) to land -
    // instead of bursts of 8 - 16 loads between the product groups: with the bursts the 36 loads of a block cost ~20 cycles of matrix
    // pipe each (both wavefronts of a SIMD in their load phase together; ablation timings in profiles/r03_notes.txt).  The last block
    // re-loads its own operands (no branch around the loads).  Same products in the same order as CM 0 / 1: the same bits.
    double xa[4][NB], xb[4][NB];
    d4 h0 = (d4){0.0, 0.0, 0.0, 0.0}, h1 = h0;
    if (B0 < B1) {
      load_a(B0, 0, KK);
      ccA = load_c(2 * B0);
      ccB = load_c(2 * B0 + 1);
#pragma unroll
      for (int r = 0; r < 4; ++r) xrow4(32 * B0 + 2 * r, xa[r]);
#pragma unroll
      for (int r = 0; r < 4; ++r) xrow4(32 * B0 + 2 * r + 1, xb[r]);
    }
    for (int B = B0; B < B1; ++B) {
      const int Bn = B + 1 < B1 ? B + 1 : B, n1 = 32 * Bn;
      d4 SA = (d4){0.0, 0.0, 0.0, 0.0}, SB = SA;
      // (TRV: the leverages of THIS block, requested in front of its 32 S products - their registers are free during the Q / T products)
      if constexpr (TRV) { h0 = buf_d4(rR, r_off, (unsigned)(32 * B) * 8u); h1 = buf_d4(rR, r_off + 32u, (unsigned)(32 * B) * 8u); }
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        const double uk = TRV ? my_u[kk * 64] : Ub[kk];
        SA = __builtin_amdgcn_mfma_f64_16x16x4f64(A[kk].x, uk, SA, 0, 0, 0);
        SB = __builtin_amdgcn_mfma_f64_16x16x4f64(A[kk].y, uk, SB, 0, 0, 0);
        A[kk] = buf_d2(rXt, xt_off, (unsigned)(4 * kk * dd.Mp + Bn * 32) * 8u);
        __builtin_amdgcn_sched_barrier(0);
      }
      double RA[4], RB[4], TA[TRV ? 4 : 1], TB[TRV ? 4 : 1];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        RA[r] = ccA[r] * SA[r] * SA[r];
        RB[r] = ccB[r] * SB[r] * SB[r];
        if constexpr (TRV) {
          TA[r] = ccA[r] * (r < 2 ? h0[2 * r] : h1[2 * r - 4]);      // tile A: rows nl + 2 r
          TB[r] = ccB[r] * (r < 2 ? h0[2 * r + 1] : h1[2 * r - 3]);  // tile B: rows nl + 2 r + 1
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      ccA = load_c(2 * Bn);
      ccB = load_c(2 * Bn + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int I = 0; I < NB; ++I) Q[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[r][I], RA[r], Q[I], 0, 0, 0);
        if constexpr (TRV) {
#pragma unroll
          for (int I = 0; I < NB; ++I) T[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[r][I], TA[r], T[I], 0, 0, 0);
        }
        xrow4(n1 + 2 * r, xa[r]);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int I = 0; I < NB; ++I) Q[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(xb[r][I], RB[r], Q[I], 0, 0, 0);
        if constexpr (TRV) {
#pragma unroll
          for (int I = 0; I < NB; ++I) T[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(xb[r][I], TB[r], T[I], 0, 0, 0);
        }
        xrow4(n1 + 2 * r + 1, xb[r]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else {
  if (B0 < B1) {
    load_a(B0, 0, KK);
  }
  for (int B = B0; B < B1; ++B) {
    const int n0 = 32 * B;
    // Q operands of tile A now (they land behind the F / S products), of tile B behind those products (they land behind tile A's Q
    // products)
    double xa[4][NB], xb[4][NB];
    if (CM == 2) {
#pragma unroll
      for (int r = 0; r < 4; ++r) xrow4(n0 + 2 * r, xa[r]);
      __builtin_amdgcn_sched_barrier(0);
    }
    d4 FA = (d4){0.0, 0.0, 0.0, 0.0}, FB = FA, SA = FA, SB = FA;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      if (CM != 2) {
        FA = __builtin_amdgcn_mfma_f64_16x16x4f64(A[kk].x, Wb[kk], FA, 0, 0, 0);
        FB = __builtin_amdgcn_mfma_f64_16x16x4f64(A[kk].y, Wb[kk], FB, 0, 0, 0);
      }
      const double uk = TRV ? my_u[kk * 64] : Ub[kk];
      SA = __builtin_amdgcn_mfma_f64_16x16x4f64(A[kk].x, uk, SA, 0, 0, 0);
      SB = __builtin_amdgcn_mfma_f64_16x16x4f64(A[kk].y, uk, SB, 0, 0, 0);
    }
    d4 cA = ccA, cB = ccB;
    d4 h0 = (d4){0.0, 0.0, 0.0, 0.0}, h1 = h0;
    if constexpr (TRV) { h0 = buf_d4(rR, r_off, (unsigned)n0 * 8u); h1 = buf_d4(rR, r_off + 32u, (unsigned)n0 * 8u); }  // rows nl .. nl+3, nl+4 .. nl+7
    __builtin_amdgcn_sched_barrier(0);
    // (with the exp of c in the pass - CM 0 / 1 - the Q operands of a tile are requested just before its c, to stay within 256 registers)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (CM == 2) xrow4(n0 + 2 * r + 1, xb[r]);
      else xrow4(n0 + 2 * r, xa[r]);
    }
    if (B + 1 < B1) load_a(B + 1, 0, KK / 2);  // (the other half once tile A's Q operands are used up: registers)
    __builtin_amdgcn_sched_barrier(0);
    if (CM != 2) {
      d4 pa, va;
      sigmoid4(FA, pa, va, cA);
      if (CM == 1) ct[(size_t)(2 * B) * 64] = cA;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double RA = cA[r] * SA[r] * SA[r];
#pragma unroll
      for (int I = 0; I < NB; ++I) Q[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[r][I], RA, Q[I], 0, 0, 0);
      if constexpr (TRV) {
        const double TA = cA[r] * (r < 2 ? h0[2 * r] : h1[2 * r - 4]);  // tile A: rows nl + 2 r
#pragma unroll
        for (int I = 0; I < NB; ++I) T[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[r][I], TA, T[I], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (CM != 2) {
#pragma unroll
      for (int r = 0; r < 4; ++r) xrow4(n0 + 2 * r + 1, xb[r]);
    }
    if (B + 1 < B1) {
      load_a(B + 1, KK / 2, KK);
      if (CM == 2) { ccA = load_c(2 * B + 2); ccB = load_c(2 * B + 3); }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (CM != 2) {
      d4 pb, vb;
      sigmoid4(FB, pb, vb, cB);
      if (CM == 1) ct[(size_t)(2 * B + 1) * 64] = cB;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double RB = cB[r] * SB[r] * SB[r];
#pragma unroll
      for (int I = 0; I < NB; ++I) Q[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(xb[r][I], RB, Q[I], 0, 0, 0);
      if constexpr (TRV) {
        const double TB = cB[r] * (r < 2 ? h0[2 * r + 1] : h1[2 * r - 3]);  // tile B: rows nl + 2 r + 1
#pragma unroll
        for (int I = 0; I < NB; ++I) T[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(xb[r][I], TB, T[I], 0, 0, 0);
      }
    }
  }
  }
  if (c0 + ci < n_chains) {
    double* __restrict__ out = qpart + ((size_t)split * n_chains + c0 + ci) * DP;
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int d = NB * (rr + 4 * r) + I;
        if (d < dd.D) out[d] = Q[I][r];
      }
    if constexpr (TRV) {  // (the layout k_trvec writes: summed over the row splits by k_reduce_tr)
      double* __restrict__ tout = trpart + ((size_t)split * n_chains + c0 + ci) * DP;
#pragma unroll
      for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int d = NB * (rr + 4 * r) + I;
          if (d < dd.D) tout[d] = T[I][r];
        }
    }
  }
}
//   CM 3  per wavefront: CM 2 when the c tiles of all its 16 chains are those of w (Chains::cstale), CM 1 otherwise.  The first momentum
//         iteration of a step: the evaluation that ended the previous step left the tiles of w behind for every chain that did not
//         just reject a proposal (both routes produce the same bits: same operands, same operation order).
template <int NB>
__global__ __launch_bounds__(256, 2) void k_mompass_trv(DevData dd, int n_chains, int nsplit, const double* __restrict__ uq, double* __restrict__ qpart,
                                                     d4* __restrict__ ctile, const double* __restrict__ R, double* __restrict__ trpart) {
  mompass_body<NB, 2, true>(dd, n_chains, nsplit, nullptr, uq, qpart, ctile, R, trpart);
}
template <int NB, int CM>
__global__ __launch_bounds__(256, 2) void k_mompass(DevData dd, int n_chains, int nsplit, const double* __restrict__ wq,
                                                 const double* __restrict__ uq, double* __restrict__ qpart, d4* __restrict__ ctile,
                                                 const int* __restrict__ cstale = nullptr) {
  if constexpr (CM == 3) {
    const int c0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
    if (c0 >= n_chains) return;
    const int c = c0 + (threadIdx.x & 15);
    const bool stale = __builtin_amdgcn_ballot_w64(c < n_chains && cstale[min(c, n_chains - 1)] != 0) != 0ull;
    if (stale) mompass_body<NB, 1>(dd, n_chains, nsplit, wq, uq, qpart, ctile);
    else mompass_body<NB, 2>(dd, n_chains, nsplit, wq, uq, qpart, ctile);
  } else {
    mompass_body<NB, CM>(dd, n_chains, nsplit, wq, uq, qpart, ctile);
  }
}

// c tiles of the chains that have just rejected a proposal (Chains::stale_list), 16 of them to a wavefront whatever their place in the
// batch.  k_mompass<.., 3> recomputes the tiles of every wavefront that holds ONE stale chain: with 3.7 % of the chains stale at a step
// that is 45 % of its wavefronts (617 us against 460 with none stale); here the same products and the same element-wise expression (same
// bits) run for the ~300 stale chains alone - ~19 wavefront groups, the rows cut into gridDim.y pieces of their own so that so few
// wavefronts still finish in microseconds - and the pass that follows finds every flag cleared.
template <int NB>
__global__ __launch_bounds__(256, 2) void k_crestore(DevData dd, int n_chains, const int* __restrict__ phase,
                                                  const double* __restrict__ wq, d4* __restrict__ ctile, int* __restrict__ cstale,
                                                  const int* __restrict__ list, const int* __restrict__ count) {
  constexpr int DP = 16 * NB;
  constexpr int KK = DP / 4;
  const int lane = threadIdx.x & 63;
  const int cnt = min(*count, n_chains);
  const int split = blockIdx.y, nsplit = gridDim.y;
  const int rr = lane >> 4, ci = lane & 15;
  // a SMALL grid whose wavefronts walk the list (gridDim.x x 64 chains at a time): a grid with room for every rejection the batch could have
  // spent its time launching workgroups that find nothing to do (32 x 16 workgroups 75 us, 8 x 16 43 us at ~300 listed chains)
  for (int g16 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 16; g16 < cnt; g16 += gridDim.x * 64) {
  const int chn = list[min(g16 + ci, cnt - 1)];
  const int cj = min(max(chn, 0), n_chains - 1);
  const bool live = g16 + ci < cnt && chn >= 0 && chn < n_chains && phase[cj] == 1;
  double Wb[KK];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) Wb[kk] = wq[(size_t)cj * DP + 4 * kk + rr];
  const int nb16 = dd.Mp / 16, nb32 = dd.Mp / 32;
  const int per = (nb32 + nsplit - 1) / nsplit;
  const int B0 = split * per, B1 = min(nb32, B0 + per);
  const double* __restrict__ xt_p = dd.Xt + (size_t)rr * dd.Mp + 2 * rm_perm16(ci);
  // the chain's slot in its own 16-chain tile group: lane (rr, chain & 15)
  d4* __restrict__ ct = ctile + (size_t)(cj >> 4) * nb16 * 64 + rr * 16 + (cj & 15);
  // (all sixteen operand loads of a block in flight together, the next block's behind this block's element-wise work: left to the
  //  compiler the loop was one load - one product pair at a time, 40 registers and ~20 us per block)
  d2 A[KK];
  auto load_a = [&](int B) {
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) A[kk] = *(const d2*)(xt_p + (size_t)(4 * kk) * dd.Mp + B * 32);
  };
  if (B0 < B1) load_a(B0);
  for (int B = B0; B < B1; ++B) {
    d4 FA = (d4){0.0, 0.0, 0.0, 0.0}, FB = FA;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      FA = __builtin_amdgcn_mfma_f64_16x16x4f64(A[kk].x, Wb[kk], FA, 0, 0, 0);
      FB = __builtin_amdgcn_mfma_f64_16x16x4f64(A[kk].y, Wb[kk], FB, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (B + 1 < B1) load_a(B + 1);
    __builtin_amdgcn_sched_barrier(0);
    d4 cA, cB, pa, va, pb, vb;
    sigmoid4(FA, pa, va, cA);  // (the function of mompass_body / k_rowpass: the same bits)
    sigmoid4(FB, pb, vb, cB);
    if (live) {
      ct[(size_t)(2 * B) * 64] = cA;
      ct[(size_t)(2 * B + 1) * 64] = cB;
    }
  }
  if (live && rr == 0 && split == 0) cstale[cj] = 0;
  }
}

// ---------------------------------------------------------------------------------------------
// K4  leverage pass on the matrix cores: h_n = x_n' G^-1 x_n for every data row, then
//       tr_d = sum_n c_n h_n x_nd          (= tr(G^-1 dG/dw_d),      rmhmc.py:67-77,148-156)
// The transposed product Y' = Ghat' X' is computed (Ghat = block-upper-triangular fold of the symmetric
// G^-1 with off-diagonal blocks doubled, so only NB(NB+1)/2 blocks are multiplied; blocks use the column
// permutation of k_assemble: block I = columns NB*m + I).  Per 16 data rows:
//   B operand  X[n0+(lane&15)][NB*(4s+(lane>>4)) + I]          (s = 0..3)   -- the only copy of X loaded
//   A operand  Ghat[NB*(4s+(lane>>4)) + I][NB*(lane&15) + J]                 -- chain constant, 80 VGPRs
//   result     Y'_J[r] = Y[n0+(lane&15)][NB*((lane>>4)+4r) + J]
// i.e. accumulator register r of tile J sits in the lane that holds x at the very same (row, column) in
// operand register (s = r, I = J): h is a per-lane dot product plus a 4-lane sum, and the trace
// accumulation re-uses the operand registers.
// ---------------------------------------------------------------------------------------------
template <int NB>
__global__ __launch_bounds__(256) void k_leverage(DevData dd, int n_chains, const int* __restrict__ phase,
                                                  const double* __restrict__ Ginv, const double* __restrict__ crow,
                                                  double* __restrict__ tr, double* __restrict__ trpart) {
  // gridDim.y > 1 (small batches, as in k_assemble): row range y writes its partial trace term to trpart[y][chain][d], summed in a
  // fixed order by k_reduce_tr
  constexpr int DP = 16 * NB;
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= n_chains) return;
  if (phase[c] != 1) return;
  const int rr = lane >> 4, ci = lane & 15;
  const double* __restrict__ Gi = Ginv + (size_t)c * DP * DP;
  double Gv[NB][4][NB];
#pragma unroll
  for (int I = 0; I < NB; ++I)
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int J = 0; J < NB; ++J) {
        const int row = NB * (4 * s + rr) + I, col = NB * ci + J;
        Gv[I][s][J] = (J >= I) ? Gi[row * DP + col] * (I == J ? 1.0 : 2.0) : 0.0;
      }
  const double* __restrict__ xp = dd.Xr + (size_t)ci * DP + NB * rr;  // row n0+ci, column NB*(4s+rr)+I
  const double* __restrict__ cp = crow + (size_t)c * dd.Mp + ci;
  double tracc[4][NB];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int J = 0; J < NB; ++J) tracc[r][J] = 0.0;

  // software pipeline over 16-row blocks (Mp is a multiple of 64, so blocks come in pairs): the operand
  // loads of the next block are issued before the MFMAs of the current one
  double XA[4][NB], XB[4][NB], cA, cB;
  auto load_block = [&](double (&X)[4][NB], double& cn, int n0) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int I = 0; I < NB; ++I) X[s][I] = xp[(size_t)n0 * DP + NB * 4 * s + I];
    cn = cp[n0];
  };
  auto compute_block = [&](const double (&X)[4][NB], double cn) {
    d4 Y[NB];
#pragma unroll
    for (int J = 0; J < NB; ++J) {
      Y[J] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int I = 0; I <= J; ++I)
#pragma unroll
        for (int s = 0; s < 4; ++s) Y[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(Gv[I][s][J], X[s][I], Y[J], 0, 0, 0);
    }
    double hp = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int J = 0; J < NB; ++J) hp = fma(Y[J][r], X[r][J], hp);
    const double ch = cn * col4_sum(hp);  // leverage of row n0+ci, times c
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int J = 0; J < NB; ++J) tracc[r][J] = fma(ch, X[r][J], tracc[r][J]);
  };
  const int ng = dd.Mp / 32, per = (ng + (int)gridDim.y - 1) / (int)gridDim.y;
  const int nbeg = 32 * min(ng, (int)blockIdx.y * per), nend = 32 * min(ng, ((int)blockIdx.y + 1) * per);
  if (nbeg < nend) load_block(XA, cA, nbeg);
  for (int n0 = nbeg; n0 < nend; n0 += 32) {
    load_block(XB, cB, n0 + 16);
    compute_block(XA, cA);
    if (n0 + 32 < nend) load_block(XA, cA, n0 + 32);
    compute_block(XB, cB);
  }
  double* __restrict__ out = gridDim.y > 1 ? trpart + ((size_t)blockIdx.y * n_chains + c) * DP : tr + (size_t)c * DP;
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int J = 0; J < NB; ++J) {
      const double a = row16_sum(tracc[r][J]);
      const int d = NB * (4 * r + rr) + J;
      if (ci == 0 && d < dd.D) out[d] = a;
    }
}

// ---------------------------------------------------------------------------------------------
// small dense kernels: one chain per wavefront (64 threads per block), DxD matrix in LDS.
// These are latency bound, so every inner loop is unrolled by 8 with all its LDS reads (ds_read_b128,
// row stride 66 doubles: conflict free) issued before the dependent FMA chain, cross-lane broadcasts use
// v_readlane (the index is wave uniform) instead of ds_bpermute, and divisions by the diagonal are
// replaced by one reciprocal square root per column.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double rdlane(double x, int k) {  // k must be wave uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), k);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(x), k);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double2 lds2(const double* p) { return *reinterpret_cast<const double2*>(p); }

// acc -= sum_{k<n} a[k]*b[k]  (a, b: LDS rows, 16-byte aligned), two accumulators
__device__ __forceinline__ double neg_dot_lds(const double* a, const double* b, int n, double acc) {
  double s0 = acc, s1 = 0.0;
  int k = 0;
  for (; k + 8 <= n; k += 8) {
    const double2 a0 = lds2(a + k), a1 = lds2(a + k + 2), a2 = lds2(a + k + 4), a3 = lds2(a + k + 6);
    const double2 b0 = lds2(b + k), b1 = lds2(b + k + 2), b2 = lds2(b + k + 4), b3 = lds2(b + k + 6);
    s0 = fma(-a0.x, b0.x, s0); s1 = fma(-a0.y, b0.y, s1);
    s0 = fma(-a1.x, b1.x, s0); s1 = fma(-a1.y, b1.y, s1);
    s0 = fma(-a2.x, b2.x, s0); s1 = fma(-a2.y, b2.y, s1);
    s0 = fma(-a3.x, b3.x, s0); s1 = fma(-a3.y, b3.y, s1);
  }
  for (; k + 2 <= n; k += 2) {
    const double2 a0 = lds2(a + k), b0 = lds2(b + k);
    s0 = fma(-a0.x, b0.x, s0); s1 = fma(-a0.y, b0.y, s1);
  }
  if (k < n) s0 = fma(-a[k], b[k], s0);
  return s0 + s1;
}

// Lower Cholesky in place, lane i owns row i (np.linalg.cholesky, rmhmc.py:60,171), blocked by 16 columns (right-looking): the column
// steps of a block only carry the dot products over the block's own columns (length < 16, for the diagonal block and the panel below it
// alike), and after each block the trailing matrix gets its rank-16 update A_IJ -= P_I P_J' on the fp64 matrix cores, operands straight
// from the LDS image (one ds_read per 16-row panel tile and k-step serves as A of tile row I and as B of tile column I).  rdiag (lane
// j): 1/L[j][j].  A pivot <= 0 or NaN yields NaN everywhere downstream (=> H is NaN => the proposal is rejected).  Rows / columns
// D..16*NB-1 are padded with the identity here.
template <int NB, bool PK = false>
__device__ __forceinline__ int chol_lds_blk(double* A, int D, int lane, double& rdiag) {
  constexpr int DPc = 16 * NB;
  int bad = 0;
  rdiag = 1.0;
  double* rowp = A + rm_row<PK>(lane);
  const int rlen = min(DPc, rm_len<PK>(lane));  // (PK: the blocks above the diagonal do not exist, and are never read)
  if (D < DPc) {
    if (lane >= D && lane < DPc)
      for (int m = 0; m < rlen; ++m) rowp[m] = (m == lane) ? 1.0 : 0.0;
    else if (lane < D)
      for (int m = D; m < rlen; ++m) rowp[m] = 0.0;
    __builtin_amdgcn_wave_barrier();
  }
#pragma unroll
  for (int kb = 0; kb < NB; ++kb) {
    const int c0 = 16 * kb;
    {
      // The 16-column panel in REGISTERS (lane = row, a[j] = column c0 + j), right-looking: the pivot and the multipliers of a column
      // are v_readlane broadcasts (the indices are compile-time constants), so the 16 dependent column steps touch no LDS and need no
      // barrier (in LDS, left-looking, a step cost ~700 cycles of round trips: 145 us of k_factor_full / k_factor_solve each).
      // Lanes above the panel (rows < c0) and beyond the matrix carry values that are never stored.
      const double* rp = A + rm_row<PK>(min(max(lane, c0), DPc - 1)) + c0;
      double a[16];
#pragma unroll
      for (int j = 0; j < 16; j += 2) { const double2 t = lds2(rp + j); a[j] = t.x; a[j + 1] = t.y; }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int jj = c0 + j;
        const double sjj = rdlane(a[j], jj);
        if (!(sjj > 0.0)) bad = 1;
        const double rinv = rsqrt(sjj);
        a[j] = (lane == jj) ? sjj * rinv : a[j] * rinv;
        if (lane == jj) rdiag = rinv;
#pragma unroll
        for (int k = j + 1; k < 16; ++k) a[k] = fma(-a[j], rdlane(a[j], c0 + k), a[k]);
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (lane >= c0 + j && lane < DPc) rowp[c0 + j] = a[j];
      __builtin_amdgcn_wave_barrier();
    }
    if (kb + 1 < NB) {
      const int kk = lane >> 4, ii = lane & 15;
      d4 acc[NB][NB];
#pragma unroll
      for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = 0; J < NB; ++J) acc[I][J] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int k0 = 0; k0 < 16; k0 += 4) {
        double op[NB];
#pragma unroll
        for (int I = kb + 1; I < NB; ++I) op[I] = A[rm_row<PK>(16 * I + ii) + c0 + k0 + kk];
#pragma unroll
        for (int I = kb + 1; I < NB; ++I)
#pragma unroll
          for (int J = kb + 1; J <= I; ++J) acc[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[I], op[J], acc[I][J], 0, 0, 0);
      }
#pragma unroll
      for (int I = kb + 1; I < NB; ++I)
#pragma unroll
        for (int J = kb + 1; J <= I; ++J)
#pragma unroll
          for (int r = 0; r < 4; ++r) A[rm_row<PK>(16 * I + kk + 4 * r) + 16 * J + ii] -= acc[I][J][r];
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (lane >= D) rdiag = 1.0;
  return bad;
}
// x = (L L')^-1 b ; lane i holds b_i on entry and x_i on return; rdiag as produced by chol_lds_blk
template <bool PK = false>
__device__ __forceinline__ double cholsolve_lds(const double* L, int D, int lane, double b, double rdiag) {
  const double* rowp = L + rm_row<PK>(lane);
  int k = 0;
  for (; k + 4 <= D; k += 4) {  // forward, column oriented; the four multipliers are fetched up front
    const double2 l01 = lds2(rowp + k), l23 = lds2(rowp + k + 2);
    const double lk[4] = {l01.x, l01.y, l23.x, l23.y};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double yk = rdlane(b, k + q) * rdlane(rdiag, k + q);
      if (lane == k + q) b = yk;
      else if (lane > k + q) b = fma(-lk[q], yk, b);
    }
  }
  for (; k < D; ++k) {
    const double yk = rdlane(b, k) * rdlane(rdiag, k);
    if (lane == k) b = yk;
    else if (lane > k) b = fma(-rowp[k], yk, b);
  }
  k = D - 1;
  for (; k >= 3; k -= 4) {  // backward with L'
    const double lk[4] = {L[rm_row<PK>(k) + lane], L[rm_row<PK>(k - 1) + lane], L[rm_row<PK>(k - 2) + lane], L[rm_row<PK>(k - 3) + lane]};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double xk = rdlane(b, k - q) * rdlane(rdiag, k - q);
      if (lane == k - q) b = xk;
      else if (lane < k - q) b = fma(-lk[q], xk, b);
    }
  }
  for (; k >= 0; --k) {
    const double xk = rdlane(b, k) * rdlane(rdiag, k);
    if (lane == k) b = xk;
    else if (lane < k) b = fma(-L[rm_row<PK>(k) + lane], xk, b);
  }
  return b;
}

// DxD matrix (row stride DP in HBM) -> LDS, sixteen row loads in flight (the kernels that start with it are one wavefront per chain
// and wait for this load before anything else: 8 KB in flight per wavefront instead of 4)
template <bool PK = false>
__device__ __forceinline__ void load_mat_lds(double* A, const double* __restrict__ G, int D, int DP, int lane) {
  // (PK: only the lower block triangle is kept, so only those columns are fetched: 20 of the 32 KB of a 64 x 64 matrix.  These kernels
  //  are one wavefront per chain and, summed over a leapfrog step, bound by the bytes of the per-chain matrices they move.)
  const int l = lane < D ? lane : 0;
  int i = 0;
  for (; i + 16 <= D; i += 16) {
    double v[16];
    const bool in = lane < rm_len<PK>(i);  // (i a multiple of 16: the sixteen rows belong to one block row)
    if (in) {
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = G[(i + q) * DP + l];
#pragma unroll
      for (int q = 0; q < 16; ++q) A[rm_row<PK>(i + q) + lane] = v[q];
    }
  }
  for (; i + 8 <= D; i += 8) {
    double v[8];
    if (lane < rm_len<PK>(i)) {
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = G[(i + q) * DP + l];
#pragma unroll
      for (int q = 0; q < 8; ++q) A[rm_row<PK>(i + q) + lane] = v[q];
    }
  }
  for (; i < D; ++i)
    if (lane < rm_len<PK>(i)) A[rm_row<PK>(i) + lane] = G[i * DP + l];
  __builtin_amdgcn_wave_barrier();
}

// position fixed point, first iterate (rmhmc.py:113-122 with FixedIter = 0): G(Pw^0) = G(w) is the
// factor already stored in the trajectory record, so u = u0 and Pw^1 = w + tau*eps*u0.
// upd_nsplit > 0: the last momentum fixed-point update p = p + tau eps/2 (grad - tr/2 + q/2) (rmhmc.py:108,110; k_mom_update with final = 1)
// is done here first, q summed from upd_nsplit row-split planes
// u0 = G^-1 p (rmhmc.py:113) as a product with the stored inverse - k_ginv_matvec's loop, bound by reading the matrix once: 66 us - instead of
// two triangular solves with the stored factor (128 dependent v_readlane steps per chain: 113 us); G^-1 = W'W is the same factor's inverse
__global__ __launch_bounds__(64) void k_pos_first(int D, int DP, Chains ch, double eps, int upd_nsplit = 0) {
  __shared__ double A[64];
  const int c = blockIdx.x, lane = threadIdx.x;
  if (c == 0 && lane == 0 && ch.stale_list) *ch.stale_count = 0;  // (k_crestore has consumed the list; k_iter_end appends at the end of the step)
  if (ch.phase[c] != 1) return;
  const double* __restrict__ Gi = ch.trj.Ginv + (size_t)c * DP * DP;
  const bool in = lane < D;
  double g[16];  // sixteen matrix rows in flight, the first sixteen requested before the partial sums (as k_mom_update_matvec)
#pragma unroll
  for (int q = 0; q < 16; ++q) g[q] = (in && q < D) ? Gi[q * DP + lane] : 0.0;  // symmetric: row j read coalesced
  double pb = in ? ch.p[(size_t)c * DP + lane] : 0.0;
  if (upd_nsplit > 0 && in) {
    const size_t o = (size_t)c * DP + lane;
    double q = 0.0;
    for (int s0 = 0; s0 < upd_nsplit; s0 += 16) {
      double t[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) t[k] = (s0 + k < upd_nsplit) ? ch.qpart[((size_t)(s0 + k) * ch.n + c) * DP + lane] : 0.0;
#pragma unroll
      for (int k = 0; k < 16; ++k)
        if (s0 + k < upd_nsplit) q += t[k];
    }
    pb = pb + (ch.tau[c] * eps * 0.5) * (ch.trj.grad[o] - 0.5 * ch.trj.tr[o] + 0.5 * q);
    ch.p[o] = pb;
  }
  double u0 = 0.0;
  A[lane] = pb;
  __builtin_amdgcn_wave_barrier();
  for (int j0 = 0; j0 < D; j0 += 16) {
    double gn[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) gn[q] = (in && j0 + 16 + q < D) ? Gi[(j0 + 16 + q) * DP + lane] : 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q)
      if (in && j0 + q < D) u0 = fma(g[q], A[j0 + q], u0);
#pragma unroll
    for (int q = 0; q < 16; ++q) g[q] = gn[q];
  }
  if (lane < D) {
    ch.u0[(size_t)c * DP + lane] = u0;
    ch.wq[(size_t)c * DP + lane] = ch.trj.w[(size_t)c * DP + lane] + ch.tau[c] * eps * u0;
  }
}

// position fixed point, iterate k>=1 (rmhmc.py:116-122): factor G(Pw^k), solve, update Pw.
// last >= 0: this is the last position iterate - the iterate is accepted as the new w and the position guard applied here (what k_pos_final
// does in a launch of its own, rmhmc.py:123-130); last = the guards flag
template <int NB>
__global__ __launch_bounds__(64) void k_factor_solve(int D, int DP, Chains ch, double eps, int last = -1) {
  __shared__ __attribute__((aligned(16))) double A[RM_PK_DOUBLES];
  const int c = blockIdx.x, lane = threadIdx.x;
  if (c == 0 && lane == 0 && ch.i8_dmax) *ch.i8_dmax = 0ull;
  if (ch.phase[c] != 1) return;
  if (lane == 0 && ch.i8_vbad) ch.i8_vbad[c] = 0;
  load_mat_lds<true>(A, ch.Gq + (size_t)c * DP * DP, D, DP, lane);
  double rdiag;
  const int bad = chol_lds_blk<NB, true>(A, D, lane, rdiag);
  const double pb = (lane < D) ? ch.p[(size_t)c * DP + lane] : 0.0;
  const double u = cholsolve_lds<true>(A, D, lane, pb, rdiag);
  double wn = 0.0;
  if (lane < D) wn = ch.trj.w[(size_t)c * DP + lane] + ch.tau[c] * (eps * 0.5) * (ch.u0[(size_t)c * DP + lane] + u);
  int st = bad ? 1 : 0;  // RMHMC_ST_NOT_PD
  if (last >= 0) {  // (k_pos_final's arithmetic: sum of squares over the lanes, sqrt, w / (3 |w|))
    const double nw = sqrt(wave_sum(wn * wn));
    if (last && nw > 10.0) { wn /= nw * 3.0; st |= 8; }  // RMHMC_ST_GUARD_W
    if (lane < D) ch.trj.w[(size_t)c * DP + lane] = wn;
  }
  if (lane < D) ch.wq[(size_t)c * DP + lane] = wn;
  if (st && lane == 0) ch.status[c] |= st;
}

// The per-chain vector kernels below run one wavefront per chain and stride the lanes over the dimensions
// (d = lane, lane+64, ...), so they serve D <= 64 and the large-D path (D <= 256) alike.
#define RM_DMAX 256
#define RM_DCH 4   // RM_DMAX / 64 dimension chunks per lane

// accept the position iterate as the new w and apply the position guard (rmhmc.py:123-130)
__global__ __launch_bounds__(64) void k_pos_final(int D, int DP, Chains ch, int guards) {
  const int c = blockIdx.x, lane = threadIdx.x;
  if (ch.phase[c] != 1) return;
  const size_t o = (size_t)c * DP;
  double ss = 0.0;
  for (int d = lane; d < D; d += 64) { const double w = ch.wq[o + d]; ss = fma(w, w, ss); }
  const double nw = sqrt(wave_sum(ss));
  const bool fire = guards && nw > 10.0;
  for (int d = lane; d < D; d += 64) {
    double w = ch.wq[o + d];
    if (fire) w /= nw * 3.0;
    ch.trj.w[o + d] = w;
    ch.wq[o + d] = w;
  }
  if (fire && lane == 0) ch.status[c] |= 8;  // RMHMC_ST_GUARD_W
}

// Lower triangle of (L L')^-1 in place of the Cholesky factor L held in the LDS image A (one wavefront, lane = row; rdiag as
// produced by chol_lds_blk).  W = L^-1 in place by 16 x 16 blocks (see below); then G^-1 = W' W on the fp64 matrix cores with
// the operands read straight from the LDS image (A[i][k] = W[m0+k][16I+i], B[k][j] = W[m0+k][16J+j]: one ds_read per 16-column
// tile serves both), lower tiles only, written back into A.
template <int NB, bool PK = false>
__device__ __forceinline__ void spd_inverse_lds(double* A, int D, int lane, double rdiag) {
  constexpr int DPc = 16 * NB;
  // W = L^-1 in place, by 16 x 16 blocks.  (A column sweep over the whole matrix - 63 dependent steps with dot products of up to 63
  // terms - was 590 of the 890 us of k_factor_full.)
  //   1. diagonal blocks W_II = L_II^-1, all NB of them in the same 15 column steps (lane = row, each lane in its own block);
  //   2. block column by block column, W_IJ = -W_II (sum_{K=J}^{I-1} L_IK W_KJ) on the fp64 matrix cores, operands straight from
  //      the LDS image; the inner sum stays in the accumulator layout, which is the B-operand layout of the product with W_II.
  // Block (I, J) of L is read for the last time when W_IJ is computed, and is overwritten by it.
  {
    const bool act = lane < DPc;            // (NB < 4: the lanes beyond the matrix read a valid row and never write)
    const int row = act ? lane : DPc - 1;
    double* rowp = A + rm_row<PK>(row);
    const int rlen = min(DPc, rm_len<PK>(row));
    const int il = row & 15, cb0 = row & ~15;  // position inside / first row and column of the lane's diagonal block
    // diagonal 1 / L_ii and the zero upper triangle (rows / columns >= D are the identity padding of
    // chol_lds_blk: they invert to themselves and are zeroed at the end)
    if (act) {
      rowp[row] = rdiag;
      for (int m = row + 1; m < rlen; ++m) rowp[m] = 0.0;  // (packed image: the row ends with its diagonal block)
    }
    __builtin_amdgcn_wave_barrier();
    for (int jl = 14; jl >= 0; --jl) {
      // W[i][j] = -(sum_{m > j} W[i][m] L[m][j]) / L[j][j] inside the diagonal block (W[i][m] = 0 for m > i), j = cb0 + jl
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int ml = 1; ml < 16; ++ml) {
        const double t = (ml > jl) ? rowp[cb0 + ml] * A[rm_row<PK>(cb0 + ml) + cb0 + jl] : 0.0;
        if (ml & 1) s0 += t; else s1 += t;
      }
      const double rj = A[rm_row<PK>(cb0 + jl) + cb0 + jl];  // W[j][j] = 1 / L[j][j], set above
      __builtin_amdgcn_wave_barrier();
      if (act && il > jl) rowp[cb0 + jl] = -(s0 + s1) * rj;
      __builtin_amdgcn_wave_barrier();
    }
    const int kk = lane >> 4, ii = lane & 15;
#pragma unroll
    for (int J = 0; J + 1 < NB; ++J) {
#pragma unroll
      for (int I = J + 1; I < NB; ++I) {
        d4 T = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int K = J; K < I; ++K)
#pragma unroll
          for (int k0 = 0; k0 < 16; k0 += 4)
            T = __builtin_amdgcn_mfma_f64_16x16x4f64(A[rm_row<PK>(16 * I + ii) + 16 * K + k0 + kk], A[rm_row<PK>(16 * K + k0 + kk) + 16 * J + ii], T, 0, 0,
                                                     0);
        d4 R = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int r = 0; r < 4; ++r) R = __builtin_amdgcn_mfma_f64_16x16x4f64(A[rm_row<PK>(16 * I + ii) + 16 * I + 4 * r + kk], T[r], R, 0, 0, 0);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 4; ++r) A[rm_row<PK>(16 * I + kk + 4 * r) + 16 * J + ii] = -R[r];
        __builtin_amdgcn_wave_barrier();
      }
    }
    // rows of the padding: W = 0 there, so that G^-1 = W' W is zero outside D x D
    if (D < DPc) {
      if (lane >= D && lane < DPc)
        for (int m = 0; m < rlen; ++m) rowp[m] = 0.0;
      __builtin_amdgcn_wave_barrier();
    }
  }
  // G^-1 = W' W, lower 16x16 tiles (I >= J); rows of W below the first row of tile J... all rows m contribute, W[m][a] = 0 for m < a
  constexpr int NT = NB * (NB + 1) / 2;
  d4 acc[NT];
#pragma unroll
  for (int q = 0; q < NT; ++q) acc[q] = (d4){0.0, 0.0, 0.0, 0.0};
  {
    const int kk = lane >> 4, ii = lane & 15;
    for (int m0 = 0; m0 < DPc; m0 += 4) {
      double op[NB];
#pragma unroll
      for (int I = 0; I < NB; ++I) op[I] = A[rm_row<PK>(m0 + kk) + 16 * I + ii];  // (PK, rows above block row I: past the row's end, not used below)
      int q = 0;
#pragma unroll
      for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = 0; J <= I; ++J) {
          if (m0 + 3 >= 16 * I) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[I], op[J], acc[q], 0, 0, 0);  // W[m][a] = 0 for m < a
          ++q;
        }
    }
  }
  __builtin_amdgcn_wave_barrier();
  // tiles back into the LDS image (lower triangle of G^-1): element (16I + (lane>>4) + 4r, 16J + (lane&15)) <-> acc[r]
  {
    int q = 0;
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
      for (int J = 0; J <= I; ++J) {
#pragma unroll
        for (int r = 0; r < 4; ++r) A[rm_row<PK>(16 * I + (lane >> 4) + 4 * r) + 16 * J + (lane & 15)] = acc[q][r];
        ++q;
      }
  }
  __builtin_amdgcn_wave_barrier();
}

// new point: factor G(w), half log-determinant, explicit inverse, log joint, and u = G^-1 p
// (rmhmc.py:137-138,158,166-171).  One DxD matrix in LDS per chain (4 chains per CU): Cholesky in place, W = L^-1 in place
// (lane = row, column by column from the right: W[i][j] = -(sum_{m>j} W[i][m] L[m][j]) / L[j][j] reads only columns > j of W and
// column j of L, which is overwritten afterwards), then G^-1 = W' W on the fp64 matrix cores with the operands read straight
// from the LDS image (A[i][k] = W[m0+k][16I+i], B[k][j] = W[m0+k][16J+j]: one ds_read per 16-column tile serves both), lower
// tiles only.  The product is exactly symmetric by construction.
template <int NB>
__global__ __launch_bounds__(64) void k_factor_full(DevData dd, Chains ch, int nsplit) {
  __shared__ __attribute__((aligned(16))) double A[RM_PK_DOUBLES];
  constexpr int DPc = 16 * NB;
  const int D = dd.D, DP = dd.DP;
  const int c = blockIdx.x, lane = threadIdx.x;
  if (c == 0 && lane == 0 && ch.i8_dmax) *ch.i8_dmax = 0ull;
  if (ch.phase[c] != 1) return;
  if (lane == 0 && ch.i8_vbad) ch.i8_vbad[c] = 0;
  load_mat_lds<true>(A, ch.Gq + (size_t)c * DP * DP, D, DP, lane);
  double rdiag;
  const int bad = chol_lds_blk<NB, true>(A, D, lane, rdiag);
  // half log det = sum log diag(L) = -sum log(1/L_jj)   (rmhmc.py:171,175)
  const double hld = -wave_sum((lane < D) ? log(rdiag) : 0.0);
  // store L (lower; the entries above the diagonal are zero since the allocation and never written, here or by copy_rec)
  double* __restrict__ Lg = ch.trj.L + (size_t)c * DP * DP;
  for (int i = 0; i < D; ++i)
    if (lane <= i) Lg[i * DP + lane] = A[rm_row<true>(i) + lane];
  __builtin_amdgcn_wave_barrier();
  spd_inverse_lds<NB, true>(A, D, lane, rdiag);
  double* __restrict__ Gi = ch.trj.Ginv + (size_t)c * DP * DP;
  const double pl = (lane < D) ? ch.p[(size_t)c * DP + lane] : 0.0;
  double u = 0.0;
  for (int i = 0; i < D; ++i) {
    const double gi = (lane <= i) ? A[rm_row<true>(i) + lane] : A[rm_row<true>(min(lane, DPc - 1)) + i];
    if (lane < D) Gi[i * DP + lane] = gi;
    u = fma((lane < D) ? gi : 0.0, rdlane(pl, i), u);  // u = G^-1 p
  }
  if (lane < D) ch.uq[(size_t)c * DP + lane] = u;
  // log joint = sum of the row-split partials + Gaussian prior (rmhmc.py:166-169, tools.py:10-14)
  double part = 0.0;
  for (int b = lane; b < nsplit; b += 64) part += ch.ljl_part[(size_t)c * nsplit + b];
  const double wl = (lane < D) ? ch.trj.w[(size_t)c * DP + lane] : 0.0;
  // gradient = X'(t - s) - w/alpha (rmhmc.py:100,140): sum of the row-split partials of k_rowpass<RP_F>
  if (lane < D) {
    double g = 0.0;
    for (int sp = 0; sp < nsplit; ++sp) g += ch.gpart[((size_t)sp * ch.n + c) * DP + lane];
    ch.trj.grad[(size_t)c * DP + lane] = g - wl * dd.inv_alpha;
  }
  part += (lane < D) ? (dd.log_prior_const - wl * wl * 0.5 * dd.inv_alpha) : 0.0;
  const double ljl = wave_sum(part);
  if (lane == 0) {
    ch.trj.hld[c] = hld;
    ch.trj.ljl[c] = ljl;
    if (bad) ch.status[c] |= 1;
  }
}

// u = G^-1 v for the momentum fixed point (rmhmc.py:104); src = p (first iterate) or PM
__global__ __launch_bounds__(64) void k_ginv_matvec(int D, int DP, Chains ch, const double* __restrict__ src) {
  const int c = blockIdx.x, lane = threadIdx.x;
  if (ch.phase[c] != 1) return;
  const double* __restrict__ Gi = ch.trj.Ginv + (size_t)c * DP * DP;
  if (D <= 64) {  // lane = dimension, sixteen rows in flight (as k_mom_update_matvec); the same sums in the same order as the loop below
    const bool in = lane < D;
    const double sv = in ? src[(size_t)c * DP + lane] : 0.0;
    double u0 = 0.0;
    for (int j0 = 0; j0 < D; j0 += 16) {
      double g[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) g[q] = (in && j0 + q < D) ? Gi[(j0 + q) * DP + lane] : 0.0;
#pragma unroll
      for (int q = 0; q < 16; ++q)
        if (in && j0 + q < D) u0 = fma(g[q], rdlane(sv, j0 + q), u0);
    }
    if (in) ch.uq[(size_t)c * DP + lane] = u0;
    return;
  }
  double u[RM_DCH] = {0.0, 0.0, 0.0, 0.0};
  for (int j = 0; j < D; ++j) {
    const double sj = src[(size_t)c * DP + j];
#pragma unroll
    for (int k = 0; k < RM_DCH; ++k) {
      const int d = lane + 64 * k;
      if (d < D) u[k] = fma(Gi[j * DP + d], sj, u[k]);  // symmetric: row j read coalesced
    }
  }
#pragma unroll
  for (int k = 0; k < RM_DCH; ++k)
    if (lane + 64 * k < D) ch.uq[(size_t)c * DP + lane + 64 * k] = u[k];
}

// PM = p + tau*eps/2 * (grad - tr/2 + q/2)   (rmhmc.py:108); final != 0: p = PM (rmhmc.py:110)
__global__ __launch_bounds__(64) void k_mom_update(int D, int DP, Chains ch, double eps, int final, int nsplit) {
  const int c = blockIdx.x, lane = threadIdx.x;
  if (ch.phase[c] != 1) return;
  const double h = ch.tau[c] * eps * 0.5;
  for (int d = lane; d < D; d += 64) {
    const size_t o = (size_t)c * DP + d;
    double q = 0.0;
    for (int s = 0; s < nsplit; ++s) q += ch.qpart[((size_t)s * ch.n + c) * DP + d];
    const double pm = ch.p[o] + h * (ch.trj.grad[o] - 0.5 * ch.trj.tr[o] + 0.5 * q);
    if (final) ch.p[o] = pm;
    else ch.PM[o] = pm;
  }
}

// k_mom_update (final = 0) followed by k_ginv_matvec on the new PM, in one launch (D <= 64: lane = dimension; the same sums in the same order)
__global__ __launch_bounds__(64) void k_mom_update_matvec(int D, int DP, Chains ch, double eps, int nsplit) {
  const int c = blockIdx.x, lane = threadIdx.x;
  if (ch.phase[c] != 1) return;
  const size_t o = (size_t)c * DP + lane;
  const double* __restrict__ Gi = ch.trj.Ginv + (size_t)c * DP * DP;
  const bool in = lane < D;
  // sixteen matrix rows in flight at a time, the first sixteen requested before the partial sums are (every chain's wavefront is resident at
  // once here, so the launch lasts as long as ONE wavefront's chain of round trips: it was one per row and one per partial)
  double g[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) g[q] = (in && q < D) ? Gi[q * DP + lane] : 0.0;  // symmetric: row j read coalesced
  double pm = 0.0;
  if (in) {
    double q = 0.0;
    for (int s0 = 0; s0 < nsplit; s0 += 16) {
      double t[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) t[k] = (s0 + k < nsplit) ? ch.qpart[((size_t)(s0 + k) * ch.n + c) * DP + lane] : 0.0;
#pragma unroll
      for (int k = 0; k < 16; ++k)
        if (s0 + k < nsplit) q += t[k];
    }
    pm = ch.p[o] + (ch.tau[c] * eps * 0.5) * (ch.trj.grad[o] - 0.5 * ch.trj.tr[o] + 0.5 * q);
    ch.PM[o] = pm;
  }
  double u = 0.0;
  for (int j0 = 0; j0 < D; j0 += 16) {
    double gn[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) gn[q] = (in && j0 + 16 + q < D) ? Gi[(j0 + 16 + q) * DP + lane] : 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q)
      if (j0 + q < D) u = fma(g[q], rdlane(pm, j0 + q), u);
#pragma unroll
    for (int q = 0; q < 16; ++q) g[q] = gn[q];
  }
  if (in) ch.uq[o] = u;
}

// explicit momentum half step at the new point (rmhmc.py:163) + step bookkeeping
// trpart != null: the trace term is summed here from its row-split planes first (k_reduce_tr's sums, in a launch of its own otherwise)
__global__ __launch_bounds__(64) void k_mom_final(int D, int DP, Chains ch, double eps, int advance, int nsplit, const double* __restrict__ trpart = nullptr) {
  const int c = blockIdx.x, lane = threadIdx.x;
  if (ch.phase[c] != 1) return;
  const double h = ch.tau[c] * eps * 0.5;
  int nonfinite = 0;
  for (int d = lane; d < D; d += 64) {
    const size_t o = (size_t)c * DP + d;
    if (trpart) {
      double s = 0.0;
      for (int sp = 0; sp < nsplit; ++sp) s += trpart[((size_t)sp * ch.n + c) * DP + d];
      ch.trj.tr[o] = s;
    }
    double q = 0.0;
    for (int s = 0; s < nsplit; ++s) q += ch.qpart[((size_t)s * ch.n + c) * DP + d];
    ch.last[o] = q;
    const double pn = advance ? ch.p[o] + h * (ch.trj.grad[o] - 0.5 * ch.trj.tr[o] + 0.5 * q) : ch.p[o];
    if (advance) ch.p[o] = pn;
    nonfinite |= !(isfinite(pn) && isfinite(ch.trj.w[o]));
  }
  const unsigned long long any = __ballot(nonfinite);
  if (lane == 0) {
    if (any) ch.status[c] |= 2;  // RMHMC_ST_NONFINITE
    if (advance) {
      ch.steps_left[c] -= 1;
      ch.steps_done[c] += 1;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// transition control (rmhmc.py:47-48,59-61,80-93 and :166-191)
// ---------------------------------------------------------------------------------------------
struct IterParams {
  unsigned flags;
  int L;                  // NumOfLeapFrogSteps
  unsigned long long seed;
  long long chain_offset;
  long long iter_limit;   // chains stop starting transitions at this iteration count
  long long burn_in;
  long long S;            // rows per chain in samples
  double* samples;        // [n][S][D] (unpadded) or nullptr
  const double *z_in, *ulen_in, *gdir_in, *uacc_in;  // explicit randomness (unit API) or nullptr
  int* done_count;
  const int* orig;        // work-sorted layout of the bulk sampler (rmhmc_hip.hip, sample_core): the chain at position c is chain orig[c] of the
                          // caller's order - its Philox key and its block of `samples` - or nullptr: position = chain
  int lower_L;            // the records' L is a lower factor with an all-zero upper triangle (generic D <= 64 kernels): copy_rec moves the
                          // lower part only
};

// lowerL: L holds a lower Cholesky factor whose upper triangle is zero in every record (the generic path, D <= 64): only the columns up
// to the diagonal are moved
__device__ __forceinline__ void copy_rec(const Rec& dst, const Rec& src, int c, int D, int DP, int lane, bool lowerL = false) {
  const size_t o = (size_t)c * DP;
  for (int d = lane; d < D; d += 64) {
    dst.w[o + d] = src.w[o + d];
    dst.grad[o + d] = src.grad[o + d];
    dst.tr[o + d] = src.tr[o + d];
  }
  const size_t m = (size_t)c * DP * DP;
  for (int d = lane; d < D; d += 64) {
    int i = 0;
    for (; i + 16 <= D; i += 16) {  // sixteen rows of both matrices in flight (the copy is latency bound: one wavefront per chain)
      double a[16], b[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) { a[q] = lowerL && d > i + 15 ? 0.0 : src.L[m + (i + q) * DP + d]; b[q] = src.Ginv[m + (i + q) * DP + d]; }
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        if (!(lowerL && d > i + 15)) dst.L[m + (i + q) * DP + d] = a[q];   // (a lower factor: zero above the diagonal on both sides, always)
        dst.Ginv[m + (i + q) * DP + d] = b[q];
      }
    }
    for (; i + 8 <= D; i += 8) {
      double a[8], b[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) { a[q] = lowerL && d > i + 7 ? 0.0 : src.L[m + (i + q) * DP + d]; b[q] = src.Ginv[m + (i + q) * DP + d]; }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (!(lowerL && d > i + 7)) dst.L[m + (i + q) * DP + d] = a[q];
        dst.Ginv[m + (i + q) * DP + d] = b[q];
      }
    }
    for (; i < D; ++i) {
      if (!(lowerL && d > i)) dst.L[m + i * DP + d] = src.L[m + i * DP + d];
      dst.Ginv[m + i * DP + d] = src.Ginv[m + i * DP + d];
    }
  }
  if (lane == 0) {
    dst.ljl[c] = src.ljl[c];
    dst.hld[c] = src.hld[c];
  }
}

// 0.5 * p' Ginv p for the chain's vector ps[0..D) in LDS
__device__ __forceinline__ double half_quadform(const double* __restrict__ Gi, int D, int DP, int lane, const double* ps) {
  if (D <= 64) {
    // lane = dimension; sixteen rows in flight at a time (the loop below is one load - one wait per row: in k_iter_begin / k_iter_end, where only
    // the chains at a transition boundary work, 64 serial round trips were most of the kernels' 110 / 88 us).  Same sums in the same order.
    const bool in = lane < D;
    double y = 0.0;
    for (int j0 = 0; j0 < D; j0 += 16) {
      double g[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) g[q] = (in && j0 + q < D) ? Gi[(size_t)(j0 + q) * DP + lane] : 0.0;
#pragma unroll
      for (int q = 0; q < 16; ++q)
        if (in && j0 + q < D) y = fma(g[q], ps[j0 + q], y);
    }
    return 0.5 * wave_sum(in ? fma(y, ps[lane], 0.0) : 0.0);
  }
  double y[RM_DCH] = {0.0, 0.0, 0.0, 0.0};
  for (int j = 0; j < D; ++j) {
    const double pj = ps[j];
#pragma unroll
    for (int k = 0; k < RM_DCH; ++k) {
      const int d = lane + 64 * k;
      if (d < D) y[k] = fma(Gi[j * DP + d], pj, y[k]);
    }
  }
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < RM_DCH; ++k)
    if (lane + 64 * k < D) s = fma(y[k], ps[lane + 64 * k], s);
  return 0.5 * wave_sum(s);
}

// D standard normals of (seed, chain, iteration) into zs[] (or the caller-supplied draws)
__device__ __forceinline__ void draw_normals(const IterParams& ip, int c, long long it, int D, int lane, double* zs, long long oc = -1) {
  const long long id = oc >= 0 ? oc : c;  // chain index in the caller's order (IterParams::orig)
  for (int d = lane; d < D; d += 64) {
    if (ip.z_in) {
      zs[d] = ip.z_in[(size_t)c * D + d];
    } else {
      double U0, U1;
      rng_block(ip.seed, (unsigned long long)(ip.chain_offset + id), (uint32_t)it, (uint32_t)(d >> 1), U0, U1);
      const double R = sqrt(-2.0 * log(U0));
      double sn, cs;
      sincos(RM_PI2 * U1, &sn, &cs);
      zs[d] = (d & 1) ? R * sn : R * cs;
    }
  }
  __builtin_amdgcn_wave_barrier();  // callers are one wavefront (a 64-thread block, or wave 0 of k_step_medium): LDS is in order per wave
}

// start of a transition for chain c, executed by ONE wavefront (zs, ps: LDS scratch of D doubles each)
__device__ __forceinline__ void iter_begin_dev(int D, int DP, const Chains& ch, const IterParams& ip, int c, int lane, double* zs, double* ps) {
  if (ch.phase[c] != 0) return;
  const long long it = ch.iter[c];
  if (it >= ip.iter_limit) return;
  // trajectory starts from the cached record of the current point (wNew = w.copy(), rmhmc.py:47).  After an ACCEPTED proposal the two
  // records are already identical (k_iter_end has just copied trj to cur), so the 2 x 64 KB copy is only made after a rejection - the
  // same per-chain flag that tells k_mompass the c tiles are not those of trj.w - and on paths that do not keep the flag (it stays 1).
  if (!ch.cstale || ch.cstale[c]) copy_rec(ch.trj, ch.cur, c, D, DP, lane, ip.lower_L != 0);
  const long long oc = ip.orig ? ip.orig[c] : c;
  // draws: z ~ randn(1,D), u_len ~ rand(), g_dir ~ randn()   (rmhmc.py:80,89,90)
  draw_normals(ip, c, it, D, lane, zs, oc);
  double u_len, g_dir;
  if (ip.z_in) {
    u_len = ip.ulen_in[c];
    g_dir = ip.gdir_in[c];
  } else {
    const unsigned long long gid = (unsigned long long)(ip.chain_offset + oc);
    double U0, U1, Ua;
    rng_block(ip.seed, gid, (uint32_t)it, 0x40000000u, u_len, Ua);
    rng_block(ip.seed, gid, (uint32_t)it, 0x40000001u, U0, U1);
    g_dir = sqrt(-2.0 * log(U0)) * cos(RM_PI2 * U1);
  }
  // momentum p = L' z (reference, rmhmc.py:60,80) or L z (corrected)
  const double* __restrict__ Lc = ch.cur.L + (size_t)c * DP * DP;
  double p[RM_DCH] = {0.0, 0.0, 0.0, 0.0};
  if (D <= 64) {  // lane = dimension, sixteen loads in flight (as half_quadform); the same sums in the same order as the loops below
    const int d = lane;
    double acc = 0.0;
    for (int j0 = 0; j0 < D; j0 += 16) {
      double l[16];
      if (ip.flags & 1u) {
#pragma unroll
        for (int q = 0; q < 16; ++q) l[q] = (j0 + q < D && d <= j0 + q) ? Lc[(size_t)(j0 + q) * DP + d] : 0.0;   // L' z: column d of row j0 + q
#pragma unroll
        for (int q = 0; q < 16; ++q)
          if (j0 + q < D && d <= j0 + q) acc = fma(l[q], zs[j0 + q], acc);
      } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) l[q] = (d < D && j0 + q <= d) ? Lc[(size_t)d * DP + j0 + q] : 0.0;            // L z: the lane's own row
#pragma unroll
        for (int q = 0; q < 16; ++q)
          if (d < D && j0 + q <= d) acc = fma(l[q], zs[j0 + q], acc);
      }
    }
    p[0] = acc;
  } else if (ip.flags & 1u) {
    for (int i = 0; i < D; ++i) {
      const double zi = zs[i];
#pragma unroll
      for (int k = 0; k < RM_DCH; ++k) {
        const int d = lane + 64 * k;
        if (d <= i) p[k] = fma(Lc[i * DP + d], zi, p[k]);
      }
    }
  } else {
    for (int j = 0; j < D; ++j) {
      const double zj = zs[j];
#pragma unroll
      for (int k = 0; k < RM_DCH; ++k) {
        const int d = lane + 64 * k;
        if (d >= j && d < D) p[k] = fma(Lc[d * DP + j], zj, p[k]);
      }
    }
  }
  int st = 0;
  if (ip.flags & 2u) {  // momentum guard, rmhmc.py:81-85
    double ss = 0.0;
#pragma unroll
    for (int k = 0; k < RM_DCH; ++k) ss = fma(p[k], p[k], ss);
    const double np_ = sqrt(wave_sum(ss));
    if (np_ > 100.0) {
#pragma unroll
      for (int k = 0; k < RM_DCH; ++k) p[k] /= np_ * 25.0;
      st = 4;
    }
  }
#pragma unroll
  for (int k = 0; k < RM_DCH; ++k) {
    const int d = lane + 64 * k;
    if (d < D) {
      ps[d] = p[k];
      ch.p[(size_t)c * DP + d] = p[k];
      ch.p0[(size_t)c * DP + d] = p[k];
    }
  }
  __builtin_amdgcn_wave_barrier();
  const double quad = half_quadform(ch.cur.Ginv + (size_t)c * DP * DP, D, DP, lane, ps);
  if (lane == 0) {
    const int ns = (int)ceil(u_len * (double)ip.L);  // rmhmc.py:89
    ch.steps_left[c] = ns;
    ch.nsteps_last[c] = ns;
    ch.tau[c] = (g_dir > 0.5) ? 1.0 : -1.0;           // rmhmc.py:90-93
    ch.Hcur[c] = -ch.cur.ljl[c] + ch.cur.hld[c] + quad;  // rmhmc.py:175-176
    ch.status[c] = st;
    ch.phase[c] = (ns > 0) ? 1 : 2;                   // 2: trajectory of zero steps, finish at once
  }
}

__global__ __launch_bounds__(64) void k_iter_begin(int D, int DP, Chains ch, IterParams ip) {
  __shared__ double zs[RM_DMAX];
  __shared__ double ps[RM_DMAX];
  iter_begin_dev(D, DP, ch, ip, blockIdx.x, threadIdx.x, zs, ps);
}

// end of a transition for chain c (rmhmc.py:166-191), executed by ONE wavefront
__device__ __forceinline__ void iter_end_dev(int D, int DP, const Chains& ch, const IterParams& ip, int c, int lane, double* ps) {
  const int ph = ch.phase[c];
  if (!(ph == 2 || (ph == 1 && ch.steps_left[c] == 0))) return;
  const long long it = ch.iter[c];
  const long long oc = ip.orig ? ip.orig[c] : c;
  for (int d = lane; d < D; d += 64) ps[d] = ch.p[(size_t)c * DP + d];
  __builtin_amdgcn_wave_barrier();
  const double quad = half_quadform(ch.trj.Ginv + (size_t)c * DP * DP, D, DP, lane, ps);
  const double Hp = -ch.trj.ljl[c] + ch.trj.hld[c] + quad;  // rmhmc.py:171-172
  const double ratio = -Hp + ch.Hcur[c];                     // rmhmc.py:179
  double u_acc;
  if (ip.z_in) {
    u_acc = ip.uacc_in[c];
  } else {
    double U0;
    rng_block(ip.seed, (unsigned long long)(ip.chain_offset + oc), (uint32_t)it, 0x40000000u, U0, u_acc);
  }
  const bool accept = (ratio > 0.0) || (ratio > log(u_acc));  // rmhmc.py:181
  if (accept) copy_rec(ch.cur, ch.trj, c, D, DP, lane, ip.lower_L != 0);
  __builtin_amdgcn_wave_barrier();  // (a lane reads back only what it wrote itself)
  if (ip.samples && it >= ip.burn_in && it - ip.burn_in < ip.S)
    for (int d = lane; d < D; d += 64)
      ip.samples[((size_t)oc * ip.S + (size_t)(it - ip.burn_in)) * D + d] = ch.cur.w[(size_t)c * DP + d];
  if (lane == 0) {
    ch.Hprop[c] = Hp;
    if (accept) ch.accepted[c] += 1;
    else if (ch.cstale) {
      ch.cstale[c] = 1;
      if (ch.stale_list) ch.stale_list[atomicAdd(ch.stale_count, 1)] = c;
    }
    ch.iter[c] = it + 1;
    ch.phase[c] = 0;
    if (it + 1 == ip.iter_limit && ip.done_count) atomicAdd(ip.done_count, 1);
  }
}

__global__ __launch_bounds__(64) void k_iter_end(int D, int DP, Chains ch, IterParams ip) {
  __shared__ double ps[RM_DMAX];
  iter_end_dev(D, DP, ch, ip, blockIdx.x, threadIdx.x, ps);
}

// trj -> cur for every chain (used after the initial point evaluation)
__global__ __launch_bounds__(64) void k_commit_all(int D, int DP, Chains ch) {
  copy_rec(ch.cur, ch.trj, blockIdx.x, D, DP, threadIdx.x);
}

// ---------------------------------------------------------------------------------------------
// plain HMC with identity mass (code/hmc.py:12-99): the separable special case, re-using the row pass
// for the gradient and the log joint.  Records: cur/trj .w .grad .ljl only.
// ---------------------------------------------------------------------------------------------
// sum the row-split partials of k_rowpass<RP_G> into trj.grad / trj.ljl (hmc.py:53,61,64-67)
__device__ __forceinline__ void hmc_finish_eval(const DevData& dd, const Chains& ch, int c, int lane, int nsplit) {
  const int D = dd.D, DP = dd.DP;
  double part = 0.0;
  for (int d = lane; d < D; d += 64) {
    const double wl = ch.trj.w[(size_t)c * DP + d];
    double g = 0.0;
    for (int sp = 0; sp < nsplit; ++sp) g += ch.gpart[((size_t)sp * ch.n + c) * DP + d];
    ch.trj.grad[(size_t)c * DP + d] = g - wl * dd.inv_alpha;
    part += dd.log_prior_const - wl * wl * 0.5 * dd.inv_alpha;
  }
  for (int b = lane; b < nsplit; b += 64) part += ch.ljl_part[(size_t)c * nsplit + b];
  const double ljl = wave_sum(part);
  if (lane == 0) ch.trj.ljl[c] = ljl;
}
__device__ __forceinline__ void hmc_copy(const Rec& dst, const Rec& src, int c, int D, int DP, int lane) {
  for (int d = lane; d < D; d += 64) {
    dst.w[(size_t)c * DP + d] = src.w[(size_t)c * DP + d];
    dst.grad[(size_t)c * DP + d] = src.grad[(size_t)c * DP + d];
  }
  if (lane == 0) dst.ljl[c] = src.ljl[c];
}
// initial evaluation at theta0: trj -> cur
__global__ __launch_bounds__(64) void k_hmc_init(DevData dd, Chains ch, int nsplit) {
  const int c = blockIdx.x, lane = threadIdx.x;
  hmc_finish_eval(dd, ch, c, lane, nsplit);
  __syncthreads();
  hmc_copy(ch.cur, ch.trj, c, dd.D, dd.DP, lane);
}
// start a transition (hmc.py:41-48,72)
__global__ __launch_bounds__(64) void k_hmc_begin(int D, int DP, Chains ch, IterParams ip) {
  __shared__ double zs[RM_DMAX];
  const int c = blockIdx.x, lane = threadIdx.x;
  if (ch.phase[c] != 0) return;
  const long long it = ch.iter[c];
  if (it >= ip.iter_limit) return;
  hmc_copy(ch.trj, ch.cur, c, D, DP, lane);
  draw_normals(ip, c, it, D, lane, zs);
  double u_len;
  if (ip.z_in) {
    u_len = ip.ulen_in[c];
  } else {
    double Ua;
    rng_block(ip.seed, (unsigned long long)(ip.chain_offset + c), (uint32_t)it, 0x40000000u, u_len, Ua);
  }
  double kin = 0.0;
  for (int d = lane; d < D; d += 64) {
    const double z = zs[d];
    kin = fma(z, z, kin);
    ch.p[(size_t)c * DP + d] = z;  // Mass = I: p = z (hmc.py:41)
  }
  kin = 0.5 * wave_sum(kin);
  if (lane == 0) {
    const int ns = (int)ceil(u_len * (double)ip.L);  // hmc.py:48
    ch.steps_left[c] = ns;
    ch.nsteps_last[c] = ns;
    ch.Hcur[c] = -ch.cur.ljl[c] + kin;  // hmc.py:72
    ch.status[c] = 0;
    ch.phase[c] = (ns > 0) ? 1 : 2;
  }
}
// first momentum half step and the position step (hmc.py:52-58); a NaN momentum ends the trajectory (:56-57)
__global__ __launch_bounds__(64) void k_hmc_pre(int D, int DP, Chains ch, double eps) {
  const int c = blockIdx.x, lane = threadIdx.x;
  if (ch.phase[c] != 1) return;
  int isnan_ = 0;
  for (int d = lane; d < D; d += 64) {
    const size_t o = (size_t)c * DP + d;
    const double p = ch.p[o] + eps * 0.5 * ch.trj.grad[o];
    ch.p[o] = p;
    isnan_ |= (p != p);
  }
  const unsigned long long nan = __ballot(isnan_);
  if (!nan)
    for (int d = lane; d < D; d += 64) ch.trj.w[(size_t)c * DP + d] += eps * ch.p[(size_t)c * DP + d];
  if (nan && lane == 0) { ch.status[c] |= 2; ch.steps_left[c] = 1; }
}
// gradient / log joint at the new position, second momentum half step (hmc.py:60-62)
__global__ __launch_bounds__(64) void k_hmc_post(DevData dd, Chains ch, double eps, int nsplit) {
  const int c = blockIdx.x, lane = threadIdx.x;
  if (ch.phase[c] != 1) return;
  hmc_finish_eval(dd, ch, c, lane, nsplit);
  __syncthreads();
  if (!(ch.status[c] & 2))
    for (int d = lane; d < dd.D; d += 64) ch.p[(size_t)c * dd.DP + d] += eps * 0.5 * ch.trj.grad[(size_t)c * dd.DP + d];
  if (lane == 0) { ch.steps_left[c] -= 1; ch.steps_done[c] += 1; }
}
// finish a transition (hmc.py:64-84)
__global__ __launch_bounds__(64) void k_hmc_end(int D, int DP, Chains ch, IterParams ip) {
  const int c = blockIdx.x, lane = threadIdx.x;
  const int ph = ch.phase[c];
  if (!(ph == 2 || (ph == 1 && ch.steps_left[c] == 0))) return;
  const long long it = ch.iter[c];
  double kin = 0.0;
  for (int d = lane; d < D; d += 64) { const double pl = ch.p[(size_t)c * DP + d]; kin = fma(pl, pl, kin); }
  const double Hp = -ch.trj.ljl[c] + 0.5 * wave_sum(kin);  // hmc.py:69
  const double ratio = -Hp + ch.Hcur[c];
  double u_acc;
  if (ip.z_in) {
    u_acc = ip.uacc_in[c];
  } else {
    double U0;
    rng_block(ip.seed, (unsigned long long)(ip.chain_offset + c), (uint32_t)it, 0x40000000u, U0, u_acc);
  }
  const bool accept = (ratio > 0.0) || (ratio > log(u_acc));  // hmc.py:77
  if (accept) hmc_copy(ch.cur, ch.trj, c, D, DP, lane);
  __syncthreads();
  if (ip.samples && it >= ip.burn_in && it - ip.burn_in < ip.S)
    for (int d = lane; d < D; d += 64)
      ip.samples[((size_t)c * ip.S + (size_t)(it - ip.burn_in)) * D + d] = ch.cur.w[(size_t)c * DP + d];
  if (lane == 0) {
    ch.Hprop[c] = Hp;
    if (accept) ch.accepted[c] += 1;
    ch.iter[c] = it + 1;
    ch.phase[c] = 0;
    if (it + 1 == ip.iter_limit && ip.done_count) atomicAdd(ip.done_count, 1);
  }
}

// ---------------------------------------------------------------------------------------------
// trace term from per-row weights: tr_d = sum_n R[c][n] x_nd (R = c_n h_n written by k_leverage_i8, or h_n alone with c_n taken here
// from the c tiles of the row pass: then neither the row pass writes nor the leverage GEMM's epilogue reads a natural-layout c).  Same MFMA mapping as the
// gradient product of k_rowpass: 16 chains per wave, row splits, partial sums to trpart (summed by k_reduce_tr).
// ---------------------------------------------------------------------------------------------
template <int NB>
__global__ __launch_bounds__(256) void k_trvec(DevData dd, int n_chains, int nsplit, const double* __restrict__ R, double* __restrict__ trpart,
                                               const d4* __restrict__ ctile = nullptr) {
  constexpr int DP = 16 * NB;
  const int lane = threadIdx.x & 63;
  const int c0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
  if (c0 >= n_chains) return;
  const int split = blockIdx.y;
  const int rr = lane >> 4, ci = lane & 15;
  const int cj = min(c0 + ci, n_chains - 1);
  // (32-row blocks, eight consecutive data rows per lane as in k_rowpass; the sum over rows does not care which tile a row is in)
  const int nb32 = dd.Mp / 32;
  const int per = (nb32 + nsplit - 1) / nsplit;
  const int B0 = split * per, B1 = min(nb32, B0 + per);
  const double* __restrict__ xr_p = dd.Xr + (size_t)(8 * rr) * DP + NB * ci;
  const double* __restrict__ rp = R + (size_t)cj * dd.Mp + 8 * rr;
  d4 T[NB];
#pragma unroll
  for (int I = 0; I < NB; ++I) T[I] = (d4){0.0, 0.0, 0.0, 0.0};
  for (int B = B0; B < B1; ++B) {
    const int n0 = 32 * B;
    d4 r0 = *(const d4*)(rp + n0), r1 = *(const d4*)(rp + n0 + 4);  // 64 contiguous bytes per lane
    if (ctile) {  // R holds h alone; c comes in the tile layout of k_rowpass<RP_F>: tile 2B = rows nl + 2r, tile 2B+1 = rows nl + 2r + 1
      const d4 cA = ctile[((size_t)(c0 >> 4) * (dd.Mp / 16) + 2 * B) * 64 + lane], cB = ctile[((size_t)(c0 >> 4) * (dd.Mp / 16) + 2 * B + 1) * 64 + lane];
      r0 *= (d4){cA[0], cB[0], cA[1], cB[1]};
      r1 *= (d4){cA[2], cB[2], cA[3], cB[3]};
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      double xb[NB];
#pragma unroll
      for (int I = 0; I < NB; ++I) xb[I] = xr_p[(size_t)(n0 + k) * DP + I];
      const double rv = k < 4 ? r0[k & 3] : r1[k & 3];
#pragma unroll
      for (int I = 0; I < NB; ++I) T[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(xb[I], rv, T[I], 0, 0, 0);
    }
  }
  if (c0 + ci < n_chains) {
    double* __restrict__ out = trpart + ((size_t)split * n_chains + c0 + ci) * DP;
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int d = NB * (rr + 4 * r) + I;
        if (d < dd.D) out[d] = T[I][r];
      }
  }
}

// ---------------------------------------------------------------------------------------------
// simplified manifold MALA (widening row 8f-4; authors_code/Bayes_Log_Reg/MCMC/BLR_mMALA_Simp.m:175-290): one point
// evaluation per transition with the RMHMC kernels, proposal and acceptance here.  Every chain does one transition
// per step, so there is no asynchronous bookkeeping.  Hcur carries LJL + log q(w'|w) of the current transition.
// ---------------------------------------------------------------------------------------------
// proposal w' = w + eps/2 G^-1 grad + sqrt(eps) G^-1 (L z)  ( = N(mean, eps G^-1), :217-219 )  ->  trj.w
__global__ __launch_bounds__(64) void k_mmala_begin(int D, int DP, Chains ch, IterParams ip, double eps, int full) {
  __shared__ double zs[RM_DMAX];
  __shared__ double ts[RM_DMAX];
  const int c = blockIdx.x, lane = threadIdx.x;
  const long long it = ch.iter[c];
  if (it >= ip.iter_limit) { if (lane == 0) ch.phase[c] = 0; return; }
  draw_normals(ip, c, it, D, lane, zs);
  const double* __restrict__ Lc = ch.cur.L + (size_t)c * DP * DP;
  const double* __restrict__ Gi = ch.cur.Ginv + (size_t)c * DP * DP;
  double zz = 0.0;
  for (int d = lane; d < D; d += 64) {  // t = L z + sqrt(eps)/2 ... : ts = sqrt(eps) L z + eps/2 grad, then w' = w + Ginv ts
    double s = 0.0;
    for (int j = 0; j <= d; ++j) s = fma(Lc[(size_t)d * DP + j], zs[j], s);
    // drift vector: gradient (BLR_mMALA_Simp.m) or gradient minus trace term (BLR_mMALA.m:231-233, see oracle/rmhmc_oracle.c mpoint_eval)
    const double dv = ch.cur.grad[(size_t)c * DP + d] - (full ? ch.cur.tr[(size_t)c * DP + d] : 0.0);
    ts[d] = sqrt(eps) * s + 0.5 * eps * dv;
    zz = fma(zs[d], zs[d], zz);
  }
  zz = wave_sum(zz);
  __syncthreads();
  double u[RM_DCH] = {0.0, 0.0, 0.0, 0.0};
  for (int j = 0; j < D; ++j) {
    const double tj = ts[j];
#pragma unroll
    for (int k = 0; k < RM_DCH; ++k) {
      const int d = lane + 64 * k;
      if (d < D) u[k] = fma(Gi[(size_t)j * DP + d], tj, u[k]);
    }
  }
#pragma unroll
  for (int k = 0; k < RM_DCH; ++k) {
    const int d = lane + 64 * k;
    if (d < D) ch.trj.w[(size_t)c * DP + d] = ch.cur.w[(size_t)c * DP + d] + u[k];
  }
  if (lane == 0) {
    ch.Hcur[c] = ch.cur.ljl[c] + ch.cur.hld[c] - 0.5 * zz;  // LJL + log q(w'|w)  (:227, constant -(D/2) log eps dropped)
    ch.status[c] = 0;
    ch.phase[c] = 1;
  }
}
// acceptance (:229-262) once the record at w' has been evaluated
__global__ __launch_bounds__(64) void k_mmala_end(int D, int DP, Chains ch, IterParams ip, double eps, int full) {
  __shared__ double ds[RM_DMAX];
  const int c = blockIdx.x, lane = threadIdx.x;
  if (ch.phase[c] != 1) return;
  const long long it = ch.iter[c];
  const double* __restrict__ Lp = ch.trj.L + (size_t)c * DP * DP;
  const double* __restrict__ Gi = ch.trj.Ginv + (size_t)c * DP * DP;
  // d = w' + eps/2 G'^-1 grad' - w
  double u[RM_DCH] = {0.0, 0.0, 0.0, 0.0};
  for (int j = 0; j < D; ++j) {
    const double gj = ch.trj.grad[(size_t)c * DP + j] - (full ? ch.trj.tr[(size_t)c * DP + j] : 0.0);
#pragma unroll
    for (int k = 0; k < RM_DCH; ++k) {
      const int d = lane + 64 * k;
      if (d < D) u[k] = fma(Gi[(size_t)j * DP + d], gj, u[k]);
    }
  }
#pragma unroll
  for (int k = 0; k < RM_DCH; ++k) {
    const int d = lane + 64 * k;
    if (d < D) ds[d] = ch.trj.w[(size_t)c * DP + d] + 0.5 * eps * u[k] - ch.cur.w[(size_t)c * DP + d];
  }
  __syncthreads();
  // |L'^T d|^2 = d' G' d
  double yy = 0.0;
  for (int j = lane; j < D; j += 64) {
    double s = 0.0;
    for (int i = j; i < D; ++i) s = fma(Lp[(size_t)i * DP + j], ds[i], s);
    yy = fma(s, s, yy);
  }
  yy = wave_sum(yy);
  const double q_rev = ch.trj.hld[c] - yy / (2.0 * eps);
  const double ratio = ch.trj.ljl[c] + q_rev - ch.Hcur[c];  // :251
  double u_acc;
  if (ip.z_in) {
    u_acc = ip.uacc_in[c];
  } else {
    double U0;
    rng_block(ip.seed, (unsigned long long)(ip.chain_offset + c), (uint32_t)it, 0x40000000u, U0, u_acc);
  }
  const bool accept = (ratio > 0.0) || (ratio > log(u_acc));
  __syncthreads();
  if (accept) copy_rec(ch.cur, ch.trj, c, D, DP, lane);
  __syncthreads();
  if (ip.samples && it >= ip.burn_in && it - ip.burn_in < ip.S)
    for (int d = lane; d < D; d += 64)
      ip.samples[((size_t)c * ip.S + (size_t)(it - ip.burn_in)) * D + d] = ch.cur.w[(size_t)c * DP + d];
  if (lane == 0) {
    ch.Hprop[c] = ratio;
    if (accept) ch.accepted[c] += 1;
    ch.iter[c] = it + 1;
    ch.phase[c] = 0;
  }
}

// ---------------------------------------------------------------------------------------------
// ESS on the device (widening row 8f-2): tools.CalculateESS(Samples, S-1), tools.py:32-74, for one series per
// wavefront.  samples[(c*S + s)*P + d]; the centred series lives in LDS; autocovariances are evaluated lag by lag
// (two per Geyer pair, tools.py:46-50) until the running-minimum pair sum (:54-60) turns non-positive, which is
// where the reference's "sum of the positive Gammas" (:62-67) ends.  Also returns mean and population variance.
// nfft = 0: linear autocovariances (the MATLAB original, ac.m:78: a 2^(k+1)-point FFT never wraps).  nfft > 0
// (RMHMC_FLAG_ESS_WRAP): the reference's Python translation takes an FFT of length nFFT = nextpow2(S)+1 (tools.py:16-23), whose
// circular autocorrelation at lag l is the linear one at l PLUS the linear one at nFFT - l; reproduced term by term.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_ess(const double* __restrict__ samples, long long S, int P, double* __restrict__ ess_out,
                                            double* __restrict__ mean_out, double* __restrict__ var_out, long long nfft) {
  extern __shared__ __attribute__((aligned(16))) double ess_x[];
  const long long idx = blockIdx.x;
  const long long c = idx / P;
  const int d = (int)(idx % P);
  const int lane = threadIdx.x;
  const double* __restrict__ x = samples + (size_t)c * S * P + d;
  double m = 0.0;
  for (long long s = lane; s < S; s += 64) m += x[(size_t)s * P];
  m = wave_sum(m) / (double)S;
  double c0 = 0.0;
  for (long long s = lane; s < S; s += 64) {
    const double v = x[(size_t)s * P] - m;
    ess_x[s] = v;
    c0 = fma(v, v, c0);
  }
  c0 = wave_sum(c0);
  __syncthreads();
  double prev = INFINITY, sum = 0.0;
  const long long half = S / 2;
  for (long long j = 0; j < half; ++j) {
    const long long l0 = 2 * j, l1 = 2 * j + 1;
    double a = 0.0, b = 0.0;
    for (long long s = lane; s + l0 < S; s += 64) a = fma(ess_x[s], ess_x[s + l0], a);
    for (long long s = lane; s + l1 < S; s += 64) b = fma(ess_x[s], ess_x[s + l1], b);
    if (nfft > 0) {  // wrapped partners of the two lags
      const long long w0 = nfft - l0, w1 = nfft - l1;
      if (l0 > 0) for (long long s = lane; s + w0 < S; s += 64) a = fma(ess_x[s], ess_x[s + w0], a);
      for (long long s = lane; s + w1 < S; s += 64) b = fma(ess_x[s], ess_x[s + w1], b);
    }
    double g = wave_sum(a + b) / c0;
    if (g > prev) g = prev;
    if (!(g > 0.0)) break;
    sum += g;
    prev = g;
  }
  double mono = -1.0 + 2.0 * sum;
  if (mono < 1.0) mono = 1.0;
  if (lane == 0) {
    if (ess_out) ess_out[idx] = (c0 > 0.0) ? (double)S / mono : NAN;
    if (mean_out) mean_out[idx] = m;
    if (var_out) var_out[idx] = c0 / (double)S;
  }
}

// generic fills
__global__ void k_fill_int(int* p, int v, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
__global__ void k_fill_ll(long long* p, long long v, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
// *out = min(*out, min_c iter[c]): completed transitions of the slowest chain (the sampler's host loop sizes its next batch of global
// steps from it: a transition takes at least one step, so the slowest chain needs at least limit - min more)
__global__ void k_min_iter(const long long* __restrict__ iter, size_t n, unsigned long long* __restrict__ out) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  unsigned long long m = i < n ? (unsigned long long)iter[i] : ~0ull;
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long x = __shfl_xor(m, o);
    m = x < m ? x : m;
  }
  if ((threadIdx.x & 63) == 0) atomicMin(out, m);
}
// T[c] = leapfrog steps chain chain_offset + c executes in transitions it0 .. it1-1: the trajectory lengths are drawn independently of
// the state (RandomStep = ceil(rand() L), rmhmc.py:89), so they are known before the run (same draw as iter_begin_dev)
__global__ void k_traj_steps(unsigned long long seed, long long chain_offset, int L, long long it0, long long it1, size_t n,
                             long long* __restrict__ T) {
  size_t c = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (c >= n) return;
  long long t = 0;
  for (long long it = it0; it < it1; ++it) {
    double u_len, ua;
    rng_block(seed, (unsigned long long)(chain_offset + (long long)c), (uint32_t)it, 0x40000000u, u_len, ua);
    t += (long long)ceil(u_len * (double)L);
  }
  T[c] = t;
}
// dst[orig[i]] = src[i]
__global__ void k_scatter_ll(long long* __restrict__ dst, const long long* __restrict__ src, const int* __restrict__ orig, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) dst[orig[i]] = src[i];
}
// a[i] = b[i] - a[i]
__global__ void k_sub_ll(long long* __restrict__ a, const long long* __restrict__ b, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) a[i] = b[i] - a[i];
}
// phase[c] = (nsteps[c] > 0) for the unit leapfrog API
__global__ void k_set_leapfrog(int n, const int* __restrict__ nsteps, const int* __restrict__ dir, Chains ch) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  ch.steps_left[c] = nsteps[c];
  ch.tau[c] = (double)dir[c];
  ch.phase[c] = nsteps[c] > 0 ? 1 : 3;  // 3 = parked
}
// after each unit leapfrog step: chains that exhausted their steps are parked
__global__ void k_park_finished(int n, Chains ch) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  if (ch.phase[c] == 1 && ch.steps_left[c] <= 0) ch.phase[c] = 3;
}
