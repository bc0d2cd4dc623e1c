// large_d.hip.h — the large-D path (64 < D <= 256, BASELINE config 5: "G spills LDS -> tiled Cholesky").
//
// The D columns are cut into nbk = DP/64 blocks of 64 (DP = 64*ceil(D/64), zero padded).  Everything that is a
// pass over X re-uses the NB = 4 matrix-core mappings of kernels.hip.h on 64-column blocks:
//   * k_assemble_pair / k_leverage_pair : one wavefront per (chain, block pair bA <= bB): the 64x64 block G_AB of the
//     metric (16 MFMA tiles, 10 on the diagonal), and the pair's contribution x_A' Ghat_AB x_B to every leverage;
//   * k_rowpass_big / k_mompass_big / k_trace_big : 16 chains per workgroup, the 4 waves of a workgroup own one
//     column block each; the K = D reduction of F = X W is finished through LDS (one barrier per 16 rows, parity
//     double-buffered), the X' R products are block-local;
//   * k_chol_big / k_inverse_big : one 256-thread workgroup per chain, right-looking blocked Cholesky of the DPxDP
//     matrix in global memory (L2 resident) with 64x64 blocks staged through LDS and every block product on the matrix
//     cores (C = A B' with both operands as row-major LDS images; wave w owns tile row w of the 64x64 result);
//     inverse via the block-triangular W = L^-1 and G^-1 = W' W.
// Vector kernels are the dimension-strided ones of kernels.hip.h.
#pragma once
#include "kernels.hip.h"

#define LD_BK 64              // column block
#define LD_LD 66              // LDS leading dimension of a staged 64x64 block

__device__ __forceinline__ void pair_from_index(int pi, int& bA, int& bB) {  // pi = bB(bB+1)/2 + bA, bA <= bB
  bB = 0;
  while ((bB + 1) * (bB + 2) / 2 <= pi) ++bB;
  bA = pi - bB * (bB + 1) / 2;
}

// ---------------------------------------------------------------------------------------------------------------
// row pass, D > 64.  Grid (ceil(n/16), nsplit), 256 threads: wave w = column block w.
// ---------------------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void k_rowpass_big(DevData dd, int n_chains, int nsplit, int nbk, const int* __restrict__ phase,
                                                     const double* __restrict__ wq, double* __restrict__ out0,
                                                     double* __restrict__ out2, double* __restrict__ gpart,
                                                     double* __restrict__ ljl_part, d4* __restrict__ ctile) {
  constexpr int NB = 4, KK = 16;
  __shared__ double red[2][4][64][4];
  const int DP = dd.DP;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c0 = blockIdx.x * 16;
  const int split = blockIdx.y;
  const int rr = lane >> 4, ci = lane & 15;
  const int cj = min(c0 + ci, n_chains - 1);
  const bool live = (c0 + ci < n_chains) && (phase[cj] == 1);
  const bool act = w < nbk;
  double Wb[KK];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) Wb[kk] = act ? wq[(size_t)cj * DP + 64 * w + 4 * kk + rr] : 0.0;
  const int nb16 = dd.Mp / 16;
  const int per = (nb16 + nsplit - 1) / nsplit;
  const int b0 = split * per, b1 = min(nb16, b0 + per);
  const double* __restrict__ xt_p = dd.Xt + (size_t)(64 * w + rr) * dd.Mp + ci;
  const double* __restrict__ xr_p = dd.Xr + (size_t)rr * DP + 64 * w + NB * ci;
  double lj = 0.0;
  d4 Gr[NB];
#pragma unroll
  for (int I = 0; I < NB; ++I) Gr[I] = (d4){0.0, 0.0, 0.0, 0.0};
  int par = 0;
  for (int b = b0; b < b1; ++b, par ^= 1) {
    const int n0 = b * 16;
    d4 F = (d4){0.0, 0.0, 0.0, 0.0};
    double xb[4][NB];
    if (act) {
      double A[KK];
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) A[kk] = xt_p[(size_t)(4 * kk) * dd.Mp + n0];
      if (MODE != RP_V) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int I = 0; I < NB; ++I) xb[r][I] = xr_p[(size_t)(n0 + 4 * r) * DP + I];
      }
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) F = __builtin_amdgcn_mfma_f64_16x16x4f64(A[kk], Wb[kk], F, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) red[par][w][lane][r] = F[r];
    }
    __syncthreads();
    d4 cc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double f = 0.0;
      for (int ww = 0; ww < nbk; ++ww) f += red[par][ww][lane][r];
      const int n = n0 + rr + 4 * r;
      const double em = exp(-f);
      const double p = 1.0 / (1.0 + em);
      const double v = p * (1.0 - p);
      cc[r] = v * (1.0 - 2.0 * p);
      const size_t o = (size_t)cj * dd.Mp + n;
      if (MODE == RP_V) {
        if (live && w == 0) out0[o] = v;
      } else {
        const double ef = exp(f);
        const double tn = dd.t[n];
        if (MODE == RP_F && live && w == 0) {
          out0[o] = v;
          out2[o] = v * (1.0 - 2.0 * p);
        }
        if (w == 0 && n < dd.M) lj += f * tn - log(1.0 + ef);
        const double rn = tn - ef / (1.0 + ef);
        if (act) {
#pragma unroll
          for (int I = 0; I < NB; ++I) Gr[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(xb[r][I], rn, Gr[I], 0, 0, 0);
        }
      }
    }
    // c in the accumulator layout, for k_mompass_big<2> at the same position (as k_rowpass / k_mompass for D <= 64)
    if (MODE == RP_F && ctile && live && w == 0) ctile[((size_t)blockIdx.x * nb16 + b) * 64 + lane] = cc;
  }
  if (MODE != RP_V) {
    lj = col4_sum(lj);
    if (live && w == 0 && rr == 0) ljl_part[(size_t)cj * nsplit + split] = lj;
    if (live && act) {
      double* __restrict__ out = gpart + ((size_t)split * n_chains + c0 + ci) * DP + 64 * w;
#pragma unroll
      for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int d = NB * (rr + 4 * r) + I;
          if (64 * w + d < dd.D) out[d] = Gr[I][r];
        }
    }
  }
}

// quadratic-term pass (k_mompass), D > 64.  CM as in k_mompass: 0 c from F, 1 the same and stored (wave 0 writes the tile), 2 c loaded: no
// F product, no exp, and half the cross-wave reduction.
template <int CM>
__global__ __launch_bounds__(256) void k_mompass_big(DevData dd, int n_chains, int nsplit, int nbk, const double* __restrict__ wq,
                                                     const double* __restrict__ uq, double* __restrict__ qpart, d4* __restrict__ ctile) {
  constexpr int NB = 4, KK = 16;
  __shared__ double red[2][4][64][8];
  const int DP = dd.DP;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c0 = blockIdx.x * 16;
  const int split = blockIdx.y;
  const int rr = lane >> 4, ci = lane & 15;
  const int cj = min(c0 + ci, n_chains - 1);
  const bool act = w < nbk;
  double Wb[KK], Ub[KK];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) {
    Wb[kk] = (act && CM != 2) ? wq[(size_t)cj * DP + 64 * w + 4 * kk + rr] : 0.0;
    Ub[kk] = act ? uq[(size_t)cj * DP + 64 * w + 4 * kk + rr] : 0.0;
  }
  d4* __restrict__ ct = ctile + (size_t)blockIdx.x * (dd.Mp / 16) * 64 + lane;
  const int nb16 = dd.Mp / 16;
  const int per = (nb16 + nsplit - 1) / nsplit;
  const int b0 = split * per, b1 = min(nb16, b0 + per);
  const double* __restrict__ xt_p = dd.Xt + (size_t)(64 * w + rr) * dd.Mp + ci;
  const double* __restrict__ xr_p = dd.Xr + (size_t)rr * DP + 64 * w + NB * ci;
  d4 Q[NB];
#pragma unroll
  for (int I = 0; I < NB; ++I) Q[I] = (d4){0.0, 0.0, 0.0, 0.0};
  int par = 0;
  for (int b = b0; b < b1; ++b, par ^= 1) {
    const int n0 = b * 16;
    double xb[4][NB];
    if (act) {
      double A[KK];
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) A[kk] = xt_p[(size_t)(4 * kk) * dd.Mp + n0];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int I = 0; I < NB; ++I) xb[r][I] = xr_p[(size_t)(n0 + 4 * r) * DP + I];
      d4 F = (d4){0.0, 0.0, 0.0, 0.0}, S = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        if (CM != 2) F = __builtin_amdgcn_mfma_f64_16x16x4f64(A[kk], Wb[kk], F, 0, 0, 0);
        S = __builtin_amdgcn_mfma_f64_16x16x4f64(A[kk], Ub[kk], S, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) { if (CM != 2) red[par][w][lane][r] = F[r]; red[par][w][lane][4 + r] = S[r]; }
    }
    d4 cc;
    if (CM == 2 && act) cc = ct[(size_t)b * 64];
    __syncthreads();
    if (act) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double f = 0.0, s = 0.0;
        for (int ww = 0; ww < nbk; ++ww) { if (CM != 2) f += red[par][ww][lane][r]; s += red[par][ww][lane][4 + r]; }
        if (CM != 2) {
          const double em = exp(-f);
          const double p = 1.0 / (1.0 + em);
          cc[r] = p * (1.0 - p) * (1.0 - 2.0 * p);
        }
        const double R = cc[r] * s * s;
#pragma unroll
        for (int I = 0; I < NB; ++I) Q[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(xb[r][I], R, Q[I], 0, 0, 0);
      }
      if (CM == 1 && w == 0) ct[(size_t)b * 64] = cc;
    }
  }
  if (act && c0 + ci < n_chains) {
    double* __restrict__ out = qpart + ((size_t)split * n_chains + c0 + ci) * DP + 64 * w;
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int d = NB * (rr + 4 * r) + I;
        if (64 * w + d < dd.D) out[d] = Q[I][r];
      }
  }
}

// trace term tr_d = sum_n c_n h_n x_nd with h_n = sum over block pairs of hpart (rmhmc.py:67-77,148-156), D > 64
__global__ __launch_bounds__(256) void k_trace_big(DevData dd, int n_chains, int nsplit, int nbk, int npairs,
                                                   const double* __restrict__ crow, const double* __restrict__ hpart,
                                                   double* __restrict__ trpart) {
  constexpr int NB = 4;
  const int DP = dd.DP;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (w >= nbk) return;
  const int c0 = blockIdx.x * 16;
  const int split = blockIdx.y;
  const int rr = lane >> 4, ci = lane & 15;
  const int cj = min(c0 + ci, n_chains - 1);
  const int nb16 = dd.Mp / 16;
  const int per = (nb16 + nsplit - 1) / nsplit;
  const int b0 = split * per, b1 = min(nb16, b0 + per);
  const double* __restrict__ xr_p = dd.Xr + (size_t)rr * DP + 64 * w + NB * ci;
  d4 T[NB];
#pragma unroll
  for (int I = 0; I < NB; ++I) T[I] = (d4){0.0, 0.0, 0.0, 0.0};
  for (int b = b0; b < b1; ++b) {
    const int n0 = b * 16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const size_t o = (size_t)cj * dd.Mp + n0 + rr + 4 * r;
      double h = 0.0;
      for (int pi = 0; pi < npairs; ++pi) h += hpart[(size_t)pi * n_chains * dd.Mp + o];
      const double R = crow[o] * h;
#pragma unroll
      for (int I = 0; I < NB; ++I)
        T[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(xr_p[(size_t)(n0 + 4 * r) * DP + I], R, T[I], 0, 0, 0);
    }
  }
  if (c0 + ci < n_chains) {
    double* __restrict__ out = trpart + ((size_t)split * n_chains + c0 + ci) * DP + 64 * w;
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int d = NB * (rr + 4 * r) + I;
        if (64 * w + d < dd.D) out[d] = T[I][r];
      }
  }
}

// trj.tr = sum of the row-split partials of k_trace_big
__global__ __launch_bounds__(64) void k_reduce_tr(int D, int DP, Chains ch, const double* __restrict__ trpart, int nsplit) {
  const int c = blockIdx.x, lane = threadIdx.x;
  if (ch.phase[c] != 1) return;
  for (int d = lane; d < D; d += 64) {
    double s = 0.0;
    for (int sp = 0; sp < nsplit; ++sp) s += trpart[((size_t)sp * ch.n + c) * DP + d];
    ch.trj.tr[(size_t)c * DP + d] = s;
  }
}

// gradient and log joint from the row-split partials of k_rowpass_big<RP_F> (what k_factor_full does for D <= 64)
__global__ __launch_bounds__(64) void k_finish_big(DevData dd, Chains ch, int nsplit) {
  const int c = blockIdx.x, lane = threadIdx.x;
  if (ch.phase[c] != 1) return;
  hmc_finish_eval(dd, ch, c, lane, nsplit);
}

// ---------------------------------------------------------------------------------------------------------------
// metric blocks and leverage contributions, one wavefront per (chain, block pair)
// ---------------------------------------------------------------------------------------------------------------
// DIAG: blockIdx.y = diagonal block (10 tiles); otherwise blockIdx.y enumerates the pairs bA < bB (16 tiles).  Two
// instantiations instead of a run-time branch: with both tile sets in one kernel the register allocation doubles.
template <bool DIAG>
__global__ __launch_bounds__(256) void k_assemble_pair(DevData dd, int n_chains, const int* __restrict__ phase,
                                                       const double* __restrict__ vrow, double* __restrict__ Gq) {
  constexpr int NB = 4;
  const int DP = dd.DP;
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= n_chains) return;
  if (phase[c] != 1) return;
  int bA, bB;
  if (DIAG) {
    bA = bB = blockIdx.y;
  } else {  // pair index p -> (bA < bB): p = bB(bB-1)/2 + bA
    bB = 1;
    while (bB * (bB + 1) / 2 <= (int)blockIdx.y) ++bB;
    bA = blockIdx.y - bB * (bB - 1) / 2;
  }
  constexpr bool diag = DIAG;
  const int rr = lane >> 4, ci = lane & 15;
  const double* __restrict__ xpA = dd.Xr + (size_t)rr * DP + 64 * bA + NB * ci;
  const double* __restrict__ xpB = dd.Xr + (size_t)rr * DP + 64 * bB + NB * ci;
  const double* __restrict__ vp = vrow + (size_t)c * dd.Mp + rr;
  d4 acc[NB][NB];
#pragma unroll
  for (int I = 0; I < NB; ++I)
#pragma unroll
    for (int J = 0; J < NB; ++J) acc[I][J] = (d4){0.0, 0.0, 0.0, 0.0};
  // software pipeline over groups of 2 MFMA k-chunks (8 data rows): the operands of the next group are requested
  // before the 2*16 MFMAs of the current one (Mp is a multiple of 64, so groups come in pairs)
  double xaA[2][NB], xbA[2][NB], vA[2], xaB[2][NB], xbB[2][NB], vB[2];
  auto load_group = [&](double (&xa)[2][NB], double (&xb)[2][NB], double (&vv)[2], int n1) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
      for (int I = 0; I < NB; ++I) {
        xa[q][I] = xpA[(size_t)(n1 + 4 * q) * DP + I];
        xb[q][I] = xpB[(size_t)(n1 + 4 * q) * DP + I];
      }
      vv[q] = vp[n1 + 4 * q];
    }
  };
  auto compute_group = [&](const double (&xa)[2][NB], const double (&xb)[2][NB], const double (&vv)[2]) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      double xs[NB];
#pragma unroll
      for (int I = 0; I < NB; ++I) xs[I] = xa[q][I] * vv[q];
      if (diag) {
#pragma unroll
        for (int I = 0; I < NB; ++I)
#pragma unroll
          for (int J = I; J < NB; ++J) acc[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(xs[I], xb[q][J], acc[I][J], 0, 0, 0);
      } else {
#pragma unroll
        for (int I = 0; I < NB; ++I)
#pragma unroll
          for (int J = 0; J < NB; ++J) acc[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(xs[I], xb[q][J], acc[I][J], 0, 0, 0);
      }
    }
  };
  load_group(xaA, xbA, vA, 0);
  for (int n1 = 0; n1 < dd.Mp; n1 += 16) {
    load_group(xaB, xbB, vB, n1 + 8);
    compute_group(xaA, xbA, vA);
    if (n1 + 16 < dd.Mp) load_group(xaA, xbA, vA, n1 + 16);
    compute_group(xaB, xbB, vB);
  }
  double* __restrict__ G = Gq + (size_t)c * DP * DP;
#pragma unroll
  for (int I = 0; I < NB; ++I)
#pragma unroll
    for (int J = 0; J < NB; ++J) {
      if (diag && J < I) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 64 * bA + NB * (rr + 4 * r) + I;
        const int col = 64 * bB + NB * ci + J;
        double val = acc[I][J][r];
        if (row == col) val += dd.inv_alpha;
        if (!diag || I != J || row >= col) {  // keep the lower copy of a diagonal tile and mirror: exactly symmetric
          G[(size_t)row * DP + col] = val;
          G[(size_t)col * DP + row] = val;
        }
      }
    }
}

// hpart[pair][chain][n] = x_A' Ghat_AB x_B  (Ghat_AB = 2 G^-1_AB for bA < bB; the folded upper triangle on the diagonal)
// DIAG: blockIdx.y = diagonal block, hpart slot blockIdx.y; otherwise blockIdx.y enumerates bA < bB, slot nbk + blockIdx.y.
template <bool DIAG>
__global__ __launch_bounds__(256) void k_leverage_pair(DevData dd, int n_chains, int nbk, const int* __restrict__ phase,
                                                       const double* __restrict__ Ginv, double* __restrict__ hpart) {
  constexpr int NB = 4;
  const int DP = dd.DP;
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= n_chains) return;
  if (phase[c] != 1) return;
  int bA, bB;
  if (DIAG) {
    bA = bB = blockIdx.y;
  } else {
    bB = 1;
    while (bB * (bB + 1) / 2 <= (int)blockIdx.y) ++bB;
    bA = blockIdx.y - bB * (bB - 1) / 2;
  }
  const int slot = DIAG ? (int)blockIdx.y : nbk + (int)blockIdx.y;
  const int rr = lane >> 4, ci = lane & 15;
  const double* __restrict__ Gi = Ginv + (size_t)c * DP * DP;
  double Gv[NB][4][NB];
#pragma unroll
  for (int I = 0; I < NB; ++I)
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int J = 0; J < NB; ++J) {
        const int row = 64 * bA + NB * (4 * s + rr) + I, col = 64 * bB + NB * ci + J;
        if (DIAG) Gv[I][s][J] = (J >= I) ? Gi[(size_t)row * DP + col] * (I == J ? 1.0 : 2.0) : 0.0;
        else Gv[I][s][J] = 2.0 * Gi[(size_t)row * DP + col];
      }
  const double* __restrict__ xpA = dd.Xr + (size_t)ci * DP + 64 * bA + NB * rr;
  const double* __restrict__ xpB = dd.Xr + (size_t)ci * DP + 64 * bB + NB * rr;
  double* __restrict__ hp_out = hpart + ((size_t)slot * n_chains + c) * dd.Mp;
  // software pipeline: the operand rows of the next 16-row block are requested before the MFMAs of the current one
  double XA0[4][NB], XA1[4][NB];
  auto load_A = [&](double (&X)[4][NB], int n0) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int I = 0; I < NB; ++I) X[s][I] = xpA[(size_t)n0 * DP + NB * 4 * s + I];
  };
  auto compute_block = [&](const double (&XA)[4][NB], int n0) {
    double XB[4][NB];
    if (!DIAG) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int I = 0; I < NB; ++I) XB[s][I] = xpB[(size_t)n0 * DP + NB * 4 * s + I];
    }
    d4 Y[NB];
#pragma unroll
    for (int J = 0; J < NB; ++J) {
      Y[J] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int I = 0; I < NB; ++I) {
        if (DIAG && I > J) continue;
#pragma unroll
        for (int s = 0; s < 4; ++s) Y[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(Gv[I][s][J], XA[s][I], Y[J], 0, 0, 0);
      }
    }
    double hp = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int J = 0; J < NB; ++J) hp = fma(Y[J][r], DIAG ? XA[r][J] : XB[r][J], hp);
    const double h = col4_sum(hp);
    if (rr == 0) hp_out[n0 + ci] = h;
  };
  load_A(XA0, 0);
  for (int n0 = 0; n0 < dd.Mp; n0 += 32) {
    load_A(XA1, n0 + 16);
    compute_block(XA0, n0);
    if (n0 + 32 < dd.Mp) load_A(XA0, n0 + 32);
    compute_block(XA1, n0 + 16);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// blocked dense algebra: one 256-thread workgroup per chain
// ---------------------------------------------------------------------------------------------------------------
// 64x64 block of a row-major matrix (leading dimension ld) -> LDS image, optionally transposed
__device__ __forceinline__ void stage_block(double* dst, const double* __restrict__ src, int ld, bool transpose) {
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    const int r = e >> 6, cc = e & 63;
    const double v = src[(size_t)r * ld + cc];
    dst[transpose ? (cc * LD_LD + r) : (r * LD_LD + cc)] = v;
  }
}
// acc[J] += (A B')[16w + ..][16J + ..]  for wave w; A, B row-major LDS images (64 x 64, K along the row)
__device__ __forceinline__ void block_gemm_nt(const double* A, const double* B, int w, int lane, d4 (&acc)[4]) {
  const int rr = lane >> 4, ci = lane & 15;
  const double* ap = A + (16 * w + ci) * LD_LD + rr;
#pragma unroll 4
  for (int s = 0; s < 16; ++s) {
    const double a = ap[4 * s];
#pragma unroll
    for (int J = 0; J < 4; ++J) {
      const double b = B[(16 * J + ci) * LD_LD + 4 * s + rr];
      acc[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[J], 0, 0, 0);
    }
  }
}
// element (16w + rr + 4r, 16J + ci) of a 64x64 block <-> acc[J][r]
__device__ __forceinline__ void store_block(double* __restrict__ dst, int ld, int w, int lane, const d4 (&acc)[4], double sign) {
  const int rr = lane >> 4, ci = lane & 15;
#pragma unroll
  for (int J = 0; J < 4; ++J)
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[(size_t)(16 * w + rr + 4 * r) * ld + 16 * J + ci] = sign * acc[J][r];
}
__device__ __forceinline__ void load_block_acc(const double* __restrict__ src, int ld, int w, int lane, d4 (&acc)[4]) {
  const int rr = lane >> 4, ci = lane & 15;
#pragma unroll
  for (int J = 0; J < 4; ++J)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[J][r] = src[(size_t)(16 * w + rr + 4 * r) * ld + 16 * J + ci];
}
__device__ __forceinline__ void zero_acc(d4 (&acc)[4]) {
#pragma unroll
  for (int J = 0; J < 4; ++J) acc[J] = (d4){0.0, 0.0, 0.0, 0.0};
}
// W = L^-1 (lower triangular, row-major image) from the Cholesky factor L in LDS; lane = row, one wavefront.
// Row i of W L = I by back substitution over the columns j = i-1 .. 0 (rows of Wm zero beyond the diagonal).
__device__ __forceinline__ void tri_inverse_lds(const double* L, double* Wm, int lane, double rdiag) {
  double* wrow = Wm + lane * LD_LD;
  for (int m = 0; m < 64; ++m) wrow[m] = (m == lane) ? rdiag : 0.0;
  __builtin_amdgcn_wave_barrier();
  for (int j = 62; j >= 0; --j) {
    // s = sum_{m=j+1}^{63} W[lane][m] L[m][j]   (terms with m > lane are zero)
    double s0 = 0.0, s1 = 0.0;
    int m = j + 1;
    if ((m & 1) && m < 64) { s0 = fma(wrow[m], L[m * LD_LD + j], s0); ++m; }
    for (; m + 4 <= 64; m += 4) {
      const double2 w01 = lds2(wrow + m), w23 = lds2(wrow + m + 2);
      s0 = fma(w01.x, L[m * LD_LD + j], s0); s1 = fma(w01.y, L[(m + 1) * LD_LD + j], s1);
      s0 = fma(w23.x, L[(m + 2) * LD_LD + j], s0); s1 = fma(w23.y, L[(m + 3) * LD_LD + j], s1);
    }
    for (; m < 64; ++m) s0 = fma(wrow[m], L[m * LD_LD + j], s0);
    const double rj = rdlane(rdiag, j);
    if (lane > j) wrow[j] = -(s0 + s1) * rj;
    __builtin_amdgcn_wave_barrier();
  }
}

// Blocked right-looking Cholesky of Gq[c] (DP x DP, in place, lower blocks), diagonal-block inverses to Wd[c][k].
// MODE 0 (position iterate, rmhmc.py:116-122): then solve G u = p and set wq = w + tau*eps/2 (u0 + u).
// MODE 1 (new point, rmhmc.py:137,171): copy L to trj.L and store the half log-determinant.
template <int MODE>
__global__ __launch_bounds__(256) void k_chol_big(DevData dd, Chains ch, int nbk, double* __restrict__ Wd, double eps) {
  __shared__ __attribute__((aligned(16))) double A[64 * LD_LD];
  __shared__ __attribute__((aligned(16))) double B[64 * LD_LD];
  __shared__ double vec[RM_DMAX];
  __shared__ double hsum;
  __shared__ int sbad;
  const int c = blockIdx.x;
  if (ch.phase[c] != 1) return;
  const int D = dd.D, DP = dd.DP;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  double* __restrict__ G = ch.Gq + (size_t)c * DP * DP;
  double* __restrict__ Wc = Wd + (size_t)c * nbk * 64 * 64;
  if (threadIdx.x == 0) { hsum = 0.0; sbad = 0; }
  __syncthreads();
  for (int k = 0; k < nbk; ++k) {
    stage_block(A, G + (size_t)(64 * k) * DP + 64 * k, DP, false);
    __syncthreads();
    if (w == 0) {
      double rdiag;
      const int bad = chol_lds_blk<4>(A, 64, lane, rdiag);
      // zero the strict upper triangle of the factor image (it is stored and multiplied as a full block)
      for (int j = lane + 1; j < 64; ++j) A[lane * LD_LD + j] = 0.0;
      __builtin_amdgcn_wave_barrier();
      tri_inverse_lds(A, B, lane, rdiag);
      const double ld = (64 * k + lane < D) ? -log(rdiag) : 0.0;  // zero-padded dimensions are not part of |G|
      const double hs = wave_sum(ld);
      if (lane == 0) { hsum += hs; if (bad) sbad = 1; }
    }
    __syncthreads();
    // L_kk and W_kk to memory
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
      const int r = e >> 6, cc = e & 63;
      G[(size_t)(64 * k + r) * DP + 64 * k + cc] = A[r * LD_LD + cc];
      Wc[(size_t)k * 4096 + e] = B[r * LD_LD + cc];
    }
    __syncthreads();
    // panel: L_ik = G_ik W_kk'
    for (int i = k + 1; i < nbk; ++i) {
      stage_block(A, G + (size_t)(64 * i) * DP + 64 * k, DP, false);
      __syncthreads();
      d4 acc[4];
      zero_acc(acc);
      block_gemm_nt(A, B, w, lane, acc);
      store_block(G + (size_t)(64 * i) * DP + 64 * k, DP, w, lane, acc, 1.0);
      __syncthreads();
    }
    // trailing update: G_ij -= L_ik L_jk'
    for (int i = k + 1; i < nbk; ++i)
      for (int j = k + 1; j <= i; ++j) {
        stage_block(A, G + (size_t)(64 * i) * DP + 64 * k, DP, false);
        stage_block(B, G + (size_t)(64 * j) * DP + 64 * k, DP, false);
        __syncthreads();
        d4 acc[4], cur[4];
        zero_acc(acc);
        block_gemm_nt(A, B, w, lane, acc);
        load_block_acc(G + (size_t)(64 * i) * DP + 64 * j, DP, w, lane, cur);
#pragma unroll
        for (int J = 0; J < 4; ++J) cur[J] -= acc[J];
        store_block(G + (size_t)(64 * i) * DP + 64 * j, DP, w, lane, cur, 1.0);
        __syncthreads();
      }
  }
  // strictly upper blocks of the factor are zero
  for (int i = 0; i < nbk; ++i)
    for (int j = i + 1; j < nbk; ++j)
      for (int e = threadIdx.x; e < 64 * 64; e += 256) G[(size_t)(64 * i + (e >> 6)) * DP + 64 * j + (e & 63)] = 0.0;
  __threadfence_block();
  __syncthreads();
  if (MODE == 1) {
    double* __restrict__ Lg = ch.trj.L + (size_t)c * DP * DP;
    for (int r = 0; r < D; ++r)
      for (int d = threadIdx.x; d < D; d += 256) Lg[(size_t)r * DP + d] = G[(size_t)r * DP + d];
    if (threadIdx.x == 0) {
      ch.trj.hld[c] = hsum;
      if (sbad) ch.status[c] |= 1;
    }
  } else {
    // solve (L L') u = p by block substitution with the stored diagonal inverses; thread t = element t
    const int t = threadIdx.x;
    vec[t] = (t < D) ? ch.p[(size_t)c * DP + t] : 0.0;
    __syncthreads();
    for (int k = 0; k < nbk; ++k) {  // forward: y_k = W_kk (b_k - sum_{j<k} L_kj y_j)
      double s = 0.0;
      if (t < 64) {
        s = vec[64 * k + t];
        const double* __restrict__ Lrow = G + (size_t)(64 * k + t) * DP;
        for (int m = 0; m < 64 * k; ++m) s = fma(-Lrow[m], vec[m], s);
      }
      __syncthreads();
      if (t < 64) vec[64 * k + t] = s;
      __syncthreads();
      double y = 0.0;
      if (t < 64) {
        const double* __restrict__ Wk = Wc + (size_t)k * 4096 + (size_t)t * 64;
        for (int m = 0; m <= t; ++m) y = fma(Wk[m], vec[64 * k + m], y);
      }
      __syncthreads();
      if (t < 64) vec[64 * k + t] = y;
      __syncthreads();
    }
    for (int k = nbk - 1; k >= 0; --k) {  // backward: x_k = W_kk' (y_k - sum_{j>k} L_jk' x_j)
      double s = 0.0;
      if (t < 64) {
        s = vec[64 * k + t];
        for (int m = 64 * (k + 1); m < 64 * nbk; ++m) s = fma(-G[(size_t)m * DP + 64 * k + t], vec[m], s);
      }
      __syncthreads();
      if (t < 64) vec[64 * k + t] = s;
      __syncthreads();
      double x = 0.0;
      if (t < 64) {
        const double* __restrict__ Wk = Wc + (size_t)k * 4096;
        for (int m = t; m < 64; ++m) x = fma(Wk[(size_t)m * 64 + t], vec[64 * k + m], x);
      }
      __syncthreads();
      if (t < 64) vec[64 * k + t] = x;
      __syncthreads();
    }
    if (t < D)
      ch.wq[(size_t)c * DP + t] = ch.trj.w[(size_t)c * DP + t] + ch.tau[c] * (eps * 0.5) * (ch.u0[(size_t)c * DP + t] + vec[t]);
    if (sbad && t == 0) ch.status[c] |= 1;
  }
}

// G^-1 from the factor: W = L^-1 block by block into Gq (workspace), then trj.Ginv = W' W.
// L is read from trj.L, the diagonal-block inverses from Wd (both written by k_chol_big<1>).
__global__ __launch_bounds__(256) void k_inverse_big(DevData dd, Chains ch, int nbk, const double* __restrict__ Wd) {
  __shared__ __attribute__((aligned(16))) double A[64 * LD_LD];
  __shared__ __attribute__((aligned(16))) double B[64 * LD_LD];
  const int c = blockIdx.x;
  if (ch.phase[c] != 1) return;
  const int DP = dd.DP;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const double* __restrict__ L = ch.trj.L + (size_t)c * DP * DP;
  double* __restrict__ W = ch.Gq + (size_t)c * DP * DP;
  double* __restrict__ Gi = ch.trj.Ginv + (size_t)c * DP * DP;
  const double* __restrict__ Wc = Wd + (size_t)c * nbk * 64 * 64;
  // phase 1: block lower triangular W
  for (int j = 0; j < nbk; ++j) {
    for (int e = threadIdx.x; e < 64 * 64; e += 256) W[(size_t)(64 * j + (e >> 6)) * DP + 64 * j + (e & 63)] = Wc[(size_t)j * 4096 + e];
    __threadfence_block();
    __syncthreads();
    for (int i = j + 1; i < nbk; ++i) {
      d4 T[4];
      zero_acc(T);
      for (int m = j; m < i; ++m) {  // T += L_im W_mj
        stage_block(A, L + (size_t)(64 * i) * DP + 64 * m, DP, false);
        stage_block(B, W + (size_t)(64 * m) * DP + 64 * j, DP, true);
        __syncthreads();
        block_gemm_nt(A, B, w, lane, T);
        __syncthreads();
      }
      // W_ij = -W_ii T : A = W_ii, B = T' (written from the accumulators)
      stage_block(A, Wc + (size_t)i * 4096, 64, false);
      {
        const int rr = lane >> 4, ci = lane & 15;
#pragma unroll
        for (int J = 0; J < 4; ++J)
#pragma unroll
          for (int r = 0; r < 4; ++r) B[(16 * J + ci) * LD_LD + 16 * w + rr + 4 * r] = T[J][r];
      }
      __syncthreads();
      d4 acc[4];
      zero_acc(acc);
      block_gemm_nt(A, B, w, lane, acc);
      store_block(W + (size_t)(64 * i) * DP + 64 * j, DP, w, lane, acc, -1.0);
      __threadfence_block();
      __syncthreads();
    }
  }
  // phase 2: G^-1_ab = sum_{k >= a} W_ka' W_kb   (a >= b), mirrored
  for (int a = 0; a < nbk; ++a)
    for (int b = 0; b <= a; ++b) {
      d4 acc[4];
      zero_acc(acc);
      for (int k = a; k < nbk; ++k) {
        stage_block(A, W + (size_t)(64 * k) * DP + 64 * a, DP, true);
        stage_block(B, W + (size_t)(64 * k) * DP + 64 * b, DP, true);
        __syncthreads();
        block_gemm_nt(A, B, w, lane, acc);
        __syncthreads();
      }
      store_block(Gi + (size_t)(64 * a) * DP + 64 * b, DP, w, lane, acc, 1.0);
      if (a != b) {
        const int rr = lane >> 4, ci = lane & 15;
#pragma unroll
        for (int J = 0; J < 4; ++J)
#pragma unroll
          for (int r = 0; r < 4; ++r) Gi[(size_t)(64 * b + 16 * J + ci) * DP + 64 * a + 16 * w + rr + 4 * r] = acc[J][r];
      }
    }
  // exact symmetry of the diagonal blocks, and the padding of G^-1 kept at zero
  __threadfence_block();
  __syncthreads();
  for (int a = 0; a < nbk; ++a)
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
      const int r = e >> 6, cc = e & 63;
      if (cc < r) {
        double* p1 = Gi + (size_t)(64 * a + r) * DP + 64 * a + cc;
        double* p2 = Gi + (size_t)(64 * a + cc) * DP + 64 * a + r;
        const double m = 0.5 * (*p1 + *p2);
        *p1 = m; *p2 = m;
      }
    }
  __threadfence_block();
  __syncthreads();
  for (int r = 0; r < DP; ++r)
    for (int d = threadIdx.x; d < DP; d += 256)
      if (r >= dd.D || d >= dd.D) Gi[(size_t)r * DP + d] = 0.0;
}

// position fixed point, first iterate (rmhmc.py:113-122 with FixedIter = 0): u0 = G^-1 p (uq, from k_ginv_matvec)
__global__ __launch_bounds__(64) void k_pos_first_big(int D, int DP, Chains ch, double eps) {
  const int c = blockIdx.x, lane = threadIdx.x;
  if (ch.phase[c] != 1) return;
  for (int d = lane; d < D; d += 64) {
    const size_t o = (size_t)c * DP + d;
    const double u0 = ch.uq[o];
    ch.u0[o] = u0;
    ch.wq[o] = ch.trj.w[o] + ch.tau[c] * eps * u0;
  }
}
