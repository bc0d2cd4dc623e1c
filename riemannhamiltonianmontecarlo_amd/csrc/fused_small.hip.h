// fused_small.hip.h — the small-problem path (BASELINE config 2: D <= 8, X resident in LDS, no MFMA).
//
// One launch advances every chain by `nsteps` global steps (momentum draw / leapfrog step / accept, exactly
// the sequence of k_iter_begin + launch_step + k_iter_end of the generic path), so a whole sampling run is
// a handful of launches instead of ~37 per leapfrog step.
//   * workgroup = 4 wavefronts = 4 chains sharing one copy of X (transposed, [8][Mp]) in LDS; one workgroup
//     per CU (LDS bound), i.e. one wave per SIMD with the full 512-VGPR budget;
//   * one chain per wavefront, lane = data row (rows lane, lane+64, ...), x_n read from LDS;
//   * all D-vectors and the packed lower-triangular 8x8 matrices are wave-uniform values replicated in
//     registers (fully unrolled, D = 8 is a compile-time constant; smaller D is zero padded), so Cholesky,
//     triangular solves and the inverse need no LDS and no cross-lane traffic;
//   * per-row scalars c_n = v_n(1-2p_n) of the trajectory point are cached in LDS for the K momentum passes;
//     the record of the chain's current point also lives in (per-wave) LDS, the trajectory's in registers;
//   * sums over data rows are per-lane partial sums followed by a wave all-reduce.
// Same algorithm, same Philox streams and same state arrays as the generic kernels (rmhmc.py:37-191), so the
// two paths are interchangeable between launches.
#pragma once
#include "kernels.hip.h"

// No implicit a*b+c contraction in this file: the global-step loop of k_fused_small may be peeled or
// versioned by the optimiser, and with fp-contract=fast the copies can round differently, which would make a
// trajectory that continues in the next launch differ in the last bit from one that stays inside a launch
// (observed: 1 ulp).  Every multiply-add that matters for speed is written as an explicit fma().
#pragma clang fp contract(off)

#define FS_D 8
#define FS_T 36                  // packed lower triangle of an 8x8 matrix
#define FS_WAVES 4
#define FS_PT 104                // doubles of one point record in LDS (w, grad, tr, L, Gi, ljl, hld; padded)
#define FS_IDX(i, j) ((i) * ((i) + 1) / 2 + (j))   // i >= j

struct FusedParams {
  IterParams ip;
  double eps;
  int K;
  int nsteps;      // global steps to run in this launch
  int DPs;         // leading dimension of the per-chain arrays in HBM (16)
  int init_eval;   // 1: only evaluate the record at cur.w (chain initialisation, so that every record a fused run ever
                   //    holds comes from the same arithmetic: checkpoints resume bit for bit)
};

// ---- wave-uniform 8x8 helpers on packed lower-triangular registers -------------------------------------
// in place Cholesky; returns 1 if a pivot is <= 0 or NaN (then everything downstream is NaN => rejection)
__device__ __forceinline__ int fs_chol(double (&A)[FS_T], double (&rd)[FS_D]) {
  int bad = 0;
#pragma unroll
  for (int j = 0; j < FS_D; ++j) {
    double s = A[FS_IDX(j, j)];
#pragma unroll
    for (int k = 0; k < j; ++k) s = fma(-A[FS_IDX(j, k)], A[FS_IDX(j, k)], s);
    if (!(s > 0.0)) bad = 1;
    const double ljj = s * rsqrt(s);
    A[FS_IDX(j, j)] = ljj;
    rd[j] = 1.0 / ljj;  // defined from the stored L_jj so that a record reloaded from memory reproduces it bit for bit
#pragma unroll
    for (int i = j + 1; i < FS_D; ++i) {
      double a = A[FS_IDX(i, j)];
#pragma unroll
      for (int k = 0; k < j; ++k) a = fma(-A[FS_IDX(i, k)], A[FS_IDX(j, k)], a);
      A[FS_IDX(i, j)] = a * rd[j];
    }
  }
  return bad;
}
// x = (L L')^-1 b
__device__ __forceinline__ void fs_solve(const double (&L)[FS_T], const double (&rd)[FS_D], const double (&b)[FS_D], double (&x)[FS_D]) {
#pragma unroll
  for (int i = 0; i < FS_D; ++i) {
    double s = b[i];
#pragma unroll
    for (int k = 0; k < i; ++k) s = fma(-L[FS_IDX(i, k)], x[k], s);
    x[i] = s * rd[i];
  }
#pragma unroll
  for (int i = FS_D - 1; i >= 0; --i) {
    double s = x[i];
#pragma unroll
    for (int k = i + 1; k < FS_D; ++k) s = fma(-L[FS_IDX(k, i)], x[k], s);
    x[i] = s * rd[i];
  }
}
// Ginv (packed lower) = (L L')^-1 via W = L^-1, Ginv = W' W
__device__ __forceinline__ void fs_inverse(const double (&L)[FS_T], const double (&rd)[FS_D], double (&Gi)[FS_T]) {
  double W[FS_T];
#pragma unroll
  for (int j = 0; j < FS_D; ++j) {
    W[FS_IDX(j, j)] = rd[j];
#pragma unroll
    for (int i = j + 1; i < FS_D; ++i) {
      double s = 0.0;
#pragma unroll
      for (int k = j; k < i; ++k) s = fma(-L[FS_IDX(i, k)], W[FS_IDX(k, j)], s);
      W[FS_IDX(i, j)] = s * rd[i];
    }
  }
#pragma unroll
  for (int a = 0; a < FS_D; ++a)
#pragma unroll
    for (int b = 0; b <= a; ++b) {
      double s = 0.0;
#pragma unroll
      for (int i = a; i < FS_D; ++i) s = fma(W[FS_IDX(i, a)], W[FS_IDX(i, b)], s);
      Gi[FS_IDX(a, b)] = s;
    }
}
// y = S x for a packed symmetric S
__device__ __forceinline__ void fs_symv(const double (&S)[FS_T], const double (&x)[FS_D], double (&y)[FS_D]) {
#pragma unroll
  for (int i = 0; i < FS_D; ++i) {
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < FS_D; ++j) s = fma((i >= j) ? S[FS_IDX(i, j)] : S[FS_IDX(j, i)], x[j], s);
    y[i] = s;
  }
}
__device__ __forceinline__ double fs_dot(const double (&a)[FS_D], const double (&b)[FS_D]) {
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < FS_D; ++i) s = fma(a[i], b[i], s);
  return s;
}
__device__ __forceinline__ void fs_load_row(const double* Xs, int Mp, int n, double (&x)[FS_D]) {
#pragma unroll
  for (int d = 0; d < FS_D; ++d) x[d] = Xs[d * Mp + n];
}
// p = sigma(f), v = p(1-p), c = v(1-2p) with every rounding spelled out: hipcc's fp-contract=fast may rewrite
// x*(1-y) as fma(-x,y,x) or not depending on the surrounding code, which would make the c_n cache rebuilt when a
// launch resumes a trajectory differ in the last bit from the one computed inside the step
__device__ __forceinline__ void fs_pvc(double f, double& p, double& v, double& c) {
  p = 1.0 / (1.0 + exp(-f));
  v = __builtin_fma(-p, p, p);
  c = __builtin_fma(-2.0 * p, v, v);
}
// Wave all-reduce of N values at once: a reduce-scatter over the top log2(P) lane bits (P = N rounded up to a
// power of two; at every step a lane keeps one half of its list and sends the other half to lane^bit), plain
// butterfly steps for the remaining lane bits, then a v_readlane broadcast.  P-1 + (6-log2 P) shuffles instead of
// 6 N, and the totals come back wave-uniform.
template <int LEN, int BIT>
struct FsReduceScatter {
  template <int P>
  static __device__ __forceinline__ void run(double (&v)[P], int lane) {
    constexpr int H = LEN / 2;
    const bool up = (lane & BIT) != 0;
#pragma unroll
    for (int i = 0; i < H; ++i) {
      const double send = up ? v[i] : v[i + H];
      const double keep = up ? v[i + H] : v[i];
      v[i] = keep + __shfl_xor(send, BIT, 64);
    }
    FsReduceScatter<H, BIT / 2>::run(v, lane);
  }
};
template <int BIT>
struct FsReduceScatter<1, BIT> {
  template <int P>
  static __device__ __forceinline__ void run(double (&v)[P], int) {
#pragma unroll
    for (int o = BIT; o >= 1; o >>= 1) v[0] += __shfl_xor(v[0], o, 64);  // remaining lane bits
  }
};
template <int N>
__device__ __forceinline__ void fs_allreduce(double (&v)[N], int lane) {
  constexpr int P = N <= 1 ? 1 : N <= 2 ? 2 : N <= 4 ? 4 : N <= 8 ? 8 : N <= 16 ? 16 : N <= 32 ? 32 : 64;
  static_assert(N <= 64, "at most 64 values");
  constexpr int SH = P == 1 ? 6 : P == 2 ? 5 : P == 4 ? 4 : P == 8 ? 3 : P == 16 ? 2 : P == 32 ? 1 : 0;
  double t[P];
#pragma unroll
  for (int i = 0; i < P; ++i) t[i] = (i < N) ? v[i] : 0.0;
  FsReduceScatter<P, 32>::run(t, lane);
  // value i now sits (fully summed) in the lanes whose top bits equal i
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = rdlane(t[0], i << SH);
}

// everything the sampler needs at a point (rmhmc.py:50-77 / :134-156), wave-uniform
struct FsPoint {
  double w[FS_D], grad[FS_D], tr[FS_D];
  double L[FS_T], rd[FS_D], Gi[FS_T];
  double ljl, hld;
};

// metric G(w) (+ I/alpha) packed lower, optionally with gradient, log joint and the c_n cache
template <bool FULL>
__device__ __forceinline__ void fs_metric(const DevData& dd, const double* Xs, const double* ts, double* cbuf, int lane,
                                          const double (&w)[FS_D], double (&G)[FS_T], double (&grad)[FS_D], double& ljl) {
  const int Mp = dd.Mp;
#pragma unroll
  for (int t = 0; t < FS_T; ++t) G[t] = 0.0;
  double g[FS_D], lj = 0.0;
#pragma unroll
  for (int d = 0; d < FS_D; ++d) g[d] = 0.0;
#pragma unroll 2
  for (int n = lane; n < Mp; n += 64) {
    double x[FS_D];
    fs_load_row(Xs, Mp, n, x);
    const double f = fs_dot(x, w);
    double p, v, cn;
    fs_pvc(f, p, v, cn);
#pragma unroll
    for (int a = 0; a < FS_D; ++a) {
      const double va = v * x[a];
#pragma unroll
      for (int b = 0; b <= a; ++b) G[FS_IDX(a, b)] = fma(va, x[b], G[FS_IDX(a, b)]);
    }
    if (FULL) {
      const double ef = exp(f);
      const double tn = ts[n];
      const double r = tn - ef / (1.0 + ef);
#pragma unroll
      for (int d = 0; d < FS_D; ++d) g[d] = fma(r, x[d], g[d]);
      if (n < dd.M) lj += f * tn - log(1.0 + ef);
      cbuf[n] = cn;
    }
  }
  if (FULL) {
    double red[FS_T + FS_D + 1];
#pragma unroll
    for (int t = 0; t < FS_T; ++t) red[t] = G[t];
#pragma unroll
    for (int d = 0; d < FS_D; ++d) red[FS_T + d] = g[d];
    red[FS_T + FS_D] = lj;
    fs_allreduce(red, lane);
#pragma unroll
    for (int t = 0; t < FS_T; ++t) G[t] = red[t];
    lj = red[FS_T + FS_D];
#pragma unroll
    for (int d = 0; d < FS_D; ++d) {
      grad[d] = red[FS_T + d] - w[d] * dd.inv_alpha;
      lj += (d < dd.D) ? (dd.log_prior_const - w[d] * w[d] * 0.5 * dd.inv_alpha) : 0.0;
    }
    ljl = lj;
  } else {
    fs_allreduce(G, lane);
  }
#pragma unroll
  for (int a = 0; a < FS_D; ++a) G[FS_IDX(a, a)] += dd.inv_alpha;
}

// u' dG/dw_d u = sum_n c_n (x_n.u)^2 x_nd with the cached c_n   (rmhmc.py:104-107,158-161)
__device__ __forceinline__ void fs_quad(const DevData& dd, const double* Xs, const double* cbuf, int lane, const double (&u)[FS_D],
                                        double (&q)[FS_D]) {
#pragma unroll
  for (int d = 0; d < FS_D; ++d) q[d] = 0.0;
#pragma unroll 4
  for (int n = lane; n < dd.Mp; n += 64) {
    double x[FS_D];
    fs_load_row(Xs, dd.Mp, n, x);
    const double s = fs_dot(x, u);
    const double r = cbuf[n] * s * s;
#pragma unroll
    for (int d = 0; d < FS_D; ++d) q[d] = fma(r, x[d], q[d]);
  }
  fs_allreduce(q, lane);
}

// full record at pt.w
__device__ __forceinline__ int fs_eval_point(const DevData& dd, const double* Xs, const double* ts, double* cbuf, int lane, FsPoint& pt) {
  double G[FS_T];
  fs_metric<true>(dd, Xs, ts, cbuf, lane, pt.w, G, pt.grad, pt.ljl);
#pragma unroll
  for (int t = 0; t < FS_T; ++t) pt.L[t] = G[t];
  const int bad = fs_chol(pt.L, pt.rd);
  double hld = 0.0;
#pragma unroll
  for (int d = 0; d < FS_D; ++d) hld -= (d < dd.D) ? log(pt.rd[d]) : 0.0;   // zero-padded dims are not part of |G|
  pt.hld = hld;
  fs_inverse(pt.L, pt.rd, pt.Gi);
  // trace term: tr_d = sum_n c_n h_n x_nd,  h_n = x_n' G^-1 x_n   (rmhmc.py:67-77,148-156)
  double tr[FS_D];
#pragma unroll
  for (int d = 0; d < FS_D; ++d) tr[d] = 0.0;
#pragma unroll 2
  for (int n = lane; n < dd.Mp; n += 64) {
    double x[FS_D], y[FS_D];
    fs_load_row(Xs, dd.Mp, n, x);
    fs_symv(pt.Gi, x, y);
    const double ch = cbuf[n] * fs_dot(x, y);
#pragma unroll
    for (int d = 0; d < FS_D; ++d) tr[d] = fma(ch, x[d], tr[d]);
  }
  fs_allreduce(tr, lane);
#pragma unroll
  for (int d = 0; d < FS_D; ++d) pt.tr[d] = tr[d];
  return bad;
}

// state arrays in HBM <-> registers (packed triangles <-> the generic DPxDP row-major matrices)
__device__ __forceinline__ void fs_load_point(const Rec& r, int c, int DP, FsPoint& pt) {
#pragma unroll
  for (int d = 0; d < FS_D; ++d) {
    pt.w[d] = r.w[(size_t)c * DP + d];
    pt.grad[d] = r.grad[(size_t)c * DP + d];
    pt.tr[d] = r.tr[(size_t)c * DP + d];
  }
#pragma unroll
  for (int i = 0; i < FS_D; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      pt.L[FS_IDX(i, j)] = r.L[(size_t)c * DP * DP + i * DP + j];
      pt.Gi[FS_IDX(i, j)] = r.Ginv[(size_t)c * DP * DP + i * DP + j];
    }
  pt.ljl = r.ljl[c];
  pt.hld = r.hld[c];
}
__device__ __forceinline__ void fs_store_point(const Rec& r, int c, int D, int DP, int lane, const FsPoint& pt) {
  if (lane != 0) return;
#pragma unroll
  for (int d = 0; d < FS_D; ++d)
    if (d < D) {
      r.w[(size_t)c * DP + d] = pt.w[d];
      r.grad[(size_t)c * DP + d] = pt.grad[d];
      r.tr[(size_t)c * DP + d] = pt.tr[d];
    }
#pragma unroll
  for (int i = 0; i < FS_D; ++i)
#pragma unroll
    for (int j = 0; j < FS_D; ++j)
      if (i < D && j < D) {
        r.L[(size_t)c * DP * DP + i * DP + j] = (j <= i) ? pt.L[FS_IDX(i, j)] : 0.0;
        r.Ginv[(size_t)c * DP * DP + i * DP + j] = (j <= i) ? pt.Gi[FS_IDX(i, j)] : pt.Gi[FS_IDX(j, i)];
      }
  r.ljl[c] = pt.ljl;
  r.hld[c] = pt.hld;
}
// the zero-padded dimensions of a record loaded from HBM: L = sqrt(1/alpha) I, G^-1 = alpha I on the padding
__device__ __forceinline__ void fs_fix_padding(const DevData& dd, FsPoint& pt) {
#pragma unroll
  for (int i = 0; i < FS_D; ++i) {
    if (i >= dd.D) {
#pragma unroll
      for (int j = 0; j < i; ++j) { pt.L[FS_IDX(i, j)] = 0.0; pt.Gi[FS_IDX(i, j)] = 0.0; }
      pt.L[FS_IDX(i, i)] = sqrt(dd.inv_alpha);
      pt.Gi[FS_IDX(i, i)] = 1.0 / dd.inv_alpha;
      pt.w[i] = 0.0; pt.grad[i] = 0.0; pt.tr[i] = 0.0;
    }
    pt.rd[i] = 1.0 / pt.L[FS_IDX(i, i)];
  }
}

// the current point's record in per-wave LDS (lane 0 writes, every lane reads the broadcast)
__device__ __forceinline__ void fs_point_to_lds(double* q, int lane, const FsPoint& pt) {
  if (lane == 0) {
#pragma unroll
    for (int d = 0; d < FS_D; ++d) { q[d] = pt.w[d]; q[8 + d] = pt.grad[d]; q[16 + d] = pt.tr[d]; }
#pragma unroll
    for (int t = 0; t < FS_T; ++t) { q[24 + t] = pt.L[t]; q[60 + t] = pt.Gi[t]; }
    q[96] = pt.ljl; q[97] = pt.hld;
  }
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void fs_point_from_lds(const double* q, FsPoint& pt) {
#pragma unroll
  for (int d = 0; d < FS_D; ++d) { pt.w[d] = q[d]; pt.grad[d] = q[8 + d]; pt.tr[d] = q[16 + d]; }
#pragma unroll
  for (int t = 0; t < FS_T; ++t) { pt.L[t] = q[24 + t]; pt.Gi[t] = q[60 + t]; }
  pt.ljl = q[96]; pt.hld = q[97];
#pragma unroll
  for (int i = 0; i < FS_D; ++i) pt.rd[i] = 1.0 / pt.L[FS_IDX(i, i)];
}

__global__ __launch_bounds__(64 * FS_WAVES) void k_fused_small(DevData dd, Chains ch, FusedParams fp) {
  extern __shared__ __attribute__((aligned(16))) double fs_lds[];
  const int Mp = dd.Mp;
  double* Xs = fs_lds;            // [8][Mp]
  double* ts = fs_lds + FS_D * Mp;  // [Mp]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double* cbuf = ts + Mp + (size_t)wave * Mp;
  double* curL = ts + Mp + (size_t)FS_WAVES * Mp + wave * FS_PT;
  for (int i = threadIdx.x; i < FS_D * Mp; i += 64 * FS_WAVES) Xs[i] = dd.Xt[i];  // rows 0..7 of Xt[DP][Mp]
  for (int i = threadIdx.x; i < Mp; i += 64 * FS_WAVES) ts[i] = dd.t[i];
  __syncthreads();
  const int c = blockIdx.x * FS_WAVES + wave;
  if (c >= ch.n) return;
  const int D = dd.D, DP = fp.DPs;
  const IterParams& ip = fp.ip;

  // ---- chain state -> registers -------------------------------------------------------------------------
  FsPoint trj;
  if (fp.init_eval) {
#pragma unroll
    for (int d = 0; d < FS_D; ++d) trj.w[d] = (d < D) ? ch.cur.w[(size_t)c * DP + d] : 0.0;
    const int bad = fs_eval_point(dd, Xs, ts, cbuf, lane, trj);
    fs_store_point(ch.cur, c, D, DP, lane, trj);
    fs_store_point(ch.trj, c, D, DP, lane, trj);
    if (lane == 0) ch.status[c] = bad ? 1 : 0;
    return;
  }
  fs_load_point(ch.cur, c, DP, trj);
  fs_fix_padding(dd, trj);
  fs_point_to_lds(curL, lane, trj);
  int phase = ch.phase[c], steps_left = ch.steps_left[c], status = ch.status[c], nsteps_last = ch.nsteps_last[c];
  long long iter = ch.iter[c], accepted = ch.accepted[c], steps_done = ch.steps_done[c];
  double p[FS_D], Hcur = ch.Hcur[c], Hprop = ch.Hprop[c], tau = ch.tau[c];
#pragma unroll
  for (int d = 0; d < FS_D; ++d) p[d] = 0.0;
  if (phase == 1) {  // mid-trajectory: resume from the trajectory record (and rebuild its c_n cache)
    fs_load_point(ch.trj, c, DP, trj);
    fs_fix_padding(dd, trj);
#pragma unroll
    for (int d = 0; d < FS_D; ++d) p[d] = (d < D) ? ch.p[(size_t)c * DP + d] : 0.0;
    for (int n = lane; n < Mp; n += 64) {
      double x[FS_D];
      fs_load_row(Xs, Mp, n, x);
      double pp, vv, cc;
      fs_pvc(fs_dot(x, trj.w), pp, vv, cc);
      cbuf[n] = cc;
    }
  }
  const double h_eps = fp.eps * 0.5;

  for (int gs = 0; gs < fp.nsteps; ++gs) {
    // ---- k_iter_begin: start a transition (rmhmc.py:47,80-93,175-176) ------------------------------------
    if (phase == 0 && iter < ip.iter_limit) {
      fs_point_from_lds(curL, trj);  // the trajectory starts from the cached record of the current point
      double zl = 0.0, u_len, g_dir;
      if (ip.z_in) {
        zl = (lane < D) ? ip.z_in[(size_t)c * D + lane] : 0.0;
        u_len = ip.ulen_in[c];
        g_dir = ip.gdir_in[c];
      } else {
        const unsigned long long gid = (unsigned long long)(ip.chain_offset + c);
        double U0, U1;
        rng_block(ip.seed, gid, (uint32_t)iter, (uint32_t)(lane >> 1), U0, U1);
        const double R = sqrt(-2.0 * log(U0));
        double sn, cs;
        sincos(RM_PI2 * U1, &sn, &cs);
        zl = (lane < D) ? ((lane & 1) ? R * sn : R * cs) : 0.0;
        double Ua;
        rng_block(ip.seed, gid, (uint32_t)iter, 0x40000000u, u_len, Ua);
        rng_block(ip.seed, gid, (uint32_t)iter, 0x40000001u, U0, U1);
        g_dir = sqrt(-2.0 * log(U0)) * cos(RM_PI2 * U1);
      }
      double z[FS_D];
#pragma unroll
      for (int d = 0; d < FS_D; ++d) z[d] = rdlane(zl, d);
      // p = L' z (reference, rmhmc.py:60,80) or L z (corrected)
#pragma unroll
      for (int j = 0; j < FS_D; ++j) {
        double s = 0.0;
        if (ip.flags & 1u) {
#pragma unroll
          for (int i = j; i < FS_D; ++i) s = fma(trj.L[FS_IDX(i, j)], z[i], s);
        } else {
#pragma unroll
          for (int k = 0; k <= j; ++k) s = fma(trj.L[FS_IDX(j, k)], z[k], s);
        }
        p[j] = (j < D) ? s : 0.0;
      }
      status = 0;
      if (ip.flags & 2u) {  // momentum guard, rmhmc.py:81-85
        const double np_ = sqrt(fs_dot(p, p));
        if (np_ > 100.0) {
#pragma unroll
          for (int d = 0; d < FS_D; ++d) p[d] /= np_ * 25.0;
          status = 4;
        }
      }
      double y[FS_D];
      fs_symv(trj.Gi, p, y);
      Hcur = -trj.ljl + trj.hld + 0.5 * fs_dot(p, y);
      if (lane == 0) {
#pragma unroll
        for (int d = 0; d < FS_D; ++d)
          if (d < D) ch.p0[(size_t)c * DP + d] = p[d];
      }
      steps_left = (int)ceil(u_len * (double)ip.L);
      nsteps_last = steps_left;
      tau = (g_dir > 0.5) ? 1.0 : -1.0;
      phase = (steps_left > 0) ? 1 : 2;
      // the c_n cache must describe the trajectory's starting point
      for (int n = lane; n < Mp; n += 64) {
        double x[FS_D];
        fs_load_row(Xs, Mp, n, x);
        double pp, vv, cc;
        fs_pvc(fs_dot(x, trj.w), pp, vv, cc);
        cbuf[n] = cc;
      }
    }
    // ---- one generalised leapfrog step (rmhmc.py:96-163) ---------------------------------------------------
    if (phase == 1) {
      const double h = tau * h_eps;
      // implicit momentum half step (rmhmc.py:102-110)
      double PM[FS_D];
#pragma unroll
      for (int d = 0; d < FS_D; ++d) PM[d] = p[d];
      for (int it = 0; it < fp.K; ++it) {
        double u[FS_D], q[FS_D];
        fs_symv(trj.Gi, PM, u);
        fs_quad(dd, Xs, cbuf, lane, u, q);
#pragma unroll
        for (int d = 0; d < FS_D; ++d) PM[d] = p[d] + h * (trj.grad[d] - 0.5 * trj.tr[d] + 0.5 * q[d]);
      }
#pragma unroll
      for (int d = 0; d < FS_D; ++d) p[d] = PM[d];
      // implicit position step (rmhmc.py:113-123); first iterate re-uses the stored factor of G(w)
      double u0[FS_D], Pw[FS_D];
      fs_solve(trj.L, trj.rd, p, u0);
#pragma unroll
      for (int d = 0; d < FS_D; ++d) Pw[d] = trj.w[d] + tau * fp.eps * u0[d];
      for (int it = 1; it < fp.K; ++it) {
        double G[FS_T], rd[FS_D], u[FS_D], gdummy[FS_D], ldummy;
        fs_metric<false>(dd, Xs, ts, cbuf, lane, Pw, G, gdummy, ldummy);
        if (fs_chol(G, rd)) status |= 1;
        fs_solve(G, rd, p, u);
#pragma unroll
        for (int d = 0; d < FS_D; ++d) Pw[d] = trj.w[d] + h * (u0[d] + u[d]);
      }
      // position guard (rmhmc.py:125-130)
      if (ip.flags & 2u) {
        const double nw = sqrt(fs_dot(Pw, Pw));
        if (nw > 10.0) {
#pragma unroll
          for (int d = 0; d < FS_D; ++d) Pw[d] /= nw * 3.0;
          status |= 8;
        }
      }
#pragma unroll
      for (int d = 0; d < FS_D; ++d) trj.w[d] = Pw[d];
      // explicit momentum half step at the new point (rmhmc.py:134-163)
      if (fs_eval_point(dd, Xs, ts, cbuf, lane, trj)) status |= 1;
      double u[FS_D], q[FS_D];
      fs_symv(trj.Gi, p, u);
      fs_quad(dd, Xs, cbuf, lane, u, q);
      bool nonfinite = false;
#pragma unroll
      for (int d = 0; d < FS_D; ++d) {
        p[d] += h * (trj.grad[d] - 0.5 * trj.tr[d] + 0.5 * q[d]);
        nonfinite = nonfinite || !(isfinite(p[d]) && isfinite(trj.w[d]));
      }
      if (nonfinite) status |= 2;
      steps_left -= 1;
      steps_done += 1;
    }
    // ---- k_iter_end: finish a transition (rmhmc.py:166-191) ---------------------------------------------------
    if (phase == 2 || (phase == 1 && steps_left == 0)) {
      double y[FS_D];
      fs_symv(trj.Gi, p, y);
      Hprop = -trj.ljl + trj.hld + 0.5 * fs_dot(p, y);
      const double ratio = -Hprop + Hcur;
      double u_acc;
      if (ip.z_in) {
        u_acc = ip.uacc_in[c];
      } else {
        double U0;
        rng_block(ip.seed, (unsigned long long)(ip.chain_offset + c), (uint32_t)iter, 0x40000000u, U0, u_acc);
      }
      const bool accept = (ratio > 0.0) || (ratio > log(u_acc));
      if (accept) { fs_point_to_lds(curL, lane, trj); accepted += 1; }
      if (ip.samples && iter >= ip.burn_in && iter - ip.burn_in < ip.S && lane == 0) {
#pragma unroll
        for (int d = 0; d < FS_D; ++d)
          if (d < D) ip.samples[((size_t)c * ip.S + (size_t)(iter - ip.burn_in)) * D + d] = curL[d];
      }
      iter += 1;
      phase = 0;
      if (iter == ip.iter_limit && ip.done_count && lane == 0) atomicAdd(ip.done_count, 1);
    }
  }

  // ---- registers -> chain state ---------------------------------------------------------------------------------
  fs_store_point(ch.trj, c, D, DP, lane, trj);
  fs_point_from_lds(curL, trj);
  fs_store_point(ch.cur, c, D, DP, lane, trj);
  if (lane == 0) {
#pragma unroll
    for (int d = 0; d < FS_D; ++d)
      if (d < D) ch.p[(size_t)c * DP + d] = p[d];
    ch.phase[c] = phase; ch.steps_left[c] = steps_left; ch.status[c] = status; ch.nsteps_last[c] = nsteps_last;
    ch.iter[c] = iter; ch.accepted[c] = accepted; ch.steps_done[c] = steps_done;
    ch.Hcur[c] = Hcur; ch.Hprop[c] = Hprop; ch.tau[c] = tau;
  }
}
