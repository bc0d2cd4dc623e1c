// medium_step.hip.h — one generalised leapfrog step in ONE launch for small batches of mid-size problems
// (8 < D <= 32, M <= 2048, few chains: the bundled data sets of the reference, australian / german / heart, run as main.py
// runs them: one chain).  The generic path needs ~37 dependent launches per step (~230 us at one chain whatever the work);
// here one 256-thread workgroup owns a chain and walks the whole step of rmhmc.py:96-163 between __syncthreads():
//   * row passes (f, p, v, c, log joint, gradient, u' dG_d u, leverages): thread = data row, the row's D values in registers,
//     per-thread partial sums reduced with the multi-value wave all-reduce of fused_small.hip.h and one LDS hop;
//   * metric assembly: the fp64 MFMA mapping of k_assemble with the data rows split over the four waves, partial tiles summed
//     through LDS;
//   * Cholesky (chol_lds_blk), solves and the inverse (spd_inverse_lds) by wave 0 on the LDS image.
// Inputs / outputs are those of step_phases() in rmhmc_hip.hip: the trajectory record, p, tau, status, step counters; with `fold` the
// launch is a whole global step (transition start, one leapfrog step, transition end), one launch instead of three.
// eval_only: just the record at trj.w (what eval_point_phases(advance = false) does), used for the sampler's initial point so
// that a resumed chain sees bit for bit the record the uninterrupted run computed inside a step.
// v and c live in LDS (they are produced and consumed inside the launch).
#pragma once
#include "kernels.hip.h"
#include "fused_small.hip.h"

#define MS_MAXMP 2048  // rows (padded) whose v and c fit the LDS budget
#define MS_GLD 34      // leading dimension of the G^-1 image

template <int NB>
constexpr int ms_lds_doubles(int Mp) {
  return 64 * RM_LD + 32 * MS_GLD + 4 * (16 * NB) * (16 * NB) + 2 * Mp + 12 * 32 + 4 * 40;
}

// RPT > 0: every thread keeps its RPT data rows (n = t, t + 256, ...; Mp <= 256 RPT) in registers for the whole launch, so the ~11 row
// passes of a step read X once; RPT = 0: rows are re-read from L2 in every pass (one row ahead).
template <int NB, int RPT>
__global__ __launch_bounds__(256) void k_step_medium(DevData dd, Chains ch, double eps, int K, int guards, int eval_only, int fold, IterParams ip) {
  constexpr int DPc = 16 * NB;
  constexpr int RR = RPT > 0 ? RPT : 1;
  constexpr int NT = NB * (NB + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int c = blockIdx.x;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int D = dd.D, M = dd.M, Mp = dd.Mp;
  // fold: the transition bookkeeping of the global step (k_iter_begin before, k_iter_end after) runs inside this launch, by wave 0
  if (fold) {
    if (wave == 0) iter_begin_dev(D, DPc, ch, ip, c, lane, sm, sm + 64);
    __syncthreads();
  }
  const bool active = ch.phase[c] == 1;
  if (!fold && !active) return;
  double* A = sm;                          // 64 x RM_LD: G / L / W / G^-1 workspace of wave 0
  double* GI = A + 64 * RM_LD;             // 32 x MS_GLD: G^-1 of the point the step works at (symmetric, full)
  double* PT = GI + 32 * MS_GLD;           // 4 x DPc x DPc: per-wave partial metric
  double* vrow = PT + 4 * DPc * DPc;       // Mp
  double* crow = vrow + Mp;                // Mp
  double* vec = crow + Mp;                 // 12 vectors of 32
  double* red = vec + 12 * 32;             // 4 x 40
  double *wv = vec, *pv = vec + 32, *PM = vec + 64, *uv = vec + 96, *gradv = vec + 128, *trv = vec + 160, *qv = vec + 192,
         *wq = vec + 224, *u0 = vec + 256, *tmp = vec + 288;
  const size_t ov = (size_t)c * DPc, om = (size_t)c * DPc * DPc;
  const double h = ch.tau[c] * eps * 0.5;
  int status = 0;

  // ---- helpers -------------------------------------------------------------------------------------------------
  // block sum of DPc per-thread partials (+ one scalar): results in tmp[0..DPc) and the return value, valid after the call
  auto block_reduce = [&](double (&acc)[DPc], double sc) -> double {
    fs_allreduce<DPc>(acc, lane);
    sc = wave_sum(sc);
    __syncthreads();  // red / tmp free
    if (lane == 0) {
#pragma unroll
      for (int d = 0; d < DPc; ++d) red[wave * 40 + d] = acc[d];
      red[wave * 40 + DPc] = sc;
    }
    __syncthreads();
    if (t <= DPc) {
      const double s = (red[t] + red[40 + t]) + (red[80 + t] + red[120 + t]);
      if (t < DPc) tmp[t] = s; else red[39] = s;
    }
    __syncthreads();
    return red[39];
  };
  // row passes.  MODE 0: v (and c) at `wp`; 1: v, c, gradient partials and log joint at `wp`; 2: q_d = sum c_n (x_n.u)^2 x_nd;
  // 3: tr_d = sum c_n (x_n' G^-1 x_n) x_nd.  Result vector in tmp[], scalar returned.
  double xk[RR][DPc], tk[RR];
  if (RPT > 0) {
#pragma unroll
    for (int r = 0; r < RR; ++r) {
      const int n = t + 256 * r;
      const bool in = n < Mp;
      tk[r] = in ? dd.t[n] : 0.0;
#pragma unroll
      for (int d = 0; d < DPc; d += 2) {
        const double2 q = in ? *(const double2*)(dd.Xr + (size_t)n * DPc + d) : make_double2(0.0, 0.0);
        xk[r][d] = q.x; xk[r][d + 1] = q.y;
      }
    }
  }
  auto row_pass = [&](int mode, const double* wp, const double* up) -> double {
    double acc[DPc];
#pragma unroll
    for (int d = 0; d < DPc; ++d) acc[d] = 0.0;
    double lj = 0.0;
    auto one_row = [&](const double (&x)[DPc], double tn, int n) {  // n < Mp
      if (mode <= 1) {
        double f = 0.0;
#pragma unroll
        for (int d = 0; d < DPc; ++d) f = fma(x[d], wp[d], f);
        const double em = exp(-f);
        const double p = 1.0 / (1.0 + em);
        const double v = p * (1.0 - p);
        vrow[n] = v;
        crow[n] = v * (1.0 - 2.0 * p);
        if (mode == 1) {
          const double ef = exp(f);
          if (n < M) lj += f * tn - log(1.0 + ef);
          const double rn = tn - ef / (1.0 + ef);  // padded rows: x = 0, no contribution
#pragma unroll
          for (int d = 0; d < DPc; ++d) acc[d] = fma(rn, x[d], acc[d]);
        }
      } else if (mode == 2) {
        double s = 0.0;
#pragma unroll
        for (int d = 0; d < DPc; ++d) s = fma(x[d], up[d], s);
        const double z = crow[n] * s * s;
#pragma unroll
        for (int d = 0; d < DPc; ++d) acc[d] = fma(z, x[d], acc[d]);
      } else {
        double hn = 0.0;
#pragma unroll
        for (int i = 0; i < DPc; ++i) {  // fully unrolled (x stays in registers); the fence keeps the 16 DPc ds_reads from being hoisted
          double y0 = 0.0, y1 = 0.0;
#pragma unroll
          for (int d = 0; d < DPc; d += 2) {
            const double2 g = *(const double2*)(GI + i * MS_GLD + d);
            y0 = fma(g.x, x[d], y0); y1 = fma(g.y, x[d + 1], y1);
          }
          hn = fma(y0 + y1, x[i], hn);
          __builtin_amdgcn_sched_barrier(0);
        }
        const double z = crow[n] * hn;
#pragma unroll
        for (int d = 0; d < DPc; ++d) acc[d] = fma(z, x[d], acc[d]);
      }
    };
    if (RPT > 0) {
#pragma unroll
      for (int r = 0; r < RR; ++r)
        if (t + 256 * r < Mp) one_row(xk[r], tk[r], t + 256 * r);
    } else {
      double xn[DPc];  // next row of this thread, loaded one trip ahead
#pragma unroll
      for (int d = 0; d < DPc; d += 2) {
        const double2 q = (t < Mp) ? *(const double2*)(dd.Xr + (size_t)t * DPc + d) : make_double2(0.0, 0.0);
        xn[d] = q.x; xn[d + 1] = q.y;
      }
      for (int n = t; n < Mp; n += 256) {
        double x[DPc];
#pragma unroll
        for (int d = 0; d < DPc; ++d) x[d] = xn[d];
        if (n + 256 < Mp) {
#pragma unroll
          for (int d = 0; d < DPc; d += 2) { const double2 q = *(const double2*)(dd.Xr + (size_t)(n + 256) * DPc + d); xn[d] = q.x; xn[d + 1] = q.y; }
        }
        one_row(x, mode == 1 ? dd.t[n] : 0.0, n);
      }
    }
    if (mode == 0) { __syncthreads(); return 0.0; }
    return block_reduce(acc, lj);
  };
  // G = X' diag(v) X + I/alpha into the LDS image A (natural row-major, both triangles), rmhmc.py:57,119,137
  auto assemble = [&]() {
    const int rr = lane >> 4, ci = lane & 15;
    d4 acc[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q) acc[q] = (d4){0.0, 0.0, 0.0, 0.0};
    for (int n0 = 4 * wave; n0 < Mp; n0 += 64) {  // four 4-row groups per trip (Mp is a multiple of 64): their loads are in flight together
      double xb[4][NB], vn[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + 16 * g + rr;
        vn[g] = vrow[n];
#pragma unroll
        for (int I = 0; I < NB; ++I) xb[g][I] = dd.Xr[(size_t)n * DPc + NB * ci + I];
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        double xa[NB];
#pragma unroll
        for (int I = 0; I < NB; ++I) xa[I] = vn[g] * xb[g][I];
        int q = 0;
#pragma unroll
        for (int I = 0; I < NB; ++I)
#pragma unroll
          for (int J = I; J < NB; ++J) { acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[I], xb[g][J], acc[q], 0, 0, 0); ++q; }
      }
    }
    double* P = PT + wave * DPc * DPc;
    int q = 0;
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
      for (int J = I; J < NB; ++J) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = NB * (rr + 4 * r) + I, col = NB * ci + J;
          if (I != J || row >= col) { P[row * DPc + col] = acc[q][r]; P[col * DPc + row] = acc[q][r]; }
        }
        ++q;
      }
    __syncthreads();
    for (int e = t; e < DPc * DPc; e += 256) {
      const int row = e / DPc, col = e % DPc;
      double g = (PT[e] + PT[DPc * DPc + e]) + (PT[2 * DPc * DPc + e] + PT[3 * DPc * DPc + e]);
      if (row == col) g += dd.inv_alpha;
      A[row * RM_LD + col] = g;
    }
    __syncthreads();
  };
  auto matvec_GI = [&](const double* src, double* dst) {  // dst = G^-1 src
    if (t < D) {
      double u = 0.0;
      for (int j = 0; j < D; ++j) u = fma(GI[j * MS_GLD + t], src[j], u);
      dst[t] = u;
    }
    __syncthreads();
  };

  if (active) {
  // ---- load the trajectory point --------------------------------------------------------------------------------
  if (t < 32) {
    const bool in = t < D;
    wv[t] = in ? ch.trj.w[ov + t] : 0.0;
    pv[t] = in ? ch.p[ov + t] : 0.0;
    gradv[t] = in ? ch.trj.grad[ov + t] : 0.0;
    trv[t] = in ? ch.trj.tr[ov + t] : 0.0;
    uv[t] = 0.0; PM[t] = 0.0; qv[t] = 0.0; wq[t] = 0.0; u0[t] = 0.0;
  }
  for (int e = t; e < 32 * MS_GLD; e += 256) {
    const int i = e / MS_GLD, j = e % MS_GLD;
    GI[e] = (i < D && j < D) ? ch.trj.Ginv[om + i * DPc + j] : 0.0;
  }
  for (int n = M + t; n < Mp; n += 256) { vrow[n] = 0.0; crow[n] = 0.0; }
  if (wave == 0) load_mat_lds(A, ch.trj.L + om, D, DPc, lane);
  __syncthreads();
  if (!eval_only) {
  row_pass(0, wv, nullptr);  // c_n at w (the momentum fixed point works at the current point, rmhmc.py:102-110)

  // ---- implicit momentum half step (rmhmc.py:102-110) ---------------------------------------------------------------
  for (int it = 0; it < K; ++it) {
    matvec_GI(it == 0 ? pv : PM, uv);
    row_pass(2, nullptr, uv);
    if (t < 32) PM[t] = (t < D) ? pv[t] + h * (gradv[t] - 0.5 * trv[t] + 0.5 * tmp[t]) : 0.0;
    __syncthreads();
  }
  if (t < 32) pv[t] = PM[t];
  __syncthreads();

  // ---- implicit position step (rmhmc.py:113-123); the first iterate re-uses the stored factor of G(w) -----------------
  if (wave == 0) {
    const double rdiag = (lane < D) ? 1.0 / A[lane * RM_LD + lane] : 1.0;
    const double x = cholsolve_lds(A, D, lane, (lane < D) ? pv[lane] : 0.0, rdiag);
    if (lane < 32) {
      u0[lane] = (lane < D) ? x : 0.0;
      wq[lane] = (lane < D) ? wv[lane] + ch.tau[c] * eps * x : 0.0;
    }
  }
  __syncthreads();
  for (int it = 1; it < K; ++it) {
    row_pass(0, wq, nullptr);
    assemble();
    if (wave == 0) {
      double rdiag;
      const int bad = chol_lds_blk<NB>(A, D, lane, rdiag);
      const double x = cholsolve_lds(A, D, lane, (lane < D) ? pv[lane] : 0.0, rdiag);
      if (lane < D) wq[lane] = wv[lane] + ch.tau[c] * (eps * 0.5) * (u0[lane] + x);
      if (bad) status |= 1;
    }
    __syncthreads();
  }
  // accept the iterate, position guard (rmhmc.py:123-130)
  if (wave == 0) {
    double ss = (lane < D) ? wq[lane] * wq[lane] : 0.0;
    const double nw = sqrt(wave_sum(ss));
    const bool fire = guards && nw > 10.0;
    if (lane < 32) wv[lane] = (lane < D) ? (fire ? wq[lane] / (nw * 3.0) : wq[lane]) : 0.0;
    if (fire) status |= 8;
  }
  __syncthreads();
  }  // !eval_only

  // ---- new point (rmhmc.py:134-161) and the explicit momentum half step (:163) -----------------------------------------
  const double ljsum = row_pass(1, wv, nullptr);
  if (t < 32) gradv[t] = (t < D) ? tmp[t] - wv[t] * dd.inv_alpha : 0.0;
  __syncthreads();
  assemble();
  double hld = 0.0;
  if (wave == 0) {
    double rdiag;
    const int bad = chol_lds_blk<NB>(A, D, lane, rdiag);
    if (bad) status |= 1;
    hld = -wave_sum((lane < D) ? log(rdiag) : 0.0);
    double* __restrict__ Lg = ch.trj.L + om;
    for (int i = 0; i < D; ++i)
      if (lane < D) Lg[i * DPc + lane] = (lane <= i) ? A[i * RM_LD + lane] : 0.0;
    __builtin_amdgcn_wave_barrier();
    spd_inverse_lds<NB>(A, D, lane, rdiag);
  }
  __syncthreads();
  for (int e = t; e < 32 * MS_GLD; e += 256) {
    const int i = e / MS_GLD, j = e % MS_GLD;
    const double g = (i < D && j < D) ? A[max(i, j) * RM_LD + min(i, j)] : 0.0;
    GI[e] = g;
    if (i < D && j < D) ch.trj.Ginv[om + i * DPc + j] = g;
  }
  __syncthreads();
  matvec_GI(pv, uv);                 // u = G^-1 p
  row_pass(2, nullptr, uv);          // quadratic term at the new point
  if (t < 32) qv[t] = tmp[t];
  row_pass(3, nullptr, nullptr);     // trace term
  if (t < 32) trv[t] = (t < D) ? tmp[t] : 0.0;
  __syncthreads();
  // explicit half step, record, bookkeeping (k_mom_final)
  int nonfinite = 0;
  if (t < D) {
    const double pn = eval_only ? pv[t] : pv[t] + h * (gradv[t] - 0.5 * trv[t] + 0.5 * qv[t]);
    if (!eval_only) ch.p[ov + t] = pn;
    ch.last[ov + t] = qv[t];
    ch.uq[ov + t] = uv[t];
    ch.trj.w[ov + t] = wv[t];
    ch.trj.grad[ov + t] = gradv[t];
    ch.trj.tr[ov + t] = trv[t];
    nonfinite = !(isfinite(pn) && isfinite(wv[t]));
  }
  if (wave == 0) {
    const unsigned long long any = __ballot(nonfinite);
    double part = (lane < D) ? (dd.log_prior_const - wv[lane] * wv[lane] * 0.5 * dd.inv_alpha) : 0.0;
    const double prior = wave_sum(part);
    if (lane == 0) {
      ch.trj.hld[c] = hld;
      ch.trj.ljl[c] = ljsum + prior;
      if (any) status |= 2;
      if (status) ch.status[c] |= status;
      if (!eval_only) {
        ch.steps_left[c] -= 1;
        ch.steps_done[c] += 1;
      }
    }
  }
  }  // active
  if (fold) {
    __syncthreads();
    if (wave == 0) iter_end_dev(D, DPc, ch, ip, c, lane, sm);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Plain HMC (code/hmc.py:48-62), the sampler the reference's main.py actually calls: a whole TRAJECTORY in one launch for
// small batches (D <= 32).  The generic path spends five launches per leapfrog step and the reference draws up to 100 steps
// per transition; here one 256-thread workgroup owns a chain, keeps w / p / gradient in LDS and does every remaining step of
// the trajectory: momentum half step, position step, gradient and log joint at the new position (thread = data row, partial
// sums through the wave all-reduce and one LDS hop), second half step.  A NaN momentum ends the trajectory as in k_hmc_pre.
// eval_only: gradient and log joint at trj.w, then trj -> cur (the sampler's initial record, same arithmetic as inside a
// trajectory).
// ---------------------------------------------------------------------------------------------------------------
// RPT > 0: every thread keeps its RPT data rows (n = t, t+256, ...; Mp <= 256 RPT) in registers for the whole trajectory;
// RPT = 0: rows are re-read from L2 at every step.
template <int NB, int RPT>
__global__ __launch_bounds__(256) void k_hmc_traj(DevData dd, Chains ch, double eps, int eval_only) {
  constexpr int DPc = 16 * NB;
  constexpr int RR = RPT > 0 ? RPT : 1;
  __shared__ double wv[32], pv[32], gv[32], tmp[32], red[4 * 40];
  __shared__ int flag;
  const int c = blockIdx.x;
  if (ch.phase[c] != 1) return;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int D = dd.D, M = dd.M, Mp = dd.Mp;
  const size_t ov = (size_t)c * DPc;
  if (t < 32) {
    const bool in = t < D;
    wv[t] = in ? ch.trj.w[ov + t] : 0.0;
    pv[t] = in ? ch.p[ov + t] : 0.0;
    gv[t] = in ? ch.trj.grad[ov + t] : 0.0;
  }
  int steps = eval_only ? 1 : ch.steps_left[c];
  int status = 0, done = 0;
  double ljl = 0.0;
  double xk[RR][DPc], tk[RR];
  if (RPT > 0) {
#pragma unroll
    for (int r = 0; r < RR; ++r) {
      const int n = t + 256 * r;
      const bool in = n < Mp;
      tk[r] = in ? dd.t[n] : 0.0;
#pragma unroll
      for (int d = 0; d < DPc; d += 2) {
        const double2 q = in ? *(const double2*)(dd.Xr + (size_t)n * DPc + d) : make_double2(0.0, 0.0);
        xk[r][d] = q.x; xk[r][d + 1] = q.y;
      }
    }
  }
  __syncthreads();
  for (int s = 0; s < steps; ++s) {
    if (!eval_only) {  // first half step and the position step (hmc.py:52-58)
      if (wave == 0) {
        double p = 0.0;
        int isnan_ = 0;
        if (lane < D) { p = pv[lane] + eps * 0.5 * gv[lane]; isnan_ = (p != p); }
        const unsigned long long nan = __ballot(isnan_);
        if (lane < D) {
          pv[lane] = p;
          if (!nan) wv[lane] += eps * p;
        }
        if (lane == 0) flag = nan ? 1 : 0;
      }
      __syncthreads();
      if (flag) status |= 2;
    }
    // gradient X'(t - e^f/(1+e^f)) and log joint sum f t - log(1+e^f) at w (hmc.py:60-61, naive forms kept: they overflow
    // exactly where the reference does)
    double acc[DPc];
#pragma unroll
    for (int d = 0; d < DPc; ++d) acc[d] = 0.0;
    double lj = 0.0;
    auto one_row = [&](const double (&x)[DPc], double tn, int n) {
      double f = 0.0;
#pragma unroll
      for (int d = 0; d < DPc; ++d) f = fma(x[d], wv[d], f);
      const double ef = exp(f);
      if (n < M) lj += f * tn - log(1.0 + ef);
      const double rn = tn - ef / (1.0 + ef);
#pragma unroll
      for (int d = 0; d < DPc; ++d) acc[d] = fma(rn, x[d], acc[d]);
    };
    if (RPT > 0) {
#pragma unroll
      for (int r = 0; r < RR; ++r) one_row(xk[r], tk[r], t + 256 * r);
    } else {
      for (int n = t; n < Mp; n += 256) {
        const double* xr = dd.Xr + (size_t)n * DPc;
        double x[DPc];
#pragma unroll
        for (int d = 0; d < DPc; d += 2) { const double2 q = *(const double2*)(xr + d); x[d] = q.x; x[d + 1] = q.y; }
        one_row(x, dd.t[n], n);
      }
    }
    fs_allreduce<DPc>(acc, lane);
    lj = wave_sum(lj);
    if (lane == 0) {
#pragma unroll
      for (int d = 0; d < DPc; ++d) red[wave * 40 + d] = acc[d];
      red[wave * 40 + DPc] = lj;
    }
    __syncthreads();
    if (wave == 0) {
      double part = 0.0;
      if (lane < D) {
        const double g = (red[lane] + red[40 + lane]) + (red[80 + lane] + red[120 + lane]) - wv[lane] * dd.inv_alpha;
        gv[lane] = g;
        part = dd.log_prior_const - wv[lane] * wv[lane] * 0.5 * dd.inv_alpha;
        if (!eval_only && !(status & 2)) pv[lane] += eps * 0.5 * g;  // second half step (hmc.py:62)
      }
      ljl = wave_sum(part) + ((red[DPc] + red[40 + DPc]) + (red[80 + DPc] + red[120 + DPc]));
    }
    ++done;
    __syncthreads();
    if (status & 2) break;  // a NaN momentum ends the trajectory (hmc.py:56-57)
  }
  if (t < D) {
    ch.trj.w[ov + t] = wv[t];
    ch.trj.grad[ov + t] = gv[t];
    if (!eval_only) ch.p[ov + t] = pv[t];
    if (eval_only) { ch.cur.w[ov + t] = wv[t]; ch.cur.grad[ov + t] = gv[t]; }
  }
  if (t == 0) {
    ch.trj.ljl[c] = ljl;
    if (eval_only) {
      ch.cur.ljl[c] = ljl;
    } else {
      if (status) ch.status[c] |= status;
      ch.steps_left[c] = 0;
      ch.steps_done[c] += done;
    }
  }
}
