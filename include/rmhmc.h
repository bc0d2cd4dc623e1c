/*
 * rmhmc.h — C-ABI of the MI355X-native RMHMC hot path (librmhmc_hip.so).
 *
 * Drop-in boundary for the generalised-leapfrog RMHMC sampler of
 * emilemathieu/RiemannHamiltonianMonteCarlo, code/rmhmc.py:13-201
 * (Bayesian logistic regression, N(0, alpha I) prior).  The reference has no
 * FFI of its own: its boundary is the in-process Python call
 *     results_beta[i], results_time[i] = RMHMC(XX, t)        (code/main.py:52)
 * so the entry points below are what a ctypes binding for that call would
 * bind.  Every entry point cites the reference block it
 * replaces.  The same symbols are exported by the CPU oracle
 * (oracle/librmhmc_oracle.so), which is test infrastructure only.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no exceptions cross the boundary.
 *   - every function returns an int status: 0 = ok, <0 = error; the message is
 *     available from rmhmc_last_error(ctx) (ctx may be NULL after a failed
 *     rmhmc_create).
 *   - all buffers are caller-owned HOST memory unless the name ends in _dev;
 *     float64, row-major, chain-major: w[n_chains*D], G[n_chains*D*D] ...
 *   - the context is opaque, owns all device memory, and is not thread-safe;
 *     the library is re-entrant across contexts.
 *   - arithmetic type: float64 (the reference is float64 throughout).
 */
#ifndef RMHMC_H
#define RMHMC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rmhmc_ctx rmhmc_ctx;

/* status codes */
#define RMHMC_OK 0
#define RMHMC_ERR_INVALID (-1)     /* bad argument / call order                */
#define RMHMC_ERR_NO_DEVICE (-2)   /* no HIP device / extension cannot run     */
#define RMHMC_ERR_RUNTIME (-3)     /* HIP runtime error                        */
#define RMHMC_ERR_UNSUPPORTED (-4) /* shape / dtype outside the built kernels  */
#define RMHMC_ERR_NOMEM (-5)

/* dtype enum (only float64 is built; the reference is float64) */
#define RMHMC_F64 0

/* flags for rmhmc_create.  RMHMC_COMPAT reproduces the Python reference
 * bit-for-behaviour; clearing a bit gives the "corrected" variant. */
#define RMHMC_FLAG_MOMENTUM_LT (1u << 0) /* p = L^T z, L lower (rmhmc.py:60,80);
                                            cleared: p = L z, Cov(p) = G       */
#define RMHMC_FLAG_GUARDS (1u << 1)      /* RENORMALIZE guards rmhmc.py:81-85,
                                            125-130                            */
#define RMHMC_COMPAT (RMHMC_FLAG_MOMENTUM_LT | RMHMC_FLAG_GUARDS)
#define RMHMC_FLAG_FP32_METRIC (1u << 4) /* EXPERIMENT (BASELINE config 5 "fp32 vs fp64 tolerance sweep"): the metric
                                            assemblies X'diag(v)X run on the fp32 matrix cores (f32 operands and
                                            accumulators); everything else stays float64.  Not reference-compatible:
                                            measured errors are in DESIGN.md.  Ignored by the D <= 8 fused path.      */
#define RMHMC_FLAG_INT8_METRIC (1u << 5) /* HIP library: compute the two O(M D^2) passes (metric assembly, leverages)
                                          on the int8 matrix cores from exact signed-byte slices of the operands
                                          (fixed point, int32 accumulation, fp64 combination).  Slices S in bits
                                          12..14 (RMHMC_FLAG_INT8_SLICES(S), 4..7, 0 = default 6): norm-wise error
                                          of G about 1e-9 / 5e-12 / 3e-14 / 1e-14 for S = 4 / 5 / 6 / 7 (fp64
                                          summation itself: ~1e-15).  Contractions longer than 2^17/S terms are
                                          summed over several launches so the int32 accumulators cannot overflow
                                          for any data.  Not used by the fused D <= 8 path; ignored by the oracle,
                                          which is always fp64. */
#define RMHMC_FLAG_INT8_SLICES(S) (((uint32_t)(S) & 7u) << 12)
#define RMHMC_FLAG_INT8_CERTIFY (1u << 7) /* with RMHMC_FLAG_INT8_METRIC: rmhmc_set_data evaluates the worst-case error bound of
                                          the fixed-point assembly for the data it is given (csrc/metric_i8.hip.h, "Error bound";
                                          e.g. one outlier row 1000x the others coarsens every other row's fixed-point grid) and,
                                          when it exceeds RMHMC_INT8_CERTIFY_TOL, runs that data set on the fp64 matrix cores
                                          instead.  Without this bit the int8 path is used unconditionally (explicit request).
                                          The Python shims set it whenever they choose the int8 path by themselves. */
#define RMHMC_INT8_CERTIFY_TOL 1e-9
#define RMHMC_FLAG_INT8_INNER_FULL (1u << 10) /* with RMHMC_FLAG_INT8_METRIC at 6 slices: sum EVERY integer GEMM from all 6 slices.  By default
                                          two kinds of quantity that only steer the trajectory and enter no Hamiltonian use the 5 most
                                          significant slices of the same operands (15 slice products instead of 21, ~5e-12 instead of
                                          ~3e-14): the metric of the position fixed-point ITERATES before the last one (rmhmc.py:116-122,
                                          FixedIter = 1 .. K-2: it only steers the next iterate) and the leverages x_n' G^-1 x_n behind the
                                          trace term tr(G^-1 dG/dw_d) (rmhmc.py:64-77,142-156: it only enters the momentum updates).  The
                                          last iterate and every evaluation point (G, log det, gradient: everything that enters the
                                          Hamiltonian and rmhmc_metric) always use all 6.  Effect on theta / p after a step: < 1e-11
                                          relative (tests/test_gpu_int8_metric.py). */
#define RMHMC_FLAG_MMALA_FULL (1u << 6)  /* rmhmc_mmala_*: the full manifold MALA of BLR_mMALA.m (drift with the metric-
                                          derivative terms) instead of the simplified one of BLR_mMALA_Simp.m */
#define RMHMC_FLAG_ESS_WRAP (1u << 9)     /* rmhmc_ess / rmhmc_sample_stats: autocorrelations exactly as the reference's PYTHON
                                            tools.ac computes them, i.e. circular with period nFFT = nextpow2(S)+1
                                            (tools.py:16-23: lag l also collects lag nFFT-l; differs from the linear ones
                                            whenever S is close to a power of two).  Cleared (default): linear
                                            autocovariances = the MATLAB original (ac.m:78, 2^(k+1)-point FFT, no wrap).  */
#define RMHMC_FLAG_ORACLE_LITERAL (1u << 8) /* oracle only: form the DxDxD
                                               InvGdG tensor and use LU
                                               inv/solve like rmhmc.py:64-77  */

/* per-chain status bits written to status_out[] */
#define RMHMC_ST_NOT_PD (1 << 0)       /* a Cholesky pivot was <=0 or NaN       */
#define RMHMC_ST_NONFINITE (1 << 1)    /* w or p became inf/NaN                 */
#define RMHMC_ST_GUARD_P (1 << 2)      /* momentum guard fired (rmhmc.py:81)    */
#define RMHMC_ST_GUARD_W (1 << 3)      /* position guard fired (rmhmc.py:125)   */

const char *rmhmc_version(void);

/* Create a context for M data rows, D dimensions, n_chains independent chains
 * on HIP device `device_id`.  Fails with RMHMC_ERR_NO_DEVICE when no GPU is
 * usable (there is no CPU fallback in the product library). */
int rmhmc_create(rmhmc_ctx **out, int32_t device_id, int64_t M, int32_t D,
                 int64_t n_chains, int32_t dtype, uint32_t flags);

/* Tuning options: every switch of the HIP library that changes its schedule or the last bits of its results, as explicit
 * arguments (the library reads NO environment variables).  rmhmc_create is rmhmc_create_opts with no options.  Keys marked
 * [create] shape allocations or the choice of kernels and can only be given to rmhmc_create_opts; the others may also be changed
 * later with rmhmc_set_option (between calls: the context is not thread-safe).  rmhmc_options writes the active set as
 * "key=value key=value ..." (rmhmc_device_info appends it too).  Unknown key / value out of range: RMHMC_ERR_INVALID.
 *
 *   key               default  meaning
 *   graph             1        replay the launches of one global step from a hipGraph (0: plain launches)
 *   sorted            1        bulk samplers lay the chains out by decreasing post-burn-in work and shrink the launches of
 *                              the tail to the chains still running (results bit-identical either way)
 *   inflight          32       at most this many global steps (or graph replays) queued on the device ahead of the host
 *                              (0: unbounded)
 *   cdyn              1        first momentum pass of a step re-uses the c = v(1-2p) tiles of chains that did not just reject
 *   crestore          1        ... and the tiles of the chains that did are recomputed for them alone (k_crestore)
 *   i8_force_rebase   0        test hook: delta assemblies treat every chain as if its fixed-point exponent had changed
 *   ccache   [create] 1        c tiles kept per position for the momentum passes (0: recomputed by every pass)
 *   medium   [create] 1        one-launch leapfrog step / HMC trajectory for small batches (8 < D <= 32, M <= 2048, <= 512 chains)
 *   fused    [create] 1        LDS-resident many-steps-per-launch kernel for D <= 8
 *   hmc_traj_maxn [create] -1  largest batch for the one-launch HMC trajectory (-1: the built-in rule)
 *   fsplit   [create] 0        row ranges per chain of the fp64 assembly of small batches (0: chosen from the batch size)
 *   nsplit_max [create] 64     cap on the row splits of the 16-chains-per-wavefront passes
 *   nsplit_waves [create] -1   wavefronts per launch those row splits aim at (-1: 2048 for D <= 64, 6144 for the blocked large-D passes)
 *   i8_tail  [create] -1       int8 path: ragged last pair block as tiles of its own: -1 when it pays, 0 never, 1 always
 *   i8_delta [create] 1        int8 path, 6 slices: the metric at the end of a leapfrog step as G(last iterate) + the assembly
 *                              of the v differences (4 slices)
 *   i8_delta_inner [create] 1  ... and the second position iterate from the first likewise
 * The CPU oracle accepts every key and ignores the values. */
typedef struct { const char *key; int64_t value; } rmhmc_option;
int rmhmc_create_opts(rmhmc_ctx **out, int32_t device_id, int64_t M, int32_t D, int64_t n_chains, int32_t dtype,
                      uint32_t flags, const rmhmc_option *opts, int32_t n_opts);
int rmhmc_set_option(rmhmc_ctx *ctx, const char *key, int64_t value);
int rmhmc_get_option(rmhmc_ctx *ctx, const char *key, int64_t *value_out);
int rmhmc_options(rmhmc_ctx *ctx, char *buf, size_t len);
void rmhmc_destroy(rmhmc_ctx *ctx);
const char *rmhmc_last_error(const rmhmc_ctx *ctx);
/* Human-readable device / build description into buf (NUL-terminated). */
int rmhmc_device_info(rmhmc_ctx *ctx, char *buf, size_t len);

/* Inputs of RMHMC(XX, t): XX float64 (M,D) C-contiguous, t float64 (M,) in
 * {0,1}; alpha = prior variance (rmhmc.py:19, hard-coded 100 there).
 * Borrowed for the duration of the call only (copied to HBM). */
int rmhmc_set_data(rmhmc_ctx *ctx, const double *X, const double *t,
                   double alpha);

/* ---- unit entry points (the reference's inline "callbacks") -------------- */

/* C1  log-joint  LJL(w) = f't - sum log(1+e^f) + log N(w;0,alpha I)
 *     rmhmc.py:31-34,166-169 ; tools.py:10-14.   w[n*D] -> ljl_out[n]        */
int rmhmc_log_posterior(rmhmc_ctx *ctx, const double *w, double *ljl_out);

/* C2+C3  gradient and metric at w.   rmhmc.py:51-60, 99-100, 134-140.
 *     G_out[n*D*D] (full symmetric) | NULL,
 *     half_logdet_out[n] = sum log diag chol(G) (rmhmc.py:171,175) | NULL,
 *     grad_out[n*D] | NULL.                                                  */
int rmhmc_metric(rmhmc_ctx *ctx, const double *w, double *G_out,
                 double *half_logdet_out, double *grad_out);

/* C4  the two contractions of the metric-derivative tensor that the sampler
 *     uses (rmhmc.py:64-77,104-107,142-161):
 *       trace_out[n*D]  = tr(G^-1 dG/dw_d)
 *       quad_out[n*D]   = u' (dG/dw_d) u  with u = G^-1 p      (NULL if p NULL)
 *     so that LastTerm_d = 0.5*quad_d.                                        */
int rmhmc_metric_terms(rmhmc_ctx *ctx, const double *w, const double *p,
                       double *trace_out, double *quad_out);

/* a3-a6  nsteps[c] generalised leapfrog steps of size eps in direction
 *     dir[c] (+1/-1) with K fixed-point iterations (rmhmc.py:96-163).
 *     w, p: in/out [n*D].  half_logdet_out[n] | NULL: at the final point.
 *     status_out[n] | NULL.                                                  */
int rmhmc_leapfrog(rmhmc_ctx *ctx, double *w, double *p, double eps,
                   const int32_t *dir, const int32_t *nsteps, int32_t K,
                   double *half_logdet_out, int32_t *status_out);

/* a1-a7  one full MCMC transition with caller-supplied randomness, in the
 *     reference's draw order (rmhmc.py:80,89,90,181):
 *       z[n*D]   ~ randn(1,D)      momentum noise
 *       u_len[n] ~ rand()          RandomStep = ceil(u_len*L)
 *       g_dir[n] ~ randn()         TimeStep = +1 iff g_dir > 0.5
 *       u_acc[n] ~ rand()          accept iff Ratio>0 or Ratio>log(u_acc)
 *     w in/out.  All outputs optional (NULL).                                 */
int rmhmc_transition(rmhmc_ctx *ctx, double *w, const double *z,
                     const double *u_len, const double *g_dir,
                     const double *u_acc, int32_t L, double eps, int32_t K,
                     int32_t *accepted_out, int32_t *nsteps_out,
                     double *H_cur_out, double *H_prop_out,
                     double *w_prop_out, double *p_prop_out,
                     double *half_logdet_prop_out, int32_t *status_out);

/* ---- bulk entry points -------------------------------------------------- */

/* a0-a8  RMHMC(XX, t, n_iter, burn_in, L, eps, K) for all chains.
 *     Randomness: Philox4x32-10 keyed by seed, counter (chain_offset+chain,
 *     iteration, draw) — identical streams whatever the sharding.
 *     theta0[n*D] | NULL (NULL: 1e-3 everywhere, rmhmc.py:27).
 *     samples_out[n*S*D], S = n_iter-burn_in; row s of chain c = state after
 *     iteration burn_in+s (row 0 is left unwritten by rmhmc.py:190-191; here
 *     it holds the state after iteration burn_in).
 *     accept_out[n] | NULL : accepted proposals over all n_iter iterations.
 *     steps_out[n]  | NULL : leapfrog steps executed after burn-in.
 *     seconds_out   | NULL : wall seconds of the post-burn-in phase
 *                            (TimeTaken, rmhmc.py:194-198).                  */
int rmhmc_sample(rmhmc_ctx *ctx, int64_t n_iter, int64_t burn_in, int32_t L,
                 double eps, int32_t K, uint64_t seed, int64_t chain_offset,
                 const double *theta0, double *samples_out,
                 int64_t *accept_out, int64_t *steps_out,
                 double *seconds_out);

/* Stateful form used by the benchmark and the multi-GPU driver: set the
 * chain state once, then advance every chain by `n_steps` leapfrog steps
 * (chains start a new transition whenever their trajectory ends, so every
 * chain executes exactly n_steps leapfrog steps).  rmhmc_chains_run is
 * synchronous: the device is idle when it returns.                          */
int rmhmc_chains_init(rmhmc_ctx *ctx, const double *theta0, uint64_t seed,
                      int64_t chain_offset, int32_t L, double eps, int32_t K);
int rmhmc_chains_run(rmhmc_ctx *ctx, int64_t n_steps);
/* Current position of every chain, completed transitions, accepted ones.    */
int rmhmc_chains_state(rmhmc_ctx *ctx, double *w_out, int64_t *iters_out,
                       int64_t *accept_out);
/* Checkpoint / resume: (w, iters, accepted) from rmhmc_chains_state is a complete checkpoint.  To resume, call
 * rmhmc_chains_init with theta0 = the saved w (same seed, chain_offset, L, eps, K) and then this function with the
 * saved counters.  Randomness is keyed by (seed, chain, iteration), so a chain that was in mid-trajectory replays
 * that transition from its start and every chain continues bit for bit as if it had never stopped.          */
int rmhmc_chains_restore(rmhmc_ctx *ctx, const int64_t *iters, const int64_t *accepted);
/* Device seconds (HIP events on the library's stream) of the kernel named
 * `which` accumulated since the last rmhmc_chains_init, and its launch
 * count; which = "assemble" | "factor" | "momentum" | "leverage" | "total". */
int rmhmc_kernel_time(rmhmc_ctx *ctx, const char *which, double *seconds_out,
                      int64_t *launches_out);

/* ---- widening, SURVEY.md 8(f)-1: plain HMC with identity mass, code/hmc.py:12-99 ------------------
 * Same data, context and conventions as above.  Differences from RMHMC that the reference has and that are
 * kept: trajectory length RandomStep = ceil(u_len*L) with L = 100 by default and no direction flip
 * (hmc.py:48), theta0 = 0 (hmc.py:27), a NaN momentum ends the trajectory (hmc.py:56-57) and the proposal
 * is then rejected.                                                                                    */

/* hmc.py:41-80 with caller-supplied randomness (draw order hmc.py:41,48,77):
 *   z[n*D] ~ randn(1,D), u_len[n] ~ rand(), u_acc[n] ~ rand().  w in/out; outputs optional.            */
int rmhmc_hmc_transition(rmhmc_ctx *ctx, double *w, const double *z, const double *u_len,
                         const double *u_acc, int32_t L, double eps, int32_t *accepted_out,
                         int32_t *nsteps_out, double *H_cur_out, double *H_prop_out,
                         double *w_prop_out, double *p_prop_out);

/* HMC(XX, t, n_iter, burn_in, L, eps) for all chains; arguments as rmhmc_sample
 * (theta0 NULL: zeros, hmc.py:27).                                                                     */
int rmhmc_hmc_sample(rmhmc_ctx *ctx, int64_t n_iter, int64_t burn_in, int32_t L, double eps,
                     uint64_t seed, int64_t chain_offset, const double *theta0,
                     double *samples_out, int64_t *accept_out, int64_t *steps_out,
                     double *seconds_out);

/* ---- widening, SURVEY.md 8(f)-2: the min-ESS half of the metric on the device ------------------------
 * Geyer initial-monotone-sequence ESS with the semantics of tools.CalculateESS(Samples, MaxLag = S-1)
 * (code/tools.py:32-74; authors_code/.../Results/CalculateESS.m), computed directly (autocovariances lag by
 * lag until the pair sum turns non-positive) instead of through an FFT of all lags: identical to the MATLAB
 * original (2^(k+1)-point FFT, ac.m:78); with RMHMC_FLAG_ESS_WRAP identical to the Python translation, whose
 * (nextpow2(S)+1)-point FFT wraps lag nFFT-l onto lag l (tools.py:23).                                  */

/* samples[n*S*P] (host, chain-major, row s = sample s) -> ess_out[n*P].  n here is any number of sample
 * blocks (it need not equal the context's n_chains).  S <= 20000.                                      */
int rmhmc_ess(rmhmc_ctx *ctx, const double *samples, int64_t n, int64_t S, int32_t P, double *ess_out);

/* rmhmc_sample without the sample transfer: per-chain posterior mean / population variance / ESS of every
 * dimension over the S = n_iter-burn_in saved states, reduced on the device (config 4 at S = 5000 would be
 * 168 GB of raw samples).  mean_out, var_out, ess_out: [n*D], each optional.                            */
int rmhmc_sample_stats(rmhmc_ctx *ctx, int64_t n_iter, int64_t burn_in, int32_t L, double eps, int32_t K,
                       uint64_t seed, int64_t chain_offset, const double *theta0, double *mean_out,
                       double *var_out, double *ess_out, int64_t *accept_out, int64_t *steps_out,
                       double *seconds_out);

/* Progress reports of the bulk samplers (rmhmc_sample*, rmhmc_hmc_sample).  The reference prints the iteration count and the
 * acceptance rate of the last window whenever IterationNum+1 is a multiple of 50 (rmhmc.py:38-45; hmc.py:85-89) and a banner
 * when burn-in completes (rmhmc.py:194-196).  With a callback set, fn(RMHMC_EV_PROGRESS, m, accepted, iterations, user) is
 * called from the calling thread for the milestones m = first, first+every, first+2 every, ...; accepted / iterations =
 * accepted proposals / completed transitions summed over all chains at that moment (the window's acceptance rate is the ratio
 * of their increments between two reports).
 *   One chain: the run is cut at every milestone, so the report comes exactly when m transitions are complete, as the
 *   reference prints it.
 *   Several chains: nobody is stopped; the report for m comes at the first host synchronisation after the SLOWEST chain has
 *   completed m transitions (the others are ahead, iterations > n m), and milestones passed since the last report are merged
 *   into one call with the largest of them.
 * A milestone equal to burn_in+1 falls where the reference prints it: rmhmc_sample* after the burn-in event (rmhmc.py:38 prints at the
 * top of the next iteration), rmhmc_hmc_sample before it (hmc.py:85-89 prints at the bottom of the iteration); rmhmc_hmc_sample
 * reports no milestone beyond burn_in+1 (hmc.py:83-89 prints during burn-in only), so nothing cuts its TimeTaken window.
 * fn(RMHMC_EV_BURNIN_DONE, burn_in+1, accepted, iterations, user) is called when the burn-in phase ends (every chain at exactly
 * burn_in+1 transitions), just before the TimeTaken timer starts.  Samples do not depend on any of it.  fn = NULL switches the
 * reports off.  The oracle accepts the call and never reports.                                                            */
typedef void (*rmhmc_progress_fn)(int32_t event, int64_t iterations_done, int64_t accepted_total, int64_t iterations_total,
                                  void *user);
#define RMHMC_EV_PROGRESS 0
#define RMHMC_EV_BURNIN_DONE 1
int rmhmc_set_progress(rmhmc_ctx *ctx, rmhmc_progress_fn fn, int64_t first, int64_t every, void *user);

/* Certificate of the int8 metric path for the data of the last rmhmc_set_data: bound_out = worst-case error of any G_ab
 * relative to sqrt(G0_aa G0_bb), G0 = X'X/4 + I/alpha (0 when the path was not requested); active_out = 1 when the int8 kernels
 * are in use, 0 when the path was not requested or RMHMC_FLAG_INT8_CERTIFY sent this data set to the fp64 kernels.  The oracle
 * reports (0, 0).                                                                                                          */
int rmhmc_int8_certificate(rmhmc_ctx *ctx, double *bound_out, int32_t *active_out);

/* ---- device-resident write-out ("one RCCL gather over xGMI at sample write-out") -----------------------
 * The same calls as rmhmc_chains_state / rmhmc_sample / rmhmc_sample_stats with every OUTPUT ARRAY in DEVICE memory of the
 * context's GPU (plain device pointers, e.g. a torch tensor's data_ptr(); same shapes, same element types, unpadded and
 * contiguous; each may be NULL).  Nothing crosses PCIe: the sampler saves straight into samples_dev, so that the multi-GPU
 * driver can hand the buffers to RCCL (ncclGather / all_gather over xGMI) without a host bounce.  Inputs (theta0) and the
 * scalar seconds_out stay host memory.  The calls are synchronous: the library's stream is idle on return, so any other
 * stream may read the buffers afterwards.  In the CPU oracle "device" memory is host memory.                              */
int rmhmc_chains_state_dev(rmhmc_ctx *ctx, double *w_dev, int64_t *iters_dev, int64_t *accept_dev);
int rmhmc_sample_dev(rmhmc_ctx *ctx, int64_t n_iter, int64_t burn_in, int32_t L, double eps, int32_t K, uint64_t seed,
                     int64_t chain_offset, const double *theta0, double *samples_dev, int64_t *accept_dev,
                     int64_t *steps_dev, double *seconds_out);
int rmhmc_sample_stats_dev(rmhmc_ctx *ctx, int64_t n_iter, int64_t burn_in, int32_t L, double eps, int32_t K,
                           uint64_t seed, int64_t chain_offset, const double *theta0, double *mean_dev,
                           double *var_dev, double *ess_dev, int64_t *accept_dev, int64_t *steps_dev,
                           double *seconds_out);

/* ---- widening, SURVEY.md 8(f)-4: simplified manifold MALA ---------------------------------------------
 * authors_code/Bayes_Log_Reg/MCMC/BLR_mMALA_Simp.m:175-290 (MATLAB only: the Python reference has no mMALA and no
 * MATLAB/Octave runs here, so this row has NO executable reference: "parity unpinned"; the GPU path is checked against
 * the oracle's restatement of the .m file and statistically against RMHMC and the paper's Table 3).
 *   proposal  w' = w + eps/2 G^-1 grad + N(0, eps G^-1)        (:217-219, eps = StepSize = 1)
 *   accept    LJL' + log q(w|w') - LJL - log q(w'|w)             (:227-254)
 * Uses the metric / gradient / Cholesky kernels of the RMHMC path; one point evaluation per transition.   */
int rmhmc_mmala_transition(rmhmc_ctx *ctx, double *w, const double *z, const double *u_acc, double eps,
                           int32_t *accepted_out, double *ratio_out, double *w_prop_out);
int rmhmc_mmala_sample(rmhmc_ctx *ctx, int64_t n_iter, int64_t burn_in, double eps, uint64_t seed,
                       int64_t chain_offset, const double *theta0, double *samples_out,
                       int64_t *accept_out, double *seconds_out);

#ifdef __cplusplus
}
#endif
#endif /* RMHMC_H */
