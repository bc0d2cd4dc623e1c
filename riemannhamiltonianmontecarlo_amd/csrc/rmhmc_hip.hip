// rmhmc_hip.hip — host side of librmhmc_hip.so: the C-ABI of include/rmhmc.h on top of the gfx950
// kernels in kernels.hip.h.  No CPU fallback: every entry point needs a HIP device.
//
// Scheduling: the generalised leapfrog of rmhmc.py:96-163 is a fixed sequence of
// kernels, each over all chains; chains are independent, so transitions are started and finished
// asynchronously (k_iter_begin / k_iter_end act only on chains whose trajectory ended) and every
// chain executes one leapfrog step per "global step" whatever its RandomStep.
#include "../../include/rmhmc.h"
#include "kernels.hip.h"
#include "fused_small.hip.h"
#include "large_d.hip.h"
#include "metric_i8.hip.h"
#include "medium_step.hip.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <vector>

namespace {

thread_local char g_err[512] = "";  // message of a failed rmhmc_create (no context exists yet); per host thread
// hipFuncAttributeMaxDynamicSharedMemorySize is per (function, device), not per context: always raise it to the hardware
// limit, so that two live contexts of different shapes cannot lower each other's launch limit
constexpr int MAX_LDS = 160 * 1024;

struct EvPair { hipEvent_t a, b; };

// The batch of chains a launch works on, with its view of the per-chain arrays and the scratch sized for it.  There is one per context
// (the whole batch); the work-sorted sampler narrows a copy of it to the prefix of chains still running (launch_global_step_prefix).
// (Rounds 1-2 could cut the batch into 2-4 groups ping-ponged over two streams; measured twice without gain on MI355X - the co-running
// light kernels are starved and the heavy ones slow down by the same total, profiles/r01_groups_sweep.txt - and removed.)
struct Group {
  Chains ch{};
  int n = 0;
  long long off = 0;
  int nsplit = 1;
  int8_t* Vs = nullptr;  // int8 metric path: slices of v, [S][nks][nCp][32]
  int8_t* Qs = nullptr;  // slices of the doubled G^-1 entries, [S][nkp][nCp][32]
  double* qscale = nullptr;
  double *Gpart = nullptr, *Rpart = nullptr;  // k-split planes of small batches ([ksplit][n][DP*DP], [ksplit][n][Mp])
  int* Tq = nullptr;      // int32 accumulators of the ragged last pair block, [pieces][S][nCp][32 ntail] (k_assemble_i8_tail)
  int tail_pieces = 1;
  int ksplit_a = 1, ksplit_l = 1;
  int fsplit = 1;  // fp64 assembly of small batches: row ranges per chain (planes in Gpart)
  int* vbad = nullptr;
  int* vexp = nullptr;   // per-chain extra binary digits of the v grid (VSlice)
  int* vexp_d = nullptr; int* rebase = nullptr; unsigned long long* dmax = nullptr;  // delta assembly (VSlice / I8Delta)
  double* Gbase = nullptr;  // large-D path: copy of the G a delta assembly adds to (Gq is factored in place)
  d4* ctile = nullptr;   // c = v(1-2p) of trj.w in the tile layout of k_mompass, [ceil(n/16)][Mp/16][64] x 4 doubles
  int nCp = 0;
};

enum Cls { HEAVY = 0, LIGHT = 1 };  // (documentation of a launch's kind: matrix-core pass over all rows / per-chain kernel)

// Tuning options (include/rmhmc.h: rmhmc_create_opts / rmhmc_set_option).  The library reads no environment variables.
struct Options {
  int64_t graph = 1, sorted = 1, inflight = 32, cdyn = 1, crestore = 1, i8_force_rebase = 0;                 // run time
  int64_t ccache = 1, medium = 1, fused = 1, hmc_traj_maxn = -1, fsplit = 0, nsplit_max = 64, nsplit_waves = -1, i8_tail = -1,   // create time
          i8_delta = 1, i8_delta_inner = 1;
};
struct OptionDesc { const char* key; int64_t Options::*slot; bool create_only; int64_t lo, hi; };
const OptionDesc kOptions[] = {
    {"graph", &Options::graph, false, 0, 1},
    {"sorted", &Options::sorted, false, 0, 1},
    {"inflight", &Options::inflight, false, 0, 1 << 20},
    {"cdyn", &Options::cdyn, false, 0, 1},
    {"crestore", &Options::crestore, false, 0, 1},
    {"i8_force_rebase", &Options::i8_force_rebase, false, 0, 1},
    {"ccache", &Options::ccache, true, 0, 1},
    {"medium", &Options::medium, true, 0, 1},
    {"fused", &Options::fused, true, 0, 1},
    {"hmc_traj_maxn", &Options::hmc_traj_maxn, true, -1, (int64_t)1 << 40},
    {"fsplit", &Options::fsplit, true, 0, 64},
    {"nsplit_max", &Options::nsplit_max, true, 1, 1 << 20},
    {"nsplit_waves", &Options::nsplit_waves, true, -1, 1 << 20},
    {"i8_tail", &Options::i8_tail, true, -1, 1},
    {"i8_delta", &Options::i8_delta, true, 0, 1},
    {"i8_delta_inner", &Options::i8_delta_inner, true, 0, 1},
};
const OptionDesc* find_option(const char* key) {
  if (!key) return nullptr;
  for (const OptionDesc& d : kOptions)
    if (!strcmp(d.key, key)) return &d;
  return nullptr;
}

}  // namespace

struct rmhmc_ctx {
  int device = 0;
  int64_t M = 0, n = 0;
  int D = 0, DP = 0, NB = 0, Mp = 0, nblk = 0;
  uint32_t flags = 0;
  double alpha = 100.0;
  hipStream_t stream = nullptr;
  Options opt{};
  std::vector<hipEvent_t> flow;  // flow control (option inflight): events recorded every few global steps, see flow_tick
  long long flow_steps = 0, flow_ticks = 0;
  int* stale_list_alloc = nullptr;  // list of the chains that have just rejected a proposal (k_crestore; in use when cdyn and crestore are on)
  DevData dd{};
  Chains ch{};  // whole-batch view (uploads / downloads)
  std::vector<Group> groups;
  std::vector<void*> allocs;
  bool have_data = false, chains_ready = false;
  int sampler = 0;           // 0: RMHMC (rmhmc.py), 1: plain HMC (hmc.py) -- selects the global step
  bool big = false;          // large-D path: 64 < D <= 256 (large_d.hip.h)
  int nbk = 1, npairs = 1;   // 64-column blocks and block pairs of the large-D path
  double *d_Wd = nullptr, *d_hpart = nullptr, *d_Gcopy = nullptr;
  bool want_G = false;
  bool fused = false;        // small-problem path: D <= 8 and X fits in LDS (fused_small.hip.h)
  size_t fused_lds = 0;
  bool medium = false;       // one-launch leapfrog step for small batches with 8 < D <= 32 (medium_step.hip.h)
  size_t medium_lds = 0;
  // (options crestore / cdyn / ccache: c = v(1-2p) kept per position in the momentum pass's tile layout; the first pass of a step re-uses
  //  the tiles of chains that did not just reject; those of the chains that did are recomputed by k_crestore, 16 to a wavefront)
  bool hmc_traj = false;     // plain HMC in small batches: one launch per trajectory (k_hmc_traj)
  // int8 metric path (metric_i8.hip.h)
  bool i8 = false;
  int i8S = 0, i8_nks = 0, i8_bn = 128, i8_chunk = 1;  // i8_chunk: k-stages (of 32) per launch
  int i8_inner_drop = 1;     // inner assemblies from S-1 slices (launch_assemble; RMHMC_FLAG_INT8_INNER_FULL: off)
  // (options i8_delta / i8_delta_inner: G at the end of a leapfrog step, and the second position iterate, as the previous iterate's G + the
  //  assembly of the v differences; i8_force_rebase: tests treat every chain as if its v exponent had changed; i8_tail: ragged last pair
  //  block as tiles of its own: -1 when it pays (launch_assemble_i8_t), 0 never, 1 always)
  int8_t* d_Zs = nullptr;
  int* d_ze = nullptr;
  int8_t* d_Zt = nullptr;   // leverage pass: x_a x_b sliced per data row, [S][nkp][NRp][32]
  int* d_zre = nullptr;
  double *d_cmin = nullptr, *d_cmax = nullptr;  // min_n |x_nd|, max_n |x_nd| per column (VSlice)
  double* d_zscale = nullptr;
  int i8_nkp = 0, i8_NRp = 0;
  I8Pairs pairs{};
  bool i8_requested = false;   // RMHMC_FLAG_INT8_METRIC given at create (i8 may be switched off by the set_data certificate)
  double i8_bound = 0.0;       // certificate of the last set_data (metric_i8.hip.h: "Error bound"), 0 when the path is not requested
  // sampler parameters of the stateful API
  int L = 6, K = 4;
  double eps = 0.5;
  uint64_t seed = 0;
  int64_t chain_offset = 0;
  // unit-API staging (device)
  double *d_z = nullptr, *d_ulen = nullptr, *d_gdir = nullptr, *d_uacc = nullptr;
  int *d_nsteps = nullptr, *d_dir = nullptr, *d_done = nullptr;
  long long* d_steps0 = nullptr;
  unsigned long long* d_miniter = nullptr;
  // work-sorted layout of the bulk sampler (sample_core)
  int* d_orig = nullptr;            // [n] position -> chain index in the caller's order
  long long* d_T = nullptr;         // [2n] scratch: trajectory-length sums / un-permuted counters
  bool sorted = false;              // in use by the running sampler (iter_params)
  bool counters_sorted = false;     // the last sampler run left its counters un-permuted in d_T
  // progress reports of the bulk samplers (rmhmc_set_progress)
  rmhmc_progress_fn progress_fn = nullptr;
  void* progress_user = nullptr;
  long long progress_first = 0, progress_every = 0, progress_next = 0;
  // timing
  bool timing = false;
  std::map<std::string, std::vector<EvPair>> events;
  std::vector<EvPair> pool;
  char err[512] = "";
};

namespace {

int fail(rmhmc_ctx* ctx, int code, const std::string& msg) {
  snprintf(ctx ? ctx->err : g_err, 512, "%s", msg.c_str());
  return code;
}

#define HIPCK(call)                                                                                        \
  do {                                                                                                     \
    hipError_t e_ = (call);                                                                                \
    if (e_ != hipSuccess)                                                                                  \
      return fail(ctx, RMHMC_ERR_RUNTIME, std::string(hipGetErrorString(e_)) + " in " #call " (" __FILE__ ":" + std::to_string(__LINE__) + ")"); \
  } while (0)

template <typename T>
int dalloc(rmhmc_ctx* ctx, T** p, size_t count) {
  void* q = nullptr;
  size_t bytes = (count ? count : 1) * sizeof(T);
  HIPCK(hipMalloc(&q, bytes));
  HIPCK(hipMemsetAsync(q, 0, bytes, ctx->stream));
  ctx->allocs.push_back(q);
  *p = (T*)q;
  return RMHMC_OK;
}

struct Timed {
  rmhmc_ctx* ctx;
  hipStream_t st;
  EvPair ev{};
  bool on;
  Timed(rmhmc_ctx* c, const char* name, hipStream_t s) : ctx(c), st(s), on(c->timing) {
    if (!on) return;
    if (!ctx->pool.empty()) { ev = ctx->pool.back(); ctx->pool.pop_back(); }
    else { (void)hipEventCreate(&ev.a); (void)hipEventCreate(&ev.b); }
    (void)hipEventRecord(ev.a, st);
    ctx->events[name].push_back(ev);
  }
  ~Timed() { if (on) (void)hipEventRecord(ev.b, st); }
};

// Launch one kernel (or a short run of them) on the batch g.  fn(stream) enqueues it; name keys the optional event timing.
template <typename F>
void launch(rmhmc_ctx* ctx, Group& g, Cls, const char* name, F&& fn) {
  (void)g;
  Timed t(ctx, name, ctx->stream);
  fn(ctx->stream);
}

// Flow control (option inflight): the host is never more than `inflight` global steps ahead of the device.  Every inflight/4 steps an
// event goes into the stream, and before the host records the fifth it waits for the first.  A run of several hundred steps used to be
// queued in one go (~45 dispatches per step: tens of thousands of AQL packets), which costs nothing on the bare runtime but is what an
// intercepting tool's proxy queue cannot take ("Profiler note"); a bounded queue is the better citizen anyway.
void flow_tick(rmhmc_ctx* ctx, long long steps = 1) {
  const long long w = ctx->opt.inflight;
  if (w <= 0) return;
  ctx->flow_steps += steps;
  const long long stride = std::max<long long>(1, w / 4);
  if (ctx->flow_steps < stride) return;
  ctx->flow_steps = 0;
  if (ctx->flow.empty()) return;  // (created by rmhmc_create_opts)
  hipEvent_t e = ctx->flow[ctx->flow_ticks % 4];
  if (ctx->flow_ticks >= 4) (void)hipEventSynchronize(e);
  (void)hipEventRecord(e, ctx->stream);
  ctx->flow_ticks++;
}

#define NB_SWITCH(ctx, ...)                                               \
  switch ((ctx)->NB) {                                                    \
    case 1: { constexpr int NB_ = 1; __VA_ARGS__; } break;                \
    case 2: { constexpr int NB_ = 2; __VA_ARGS__; } break;                \
    case 3: { constexpr int NB_ = 3; __VA_ARGS__; } break;                \
    default: { constexpr int NB_ = 4; __VA_ARGS__; } break;               \
  }

#define I8_SWITCH(ctx, ...) I8_SWITCH_S((ctx)->i8S, __VA_ARGS__)
#define I8_SWITCH_S(sval, ...)                                                     \
  switch (sval) {                                                                  \
    case 4: { constexpr int S_ = 4, WN_ = 4, TN_ = 1; __VA_ARGS__; } break;        \
    case 5: { constexpr int S_ = 5, WN_ = 4, TN_ = 1; __VA_ARGS__; } break;        \
    case 6: { constexpr int S_ = 6, WN_ = 4, TN_ = 1; __VA_ARGS__; } break;        \
    default: { constexpr int S_ = 7, WN_ = 2, TN_ = 1; __VA_ARGS__; } break;       \
  }

// Delta assembly of the evaluation that ends a leapfrog step (I8Delta in metric_i8.hip.h): the last position iterate has left its N in
// the slice planes and its G - summed from all six slices - in Gq, and v moves by 1e-6 between the two points, so the difference
// needs four slices (10 slice products) where the full assembly needs six (21).
static bool use_delta(const rmhmc_ctx* ctx, const Group& g) {
  return ctx->i8 && ctx->opt.i8_delta && (!ctx->big || g.Gbase) && ctx->i8S == 6 && g.ctile && g.ksplit_a <= 1 && ctx->K >= 2 && g.dmax;
}
// The second position iterate as a delta of the first (both inner iterates on five slices: it = 2 < K - 1; later inner iterates would need
// the N of a predecessor whose planes hold differences)
static bool use_delta_inner(const rmhmc_ctx* ctx, const Group& g, int it) {
  return use_delta(ctx, g) && ctx->opt.i8_delta_inner && ctx->i8_inner_drop && it == 2 && it < ctx->K - 1;
}
template <int MODE>
void launch_rowpass(rmhmc_ctx* ctx, Group& g, const double* w, double* out0, double* out2 = nullptr, bool delta = false) {
  launch(ctx, g, HEAVY, "rowpass", [&](hipStream_t st) {
    if (ctx->big) {
      dim3 grid((unsigned)((g.n + 15) / 16), g.nsplit);
      hipLaunchKernelGGL((k_rowpass_big<MODE>), grid, dim3(256), 0, st, ctx->dd, g.n, g.nsplit, ctx->nbk, g.ch.phase, w, out0, out2,
                         g.ch.gpart, g.ch.ljl_part, g.ctile);
      return;
    }
    dim3 grid((unsigned)((g.n + 63) / 64), g.nsplit);
    if (ctx->i8 && MODE != RP_G) {  // int8 metric path: v goes out as byte slices (no fp64 row vector, no k_vsplit)
      // (vbad and dmax are clear: the factor kernel behind the previous assembly has reset them, Chains::i8_vbad / i8_dmax)
      const VSlice vs{g.Vs, g.vbad, ctx->i8_nks, g.nCp, ctx->i8S, g.vexp, ctx->d_cmin, ctx->d_cmax};
      // (with c tiles nobody reads a natural-layout c on this path: k_mompass and k_trvec take the tiles)
      if (MODE != RP_G && delta) {
        VSlice vd = vs;
        vd.vexp_d = g.vexp_d; vd.rebase = g.rebase; vd.dmax = g.dmax; vd.force_rebase = (int)ctx->opt.i8_force_rebase;
        if (MODE == RP_F) {
          NB_SWITCH(ctx, hipLaunchKernelGGL((k_rowpass<NB_, RP_F, 6, false, true>), grid, dim3(256), 0, st, ctx->dd, g.n, g.nsplit, g.ch.phase, w,
                                            out0, nullptr, g.ch.gpart, g.ch.ljl_part, vd, g.ctile, g.ch.cstale));
        } else {
          NB_SWITCH(ctx, hipLaunchKernelGGL((k_rowpass<NB_, RP_V, 6, true, true>), grid, dim3(256), 0, st, ctx->dd, g.n, g.nsplit, g.ch.phase, w,
                                            out0, out2, g.ch.gpart, g.ch.ljl_part, vd, g.ctile, g.ch.cstale));
        }
      } else if (g.ctile && MODE == RP_F) {
        I8_SWITCH(ctx, NB_SWITCH(ctx, hipLaunchKernelGGL((k_rowpass<NB_, MODE, S_, false>), grid, dim3(256), 0, st, ctx->dd, g.n, g.nsplit, g.ch.phase, w,
                                                        out0, nullptr, g.ch.gpart, g.ch.ljl_part, vs, g.ctile, g.ch.cstale)));
      } else {
        I8_SWITCH(ctx, NB_SWITCH(ctx, hipLaunchKernelGGL((k_rowpass<NB_, MODE, S_>), grid, dim3(256), 0, st, ctx->dd, g.n, g.nsplit, g.ch.phase, w, out0,
                                                        out2, g.ch.gpart, g.ch.ljl_part, vs, g.ctile, g.ch.cstale)));
      }
      return;
    }
    NB_SWITCH(ctx, hipLaunchKernelGGL((k_rowpass<NB_, MODE>), grid, dim3(256), 0, st, ctx->dd, g.n, g.nsplit, g.ch.phase, w, out0,
                                      out2, g.ch.gpart, g.ch.ljl_part, VSlice{}, g.ctile, g.ch.cstale));
  });
}

// int8 metric path: cut v into slices, then the sliced GEMM against the fixed slices of x_a x_b (metric_i8.hip.h)
template <int S, int WN, int TN>
void launch_assemble_i8_t(rmhmc_ctx* ctx, Group& g, const double* v, hipStream_t st, int part, bool delta = false, int seff = 6, int need_lo = 0,
                          int need_hi = 0) {
  // delta: this instantiation on the V planes seff - S .. seff - 1 (seff = 6: all digits count; 5: an inner iterate), added to Gq if the
  // digits *dmax asks for lie in [need_lo, need_hi] (I8Delta)
  const size_t vplane = (size_t)ctx->i8_nks * g.nCp * 32;
  const int8_t* const Vs = g.Vs + (delta ? (size_t)(seff - S) * vplane : 0);
  const I8Delta dl{delta ? g.dmax : nullptr, delta ? g.rebase : nullptr, delta ? std::ldexp(1.0, -8 * (seff - S)) : 1.0, need_lo, need_hi,
                   (delta && ctx->big) ? g.Gbase : nullptr};
  const int* const vexp = delta ? g.vexp_d : g.vexp;
  const int acc0 = delta ? 2 : 0;
  if (part == 0) {
    if (delta) {  // (large-D path: the planes hold N of the previous iterate, they get the digits of the difference)
      (void)hipMemsetAsync(g.dmax, 0, sizeof(unsigned long long), st);
      hipLaunchKernelGGL((k_vsplit<S, true>), dim3((unsigned)((g.n + 7) / 8)), dim3(256), 0, st, v, ctx->Mp, g.n, g.ch.phase, ctx->i8_nks, g.nCp, g.Vs,
                         g.vbad, ctx->D, ctx->DP, ctx->dd.inv_alpha, g.ch.Gq, g.vexp, VDelta{g.vexp_d, g.rebase, g.dmax, (int)ctx->opt.i8_force_rebase});
      return;
    }
    hipLaunchKernelGGL((k_vsplit<S>), dim3((unsigned)((g.n + 7) / 8)), dim3(256), 0, st, v, ctx->Mp, g.n, g.ch.phase, ctx->i8_nks, g.nCp, g.Vs, g.vbad,
                       ctx->D, ctx->DP, ctx->dd.inv_alpha, g.ch.Gq, g.vexp, VDelta{});
    return;
  }
  const int nCB = g.nCp / I8_BM, nPB = ctx->pairs.NPp / (32 * TN * WN);
  constexpr int lds = i8_lds_bytes<S, WN, TN>();
  const unsigned nblk = (unsigned)(nCB < 8 ? nCB * nPB : (nCB + 7) / 8 * 8 * nPB);  // (fewer than 8 chain blocks: tiles are dealt round)
  if (g.ksplit_a > 1) {  // small batch: too few tiles to fill the chip, so the k range is cut into planes that are summed afterwards
    const size_t plane = (size_t)g.n * ctx->DP * ctx->DP;
    hipLaunchKernelGGL((k_assemble_i8<S, WN, TN>), dim3(nblk, (unsigned)g.ksplit_a), dim3(128 * WN), lds, st, g.Vs, ctx->d_Zs, g.nCp, ctx->i8_nks,
                       0, ctx->i8_nks, 0, ctx->pairs, g.n, g.ch.phase, g.vbad, ctx->DP, ctx->dd.inv_alpha, g.Gpart, plane, g.vexp, nPB,
                       I8Delta{nullptr, nullptr, 1.0, 0, 0, nullptr});
    hipLaunchKernelGGL(k_sum_planes, dim3((unsigned)((plane + 255) / 256)), dim3(256), 0, st, g.ch.Gq, g.Gpart, g.ksplit_a, plane, plane);
    return;
  }
  // Ragged last pair block (D = 64: 2080 pairs = 16 blocks of 128 + 32): once the full blocks alone fill the chip, the rest goes to
  // k_assemble_i8_tail (bit-identical results, see there).  option i8_tail = 0 / 1: never / whenever there is a ragged block.
  const int nPBfull = ctx->pairs.NP / (32 * TN * WN);
  const bool tail = WN == 4 && g.Tq && nPBfull < nPB && (ctx->opt.i8_tail == 1 || (ctx->opt.i8_tail < 0 && (long long)nCB * nPBfull >= 256));
  const int npb = tail ? nPBfull : nPB;
  const unsigned nblk_main = (unsigned)(nCB < 8 ? nCB * npb : (nCB + 7) / 8 * 8 * npb);
  const int pb32_0 = nPBfull * TN * WN, ntail = (ctx->pairs.NP - pb32_0 * 32 + 31) / 32;
  // the k range in pieces whose int32 sums cannot overflow whatever the data (one piece up to M = 21845 at 6 slices)
  for (int ks0 = 0; ks0 < ctx->i8_nks; ks0 += ctx->i8_chunk) {
    const int nk = std::min(ctx->i8_chunk, ctx->i8_nks - ks0);
    if (nblk_main)
      hipLaunchKernelGGL((k_assemble_i8<S, WN, TN>), dim3(nblk_main), dim3(128 * WN), lds, st, Vs, ctx->d_Zs, g.nCp,
                         ctx->i8_nks, ks0, nk, (ks0 > 0 ? 1 : 0) | acc0, ctx->pairs, g.n, g.ch.phase, g.vbad, ctx->DP, ctx->dd.inv_alpha, g.ch.Gq, (size_t)0, vexp, npb, dl);
    if (tail) {
      const int pieces = std::max(1, std::min({g.tail_pieces, nk / 8, (int)(256 / std::max(1, nCB * ntail))}));
      hipLaunchKernelGGL((k_assemble_i8_tail<S>), dim3((unsigned)(nCB * ntail), (unsigned)pieces), dim3(128), (i8_lds_bytes<S, 1, 1>()), st, Vs, ctx->d_Zs,
                         g.nCp, ctx->pairs.NPp, ctx->pairs.NP, ctx->i8_nks, ks0, nk, g.n, pb32_0, ntail, g.Tq, dl);
      hipLaunchKernelGGL((k_assemble_i8_tailsum<S>), dim3((unsigned)(((size_t)g.n * 32 * ntail + 255) / 256)), dim3(256), 0, st, g.Tq, pieces, g.nCp, ntail,
                         pb32_0, (ks0 > 0 ? 1 : 0) | acc0, ctx->pairs, g.n, g.ch.phase, g.vbad, ctx->DP, ctx->dd.inv_alpha, g.ch.Gq, vexp, dl);
    }
  }
}
// Delta assembly (use_delta: S = 6, the 8-wave tile, no k-split planes): one launch each of the main tiles, the ragged pair block's tiles and their
// sum; the kernels pick the slice count the difference needs from *dmax themselves (k_assemble_i8_sel).  seff 6: full accuracy; 5: inner iterate
void launch_assemble_i8_delta(rmhmc_ctx* ctx, Group& g, hipStream_t st, int seff) {
  constexpr int WN = 4, TN = 1;
  const size_t vplane = (size_t)ctx->i8_nks * g.nCp * 32;
  const I8Delta dl{g.dmax, g.rebase, 1.0, 0, 0, ctx->big ? g.Gbase : nullptr};
  const int nCB = g.nCp / I8_BM, nPB = ctx->pairs.NPp / (32 * TN * WN);
  constexpr int lds = i8_lds_bytes<6, WN, TN>() > i8_lds_bytes<4, WN, TN>() ? i8_lds_bytes<6, WN, TN>() : i8_lds_bytes<4, WN, TN>();
  static_assert(lds >= (i8_lds_bytes<5, WN, TN>()), "dynamic LDS of the widest instantiation");
  const int nPBfull = ctx->pairs.NP / (32 * TN * WN);
  const bool tail = g.Tq && nPBfull < nPB && (ctx->opt.i8_tail == 1 || (ctx->opt.i8_tail < 0 && (long long)nCB * nPBfull >= 256));
  const int npb = tail ? nPBfull : nPB;
  const unsigned nblk_main = (unsigned)(nCB < 8 ? nCB * npb : (nCB + 7) / 8 * 8 * npb);
  const int pb32_0 = nPBfull * TN * WN, ntail = (ctx->pairs.NP - pb32_0 * 32 + 31) / 32;
  constexpr int lds_t = i8_lds_bytes<6, 1, 1>() > i8_lds_bytes<4, 1, 1>() ? i8_lds_bytes<6, 1, 1>() : i8_lds_bytes<4, 1, 1>();
  static_assert(lds_t >= (i8_lds_bytes<5, 1, 1>()), "dynamic LDS of the widest tail instantiation");
  for (int ks0 = 0; ks0 < ctx->i8_nks; ks0 += ctx->i8_chunk) {
    const int nk = std::min(ctx->i8_chunk, ctx->i8_nks - ks0);
    if (nblk_main)
      hipLaunchKernelGGL((k_assemble_i8_sel<WN, TN>), dim3(nblk_main), dim3(128 * WN), lds, st, g.Vs, vplane, seff, ctx->d_Zs, g.nCp, ctx->i8_nks, ks0, nk,
                         (ks0 > 0 ? 1 : 0) | 2, ctx->pairs, g.n, g.ch.phase, g.vbad, ctx->DP, ctx->dd.inv_alpha, g.ch.Gq, (size_t)0, g.vexp_d, npb, dl);
    if (tail) {
      const int pieces = std::max(1, std::min({g.tail_pieces, nk / 8, (int)(256 / std::max(1, nCB * ntail))}));
      hipLaunchKernelGGL(k_assemble_i8_tail_sel, dim3((unsigned)(nCB * ntail), (unsigned)pieces), dim3(128), lds_t, st, g.Vs, vplane, seff, ctx->d_Zs,
                         g.nCp, ctx->pairs.NPp, ctx->pairs.NP, ctx->i8_nks, ks0, nk, g.n, pb32_0, ntail, g.Tq, dl);
      hipLaunchKernelGGL(k_assemble_i8_tailsum_sel, dim3((unsigned)(((size_t)g.n * 32 * ntail + 255) / 256)), dim3(256), 0, st, seff, g.Tq, pieces, g.nCp, ntail,
                         pb32_0, (ks0 > 0 ? 1 : 0) | 2, ctx->pairs, g.n, g.ch.phase, g.vbad, ctx->DP, ctx->dd.inv_alpha, g.ch.Gq, g.vexp_d, dl);
    }
  }
}
// leverage pass of the int8 path: part 0 cuts G^-1 into slices, part 1 is the GEMM (R = c .* h into rv0, v is dead by then)
template <int S, int WN, int TN>
void launch_leverage_i8_t(rmhmc_ctx* ctx, Group& g, hipStream_t st, int part) {
  if (part == 0) {
    hipLaunchKernelGGL((k_qsplit<S>), dim3((unsigned)g.n), dim3(256), 0, st, g.ch.trj.Ginv, ctx->DP, ctx->pairs, g.ch.phase, ctx->i8_nkp, g.nCp,
                       g.Qs, g.qscale);
    return;
  }
  const int nCB = g.nCp / I8_BM, nRB = ctx->i8_NRp / (32 * TN * WN);
  constexpr int lds = i8_lds_bytes<S, WN, TN>();
  const unsigned nblk = (unsigned)(nCB < 8 ? nCB * nRB : (nCB + 7) / 8 * 8 * nRB);
  if (g.ksplit_l > 1) {
    const size_t plane = (size_t)g.n * ctx->Mp;
    hipLaunchKernelGGL((k_leverage_i8<S, WN, TN>), dim3(nblk, (unsigned)g.ksplit_l), dim3(128 * WN), lds, st, g.Qs, ctx->d_Zt, g.nCp, ctx->i8_NRp,
                       ctx->i8_nkp, 0, ctx->i8_nkp, 0, (ctx->big || g.ctile) ? 0 : 1, g.n, ctx->Mp, g.ch.phase, g.qscale, ctx->d_zscale, g.ch.rv2, g.Rpart, plane);
    hipLaunchKernelGGL(k_sum_planes, dim3((unsigned)((plane + 255) / 256)), dim3(256), 0, st, g.ch.rv0, g.Rpart, g.ksplit_l, plane, plane);
    return;
  }
  for (int kp0 = 0; kp0 < ctx->i8_nkp; kp0 += ctx->i8_chunk) {
    const int nk = std::min(ctx->i8_chunk, ctx->i8_nkp - kp0);
    hipLaunchKernelGGL((k_leverage_i8<S, WN, TN>), dim3(nblk), dim3(128 * WN), lds, st, g.Qs, ctx->d_Zt, g.nCp,
                       ctx->i8_NRp, ctx->i8_nkp, kp0, nk, kp0 > 0 ? 1 : 0, (ctx->big || g.ctile) ? 0 : 1, g.n, ctx->Mp, g.ch.phase, g.qscale, ctx->d_zscale,
                       g.ch.rv2, g.ch.rv0, (size_t)0);
  }
}
// inner: an assembly whose G only steers a fixed-point iterate (the position iterations before the last, rmhmc.py:116-122).  With
// ctx->i8_inner_drop it is summed from the S-1 most significant slices of the same operands (balanced digits: dropping the last digit
// IS rounding to the coarser grid), 15 slice products instead of 21.
// keep_base (large-D path): this G is the base of a delta assembly to come; Gq will be factored in place, so a copy is kept
void launch_assemble(rmhmc_ctx* ctx, Group& g, const double* v, bool inner = false, bool delta = false, bool keep_base = false) {
  if (ctx->i8 && delta && ctx->big)
    launch(ctx, g, HEAVY, "vsplit", [&](hipStream_t st) { launch_assemble_i8_t<6, 4, 1>(ctx, g, v, st, 0, true); });
  if (ctx->i8 && delta && inner) {  // (use_delta_inner: five-slice accuracy)
    launch(ctx, g, HEAVY, "assemble_i8_inner_delta", [&](hipStream_t st) { launch_assemble_i8_delta(ctx, g, st, 5); });
    return;
  }
  if (ctx->i8 && delta) {  // (use_delta: S = 6, WN = 4)
    launch(ctx, g, HEAVY, "assemble_i8_delta", [&](hipStream_t st) { launch_assemble_i8_delta(ctx, g, st, 6); });
    return;
  }
  if (ctx->i8) {
    if (ctx->big)  // (the generic row pass already wrote the slices)
      launch(ctx, g, HEAVY, "vsplit", [&](hipStream_t st) { I8_SWITCH(ctx, (launch_assemble_i8_t<S_, WN_, TN_>(ctx, g, v, st, 0))); });
    const int suse = (inner && ctx->i8_inner_drop && ctx->i8S == 6) ? 5 : ctx->i8S;  // (5 -> 4 costs parity: 2e-9 on theta at M = 97)
    launch(ctx, g, HEAVY, suse == ctx->i8S ? "assemble_i8" : "assemble_i8_inner", [&](hipStream_t st) {
      I8_SWITCH_S(suse, (launch_assemble_i8_t<S_, WN_, TN_>(ctx, g, v, st, 1)));
      if (keep_base && ctx->big && g.Gbase)
        (void)hipMemcpyAsync(g.Gbase, g.ch.Gq, sizeof(double) * (size_t)g.n * ctx->DP * ctx->DP, hipMemcpyDeviceToDevice, st);
    });
    return;
  }
  launch(ctx, g, HEAVY, "assemble", [&](hipStream_t st) {
    if (ctx->flags & RMHMC_FLAG_FP32_METRIC) {  // precision experiment: fp32 matrix cores for the metric only
      if (ctx->big) {
        hipLaunchKernelGGL((k_assemble_f32<4>), dim3((unsigned)((g.n + 3) / 4), ctx->npairs), dim3(256), 0, st, ctx->dd, g.n, g.ch.phase, v,
                           g.ch.Gq, ctx->nbk);
      } else {
        NB_SWITCH(ctx, hipLaunchKernelGGL((k_assemble_f32<NB_>), dim3((unsigned)((g.n + 3) / 4)), dim3(256), 0, st, ctx->dd, g.n,
                                          g.ch.phase, v, g.ch.Gq, 1));
      }
      return;
    }
    if (ctx->big) {
      hipLaunchKernelGGL(k_assemble_pair<false>, dim3((unsigned)((g.n + 3) / 4), ctx->npairs - ctx->nbk), dim3(256), 0, st, ctx->dd, g.n,
                         g.ch.phase, v, g.ch.Gq);
      hipLaunchKernelGGL(k_assemble_pair<true>, dim3((unsigned)((g.n + 3) / 4), ctx->nbk), dim3(256), 0, st, ctx->dd, g.n, g.ch.phase, v,
                         g.ch.Gq);
      return;
    }
    if (g.fsplit > 1) {  // small batch: row ranges into planes, summed in a fixed order
      const size_t plane = (size_t)g.n * ctx->DP * ctx->DP;
      dim3 grid((unsigned)((g.n + 3) / 4), (unsigned)g.fsplit);
      NB_SWITCH(ctx, hipLaunchKernelGGL((k_assemble<NB_>), grid, dim3(256), 0, st, ctx->dd, g.n, g.ch.phase, v, g.Gpart, plane));
      hipLaunchKernelGGL(k_sum_planes, dim3((unsigned)((plane + 255) / 256)), dim3(256), 0, st, g.ch.Gq, g.Gpart, g.fsplit, plane, plane);
      return;
    }
    dim3 grid((unsigned)((g.n + 3) / 4));
    NB_SWITCH(ctx, hipLaunchKernelGGL((k_assemble<NB_>), grid, dim3(256), 0, st, ctx->dd, g.n, g.ch.phase, v, g.ch.Gq, (size_t)0));
  });
}

// q partials of u' dG/dw_d u for every chain (u = ch.uq, w as given); summed by k_mom_update / k_mom_final.
// cmode (generic path): 1 = first pass at this w, c is computed and kept; 2 = c of this w is at hand (k_mompass in kernels.hip.h)
void launch_mompass(rmhmc_ctx* ctx, Group& g, const double* w, int cmode) {
  launch(ctx, g, HEAVY, "mompass", [&](hipStream_t st) {
    if (!ctx->opt.ccache) cmode = 0;
    if (ctx->big) {
      dim3 grid((unsigned)((g.n + 15) / 16), g.nsplit);
      switch (cmode) {
        case 1: hipLaunchKernelGGL(k_mompass_big<1>, grid, dim3(256), 0, st, ctx->dd, g.n, g.nsplit, ctx->nbk, w, g.ch.uq, g.ch.qpart, g.ctile); break;
        case 2: hipLaunchKernelGGL(k_mompass_big<2>, grid, dim3(256), 0, st, ctx->dd, g.n, g.nsplit, ctx->nbk, w, g.ch.uq, g.ch.qpart, g.ctile); break;
        default: hipLaunchKernelGGL(k_mompass_big<0>, grid, dim3(256), 0, st, ctx->dd, g.n, g.nsplit, ctx->nbk, w, g.ch.uq, g.ch.qpart, g.ctile); break;
      }
      return;
    }
    dim3 grid((unsigned)((g.n + 63) / 64), g.nsplit);
    switch (cmode) {
      case 1:  // first pass of a step at this w: the tiles are at hand unless the chain has just rejected a proposal (k_mompass<.., 3>)
        if (ctx->opt.cdyn) { NB_SWITCH(ctx, hipLaunchKernelGGL((k_mompass<NB_, 3>), grid, dim3(256), 0, st, ctx->dd, g.n, g.nsplit, w, g.ch.uq, g.ch.qpart, g.ctile, g.ch.cstale)); }
        else { NB_SWITCH(ctx, hipLaunchKernelGGL((k_mompass<NB_, 1>), grid, dim3(256), 0, st, ctx->dd, g.n, g.nsplit, w, g.ch.uq, g.ch.qpart, g.ctile)); }
        break;
      case 2: NB_SWITCH(ctx, hipLaunchKernelGGL((k_mompass<NB_, 2>), grid, dim3(256), 0, st, ctx->dd, g.n, g.nsplit, w, g.ch.uq, g.ch.qpart, g.ctile)); break;
      default: NB_SWITCH(ctx, hipLaunchKernelGGL((k_mompass<NB_, 0>), grid, dim3(256), 0, st, ctx->dd, g.n, g.nsplit, w, g.ch.uq, g.ch.qpart, g.ctile)); break;
    }
  });
}

// The evaluation's momentum pass and the trace product are both X' R over the same rows and c tiles: on the generic int8 path with c tiles the
// leverage GEMM runs first and k_mompass_trv does both products in one pass (fused_trace); otherwise k_trvec follows the GEMM.
static bool fused_trace(const rmhmc_ctx* ctx, const Group& g) { return ctx->i8 && !ctx->big && g.ctile != nullptr && ctx->opt.ccache; }

// part: 0 everything; 1 the leverage GEMM alone (h -> rv0); 2 the reduction of the trace partials alone
void launch_leverage(rmhmc_ctx* ctx, Group& g, int part = 0) {
  if (ctx->i8 && part == 2) {
    launch(ctx, g, LIGHT, "small", [&](hipStream_t st) {
      hipLaunchKernelGGL(k_reduce_tr, dim3((unsigned)g.n), dim3(64), 0, st, ctx->D, ctx->DP, g.ch, g.ch.gpart, g.nsplit);
    });
    return;
  }
  if (ctx->i8) {  // h_n as the transposed sliced GEMM, then tr = X' (c .* h) on the fp64 matrix cores
    launch(ctx, g, HEAVY, "qsplit", [&](hipStream_t st) { I8_SWITCH(ctx, (launch_leverage_i8_t<S_, WN_, TN_>(ctx, g, st, 0))); });
    // The leverages h_n = x_n' G^-1 x_n enter the trace term only, which steers the momentum and appears in no Hamiltonian: like the metric
    // of an inner position iterate they are summed from the S - 1 most significant slices of the same operands (15 slice products
    // instead of 21; h to ~3e-12 norm-wise, theta / p after a step move by < 1e-11; RMHMC_FLAG_INT8_INNER_FULL: all S)
    const int suse = (ctx->i8_inner_drop && ctx->i8S == 6) ? 5 : ctx->i8S;
    launch(ctx, g, HEAVY, "leverage_i8", [&](hipStream_t st) { I8_SWITCH_S(suse, (launch_leverage_i8_t<S_, WN_, TN_>(ctx, g, st, 1))); });
    if (part == 1) return;
    launch(ctx, g, HEAVY, "trvec", [&](hipStream_t st) {
      if (ctx->big) {  // rv0 holds h (one "pair" plane), the large-D trace kernel multiplies by c itself
        dim3 grid((unsigned)((g.n + 15) / 16), g.nsplit);
        hipLaunchKernelGGL(k_trace_big, grid, dim3(256), 0, st, ctx->dd, g.n, g.nsplit, ctx->nbk, 1, g.ch.rv2, g.ch.rv0, g.ch.gpart);
        return;
      }
      dim3 grid((unsigned)((g.n + 63) / 64), g.nsplit);
      NB_SWITCH(ctx, hipLaunchKernelGGL((k_trvec<NB_>), grid, dim3(256), 0, st, ctx->dd, g.n, g.nsplit, g.ch.rv0, g.ch.gpart, (const d4*)g.ctile));
    });
    launch(ctx, g, LIGHT, "small", [&](hipStream_t st) {
      hipLaunchKernelGGL(k_reduce_tr, dim3((unsigned)g.n), dim3(64), 0, st, ctx->D, ctx->DP, g.ch, g.ch.gpart, g.nsplit);
    });
    return;
  }
  if (ctx->big) {  // per block pair leverage contributions, then the trace GEMM over 16 chains per workgroup
    double* hpart = ctx->d_hpart + (size_t)g.off * ctx->Mp;  // [pair][n][Mp] of this group (single group: off = 0)
    launch(ctx, g, HEAVY, "leverage", [&](hipStream_t st) {
      hipLaunchKernelGGL(k_leverage_pair<false>, dim3((unsigned)((g.n + 3) / 4), ctx->npairs - ctx->nbk), dim3(256), 0, st, ctx->dd, g.n,
                         ctx->nbk, g.ch.phase, g.ch.trj.Ginv, hpart);
      hipLaunchKernelGGL(k_leverage_pair<true>, dim3((unsigned)((g.n + 3) / 4), ctx->nbk), dim3(256), 0, st, ctx->dd, g.n, ctx->nbk,
                         g.ch.phase, g.ch.trj.Ginv, hpart);
    });
    launch(ctx, g, HEAVY, "leverage", [&](hipStream_t st) {
      dim3 grid((unsigned)((g.n + 15) / 16), g.nsplit);
      hipLaunchKernelGGL(k_trace_big, grid, dim3(256), 0, st, ctx->dd, g.n, g.nsplit, ctx->nbk, ctx->npairs, g.ch.rv2, hpart, g.ch.gpart);
    });
    launch(ctx, g, LIGHT, "small", [&](hipStream_t st) {
      hipLaunchKernelGGL(k_reduce_tr, dim3((unsigned)g.n), dim3(64), 0, st, ctx->D, ctx->DP, g.ch, g.ch.gpart, g.nsplit);
    });
    return;
  }
  const int fs = std::min(g.fsplit, g.nsplit);  // (the partials go to gpart, which holds nsplit planes)
  launch(ctx, g, HEAVY, "leverage", [&](hipStream_t st) {
    dim3 grid((unsigned)((g.n + 3) / 4), (unsigned)fs);
    NB_SWITCH(ctx, hipLaunchKernelGGL((k_leverage<NB_>), grid, dim3(256), 0, st, ctx->dd, g.n, g.ch.phase, g.ch.trj.Ginv, g.ch.rv2,
                                      g.ch.trj.tr, g.ch.gpart));
  });
  if (fs > 1)
    launch(ctx, g, LIGHT, "small", [&](hipStream_t st) {
      hipLaunchKernelGGL(k_reduce_tr, dim3((unsigned)g.n), dim3(64), 0, st, ctx->D, ctx->DP, g.ch, g.ch.gpart, fs);
    });
}

// one wavefront (64-thread block) per chain
#define SMALL(ctx, g, name, kern, ...)                                                                     \
  launch(ctx, g, LIGHT, name, [&](hipStream_t st_) { hipLaunchKernelGGL(kern, dim3((unsigned)(g).n), dim3(64), 0, st_, __VA_ARGS__); })

// one 256-thread workgroup per chain (blocked dense algebra of the large-D path)
#define BIG(ctx, g, name, kern, ...)                                                                       \
  launch(ctx, g, LIGHT, name, [&](hipStream_t st_) { hipLaunchKernelGGL(kern, dim3((unsigned)(g).n), dim3(256), 0, st_, __VA_ARGS__); })

// A phase is one launch per group; phases are issued group-alternating so that, with two groups, the main
// stream sees heavy(A), heavy(B), heavy(A), ... and the light kernels of a group overlap the other's heavy one.
using Phase = std::function<void(Group&)>;
void run_phases(rmhmc_ctx* ctx, const std::vector<Phase>& phases) {
  for (const Phase& ph : phases)
    for (Group& g : ctx->groups) ph(g);
}

// Evaluate the point record at trj.w for every chain in phase 1 (rmhmc.py:134-161; with advance the
// explicit momentum half step :163 too): v, r, c, log-joint partials -> G and gradient on the matrix cores
// -> factor / inverse / u = G^-1 p -> quadratic term -> leverage pass (trace term) -> momentum update.
// mode 0: everything; 1: metric, factor, inverse, gradient, log joint only (simplified mMALA); 2: mode 1 + the trace term (full mMALA)
void eval_point_phases(rmhmc_ctx* ctx, std::vector<Phase>& ph, bool advance, int mode = 0) {
  // (advance: the evaluation that ends a leapfrog step - the last position iterate's slices and G are at hand)
  ph.push_back([ctx, advance](Group& g) { launch_rowpass<RP_F>(ctx, g, g.ch.trj.w, g.ch.rv0, g.ch.rv2, advance && use_delta(ctx, g)); });
  if (ctx->big) ph.push_back([ctx](Group& g) { SMALL(ctx, g, "small", k_finish_big, ctx->dd, g.ch, g.nsplit); });
  ph.push_back([ctx, advance](Group& g) { launch_assemble(ctx, g, g.ch.rv0, false, advance && use_delta(ctx, g)); });
  if (ctx->big) {
    ph.push_back([ctx](Group& g) {
      if (ctx->want_G)
        (void)hipMemcpyAsync(ctx->d_Gcopy + (size_t)g.off * ctx->DP * ctx->DP, g.ch.Gq, sizeof(double) * (size_t)g.n * ctx->DP * ctx->DP,
                             hipMemcpyDeviceToDevice, ctx->stream);
      BIG(ctx, g, "factor", k_chol_big<1>, ctx->dd, g.ch, ctx->nbk, ctx->d_Wd + (size_t)g.off * ctx->nbk * 4096, ctx->eps);
    });
    ph.push_back([ctx](Group& g) { BIG(ctx, g, "factor", k_inverse_big, ctx->dd, g.ch, ctx->nbk, ctx->d_Wd + (size_t)g.off * ctx->nbk * 4096); });
    ph.push_back([ctx](Group& g) { SMALL(ctx, g, "small", k_ginv_matvec, ctx->D, ctx->DP, g.ch, g.ch.p); });
  } else {
    ph.push_back([ctx](Group& g) {
      launch(ctx, g, LIGHT, "factor", [&](hipStream_t st) {
        NB_SWITCH(ctx, hipLaunchKernelGGL((k_factor_full<NB_>), dim3((unsigned)g.n), dim3(64), 0, st, ctx->dd, g.ch, g.nsplit));
      });
    });
  }
  if (mode == 1) return;  // simplified mMALA needs neither the quadratic nor the trace term
  if (mode == 2) {        // full mMALA: the trace term enters the drift, the quadratic term does not exist
    ph.push_back([ctx](Group& g) { launch_leverage(ctx, g); });
    return;
  }
  ph.push_back([ctx](Group& g) {
    if (fused_trace(ctx, g)) {  // leverage GEMM first, then ONE pass for the quadratic term and the trace term
      launch_leverage(ctx, g, 1);
      launch(ctx, g, HEAVY, "mompass", [&](hipStream_t st) {
        dim3 grid((unsigned)((g.n + 63) / 64), g.nsplit);
        NB_SWITCH(ctx, hipLaunchKernelGGL((k_mompass_trv<NB_>), grid, dim3(256), 0, st, ctx->dd, g.n, g.nsplit, g.ch.uq, g.ch.qpart, g.ctile, g.ch.rv0, g.ch.gpart));
      });  // (k_mom_final below sums the trace partials itself)
    } else {
      launch_mompass(ctx, g, g.ch.trj.w, 2);  // (the row pass above has just stored c for trj.w)
      launch_leverage(ctx, g);
    }
  });
  ph.push_back([ctx, advance](Group& g) {
    SMALL(ctx, g, "small", k_mom_final, ctx->D, ctx->DP, g.ch, ctx->eps, advance ? 1 : 0, g.nsplit, fused_trace(ctx, g) ? g.ch.gpart : (const double*)nullptr);
  });
}

// one-launch step / evaluation / folded global step for small batches (medium_step.hip.h).  Data rows per thread stay in registers
// when at most 64 doubles are needed for them.
template <int NB>
void launch_step_medium_nb(rmhmc_ctx* ctx, Group& g, hipStream_t st, int guards, int eval_only, int fold, const IterParams& ip) {
  const int rpt = (ctx->Mp + 255) / 256;
  const dim3 grid((unsigned)g.n), block(256);
  const size_t lds = ctx->medium_lds;
  switch (rpt * 16 * NB <= 64 ? rpt : 0) {
    case 1: hipLaunchKernelGGL((k_step_medium<NB, 1>), grid, block, lds, st, ctx->dd, g.ch, ctx->eps, ctx->K, guards, eval_only, fold, ip); break;
    case 2: hipLaunchKernelGGL((k_step_medium<NB, 2>), grid, block, lds, st, ctx->dd, g.ch, ctx->eps, ctx->K, guards, eval_only, fold, ip); break;
    case 3: hipLaunchKernelGGL((k_step_medium<NB, (NB == 1 ? 3 : 0)>), grid, block, lds, st, ctx->dd, g.ch, ctx->eps, ctx->K, guards, eval_only, fold, ip); break;
    case 4: hipLaunchKernelGGL((k_step_medium<NB, (NB == 1 ? 4 : 0)>), grid, block, lds, st, ctx->dd, g.ch, ctx->eps, ctx->K, guards, eval_only, fold, ip); break;
    default: hipLaunchKernelGGL((k_step_medium<NB, 0>), grid, block, lds, st, ctx->dd, g.ch, ctx->eps, ctx->K, guards, eval_only, fold, ip); break;
  }
}
void launch_step_medium(rmhmc_ctx* ctx, Group& g, hipStream_t st, int guards, int eval_only, int fold, const IterParams& ip) {
  if (ctx->NB == 1) launch_step_medium_nb<1>(ctx, g, st, guards, eval_only, fold, ip);
  else launch_step_medium_nb<2>(ctx, g, st, guards, eval_only, fold, ip);
}

// One generalised leapfrog step for every chain in phase 1 (rmhmc.py:96-163).
void step_phases(rmhmc_ctx* ctx, std::vector<Phase>& ph) {
  const int D = ctx->D, DP = ctx->DP, K = ctx->K;
  const double eps = ctx->eps;
  if (ctx->medium) {  // the whole step in one launch, one workgroup per chain
    const int guards = (ctx->flags & RMHMC_FLAG_GUARDS) ? 1 : 0;
    ph.push_back([=](Group& g) {
      launch(ctx, g, HEAVY, "medium", [&](hipStream_t st) {
        launch_step_medium(ctx, g, st, guards, 0, 0, IterParams{});
      });
    });
    return;
  }
  // c tiles of the chains whose last proposal was rejected (their trj has fallen back to cur): recomputed for them alone, so that the
  // first momentum pass finds every chain's tiles at hand
  if (!ctx->big && ctx->opt.cdyn)
    ph.push_back([=](Group& g) {
      if (!g.ctile || !g.ch.stale_list) return;
      launch(ctx, g, HEAVY, "mompass", [&](hipStream_t st) {
        // row pieces of this kernel's own, and a SMALL grid whose wavefronts walk the list (8 x 64 chains at a time): a wavefront alone on its
        // SIMD takes ~4.3 us per 32-row block (load - product - exp latencies with nothing to hide them), and workgroups that only read the
        // count and return are not free either.  Measured at config 3 (~300 listed chains = 19 wavefront groups per step, one box,
        // tools/crs_sweep.sh): grid 32 x 16 (round 2: room for 2048 chains, the rest left to k_mompass<.., 3>) 75.6 us, 8 x 16 43, 8 x 32 29.3,
        // 8 x 64 31.9, 4 x 64 28.6, 2 x 128 34.1.
        const int rsplit = std::max(1, std::min(32, ctx->Mp / 32 / 4));
        dim3 grid((unsigned)std::min((g.n + 63) / 64, 8), (unsigned)rsplit);
        NB_SWITCH(ctx, hipLaunchKernelGGL((k_crestore<NB_>), grid, dim3(256), 0, st, ctx->dd, g.n, g.ch.phase, g.ch.trj.w, g.ctile, g.ch.cstale,
                                          g.ch.stale_list, g.ch.stale_count));
        // (the count is reset by k_pos_first, later in the step)
      });
    });
  // implicit momentum half step: K fixed-point iterations (rmhmc.py:102-110).  D <= 64: the update that ends an iteration and the G^-1 PM
  // product that starts the next are one launch (k_mom_update_matvec), the last update is done by k_pos_first
  const bool fuse = !ctx->big;
  for (int it = 0; it < K; ++it) {
    if (it == 0 || !fuse) ph.push_back([=](Group& g) { SMALL(ctx, g, "small", k_ginv_matvec, D, DP, g.ch, it == 0 ? g.ch.p : g.ch.PM); });
    ph.push_back([=](Group& g) { launch_mompass(ctx, g, g.ch.trj.w, it == 0 ? 1 : 2); });
    if (!fuse) ph.push_back([=](Group& g) { SMALL(ctx, g, "small", k_mom_update, D, DP, g.ch, eps, it == K - 1 ? 1 : 0, g.nsplit); });
    else if (it < K - 1) ph.push_back([=](Group& g) { SMALL(ctx, g, "small", k_mom_update_matvec, D, DP, g.ch, eps, g.nsplit); });
  }
  // implicit position step: K fixed-point iterations (rmhmc.py:113-123); the first one re-uses the
  // stored factor of G(w)
  if (ctx->big) {
    ph.push_back([=](Group& g) { SMALL(ctx, g, "small", k_ginv_matvec, D, DP, g.ch, g.ch.p); });
    ph.push_back([=](Group& g) { SMALL(ctx, g, "small", k_pos_first_big, D, DP, g.ch, eps); });
  } else {
    ph.push_back([=](Group& g) { SMALL(ctx, g, "factor", k_pos_first, D, DP, g.ch, eps, g.nsplit); });
  }
  const int guards = (ctx->flags & RMHMC_FLAG_GUARDS) ? 1 : 0;
  for (int it = 1; it < K; ++it) {
    ph.push_back([=](Group& g) { launch_rowpass<RP_V>(ctx, g, g.ch.wq, g.ch.rv0, nullptr, use_delta_inner(ctx, g, it)); });
    ph.push_back([=](Group& g) {
      const bool base = (it == 1 && use_delta_inner(ctx, g, 2)) || (it == K - 1 && use_delta(ctx, g));
      launch_assemble(ctx, g, g.ch.rv0, it < K - 1, use_delta_inner(ctx, g, it), base);
    });
    if (ctx->big)
      ph.push_back([=](Group& g) { BIG(ctx, g, "factor", k_chol_big<0>, ctx->dd, g.ch, ctx->nbk, ctx->d_Wd + (size_t)g.off * ctx->nbk * 4096, eps); });
    else
      ph.push_back([=](Group& g) {
        launch(ctx, g, LIGHT, "factor", [&](hipStream_t st) {  // (the last iterate: accepted as the new w, position guard, in the same launch)
          NB_SWITCH(ctx, hipLaunchKernelGGL((k_factor_solve<NB_>), dim3((unsigned)g.n), dim3(64), 0, st, D, DP, g.ch, eps, it == K - 1 ? guards : -1));
        });
      });
  }
  if (ctx->big || K < 2) ph.push_back([=](Group& g) { SMALL(ctx, g, "small", k_pos_final, D, DP, g.ch, guards); });
  // explicit momentum half step at the new point (rmhmc.py:134-163)
  eval_point_phases(ctx, ph, true);
}

void launch_eval_point(rmhmc_ctx* ctx) {
  std::vector<Phase> ph;
  eval_point_phases(ctx, ph, false);
  run_phases(ctx, ph);
}
void launch_step(rmhmc_ctx* ctx) {
  std::vector<Phase> ph;
  step_phases(ctx, ph);
  run_phases(ctx, ph);
}

struct IterBase {
  long long limit, burn_in, S;
  double* samples;
  bool explicit_rng;
  bool count_done;
};

IterParams iter_params(rmhmc_ctx* ctx, const Group& g, const IterBase& b) {
  IterParams ip{};
  ip.flags = ctx->flags;
  ip.L = ctx->L;
  ip.seed = ctx->seed;
  ip.chain_offset = ctx->chain_offset + g.off;
  ip.iter_limit = b.limit;
  ip.burn_in = b.burn_in;
  ip.S = b.S;
  ip.samples = b.samples ? b.samples + (size_t)g.off * b.S * ctx->D : nullptr;
  if (ctx->sorted) {  // (single group: g.off = 0) chain ids and sample blocks through the position -> chain map
    ip.orig = ctx->d_orig;
    ip.chain_offset = ctx->chain_offset;
    ip.samples = b.samples;
  }
  if (b.explicit_rng) {
    ip.z_in = ctx->d_z + (size_t)g.off * ctx->D; ip.ulen_in = ctx->d_ulen + g.off; ip.gdir_in = ctx->d_gdir + g.off; ip.uacc_in = ctx->d_uacc + g.off;
  }
  ip.done_count = b.count_done ? ctx->d_done : nullptr;
  ip.lower_L = (!ctx->big && !ctx->medium && !ctx->fused && ctx->sampler == 0) ? 1 : 0;  // (k_factor_full is the only writer of trj.L there)
  return ip;
}

void launch_iter_begin(rmhmc_ctx* ctx, const IterBase& b) {
  for (Group& g : ctx->groups) { IterParams ip = iter_params(ctx, g, b); SMALL(ctx, g, "small", k_iter_begin, ctx->D, ctx->DP, g.ch, ip); }
}
void launch_iter_end(rmhmc_ctx* ctx, const IterBase& b) {
  for (Group& g : ctx->groups) { IterParams ip = iter_params(ctx, g, b); SMALL(ctx, g, "small", k_iter_end, ctx->D, ctx->DP, g.ch, ip); }
}

// plain HMC (hmc.py:38-84): begin / half step + position / gradient pass / half step / end
template <int NB>
void launch_hmc_traj_nb(rmhmc_ctx* ctx, Group& g, int eval_only, hipStream_t st) {
  const int rpt = (ctx->Mp + 255) / 256;  // data rows per thread; up to 4 of them stay in registers for the whole trajectory
  const dim3 grid((unsigned)g.n), block(256);
  switch (rpt <= 4 ? rpt : 0) {
    case 1: hipLaunchKernelGGL((k_hmc_traj<NB, 1>), grid, block, 0, st, ctx->dd, g.ch, ctx->eps, eval_only); break;
    case 2: hipLaunchKernelGGL((k_hmc_traj<NB, 2>), grid, block, 0, st, ctx->dd, g.ch, ctx->eps, eval_only); break;
    case 3: hipLaunchKernelGGL((k_hmc_traj<NB, 3>), grid, block, 0, st, ctx->dd, g.ch, ctx->eps, eval_only); break;
    case 4: hipLaunchKernelGGL((k_hmc_traj<NB, 4>), grid, block, 0, st, ctx->dd, g.ch, ctx->eps, eval_only); break;
    default: hipLaunchKernelGGL((k_hmc_traj<NB, 0>), grid, block, 0, st, ctx->dd, g.ch, ctx->eps, eval_only); break;
  }
}
void launch_hmc_traj(rmhmc_ctx* ctx, Group& g, int eval_only) {
  launch(ctx, g, HEAVY, "medium", [&](hipStream_t st) {
    if (ctx->NB == 1) launch_hmc_traj_nb<1>(ctx, g, eval_only, st);
    else launch_hmc_traj_nb<2>(ctx, g, eval_only, st);
  });
}

void launch_hmc_global_step(rmhmc_ctx* ctx, const IterBase& b) {
  std::vector<Phase> ph;
  const double eps = ctx->eps;
  ph.push_back([ctx, b](Group& g) { IterParams ip = iter_params(ctx, g, b); SMALL(ctx, g, "small", k_hmc_begin, ctx->D, ctx->DP, g.ch, ip); });
  if (ctx->hmc_traj) {  // the whole trajectory of every chain in one launch
    ph.push_back([ctx](Group& g) { launch_hmc_traj(ctx, g, 0); });
    ph.push_back([ctx, b](Group& g) { IterParams ip = iter_params(ctx, g, b); SMALL(ctx, g, "small", k_hmc_end, ctx->D, ctx->DP, g.ch, ip); });
    run_phases(ctx, ph);
    return;
  }
  ph.push_back([=](Group& g) { SMALL(ctx, g, "small", k_hmc_pre, ctx->D, ctx->DP, g.ch, eps); });
  ph.push_back([ctx](Group& g) { launch_rowpass<RP_G>(ctx, g, g.ch.trj.w, nullptr); });
  ph.push_back([=](Group& g) { SMALL(ctx, g, "small", k_hmc_post, ctx->dd, g.ch, eps, g.nsplit); });
  ph.push_back([ctx, b](Group& g) { IterParams ip = iter_params(ctx, g, b); SMALL(ctx, g, "small", k_hmc_end, ctx->D, ctx->DP, g.ch, ip); });
  run_phases(ctx, ph);
}

void launch_global_step(rmhmc_ctx* ctx, const IterBase& b) {
  if (ctx->sampler == 1) { launch_hmc_global_step(ctx, b); return; }
  if (ctx->medium) {  // transition start, one leapfrog step and transition end of every chain in ONE launch
    const int guards = (ctx->flags & RMHMC_FLAG_GUARDS) ? 1 : 0;
    for (Group& g : ctx->groups) {
      const IterParams ip = iter_params(ctx, g, b);
      launch(ctx, g, HEAVY, "medium", [&](hipStream_t st) {
        launch_step_medium(ctx, g, st, guards, 0, 1, ip);
      });
    }
    return;
  }
  std::vector<Phase> ph;
  ph.push_back([ctx, b](Group& g) { IterParams ip = iter_params(ctx, g, b); SMALL(ctx, g, "small", k_iter_begin, ctx->D, ctx->DP, g.ch, ip); });
  step_phases(ctx, ph);
  ph.push_back([ctx, b](Group& g) { IterParams ip = iter_params(ctx, g, b); SMALL(ctx, g, "small", k_iter_end, ctx->D, ctx->DP, g.ch, ip); });
  run_phases(ctx, ph);
}

// Small-problem path: one launch = `nsteps` global steps of every chain (fused_small.hip.h).
void launch_fused(rmhmc_ctx* ctx, const IterBase& b, long long nsteps) {
  while (nsteps > 0) {
    const int chunk = (int)std::min<long long>(nsteps, 4096);
    for (Group& g : ctx->groups) {
      FusedParams fp{};
      fp.ip = iter_params(ctx, g, b);
      fp.eps = ctx->eps; fp.K = ctx->K; fp.nsteps = chunk; fp.DPs = ctx->DP; fp.init_eval = 0;
      launch(ctx, g, HEAVY, "fused", [&](hipStream_t st) {
        hipLaunchKernelGGL(k_fused_small, dim3((unsigned)((g.n + FS_WAVES - 1) / FS_WAVES)), dim3(64 * FS_WAVES), ctx->fused_lds, st,
                           ctx->dd, g.ch, fp);
      });
    }
    nsteps -= chunk;
    if (nsteps > 0) flow_tick(ctx, ctx->opt.inflight);  // (one launch = up to 4096 steps: at most four launches queued ahead)
  }
}

// whole-batch helpers on the main stream (callers fork/join around group work)
void fill_int(rmhmc_ctx* ctx, int* p, int v, size_t n) {
  hipLaunchKernelGGL(k_fill_int, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, p, v, n);
}
void fill_ll(rmhmc_ctx* ctx, long long* p, long long v, size_t n) {
  hipLaunchKernelGGL(k_fill_ll, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, p, v, n);
}

// host [n][D] -> device [n][DP] (padding stays zero) and back
int upload_vec(rmhmc_ctx* ctx, double* dst, const double* src) {
  HIPCK(hipMemcpy2DAsync(dst, ctx->DP * sizeof(double), src, ctx->D * sizeof(double), ctx->D * sizeof(double), ctx->n,
                         hipMemcpyHostToDevice, ctx->stream));
  return RMHMC_OK;
}
int download_vec(rmhmc_ctx* ctx, double* dst, const double* src) {
  HIPCK(hipMemcpy2DAsync(dst, ctx->D * sizeof(double), src, ctx->DP * sizeof(double), ctx->D * sizeof(double), ctx->n,
                         hipMemcpyDeviceToHost, ctx->stream));
  return RMHMC_OK;
}
template <typename T>
int download(rmhmc_ctx* ctx, T* dst, const T* src, size_t count) {
  HIPCK(hipMemcpyAsync(dst, src, count * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
  return RMHMC_OK;
}
template <typename T>
int upload(rmhmc_ctx* ctx, T* dst, const T* src, size_t count) {
  HIPCK(hipMemcpyAsync(dst, src, count * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
  return RMHMC_OK;
}

int sync(rmhmc_ctx* ctx) {
  HIPCK(hipStreamSynchronize(ctx->stream));
  HIPCK(hipGetLastError());
  return RMHMC_OK;
}

#define RC(x) do { int rc_ = (x); if (rc_ != RMHMC_OK) return rc_; } while (0)
#define NEED_DATA(ctx)                                                              \
  do {                                                                              \
    if (!(ctx)) return fail(nullptr, RMHMC_ERR_INVALID, "null context");            \
    if (!(ctx)->have_data) return fail(ctx, RMHMC_ERR_INVALID, "rmhmc_set_data has not been called"); \
    HIPCK(hipSetDevice((ctx)->device));                                             \
  } while (0)

// upload w into trj.w, zero (or upload) p, mark every chain active, evaluate the record.
// Leaves the streams forked: callers join before downloading.
int eval_at(rmhmc_ctx* ctx, const double* w, const double* p, bool sampler_init = false) {
  Chains& ch = ctx->ch;
  RC(upload_vec(ctx, ch.trj.w, w));
  if (p) RC(upload_vec(ctx, ch.p, p));
  else HIPCK(hipMemsetAsync(ch.p, 0, sizeof(double) * ctx->n * ctx->DP, ctx->stream));
  fill_int(ctx, ch.phase, 1, ctx->n);
  fill_int(ctx, ch.status, 0, ctx->n);
  if (sampler_init && ctx->medium) {  // same arithmetic as inside the one-launch steps (bit-exact checkpoint / resume)
    for (Group& g : ctx->groups)
      launch(ctx, g, HEAVY, "medium", [&](hipStream_t st) {
        launch_step_medium(ctx, g, st, 0, 1, 0, IterParams{});
      });
    return RMHMC_OK;
  }
  launch_eval_point(ctx);
  return RMHMC_OK;
}

// view of the per-chain arrays for chains [off, off+n)
Chains chains_view(const rmhmc_ctx* ctx, long long off, int n) {
  const size_t DP = ctx->DP, Mp = ctx->Mp;
  Chains v = ctx->ch;
  auto vec = [&](double* p) { return p + off * DP; };
  auto mat = [&](double* p) { return p + off * DP * DP; };
  for (Rec* r : {&v.cur, &v.trj}) {
    r->w = vec(r->w); r->grad = vec(r->grad); r->tr = vec(r->tr); r->L = mat(r->L); r->Ginv = mat(r->Ginv);
    r->ljl += off; r->hld += off;
  }
  v.p = vec(v.p); v.p0 = vec(v.p0); v.Hcur += off; v.Hprop += off; v.tau += off;
  v.steps_left += off; v.phase += off; v.status += off; v.nsteps_last += off; v.cstale += off; v.stale_list += off;
  v.iter += off; v.accepted += off; v.steps_done += off;
  v.wq = vec(v.wq); v.uq = vec(v.uq); v.PM = vec(v.PM); v.u0 = vec(v.u0); v.q = vec(v.q); v.last = vec(v.last);
  v.Gq = mat(v.Gq); v.rv0 += off * Mp; v.rv2 += off * Mp;
  v.n = n;
  return v;
}

// Run-time options that live in device-visible state.  The list of the chains that have just rejected a proposal (k_iter_end appends,
// k_crestore consumes) exists on the generic multi-launch path with c tiles when cdyn and crestore are on.
void apply_runtime_options(rmhmc_ctx* ctx) {
  for (Group& g : ctx->groups) {
    const bool ok = !ctx->big && !ctx->medium && !ctx->fused && g.ctile && ctx->opt.cdyn && ctx->opt.crestore;
    g.ch.stale_list = ok ? ctx->stale_list_alloc : nullptr;
    g.ch.stale_count = ctx->ch.stale_count;
  }
  ctx->ch.stale_list = nullptr;
}

}  // namespace

// =================================================================================================
// C-ABI
// =================================================================================================
extern "C" {

const char* rmhmc_version(void) { return "rmhmc-hip 0.1 (gfx950, fp64 MFMA)"; }
const char* rmhmc_last_error(const rmhmc_ctx* ctx) { return ctx ? ctx->err : g_err; }

int rmhmc_create(rmhmc_ctx** out, int32_t device_id, int64_t M, int32_t D, int64_t n_chains, int32_t dtype, uint32_t flags) {
  return rmhmc_create_opts(out, device_id, M, D, n_chains, dtype, flags, nullptr, 0);
}

int rmhmc_create_opts(rmhmc_ctx** out, int32_t device_id, int64_t M, int32_t D, int64_t n_chains, int32_t dtype, uint32_t flags,
                      const rmhmc_option* opts, int32_t n_opts) {
  rmhmc_ctx* ctx = nullptr;  // for the macros: errors go to the global message
  Options opt{};
  if (n_opts < 0 || (n_opts > 0 && !opts)) return fail(nullptr, RMHMC_ERR_INVALID, "rmhmc_create_opts: bad option array");
  for (int i = 0; i < n_opts; ++i) {
    const OptionDesc* d = find_option(opts[i].key);
    if (!d) return fail(nullptr, RMHMC_ERR_INVALID, std::string("rmhmc_create_opts: unknown option '") + (opts[i].key ? opts[i].key : "(null)") + "'");
    if (opts[i].value < d->lo || opts[i].value > d->hi)
      return fail(nullptr, RMHMC_ERR_INVALID, std::string("rmhmc_create_opts: value out of range for option '") + d->key + "'");
    opt.*(d->slot) = opts[i].value;
  }
  if (!out || M <= 0 || D <= 0 || n_chains <= 0) return fail(nullptr, RMHMC_ERR_INVALID, "rmhmc_create: bad shape");
  if (dtype != RMHMC_F64) return fail(nullptr, RMHMC_ERR_UNSUPPORTED, "rmhmc_create: only float64 is built (the reference is float64)");
  if (D > 256) return fail(nullptr, RMHMC_ERR_UNSUPPORTED, "rmhmc_create: D > 256 is not supported (64 < D <= 256 uses the blocked large-D path)");
  if (flags & RMHMC_FLAG_ORACLE_LITERAL) return fail(nullptr, RMHMC_ERR_UNSUPPORTED, "rmhmc_create: the literal variant exists only in the CPU oracle");
  if (M > (int64_t)1 << 30 || n_chains > (int64_t)1 << 30) return fail(nullptr, RMHMC_ERR_UNSUPPORTED, "rmhmc_create: M or n_chains too large");
  // (the row passes of the D <= 64 path address X, a chain group's c tiles and its leverages with 32-bit byte offsets from a buffer
  //  descriptor's base: buf_rsrc in kernels.hip.h)
  if (D <= 64 && (M + 63) / 64 * 64 * (int64_t)(16 * ((D + 15) / 16)) * 8 >= (int64_t)1 << 32)
    return fail(nullptr, RMHMC_ERR_UNSUPPORTED, "rmhmc_create: the data matrix of the D <= 64 path must stay below 4 GB");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, RMHMC_ERR_NO_DEVICE, "rmhmc_create: no HIP device available (this library has no CPU fallback)");
  if (device_id < 0 || device_id >= ndev) return fail(nullptr, RMHMC_ERR_NO_DEVICE, "rmhmc_create: device ordinal out of range");
  HIPCK(hipSetDevice(device_id));
  hipDeviceProp_t prop;
  HIPCK(hipGetDeviceProperties(&prop, device_id));
  if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
    return fail(nullptr, RMHMC_ERR_NO_DEVICE, std::string("rmhmc_create: kernels are built for gfx950 only, device is ") + prop.gcnArchName);
  ctx = new rmhmc_ctx();
  ctx->device = device_id;
  ctx->M = M; ctx->D = D; ctx->n = n_chains; ctx->flags = flags; ctx->opt = opt;
  ctx->NB = (D + 15) / 16; ctx->DP = 16 * ctx->NB;
  if (D > 64) {  // large-D path: 64-column blocks, NB = 4 tiles inside a block
    ctx->big = true;
    ctx->nbk = (D + 63) / 64;
    ctx->npairs = ctx->nbk * (ctx->nbk + 1) / 2;
    ctx->DP = 64 * ctx->nbk;
    ctx->NB = 4;
  }
  ctx->Mp = (int)((M + 63) / 64 * 64); ctx->nblk = ctx->Mp / 64;
  if (flags & RMHMC_FLAG_INT8_INNER_FULL) ctx->i8_inner_drop = 0;
  int rc = RMHMC_OK;
  auto body = [&]() -> int {
    HIPCK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    ctx->flow.resize(4);  // flow control events (flow_tick)
    for (auto& e : ctx->flow) { e = nullptr; HIPCK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); }
    const size_t n = n_chains, DP = ctx->DP, Mp = ctx->Mp;
    double *Xr, *Xt, *t;
    RC(dalloc(ctx, &Xr, Mp * DP)); RC(dalloc(ctx, &Xt, DP * Mp)); RC(dalloc(ctx, &t, Mp));
    ctx->dd.Xr = Xr; ctx->dd.Xt = Xt; ctx->dd.t = t;
    ctx->dd.M = (int)M; ctx->dd.Mp = ctx->Mp; ctx->dd.D = D; ctx->dd.DP = ctx->DP; ctx->dd.nblk = ctx->nblk;
    Chains& ch = ctx->ch;
    ch.n = (int)n;
    for (Rec* r : {&ch.cur, &ch.trj}) {
      RC(dalloc(ctx, &r->w, n * DP)); RC(dalloc(ctx, &r->grad, n * DP)); RC(dalloc(ctx, &r->tr, n * DP));
      RC(dalloc(ctx, &r->L, n * DP * DP)); RC(dalloc(ctx, &r->Ginv, n * DP * DP));
      RC(dalloc(ctx, &r->ljl, n)); RC(dalloc(ctx, &r->hld, n));
    }
    RC(dalloc(ctx, &ch.p, n * DP)); RC(dalloc(ctx, &ch.p0, n * DP)); RC(dalloc(ctx, &ch.Hcur, n)); RC(dalloc(ctx, &ch.Hprop, n));
    RC(dalloc(ctx, &ch.tau, n)); RC(dalloc(ctx, &ch.steps_left, n)); RC(dalloc(ctx, &ch.phase, n)); RC(dalloc(ctx, &ch.status, n)); RC(dalloc(ctx, &ch.cstale, n));
    RC(dalloc(ctx, &ch.stale_list, n)); RC(dalloc(ctx, &ch.stale_count, (size_t)1));
    ctx->stale_list_alloc = ch.stale_list;
    RC(dalloc(ctx, &ch.nsteps_last, n)); RC(dalloc(ctx, &ch.iter, n)); RC(dalloc(ctx, &ch.accepted, n)); RC(dalloc(ctx, &ch.steps_done, n));
    RC(dalloc(ctx, &ch.wq, n * DP)); RC(dalloc(ctx, &ch.uq, n * DP)); RC(dalloc(ctx, &ch.PM, n * DP)); RC(dalloc(ctx, &ch.u0, n * DP));
    RC(dalloc(ctx, &ch.q, n * DP)); RC(dalloc(ctx, &ch.last, n * DP)); RC(dalloc(ctx, &ch.Gq, n * DP * DP));
    RC(dalloc(ctx, &ch.rv0, n * Mp)); RC(dalloc(ctx, &ch.rv2, n * Mp));
    ctx->groups.resize(1);
    {
      Group& g = ctx->groups[0];
      g.off = 0;
      g.n = (int)n_chains;
      // row splits of the 16-chains-per-wave passes (option nsplit_waves).  D <= 64: ~2048 wavefronts per launch = ONE round of two
      // four-wave workgroups per CU - measured against the 6144 of rounds 1-2 (three rounds) on one box, interleaved: 14.73-14.92
      // against 14.99-15.08 ms per step at config 3, +2.7 % steps/s at 4096 chains, +6 % at 2048 and 1024 (fewer partial sums to write and
      // to add up, fewer prologues); the blocked large-D passes keep 6144 (config 5: 472.8 against 475.4 ms per step).
      const long long cgroups = (g.n + 15) / 16, nb16 = ctx->Mp / 16;
      const long long target = ctx->opt.nsplit_waves > 0 ? ctx->opt.nsplit_waves : (ctx->big ? 6144 : 2048);
      long long ns = (target + cgroups - 1) / cgroups;
      if (ns < 1) ns = 1;
      if (ns > nb16) ns = nb16;
      // ... but no more than 64 splits (option nsplit_max): the consumers sum the partials serially.  (Round 1 kept up to Mp/16 splits for
      // long data sets in small batches, when the one-chain-per-wave assembly dominated those shapes anyway; with the row ranges of
      // k_assemble / k_leverage it is the serial sums that cost: D 64, M 10000, 64 / 128 / 256 chains: 2.44 / 2.28 / 2.52 -> 1.69 / 1.49 /
      // 1.99 ms per step, the int8 path at 128-512 chains 10-30 % less; profiles/r02_fp64_batch_sweep.txt)
      if (ns > ctx->opt.nsplit_max) ns = ctx->opt.nsplit_max;
      g.nsplit = (int)ns;
      g.ch = chains_view(ctx, g.off, g.n);
      RC(dalloc(ctx, &g.ch.qpart, (size_t)g.nsplit * g.n * DP));
      RC(dalloc(ctx, &g.ch.gpart, (size_t)g.nsplit * g.n * DP));
      RC(dalloc(ctx, &g.ch.ljl_part, (size_t)g.n * g.nsplit));
      if (ctx->opt.ccache) RC(dalloc(ctx, &g.ctile, (size_t)((g.n + 15) / 16) * (ctx->Mp / 16) * 64));
      if (!ctx->big) {
        // fp64 assembly (k_assemble: one chain per wavefront over all M rows): below ~1024 chains the launch has fewer wavefronts than
        // the chip has SIMDs, so the rows are cut until ~2048 wavefronts exist (at least 256 rows per range, at most 16 ranges).
        // D 64, M 10000, 512 chains: the step took longer than with 1024 chains (13.0 vs 7.6 ms, profiles/r01_i8_threshold.txt).
        const long long waves = g.n;
        long long fs = waves >= 1024 ? 1 : std::min<long long>(16, (2048 + waves - 1) / waves);
        fs = std::min<long long>(fs, std::max(1, ctx->Mp / 256));
        if (ctx->opt.fsplit >= 1) fs = ctx->opt.fsplit;
        g.fsplit = (int)fs;
      }
    }
    int i8_slices = (int)((flags >> 12) & 7u);
    if (i8_slices == 0) i8_slices = 6;
    if (i8_slices < 4) i8_slices = 4;
    // int32 accumulators: a weight-g set sums (g+1) K products of two bytes, |.| <= 2^14 each, so one launch covers at most
    // i8_chunk stages of 32 (21845 rows at 6 slices); longer contractions are summed over several launches in fp64.
    if (flags & RMHMC_FLAG_INT8_METRIC) {
      const int S = i8_slices;
      ctx->i8_chunk = std::max(1, (int)(2147483647.0 / (S * 16384.0)) / 32);
      ctx->i8 = true;
      ctx->i8_requested = true;
      ctx->i8S = S;
      ctx->i8_bn = S <= 6 ? 128 : 64;
      ctx->i8_nks = (int)((M + 31) / 32);
      const int NP = D * (D + 1) / 2, NPp = (NP + ctx->i8_bn - 1) / ctx->i8_bn * ctx->i8_bn;
      std::vector<short> pa(NPp, 0), pb(NPp, 0);
      for (int a = 0, q = 0; a < D; ++a)  // rows of the lower triangle
        for (int b = 0; b <= a; ++b, ++q) { pa[q] = (short)a; pb[q] = (short)b; }
      short *d_pa, *d_pb; double* d_scale;
      RC(dalloc(ctx, &d_pa, (size_t)NPp)); RC(dalloc(ctx, &d_pb, (size_t)NPp)); RC(dalloc(ctx, &d_scale, (size_t)NPp));
      RC(dalloc(ctx, &ctx->d_ze, (size_t)NPp));
      HIPCK(hipMemcpyAsync(d_pa, pa.data(), NPp * sizeof(short), hipMemcpyHostToDevice, ctx->stream));
      HIPCK(hipMemcpyAsync(d_pb, pb.data(), NPp * sizeof(short), hipMemcpyHostToDevice, ctx->stream));
      RC(sync(ctx));
      int* d_cexp;
      RC(dalloc(ctx, &d_cexp, (size_t)ctx->DP));
      RC(dalloc(ctx, &ctx->d_cmin, (size_t)ctx->DP)); RC(dalloc(ctx, &ctx->d_cmax, (size_t)ctx->DP));
      ctx->pairs = I8Pairs{d_pa, d_pb, d_scale, NP, NPp, d_cexp};
      RC(dalloc(ctx, &ctx->d_Zs, (size_t)S * ctx->i8_nks * NPp * 32));
      ctx->i8_nkp = (NP + 31) / 32;
      ctx->i8_NRp = (ctx->Mp + ctx->i8_bn - 1) / ctx->i8_bn * ctx->i8_bn;
      RC(dalloc(ctx, &ctx->d_Zt, (size_t)S * ctx->i8_nkp * ctx->i8_NRp * 32));
      RC(dalloc(ctx, &ctx->d_zre, (size_t)ctx->i8_NRp)); RC(dalloc(ctx, &ctx->d_zscale, (size_t)ctx->i8_NRp));
      for (Group& g : ctx->groups) {
        g.nCp = (g.n + I8_BM - 1) / I8_BM * I8_BM;
        RC(dalloc(ctx, &g.Vs, (size_t)S * ctx->i8_nks * g.nCp * 32));
        RC(dalloc(ctx, &g.vbad, (size_t)g.nCp));
        RC(dalloc(ctx, &g.vexp, (size_t)g.nCp));
        RC(dalloc(ctx, &g.vexp_d, (size_t)g.nCp)); RC(dalloc(ctx, &g.rebase, (size_t)g.nCp)); RC(dalloc(ctx, &g.dmax, (size_t)1));
        if (!ctx->big) { g.ch.i8_vbad = g.vbad; g.ch.i8_dmax = g.dmax; }
        if (ctx->big && ctx->opt.i8_delta && S == 6) RC(dalloc(ctx, &g.Gbase, (size_t)g.n * ctx->DP * ctx->DP));
        RC(dalloc(ctx, &g.Qs, (size_t)S * ctx->i8_nkp * g.nCp * 32));
        RC(dalloc(ctx, &g.qscale, (size_t)g.nCp));
        // small batches: cut the k range so that about 256 workgroups exist (at least 8 stages per piece, at most 16 pieces; only
        // when the whole range fits one overflow-safe launch)
        auto pieces = [&](long long tiles, int stages) {
          long long k = std::min<long long>(16, 256 / std::max<long long>(1, tiles));
          k = std::min<long long>(k, stages / 8);
          if (k < 2 || stages > ctx->i8_chunk || ctx->big) return 1;  // (large-D: the identity padding of G lives in Gq itself)
          const int per = (int)((stages + k - 1) / k);
          return (stages + per - 1) / per;
        };
        g.ksplit_a = pieces((long long)(g.nCp / I8_BM) * (NPp / ctx->i8_bn), ctx->i8_nks);
        g.ksplit_l = pieces((long long)(g.nCp / I8_BM) * (ctx->i8_NRp / ctx->i8_bn), ctx->i8_nkp);
        if (g.ksplit_a > 1) RC(dalloc(ctx, &g.Gpart, (size_t)g.ksplit_a * g.n * ctx->DP * ctx->DP));
        if (g.ksplit_a == 1 && ctx->i8_bn == 128 && NP % 128 != 0 && ctx->opt.i8_tail != 0) {
          const int ntail = (NP % 128 + 31) / 32;
          g.tail_pieces = (int)std::max<long long>(1, std::min<long long>(8, 256 / ((long long)(g.nCp / I8_BM) * ntail)));
          RC(dalloc(ctx, &g.Tq, (size_t)g.tail_pieces * S * g.nCp * 32 * ntail));
        }
        if (g.ksplit_l > 1) RC(dalloc(ctx, &g.Rpart, (size_t)g.ksplit_l * g.n * ctx->Mp));
      }
      I8_SWITCH(ctx, {
        constexpr int lds = i8_lds_bytes<S_, WN_, TN_>();
        // (the instantiation's own, fixed size: these kernels also hold a few KB of static LDS, and static + dynamic must stay within 160 KB)
        auto kfn = k_assemble_i8<S_, WN_, TN_>;
        HIPCK(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        auto kfn2 = k_leverage_i8<S_, WN_, TN_>;
        HIPCK(hipFuncSetAttribute((const void*)kfn2, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        auto kfn3 = k_assemble_i8_tail<S_>;
        HIPCK(hipFuncSetAttribute((const void*)kfn3, hipFuncAttributeMaxDynamicSharedMemorySize, (i8_lds_bytes<S_, 1, 1>())));
      });
      if (S == 6) {  // the one-launch delta assembly (launch_assemble_i8_delta): dynamic LDS of its widest instantiation
        constexpr int ld = i8_lds_bytes<6, 4, 1>() > i8_lds_bytes<4, 4, 1>() ? i8_lds_bytes<6, 4, 1>() : i8_lds_bytes<4, 4, 1>();
        constexpr int ldt = i8_lds_bytes<6, 1, 1>() > i8_lds_bytes<4, 1, 1>() ? i8_lds_bytes<6, 1, 1>() : i8_lds_bytes<4, 1, 1>();
        HIPCK(hipFuncSetAttribute((const void*)k_assemble_i8_sel<4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, ld));
        HIPCK(hipFuncSetAttribute((const void*)k_assemble_i8_tail_sel, hipFuncAttributeMaxDynamicSharedMemorySize, ldt));
      }
      if (S == 6)  // the instantiations of the inner assemblies (launch_assemble)
        for (int sv = S - 2; sv < S; ++sv)
          I8_SWITCH_S(sv, {
            auto kfn = k_assemble_i8<S_, WN_, TN_>;
            HIPCK(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (i8_lds_bytes<S_, WN_, TN_>())));
            auto kfn2 = k_leverage_i8<S_, WN_, TN_>;
            HIPCK(hipFuncSetAttribute((const void*)kfn2, hipFuncAttributeMaxDynamicSharedMemorySize, (i8_lds_bytes<S_, WN_, TN_>())));
            auto kfn3 = k_assemble_i8_tail<S_>;
            HIPCK(hipFuncSetAttribute((const void*)kfn3, hipFuncAttributeMaxDynamicSharedMemorySize, (i8_lds_bytes<S_, 1, 1>())));
          });
    }
    for (Group& g : ctx->groups) {  // planes of the fp64 small-batch assembly (shared with the int8 k-split planes, whichever is larger)
      const int need = std::max(g.fsplit, g.ksplit_a);
      if (g.fsplit > 1 && (!g.Gpart || g.fsplit > g.ksplit_a)) RC(dalloc(ctx, &g.Gpart, (size_t)need * g.n * ctx->DP * ctx->DP));
    }
    RC(dalloc(ctx, &ctx->d_z, n * (size_t)D)); RC(dalloc(ctx, &ctx->d_ulen, n)); RC(dalloc(ctx, &ctx->d_gdir, n)); RC(dalloc(ctx, &ctx->d_uacc, n));
    if (ctx->big) {
      RC(dalloc(ctx, &ctx->d_Wd, n * (size_t)ctx->nbk * 4096));
      // (the fp64 leverage pass of the large-D path; with the int8 path it is allocated only if the certificate sends set_data back to fp64)
      if (!(flags & RMHMC_FLAG_INT8_METRIC)) RC(dalloc(ctx, &ctx->d_hpart, (size_t)ctx->npairs * n * Mp));
    }
    fill_int(ctx, ch.cstale, 1, n);  // (no c tiles yet; the first evaluation clears it)
    RC(dalloc(ctx, &ctx->d_nsteps, n)); RC(dalloc(ctx, &ctx->d_dir, n)); RC(dalloc(ctx, &ctx->d_done, 1)); RC(dalloc(ctx, &ctx->d_steps0, n));
    RC(dalloc(ctx, &ctx->d_miniter, 1));
    RC(dalloc(ctx, &ctx->d_orig, n)); RC(dalloc(ctx, &ctx->d_T, 2 * n));
    {  // mid-size problems in small batches: one launch per leapfrog step (option medium = 0 disables it)
      // measured per global step at one chain (tools/bench_single.py): australian (D = 15) 108 us vs 218 us generic, heart (D = 14)
      // 85 vs 154, german (D = 25) 236 vs 386
      const bool on = ctx->opt.medium && !ctx->big && D > FS_D && D <= 32 && ctx->Mp <= MS_MAXMP && n_chains <= 512;
      if (on) {
        const size_t lds = sizeof(double) * (ctx->NB == 1 ? ms_lds_doubles<1>(ctx->Mp) : ms_lds_doubles<2>(ctx->Mp));
        if (ctx->NB == 1) {
          HIPCK(hipFuncSetAttribute((const void*)k_step_medium<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, MAX_LDS));
          HIPCK(hipFuncSetAttribute((const void*)k_step_medium<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, MAX_LDS));
          HIPCK(hipFuncSetAttribute((const void*)k_step_medium<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, MAX_LDS));
          HIPCK(hipFuncSetAttribute((const void*)k_step_medium<1, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, MAX_LDS));
          HIPCK(hipFuncSetAttribute((const void*)k_step_medium<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, MAX_LDS));
        } else {
          HIPCK(hipFuncSetAttribute((const void*)k_step_medium<2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, MAX_LDS));
          HIPCK(hipFuncSetAttribute((const void*)k_step_medium<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, MAX_LDS));
          HIPCK(hipFuncSetAttribute((const void*)k_step_medium<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, MAX_LDS));
        }
        ctx->medium = true;
        ctx->medium_lds = lds;
      }
    }
    {  // plain HMC in small batches: one launch per trajectory (option medium = 0 disables it too)
      // any batch for short data sets (rows in registers; australian, tools/bench_hmc_batch.py: 2048 chains 89 M leapfrog-steps/s vs 39 M
      // generic, 512 chains 59 M vs 9 M; at 8192 chains the generic path has caught up since its row passes run in one round of
      // workgroups - 107 M vs 114 M), small batches otherwise
      long long maxn = ctx->Mp <= 1024 ? (1ll << 40) : 512;
      if (ctx->opt.hmc_traj_maxn >= 0) maxn = ctx->opt.hmc_traj_maxn;
      ctx->hmc_traj = ctx->opt.medium && !ctx->big && D <= 32 && n_chains <= maxn;
    }
    {  // small-problem path eligibility (option fused = 0 disables it)
      const size_t lds = ((size_t)(FS_D + 1 + FS_WAVES) * ctx->Mp + (size_t)FS_WAVES * FS_PT) * sizeof(double);
      if (ctx->opt.fused && D <= FS_D && lds <= 160 * 1024) {
        HIPCK(hipFuncSetAttribute((const void*)k_fused_small, hipFuncAttributeMaxDynamicSharedMemorySize, MAX_LDS));
        ctx->fused = true;
        ctx->fused_lds = lds;
      }
    }
    RC(sync(ctx));
    return RMHMC_OK;
  };
  rc = body();
  if (rc != RMHMC_OK) {
    snprintf(g_err, 512, "%s", ctx->err);
    rmhmc_destroy(ctx);
    return rc;
  }
  apply_runtime_options(ctx);
  *out = ctx;
  return RMHMC_OK;
}

void rmhmc_destroy(rmhmc_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  for (auto& e : ctx->flow) if (e) (void)hipEventDestroy(e);
  for (auto& kv : ctx->events) for (auto& e : kv.second) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  for (auto& e : ctx->pool) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  for (void* p : ctx->allocs) (void)hipFree(p);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

int rmhmc_set_option(rmhmc_ctx* ctx, const char* key, int64_t value) {
  if (!ctx) return fail(nullptr, RMHMC_ERR_INVALID, "set_option: null context");
  const OptionDesc* d = find_option(key);
  if (!d) return fail(ctx, RMHMC_ERR_INVALID, std::string("set_option: unknown option '") + (key ? key : "(null)") + "'");
  if (d->create_only) return fail(ctx, RMHMC_ERR_INVALID, std::string("set_option: '") + key + "' shapes the context and can only be given to rmhmc_create_opts");
  if (value < d->lo || value > d->hi) return fail(ctx, RMHMC_ERR_INVALID, std::string("set_option: value out of range for '") + key + "'");
  HIPCK(hipSetDevice(ctx->device));
  HIPCK(hipStreamSynchronize(ctx->stream));
  ctx->opt.*(d->slot) = value;
  apply_runtime_options(ctx);
  if (ctx->stale_list_alloc) {  // a list left over from a run with the other setting must not be consumed
    HIPCK(hipMemsetAsync(ctx->ch.stale_count, 0, sizeof(int), ctx->stream));
    fill_int(ctx, ctx->ch.cstale, 1, ctx->n);  // (every chain's tiles count as stale: the next first pass recomputes them)
    HIPCK(hipStreamSynchronize(ctx->stream));
  }
  return RMHMC_OK;
}
int rmhmc_get_option(rmhmc_ctx* ctx, const char* key, int64_t* value_out) {
  if (!ctx || !value_out) return fail(ctx, RMHMC_ERR_INVALID, "get_option: null pointer");
  const OptionDesc* d = find_option(key);
  if (!d) return fail(ctx, RMHMC_ERR_INVALID, std::string("get_option: unknown option '") + (key ? key : "(null)") + "'");
  *value_out = ctx->opt.*(d->slot);
  return RMHMC_OK;
}
int rmhmc_options(rmhmc_ctx* ctx, char* buf, size_t len) {
  if (!ctx || !buf || !len) return fail(ctx, RMHMC_ERR_INVALID, "options: bad argument");
  std::string o;
  for (const OptionDesc& d : kOptions) o += (o.empty() ? "" : " ") + std::string(d.key) + "=" + std::to_string(ctx->opt.*(d.slot));
  snprintf(buf, len, "%s", o.c_str());
  return RMHMC_OK;
}

int rmhmc_device_info(rmhmc_ctx* ctx, char* buf, size_t len) {
  if (!ctx || !buf) return fail(ctx, RMHMC_ERR_INVALID, "device_info: bad argument");
  hipDeviceProp_t prop;
  HIPCK(hipGetDeviceProperties(&prop, ctx->device));
  snprintf(buf, len, "%s %s, %d CUs, %.0f MHz, %.1f GiB; M=%lld (padded %d) D=%d (padded %d, %d MFMA tiles) chains=%lld%s",
           prop.name, prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000.0, prop.totalGlobalMem / 1073741824.0,
           (long long)ctx->M, ctx->Mp, ctx->D, ctx->DP, ctx->NB * (ctx->NB + 1) / 2, (long long)ctx->n,
           ctx->fused ? ", fused small-problem path" : ctx->medium ? ", one-launch step" : ctx->big ? ", blocked large-D path" : "");
  if (ctx->i8_requested) {
    const size_t k = strlen(buf);
    snprintf(buf + k, len > k ? len - k : 0, "; int8 metric path %d slices: %s (certificate %.2e%s)", ctx->i8S,
             ctx->i8 ? "active" : "NOT certified for this data, fp64 matrix cores used", ctx->i8_bound,
             ctx->have_data ? "" : ", no data yet");
  }
  {
    const size_t k = strlen(buf);
    if (len > k + 12) {
      snprintf(buf + k, len - k, "; options: ");
      const size_t k2 = strlen(buf);
      (void)rmhmc_options(ctx, buf + k2, len - k2);
    }
  }
  return RMHMC_OK;
}

int rmhmc_set_data(rmhmc_ctx* ctx, const double* X, const double* t, double alpha) {
  if (!ctx || !X || !t || !(alpha > 0)) return fail(ctx, RMHMC_ERR_INVALID, "set_data: bad argument");
  HIPCK(hipSetDevice(ctx->device));
  const size_t M = ctx->M, D = ctx->D, DP = ctx->DP, Mp = ctx->Mp;
  std::vector<double> xr(Mp * DP, 0.0), xt(DP * Mp, 0.0), tt(Mp, 0.0);
  for (size_t n = 0; n < M; ++n) {
    for (size_t d = 0; d < D; ++d) {
      xr[n * DP + d] = X[n * D + d];
      xt[d * Mp + n] = X[n * D + d];
    }
    tt[n] = t[n];
  }
  HIPCK(hipMemcpyAsync((void*)ctx->dd.Xr, xr.data(), xr.size() * 8, hipMemcpyHostToDevice, ctx->stream));
  HIPCK(hipMemcpyAsync((void*)ctx->dd.Xt, xt.data(), xt.size() * 8, hipMemcpyHostToDevice, ctx->stream));
  HIPCK(hipMemcpyAsync((void*)ctx->dd.t, tt.data(), tt.size() * 8, hipMemcpyHostToDevice, ctx->stream));
  if (ctx->i8_requested) {  // fixed operand of the int8 metric path: slices of x_a x_b for every column pair
    std::vector<int> cexp(DP, 0);  // power-of-two column equilibration of the leverage pass: max_n |x_na| < 2^cexp[a]
    std::vector<double> cmin(DP, 0.0), cmax(DP, 0.0);
    for (size_t d = 0; d < D; ++d) {
      double m = 0.0, lo = INFINITY;
      for (size_t n = 0; n < M; ++n) { const double a = std::fabs(X[n * D + d]); m = std::max(m, a); lo = std::min(lo, a); }
      if (m > 0.0 && m < 1e300) (void)std::frexp(m, &cexp[d]);
      if (m < 1e300) { cmin[d] = lo; cmax[d] = m; }
    }
    HIPCK(hipMemcpyAsync((void*)ctx->pairs.cexp, cexp.data(), sizeof(int) * DP, hipMemcpyHostToDevice, ctx->stream));
    HIPCK(hipMemcpyAsync(ctx->d_cmin, cmin.data(), sizeof(double) * DP, hipMemcpyHostToDevice, ctx->stream));
    HIPCK(hipMemcpyAsync(ctx->d_cmax, cmax.data(), sizeof(double) * DP, hipMemcpyHostToDevice, ctx->stream));
    RC(sync(ctx));  // (cexp is a local)
    hipLaunchKernelGGL(k_zmax, dim3((unsigned)ctx->pairs.NPp), dim3(256), 0, ctx->stream, ctx->dd.Xt, (int)M, (int)Mp, ctx->pairs, ctx->d_ze,
                       (double*)ctx->pairs.scale);
    const dim3 grid((unsigned)((ctx->i8_nks * 8 + 255) / 256), (unsigned)ctx->pairs.NPp);
    I8_SWITCH(ctx, hipLaunchKernelGGL((k_zsplit<S_>), grid, dim3(256), 0, ctx->stream, ctx->dd.Xt, (int)M, (int)Mp, ctx->pairs, ctx->d_ze,
                                      ctx->i8_nks, ctx->d_Zs); (void)WN_; (void)TN_);
    hipLaunchKernelGGL(k_zrowmax, dim3((unsigned)((ctx->i8_NRp + 255) / 256)), dim3(256), 0, ctx->stream, ctx->dd.Xr, (int)M, (int)D, (int)DP,
                       ctx->i8_NRp, ctx->pairs.cexp, ctx->d_zre, ctx->d_zscale);
    const dim3 gridt((unsigned)(ctx->i8_NRp / 32), (unsigned)ctx->i8_nkp);
    I8_SWITCH(ctx, hipLaunchKernelGGL((k_ztsplit<S_>), gridt, dim3(256), 0, ctx->stream, ctx->dd.Xr, (int)M, (int)DP, ctx->pairs, ctx->d_zre,
                                      ctx->i8_nkp, ctx->i8_NRp, ctx->d_Zt); (void)WN_; (void)TN_);
    HIPCK(hipGetLastError());
  }
  RC(sync(ctx));
  if (ctx->i8_requested) {
    // Certificate of the fixed-point assembly for THIS data (metric_i8.hip.h, "Error bound"): worst-case error of G_ab relative to
    // sqrt(G0_aa G0_bb), G0 = X'X/4 + I/alpha (the metric at w = 0), maximised over the column pairs.
    std::vector<int> ze(ctx->pairs.NP);
    HIPCK(hipMemcpy(ze.data(), ctx->d_ze, sizeof(int) * ze.size(), hipMemcpyDeviceToHost));
    std::vector<double> g0(D, 1.0 / alpha);
    for (size_t n = 0; n < M; ++n)
      for (size_t d = 0; d < D; ++d) g0[d] += 0.25 * X[n * D + d] * X[n * D + d];
    double bound = 0.0;
    for (size_t a = 0, q = 0; a < D; ++a)
      for (size_t b = 0; b <= a; ++b, ++q)
        bound = std::max(bound, std::ldexp((double)ctx->i8S * (double)M, ze[q] - 8 * ctx->i8S) / std::sqrt(g0[a] * g0[b]));
    // (delta assembly at the end of a step: the slice products dropped from the difference add to those dropped from the base matrix)
    if (ctx->opt.i8_delta && ctx->i8S == 6) bound *= (2.0 * ctx->i8S - 1.0) / ctx->i8S;
    ctx->i8_bound = bound;
    const bool ok = !(ctx->flags & RMHMC_FLAG_INT8_CERTIFY) || bound <= RMHMC_INT8_CERTIFY_TOL;
    if (!ok && ctx->big && !ctx->d_hpart) RC(dalloc(ctx, &ctx->d_hpart, (size_t)ctx->npairs * ctx->n * Mp));
    ctx->i8 = ok;  // not certified: this data set runs on the fp64 matrix cores (same context, same results to fp64 rounding)
    RC(sync(ctx));
  }
  ctx->alpha = alpha;
  ctx->dd.inv_alpha = 1.0 / alpha;
  ctx->dd.log_prior_const = -0.5 * std::log(2.0 * M_PI * alpha);
  ctx->have_data = true;
  ctx->chains_ready = false;
  return RMHMC_OK;
}

// ---- unit entry points --------------------------------------------------------------------------
int rmhmc_log_posterior(rmhmc_ctx* ctx, const double* w, double* ljl_out) {
  NEED_DATA(ctx);
  if (!w || !ljl_out) return fail(ctx, RMHMC_ERR_INVALID, "log_posterior: null pointer");
  ctx->chains_ready = false;
  RC(eval_at(ctx, w, nullptr));
  RC(download(ctx, ljl_out, ctx->ch.trj.ljl, ctx->n));
  return sync(ctx);
}

int rmhmc_metric(rmhmc_ctx* ctx, const double* w, double* G_out, double* half_logdet_out, double* grad_out) {
  NEED_DATA(ctx);
  if (!w) return fail(ctx, RMHMC_ERR_INVALID, "metric: null pointer");
  ctx->chains_ready = false;
  if (ctx->big && G_out && !ctx->d_Gcopy) RC(dalloc(ctx, &ctx->d_Gcopy, (size_t)ctx->n * ctx->DP * ctx->DP));
  ctx->want_G = ctx->big && G_out;
  int rc_eval = eval_at(ctx, w, nullptr);
  ctx->want_G = false;
  RC(rc_eval);
  if (G_out)
    for (int64_t c = 0; c < ctx->n; ++c)  // strip the padding: [DP][DP] -> [D][D]
      HIPCK(hipMemcpy2DAsync(G_out + c * ctx->D * ctx->D, ctx->D * 8, (ctx->big ? ctx->d_Gcopy : ctx->ch.Gq) + c * ctx->DP * ctx->DP, ctx->DP * 8, ctx->D * 8, ctx->D,
                             hipMemcpyDeviceToHost, ctx->stream));
  if (half_logdet_out) RC(download(ctx, half_logdet_out, ctx->ch.trj.hld, ctx->n));
  if (grad_out) RC(download_vec(ctx, grad_out, ctx->ch.trj.grad));
  RC(sync(ctx));
  if (G_out && ctx->i8) {  // the int8 assembly writes the lower triangle only
    const int64_t D = ctx->D;
    for (int64_t c = 0; c < ctx->n; ++c)
      for (int64_t a = 0; a < D; ++a)
        for (int64_t b = a + 1; b < D; ++b) G_out[(c * D + a) * D + b] = G_out[(c * D + b) * D + a];
  }
  return RMHMC_OK;
}

int rmhmc_metric_terms(rmhmc_ctx* ctx, const double* w, const double* p, double* trace_out, double* quad_out) {
  NEED_DATA(ctx);
  if (!w) return fail(ctx, RMHMC_ERR_INVALID, "metric_terms: null pointer");
  ctx->chains_ready = false;
  RC(eval_at(ctx, w, p));
  if (trace_out) RC(download_vec(ctx, trace_out, ctx->ch.trj.tr));
  if (quad_out && p) RC(download_vec(ctx, quad_out, ctx->ch.last));
  return sync(ctx);
}

int rmhmc_leapfrog(rmhmc_ctx* ctx, double* w, double* p, double eps, const int32_t* dir, const int32_t* nsteps, int32_t K,
                   double* half_logdet_out, int32_t* status_out) {
  NEED_DATA(ctx);
  if (!w || !p || !dir || !nsteps || K < 1) return fail(ctx, RMHMC_ERR_INVALID, "leapfrog: null pointer or K < 1");
  ctx->chains_ready = false;
  int maxs = 0;
  for (int64_t c = 0; c < ctx->n; ++c) {
    if (nsteps[c] < 0 || (dir[c] != 1 && dir[c] != -1)) return fail(ctx, RMHMC_ERR_INVALID, "leapfrog: nsteps >= 0 and dir = +-1 required");
    if (nsteps[c] > maxs) maxs = nsteps[c];
  }
  ctx->eps = eps; ctx->K = K;
  RC(upload(ctx, ctx->d_nsteps, nsteps, ctx->n));
  RC(upload(ctx, ctx->d_dir, dir, ctx->n));
  RC(eval_at(ctx, w, p));
  for (Group& g : ctx->groups)
    launch(ctx, g, LIGHT, "small", [&](hipStream_t st) {
      hipLaunchKernelGGL(k_set_leapfrog, dim3((unsigned)((g.n + 255) / 256)), dim3(256), 0, st, g.n, ctx->d_nsteps + g.off, ctx->d_dir + g.off, g.ch);
    });
  for (int s = 0; s < maxs; ++s) {
    launch_step(ctx);
    for (Group& g : ctx->groups)
      launch(ctx, g, LIGHT, "small", [&](hipStream_t st) {
        hipLaunchKernelGGL(k_park_finished, dim3((unsigned)((g.n + 255) / 256)), dim3(256), 0, st, g.n, g.ch);
      });
  }
  RC(download_vec(ctx, w, ctx->ch.trj.w));
  RC(download_vec(ctx, p, ctx->ch.p));
  if (half_logdet_out) RC(download(ctx, half_logdet_out, ctx->ch.trj.hld, ctx->n));
  if (status_out) RC(download(ctx, status_out, ctx->ch.status, ctx->n));
  return sync(ctx);
}

// shared by transition / sample / chains_init: evaluate the record at theta0 and commit it as the
// current point of every chain
static int init_chains(rmhmc_ctx* ctx, const double* theta0_host /* [n][D] or NULL */) {
  std::vector<double> th;
  if (!theta0_host) {
    th.assign((size_t)ctx->n * ctx->D, 1e-3);  // rmhmc.py:27
    theta0_host = th.data();
  }
  if (ctx->fused) {  // the fused path evaluates its own initial record (same arithmetic as inside its steps)
    RC(upload_vec(ctx, ctx->ch.cur.w, theta0_host));
    for (Group& g : ctx->groups) {
      FusedParams fp{};
      fp.ip = iter_params(ctx, g, IterBase{0, 0, 0, nullptr, false, false});
      fp.eps = ctx->eps; fp.K = ctx->K; fp.nsteps = 0; fp.DPs = ctx->DP; fp.init_eval = 1;
      launch(ctx, g, HEAVY, "fused", [&](hipStream_t st) {
        hipLaunchKernelGGL(k_fused_small, dim3((unsigned)((g.n + FS_WAVES - 1) / FS_WAVES)), dim3(64 * FS_WAVES), ctx->fused_lds, st,
                           ctx->dd, g.ch, fp);
      });
    }
  } else {
    RC(eval_at(ctx, theta0_host, nullptr, true));
    for (Group& g : ctx->groups) SMALL(ctx, g, "small", k_commit_all, ctx->D, ctx->DP, g.ch);
  }
  fill_int(ctx, ctx->ch.phase, 0, ctx->n);
  fill_int(ctx, ctx->ch.steps_left, 0, ctx->n);
  fill_int(ctx, ctx->ch.status, 0, ctx->n);
  fill_ll(ctx, ctx->ch.iter, 0, ctx->n);
  fill_ll(ctx, ctx->ch.accepted, 0, ctx->n);
  fill_ll(ctx, ctx->ch.steps_done, 0, ctx->n);
  HIPCK(hipMemsetAsync(ctx->d_done, 0, sizeof(int), ctx->stream));
  return sync(ctx);  // theta0 staging vector goes out of scope
}

int rmhmc_transition(rmhmc_ctx* ctx, double* w, const double* z, const double* u_len, const double* g_dir, const double* u_acc,
                     int32_t L, double eps, int32_t K, int32_t* accepted_out, int32_t* nsteps_out, double* H_cur_out,
                     double* H_prop_out, double* w_prop_out, double* p_prop_out, double* half_logdet_prop_out, int32_t* status_out) {
  NEED_DATA(ctx);
  if (!w || !z || !u_len || !g_dir || !u_acc || L < 1 || K < 1) return fail(ctx, RMHMC_ERR_INVALID, "transition: null pointer, L < 1 or K < 1");
  ctx->chains_ready = false;
  ctx->L = L; ctx->eps = eps; ctx->K = K;
  RC(init_chains(ctx, w));
  RC(upload(ctx, ctx->d_z, z, (size_t)ctx->n * ctx->D));
  RC(upload(ctx, ctx->d_ulen, u_len, ctx->n));
  RC(upload(ctx, ctx->d_gdir, g_dir, ctx->n));
  RC(upload(ctx, ctx->d_uacc, u_acc, ctx->n));
  const IterBase ib{1, 0, 0, nullptr, true, false};
  if (ctx->fused) {
    launch_fused(ctx, ib, L);
  } else {
    launch_iter_begin(ctx, ib);
    launch_iter_end(ctx, ib);  // trajectories of zero steps
    for (int s = 0; s < L; ++s) {
      launch_step(ctx);
      launch_iter_end(ctx, ib);
    }
  }
  std::vector<long long> acc(ctx->n);
  RC(download_vec(ctx, w, ctx->ch.cur.w));
  RC(download(ctx, acc.data(), ctx->ch.accepted, ctx->n));
  if (nsteps_out) RC(download(ctx, nsteps_out, ctx->ch.nsteps_last, ctx->n));
  if (H_cur_out) RC(download(ctx, H_cur_out, ctx->ch.Hcur, ctx->n));
  if (H_prop_out) RC(download(ctx, H_prop_out, ctx->ch.Hprop, ctx->n));
  if (w_prop_out) RC(download_vec(ctx, w_prop_out, ctx->ch.trj.w));
  if (p_prop_out) RC(download_vec(ctx, p_prop_out, ctx->ch.p));
  if (half_logdet_prop_out) RC(download(ctx, half_logdet_prop_out, ctx->ch.trj.hld, ctx->n));
  if (status_out) RC(download(ctx, status_out, ctx->ch.status, ctx->n));
  RC(sync(ctx));
  if (accepted_out) for (int64_t c = 0; c < ctx->n; ++c) accepted_out[c] = (int32_t)acc[c];
  return RMHMC_OK;
}

// ---- bulk entry points --------------------------------------------------------------------------
// The ~40 launches of one global step captured once into a hipGraph and replayed: the generic path is launch bound
// for small batches (config 1: 0.5 ms per step of one chain, nearly all of it launch latency).  Option graph = 0 disables.
struct StepGraph {
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  ~StepGraph() {
    if (exec) (void)hipGraphExecDestroy(exec);
    if (graph) (void)hipGraphDestroy(graph);
  }
};
static bool step_graph_usable(const rmhmc_ctx* ctx, long long nsteps) {
  if (ctx->groups.size() != 1 || ctx->timing || nsteps < 8) return false;
  if (ctx->fused && ctx->sampler == 0) return false;
  if (ctx->medium && ctx->sampler == 0) return false;  // (the global step is ONE launch there: a one-node graph only costs its instantiation)
  return ctx->opt.graph != 0;
}
static bool build_step_graph(rmhmc_ctx* ctx, const IterBase& ib, StepGraph& sg) {
  if (hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) return false;
  launch_global_step(ctx, ib);
  if (hipStreamEndCapture(ctx->stream, &sg.graph) != hipSuccess || !sg.graph) { (void)hipGetLastError(); return false; }
  if (hipGraphInstantiate(&sg.exec, sg.graph, nullptr, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); sg.exec = nullptr; return false; }
  return true;
}
// nsteps global steps of the generic path (graph replay when it pays, plain launches otherwise)
static void run_generic_steps(rmhmc_ctx* ctx, const IterBase& ib, long long nsteps, StepGraph* sg) {
  if (sg && sg->exec) {
    for (long long s = 0; s < nsteps; ++s) { (void)hipGraphLaunch(sg->exec, ctx->stream); flow_tick(ctx); }
  } else {
    for (long long s = 0; s < nsteps; ++s) { launch_global_step(ctx, ib); flow_tick(ctx); }
  }
}

// number of chains that reached the iteration limit and the completed transitions of the slowest chain, in one round trip
static int poll_progress(rmhmc_ctx* ctx, int* done, long long* min_iter) {
  HIPCK(hipMemsetAsync(ctx->d_miniter, 0xff, sizeof(unsigned long long), ctx->stream));
  hipLaunchKernelGGL(k_min_iter, dim3((unsigned)((ctx->n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->ch.iter, (size_t)ctx->n, ctx->d_miniter);
  unsigned long long mi = 0;
  HIPCK(hipMemcpyAsync(done, ctx->d_done, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  HIPCK(hipMemcpyAsync(&mi, ctx->d_miniter, sizeof(mi), hipMemcpyDeviceToHost, ctx->stream));
  RC(sync(ctx));
  *min_iter = (long long)mi;
  return RMHMC_OK;
}

// rmhmc_set_progress: hand the caller the counters at a moment when every chain has completed at least `iters` transitions
static int report_progress(rmhmc_ctx* ctx, int event, long long iters) {
  std::vector<long long> acc(ctx->n), it(ctx->n);
  RC(download(ctx, acc.data(), ctx->ch.accepted, ctx->n));
  RC(download(ctx, it.data(), ctx->ch.iter, ctx->n));
  RC(sync(ctx));
  long long tot = 0, tit = 0;
  for (long long a : acc) tot += a;
  for (long long a : it) tit += a;
  ctx->progress_fn(event, iters, tot, tit, ctx->progress_user);
  return RMHMC_OK;
}
// Batches of several chains report without stopping anybody: whenever the host looks at the device anyway, the milestones the
// SLOWEST chain has passed since the last report are reported (once, with the largest of them).  Cutting a batch at every
// milestone, as the one-chain mode does to reproduce the reference's print-out exactly, would make every chain wait for the
// slowest fifty iterations at a time (+27 % at 8192 chains).
static bool progress_ticking(const rmhmc_ctx* ctx) { return ctx->progress_fn && ctx->progress_every > 0 && ctx->n > 1; }
static int progress_fire(rmhmc_ctx* ctx, long long min_iter) {
  if (!progress_ticking(ctx)) return RMHMC_OK;
  long long last = -1;
  while (ctx->progress_next <= min_iter) { last = ctx->progress_next; ctx->progress_next += ctx->progress_every; }
  if (last >= 0) RC(report_progress(ctx, RMHMC_EV_PROGRESS, last));
  return RMHMC_OK;
}

static int run_until_done(rmhmc_ctx* ctx, const IterBase& ib, long long min_steps) {
  // Every chain needs at least min_steps more global steps.  Afterwards the host looks at the device state and issues, each time,
  // as many steps as the slowest chain is certain to need: (limit - its completed transitions), a transition taking >= 1 step.
  // The remaining work shrinks geometrically (a transition averages (L+1)/2 steps), so a run costs O(log) host round trips
  // instead of one per 4 steps (thousands of syncs inside the TimeTaken window of the one-launch paths).
  int done = 0;
  long long s = min_steps, min_iter = 0;
  const long long poll = 4;
  const bool fused = ctx->fused && ctx->sampler == 0;
  const bool ticking = progress_ticking(ctx);
  const long long chunk = ticking ? std::max<long long>(8, ctx->progress_every * (ctx->L + 1) / 2) : min_steps;  // ~ one report per chunk
  StepGraph sg;
  if (!fused && step_graph_usable(ctx, std::min(min_steps, chunk))) (void)build_step_graph(ctx, ib, sg);
  for (long long left = min_steps; left > 0;) {
    const long long k = std::min(left, chunk);
    if (fused) launch_fused(ctx, ib, k);
    else run_generic_steps(ctx, ib, k, &sg);
    left -= k;
    if (ticking && left > 0) {
      RC(poll_progress(ctx, &done, &min_iter));
      RC(progress_fire(ctx, min_iter));
    }
  }
  for (;;) {
    RC(poll_progress(ctx, &done, &min_iter));
    RC(progress_fire(ctx, min_iter));
    if (done >= ctx->n) break;
    long long next = std::max(poll, ib.limit - min_iter);
    if (ticking) next = std::min(next, chunk);
    if (fused) launch_fused(ctx, ib, next);
    else run_generic_steps(ctx, ib, next, &sg);
    s += next;
    if (s > min_steps * (long long)ctx->L + 1000000) return fail(ctx, RMHMC_ERR_RUNTIME, "sampler did not terminate");
  }
  return RMHMC_OK;
}

// Every chain from exactly `from` to exactly ib.limit completed transitions.  With a progress callback and ONE chain the run is cut
// at the milestones first, first+every, ...: the callback gets the exact counters there, as the reference prints them, and the run goes
// on; the results do not depend on it (the randomness is keyed by chain and iteration).  Several chains: see progress_fire.
// A milestone that coincides with a phase boundary (burn_in + 1 completed transitions) is reported on the side of the burn-in banner
// where the reference prints it: rmhmc.py:38-45 prints at the TOP of the next iteration, i.e. after the banner of :194-196
// (at_from of phase B); hmc.py:85-94 prints at the bottom of the iteration itself, just before the banner (at_limit of phase A).
static int run_phase(rmhmc_ctx* ctx, const IterBase& ib, long long from, bool at_from = false, bool at_limit = false) {
  if (ctx->progress_fn && ctx->progress_every > 0 && ctx->n == 1) {
    long long m = ctx->progress_first;
    if (from > m) m += ((from - m + ctx->progress_every - 1) / ctx->progress_every) * ctx->progress_every;   // first milestone >= from
    if (m == from) {
      if (at_from) RC(report_progress(ctx, RMHMC_EV_PROGRESS, m));
      m += ctx->progress_every;
    }
    for (; m < ib.limit || (at_limit && m == ib.limit); m += ctx->progress_every) {
      IterBase seg = ib;
      seg.limit = m;
      HIPCK(hipMemsetAsync(ctx->d_done, 0, sizeof(int), ctx->stream));
      RC(run_until_done(ctx, seg, m - from));
      RC(report_progress(ctx, RMHMC_EV_PROGRESS, m));
      from = m;
    }
    HIPCK(hipMemsetAsync(ctx->d_done, 0, sizeof(int), ctx->stream));
  }
  if (ib.limit > from) RC(run_until_done(ctx, ib, ib.limit - from));
  return RMHMC_OK;
}

// One global step on the first na chains only (na a multiple of 128 or n).  The single group's view is narrowed for the launches of the
// step: every kernel takes its chain count and its partial-plane strides from the group, so producers and consumers agree.
static void launch_global_step_prefix(rmhmc_ctx* ctx, const IterBase& ib, int na) {
  Group& g = ctx->groups[0];
  const Group saved = g;
  g.n = na; g.ch.n = na;
  if (g.nCp) g.nCp = (na + I8_BM - 1) / I8_BM * I8_BM;
  // (row ranges, k-split planes and row splits stay as chosen for the whole batch: every chain's sums keep their order, so the
  // results are bit-identical to the unsorted run)
  launch_global_step(ctx, ib);
  g = saved;
}

// Work-sorted phase B.  The trajectory lengths do not depend on the state (RandomStep = ceil(rand() L), rmhmc.py:89), so the number of
// leapfrog steps T_c every chain needs for its post-burn-in transitions is known beforehand.  With the chains laid out in order of
// decreasing T_c, the chains still running after s global steps are a prefix of the batch, and the launches of the tail shrink with it:
// the phase costs sum_c T_c chain-steps instead of n max_c T_c (8192 chains, 199 transitions: 791 global steps for a mean of 696, i.e.
// 12 % of the TimeTaken window spent on finished chains).  Samples do not change: chains are independent and their randomness is keyed
// by the chain's index in the caller's order (IterParams::orig).
static int run_sorted_phase(rmhmc_ctx* ctx, const IterBase& ib, const std::vector<long long>& T /* descending */) {
  const int n = (int)ctx->n;
  const long long Tmin = T[n - 1], Tmax = T[0];
  const bool ticking = progress_ticking(ctx);
  const long long chunk = ticking ? std::max<long long>(8, ctx->progress_every * (ctx->L + 1) / 2) : Tmax + 1;
  int done = 0;
  long long mi = 0;
  {
    StepGraph sg;
    if (step_graph_usable(ctx, std::min(Tmin, chunk))) (void)build_step_graph(ctx, ib, sg);
    for (long long left = Tmin; left > 0;) {
      const long long k = std::min(left, chunk);
      run_generic_steps(ctx, ib, k, &sg);
      left -= k;
      if (ticking) { RC(poll_progress(ctx, &done, &mi)); RC(progress_fire(ctx, mi)); }
    }
    HIPCK(hipStreamSynchronize(ctx->stream));  // (the graph goes out of scope)
  }
  int na = n;
  for (long long s = Tmin; s < Tmax; ++s) {
    while (na > 0 && T[na - 1] <= s) --na;  // chains with T > s are still running: positions [0, na)
    const int nar = std::min(n, (na + 127) / 128 * 128);
    if (nar == n) launch_global_step(ctx, ib);
    else launch_global_step_prefix(ctx, ib, nar);
    flow_tick(ctx);
    if (ticking && (s - Tmin) % chunk == chunk - 1) { RC(poll_progress(ctx, &done, &mi)); RC(progress_fire(ctx, mi)); }
  }
  RC(poll_progress(ctx, &done, &mi));
  RC(progress_fire(ctx, mi));
  if (done < n) RC(run_until_done(ctx, ib, 1));  // (safety net: never taken if the schedule above is right)
  return RMHMC_OK;
}

// Runs the sampler; the saved states go to the device buffer d_samples ([n][S][D], caller-provided).  Per-chain counters are left
// on the device: ch.accepted, and the post-burn-in leapfrog steps in d_steps0 (steps_done at the end minus at the burn-in mark).
static int sample_core(rmhmc_ctx* ctx, int64_t n_iter, int64_t burn_in, const double* theta0, double* d_samples, double* seconds_out) {
  const long long S = n_iter - burn_in;
  const int n = (int)ctx->n;
  // work-sorted layout (see run_sorted_phase): generic and one-launch stepping paths of the RMHMC sampler, one chain group
  std::vector<long long> T;
  std::vector<double> th_perm;
  bool sorted = ctx->sampler == 0 && ctx->groups.size() == 1 && !ctx->fused && n >= 2 && n_iter > burn_in + 1;
  ctx->progress_next = ctx->progress_first;
  sorted = sorted && ctx->opt.sorted;
  if (sorted) {
    hipLaunchKernelGGL(k_traj_steps, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (unsigned long long)ctx->seed,
                       (long long)ctx->chain_offset, ctx->L, (long long)burn_in + 1, (long long)n_iter, (size_t)n, ctx->d_T);
    std::vector<long long> t0(n);
    RC(download(ctx, t0.data(), ctx->d_T, (size_t)n));
    RC(sync(ctx));
    std::vector<int> orig(n);
    for (int i = 0; i < n; ++i) orig[i] = i;
    std::stable_sort(orig.begin(), orig.end(), [&](int a, int b) { return t0[a] > t0[b]; });
    T.resize(n);
    for (int i = 0; i < n; ++i) T[i] = t0[orig[i]];
    RC(upload(ctx, ctx->d_orig, orig.data(), (size_t)n));
    RC(sync(ctx));
    th_perm.resize((size_t)n * ctx->D);
    for (int i = 0; i < n; ++i)
      for (int d = 0; d < ctx->D; ++d) th_perm[(size_t)i * ctx->D + d] = theta0 ? theta0[(size_t)orig[i] * ctx->D + d] : 1e-3;  // rmhmc.py:27
    theta0 = th_perm.data();
  }
  ctx->sorted = sorted;
  int rc = [&]() -> int {
    RC(init_chains(ctx, theta0));
    // phase A: every chain completes transitions 0..burn_in; chains that get there first wait, so that
    // the timed phase B covers exactly the post-burn-in transitions (TimeTaken, rmhmc.py:194-198)
    const IterBase ipA{burn_in + 1, burn_in, S, d_samples, false, true};
    RC(run_phase(ctx, ipA, 0));
    HIPCK(hipMemcpyAsync(ctx->d_steps0, ctx->ch.steps_done, sizeof(long long) * ctx->n, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCK(hipMemsetAsync(ctx->d_done, 0, sizeof(int), ctx->stream));
    RC(sync(ctx));
    if (ctx->progress_fn) RC(report_progress(ctx, RMHMC_EV_BURNIN_DONE, burn_in + 1));  // rmhmc.py:194-196: banner, then the timer starts
    const auto t0 = std::chrono::steady_clock::now();
    if (n_iter > burn_in + 1) {
      const IterBase ipB{n_iter, burn_in, S, d_samples, false, true};
      if (sorted) RC(run_sorted_phase(ctx, ipB, T));
      else RC(run_phase(ctx, ipB, burn_in + 1, true));
    }
    RC(sync(ctx));
    if (seconds_out) *seconds_out = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    hipLaunchKernelGGL(k_sub_ll, dim3((unsigned)((ctx->n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_steps0, ctx->ch.steps_done, (size_t)ctx->n);
    if (sorted) {  // counters back into the caller's order: d_T[0..n) accepted, d_T[n..2n) post-burn-in steps
      const dim3 grid((unsigned)((n + 255) / 256));
      hipLaunchKernelGGL(k_scatter_ll, grid, dim3(256), 0, ctx->stream, ctx->d_T, ctx->ch.accepted, ctx->d_orig, (size_t)n);
      hipLaunchKernelGGL(k_scatter_ll, grid, dim3(256), 0, ctx->stream, ctx->d_T + n, ctx->d_steps0, ctx->d_orig, (size_t)n);
    }
    return RMHMC_OK;
  }();
  ctx->sorted = false;
  ctx->counters_sorted = sorted && rc == RMHMC_OK;
  return rc;
}
// the counters of sample_core to host or device int64 arrays (either may be NULL)
static int sample_counters(rmhmc_ctx* ctx, int64_t* accept_out, int64_t* steps_out, hipMemcpyKind kind) {
  static_assert(sizeof(long long) == sizeof(int64_t), "int64");
  const long long* acc = ctx->counters_sorted ? ctx->d_T : ctx->ch.accepted;
  const long long* stp = ctx->counters_sorted ? ctx->d_T + ctx->n : ctx->d_steps0;
  if (accept_out) HIPCK(hipMemcpyAsync(accept_out, acc, sizeof(int64_t) * ctx->n, kind, ctx->stream));
  if (steps_out) HIPCK(hipMemcpyAsync(steps_out, stp, sizeof(int64_t) * ctx->n, kind, ctx->stream));
  return RMHMC_OK;
}

static int sample_impl(rmhmc_ctx* ctx, int64_t n_iter, int64_t burn_in, int32_t L, double eps, int32_t K, uint64_t seed, int64_t chain_offset,
                       const double* theta0, double* samples_out, int64_t* accept_out, int64_t* steps_out, double* seconds_out, bool dev) {
  NEED_DATA(ctx);
  if (!samples_out || burn_in < 0 || burn_in >= n_iter || L < 1 || K < 1)
    return fail(ctx, RMHMC_ERR_INVALID, "sample: need samples_out, 0 <= burn_in < n_iter, L >= 1, K >= 1");
  ctx->chains_ready = false;
  ctx->L = L; ctx->eps = eps; ctx->K = K; ctx->seed = seed; ctx->chain_offset = chain_offset;
  const size_t count = (size_t)ctx->n * (n_iter - burn_in) * ctx->D;
  double* d_samples = dev ? samples_out : nullptr;  // device-resident write-out: the sampler saves straight into the caller's HBM buffer
  if (!dev) HIPCK(hipMalloc((void**)&d_samples, sizeof(double) * count));
  int rc = [&]() -> int {
    RC(sample_core(ctx, n_iter, burn_in, theta0, d_samples, seconds_out));
    if (!dev) HIPCK(hipMemcpyAsync(samples_out, d_samples, sizeof(double) * count, hipMemcpyDeviceToHost, ctx->stream));
    RC(sample_counters(ctx, accept_out, steps_out, dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost));
    return sync(ctx);
  }();
  if (!dev && d_samples) (void)hipFree(d_samples);
  return rc;
}

int rmhmc_sample(rmhmc_ctx* ctx, int64_t n_iter, int64_t burn_in, int32_t L, double eps, int32_t K, uint64_t seed,
                 int64_t chain_offset, const double* theta0, double* samples_out, int64_t* accept_out, int64_t* steps_out,
                 double* seconds_out) {
  return sample_impl(ctx, n_iter, burn_in, L, eps, K, seed, chain_offset, theta0, samples_out, accept_out, steps_out, seconds_out, false);
}
int rmhmc_sample_dev(rmhmc_ctx* ctx, int64_t n_iter, int64_t burn_in, int32_t L, double eps, int32_t K, uint64_t seed,
                     int64_t chain_offset, const double* theta0, double* samples_dev, int64_t* accept_dev, int64_t* steps_dev,
                     double* seconds_out) {
  return sample_impl(ctx, n_iter, burn_in, L, eps, K, seed, chain_offset, theta0, samples_dev, accept_dev, steps_dev, seconds_out, true);
}

// ---- ESS / posterior summaries on the device (tools.py:32-74) -------------------------------------------
static int launch_ess(rmhmc_ctx* ctx, const double* d_samples, long long nblocks, long long S, int P, double* d_ess, double* d_mean, double* d_var) {
  if (S < 2 || S > 20000) return fail(ctx, RMHMC_ERR_UNSUPPORTED, "ess: 2 <= S <= 20000 samples per chain (the centred series is held in LDS)");
  HIPCK(hipFuncSetAttribute((const void*)k_ess, hipFuncAttributeMaxDynamicSharedMemorySize, MAX_LDS));  // per device: set where it is used
  long long nfft = 0;
  if (ctx->flags & RMHMC_FLAG_ESS_WRAP) { nfft = 1; while (nfft < S) nfft *= 2; nfft += 1; }  // tools.py:16-23
  hipLaunchKernelGGL(k_ess, dim3((unsigned)(nblocks * P)), dim3(64), (size_t)S * sizeof(double), ctx->stream, d_samples, S, P, d_ess, d_mean, d_var, nfft);
  return RMHMC_OK;
}

int rmhmc_ess(rmhmc_ctx* ctx, const double* samples, int64_t n, int64_t S, int32_t P, double* ess_out) {
  if (!ctx || !samples || !ess_out || n < 1 || P < 1 || n * (int64_t)P > 0x7fffffffLL) return fail(ctx, RMHMC_ERR_INVALID, "ess: bad argument");
  HIPCK(hipSetDevice(ctx->device));
  double *d_s = nullptr, *d_e = nullptr;
  HIPCK(hipMalloc((void**)&d_s, sizeof(double) * (size_t)n * S * P));
  int rc = [&]() -> int {
    HIPCK(hipMalloc((void**)&d_e, sizeof(double) * (size_t)n * P));
    HIPCK(hipMemcpyAsync(d_s, samples, sizeof(double) * (size_t)n * S * P, hipMemcpyHostToDevice, ctx->stream));
    RC(launch_ess(ctx, d_s, n, S, P, d_e, nullptr, nullptr));
    HIPCK(hipMemcpyAsync(ess_out, d_e, sizeof(double) * (size_t)n * P, hipMemcpyDeviceToHost, ctx->stream));
    return sync(ctx);
  }();
  (void)hipFree(d_s);
  if (d_e) (void)hipFree(d_e);
  return rc;
}

static int sample_stats_impl(rmhmc_ctx* ctx, int64_t n_iter, int64_t burn_in, int32_t L, double eps, int32_t K, uint64_t seed,
                             int64_t chain_offset, const double* theta0, double* mean_out, double* var_out, double* ess_out,
                             int64_t* accept_out, int64_t* steps_out, double* seconds_out, bool dev) {
  NEED_DATA(ctx);
  if (burn_in < 0 || burn_in >= n_iter || L < 1 || K < 1)
    return fail(ctx, RMHMC_ERR_INVALID, "sample_stats: need 0 <= burn_in < n_iter, L >= 1, K >= 1");
  ctx->chains_ready = false;
  ctx->L = L; ctx->eps = eps; ctx->K = K; ctx->seed = seed; ctx->chain_offset = chain_offset;
  const long long S = n_iter - burn_in;
  double* d_samples = nullptr;
  double* d_out = nullptr;  // [3][n][D]: ess, mean, var
  HIPCK(hipMalloc((void**)&d_samples, sizeof(double) * (size_t)ctx->n * S * ctx->D));
  int rc = [&]() -> int {
    RC(sample_core(ctx, n_iter, burn_in, theta0, d_samples, seconds_out));
    const size_t nd = (size_t)ctx->n * ctx->D;
    HIPCK(hipMalloc((void**)&d_out, sizeof(double) * 3 * nd));
    RC(launch_ess(ctx, d_samples, ctx->n, S, ctx->D, d_out, d_out + nd, d_out + 2 * nd));
    const hipMemcpyKind kind = dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    if (ess_out) HIPCK(hipMemcpyAsync(ess_out, d_out, sizeof(double) * nd, kind, ctx->stream));
    if (mean_out) HIPCK(hipMemcpyAsync(mean_out, d_out + nd, sizeof(double) * nd, kind, ctx->stream));
    if (var_out) HIPCK(hipMemcpyAsync(var_out, d_out + 2 * nd, sizeof(double) * nd, kind, ctx->stream));
    RC(sample_counters(ctx, accept_out, steps_out, kind));
    return sync(ctx);
  }();
  if (d_samples) (void)hipFree(d_samples);
  if (d_out) (void)hipFree(d_out);
  return rc;
}
int rmhmc_sample_stats(rmhmc_ctx* ctx, int64_t n_iter, int64_t burn_in, int32_t L, double eps, int32_t K, uint64_t seed,
                       int64_t chain_offset, const double* theta0, double* mean_out, double* var_out, double* ess_out,
                       int64_t* accept_out, int64_t* steps_out, double* seconds_out) {
  return sample_stats_impl(ctx, n_iter, burn_in, L, eps, K, seed, chain_offset, theta0, mean_out, var_out, ess_out, accept_out, steps_out,
                           seconds_out, false);
}
int rmhmc_sample_stats_dev(rmhmc_ctx* ctx, int64_t n_iter, int64_t burn_in, int32_t L, double eps, int32_t K, uint64_t seed,
                           int64_t chain_offset, const double* theta0, double* mean_dev, double* var_dev, double* ess_dev,
                           int64_t* accept_dev, int64_t* steps_dev, double* seconds_out) {
  return sample_stats_impl(ctx, n_iter, burn_in, L, eps, K, seed, chain_offset, theta0, mean_dev, var_dev, ess_dev, accept_dev, steps_dev,
                           seconds_out, true);
}

// ---- plain HMC (code/hmc.py) -------------------------------------------------------------------------
// evaluate gradient and log joint at theta0 and commit them as the current point of every chain
static int hmc_init_chains(rmhmc_ctx* ctx, const double* theta0_host /* [n][D] or NULL: zeros, hmc.py:27 */) {
  std::vector<double> th;
  if (!theta0_host) { th.assign((size_t)ctx->n * ctx->D, 0.0); theta0_host = th.data(); }
  RC(upload_vec(ctx, ctx->ch.trj.w, theta0_host));
  fill_int(ctx, ctx->ch.phase, 1, ctx->n);
  if (ctx->hmc_traj) {
    for (Group& g : ctx->groups) launch_hmc_traj(ctx, g, 1);
  } else {
    for (Group& g : ctx->groups) launch_rowpass<RP_G>(ctx, g, g.ch.trj.w, nullptr);
    for (Group& g : ctx->groups) SMALL(ctx, g, "small", k_hmc_init, ctx->dd, g.ch, g.nsplit);
  }
  fill_int(ctx, ctx->ch.phase, 0, ctx->n);
  fill_int(ctx, ctx->ch.steps_left, 0, ctx->n);
  fill_int(ctx, ctx->ch.status, 0, ctx->n);
  fill_ll(ctx, ctx->ch.iter, 0, ctx->n);
  fill_ll(ctx, ctx->ch.accepted, 0, ctx->n);
  fill_ll(ctx, ctx->ch.steps_done, 0, ctx->n);
  HIPCK(hipMemsetAsync(ctx->d_done, 0, sizeof(int), ctx->stream));
  return sync(ctx);
}

int rmhmc_hmc_transition(rmhmc_ctx* ctx, double* w, const double* z, const double* u_len, const double* u_acc, int32_t L, double eps,
                         int32_t* accepted_out, int32_t* nsteps_out, double* H_cur_out, double* H_prop_out, double* w_prop_out,
                         double* p_prop_out) {
  NEED_DATA(ctx);
  if (!w || !z || !u_len || !u_acc || L < 1) return fail(ctx, RMHMC_ERR_INVALID, "hmc_transition: null pointer or L < 1");
  ctx->chains_ready = false;
  ctx->L = L; ctx->eps = eps; ctx->sampler = 1;
  int rc = [&]() -> int {
    RC(hmc_init_chains(ctx, w));
    RC(upload(ctx, ctx->d_z, z, (size_t)ctx->n * ctx->D));
    RC(upload(ctx, ctx->d_ulen, u_len, ctx->n));
    RC(upload(ctx, ctx->d_uacc, u_acc, ctx->n));
    const IterBase ib{1, 0, 0, nullptr, true, false};
    for (int s = 0; s < L; ++s) launch_hmc_global_step(ctx, ib);
    std::vector<long long> acc(ctx->n);
    RC(download_vec(ctx, w, ctx->ch.cur.w));
    RC(download(ctx, acc.data(), ctx->ch.accepted, ctx->n));
    if (nsteps_out) RC(download(ctx, nsteps_out, ctx->ch.nsteps_last, ctx->n));
    if (H_cur_out) RC(download(ctx, H_cur_out, ctx->ch.Hcur, ctx->n));
    if (H_prop_out) RC(download(ctx, H_prop_out, ctx->ch.Hprop, ctx->n));
    if (w_prop_out) RC(download_vec(ctx, w_prop_out, ctx->ch.trj.w));
    if (p_prop_out) RC(download_vec(ctx, p_prop_out, ctx->ch.p));
    RC(sync(ctx));
    if (accepted_out) for (int64_t c = 0; c < ctx->n; ++c) accepted_out[c] = (int32_t)acc[c];
    return RMHMC_OK;
  }();
  ctx->sampler = 0;
  return rc;
}

int rmhmc_hmc_sample(rmhmc_ctx* ctx, int64_t n_iter, int64_t burn_in, int32_t L, double eps, uint64_t seed, int64_t chain_offset,
                     const double* theta0, double* samples_out, int64_t* accept_out, int64_t* steps_out, double* seconds_out) {
  NEED_DATA(ctx);
  if (!samples_out || burn_in < 0 || burn_in >= n_iter || L < 1)
    return fail(ctx, RMHMC_ERR_INVALID, "hmc_sample: need samples_out, 0 <= burn_in < n_iter, L >= 1");
  ctx->chains_ready = false;
  ctx->L = L; ctx->eps = eps; ctx->seed = seed; ctx->chain_offset = chain_offset;
  const long long S = n_iter - burn_in;
  double* d_samples = nullptr;
  HIPCK(hipMalloc((void**)&d_samples, sizeof(double) * (size_t)ctx->n * S * ctx->D));
  ctx->sampler = 1;  // (after the allocation: an early return above must not leave the context in HMC mode)
  ctx->progress_next = ctx->progress_first;
  int rc = [&]() -> int {
    RC(hmc_init_chains(ctx, theta0));
    const IterBase ipA{burn_in + 1, burn_in, S, d_samples, false, true};
    RC(run_phase(ctx, ipA, 0, false, true));
    HIPCK(hipMemcpyAsync(ctx->d_steps0, ctx->ch.steps_done, sizeof(long long) * ctx->n, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCK(hipMemsetAsync(ctx->d_done, 0, sizeof(int), ctx->stream));
    RC(sync(ctx));
    if (ctx->progress_fn) RC(report_progress(ctx, RMHMC_EV_BURNIN_DONE, burn_in + 1));  // hmc.py:92-94
    const auto t0 = std::chrono::steady_clock::now();
    if (n_iter > burn_in + 1) {
      // hmc.py:83-89 reports during burn-in only (`elif` of the save branch): no milestones, hence no cuts, inside TimeTaken
      const rmhmc_progress_fn fn = ctx->progress_fn;
      ctx->progress_fn = nullptr;
      const IterBase ipB{n_iter, burn_in, S, d_samples, false, true};
      const int rcB = run_phase(ctx, ipB, burn_in + 1);
      ctx->progress_fn = fn;
      RC(rcB);
    }
    RC(sync(ctx));
    if (seconds_out) *seconds_out = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    HIPCK(hipMemcpyAsync(samples_out, d_samples, sizeof(double) * (size_t)ctx->n * S * ctx->D, hipMemcpyDeviceToHost, ctx->stream));
    std::vector<long long> a(ctx->n), s1(ctx->n), s0(ctx->n);
    RC(download(ctx, a.data(), ctx->ch.accepted, ctx->n));
    RC(download(ctx, s1.data(), ctx->ch.steps_done, ctx->n));
    RC(download(ctx, s0.data(), ctx->d_steps0, ctx->n));
    RC(sync(ctx));
    for (int64_t c = 0; c < ctx->n; ++c) {
      if (accept_out) accept_out[c] = a[c];
      if (steps_out) steps_out[c] = s1[c] - s0[c];
    }
    return RMHMC_OK;
  }();
  (void)hipFree(d_samples);
  ctx->sampler = 0;
  return rc;
}

// ---- simplified manifold MALA (BLR_mMALA_Simp.m) -------------------------------------------------------------
static void launch_mmala_step(rmhmc_ctx* ctx, const IterBase& b) {
  std::vector<Phase> ph;
  const double eps = ctx->eps;
  const int full = (ctx->flags & RMHMC_FLAG_MMALA_FULL) ? 1 : 0;
  ph.push_back([ctx, b, eps, full](Group& g) { IterParams ip = iter_params(ctx, g, b); SMALL(ctx, g, "small", k_mmala_begin, ctx->D, ctx->DP, g.ch, ip, eps, full); });
  eval_point_phases(ctx, ph, false, full ? 2 : 1);
  ph.push_back([ctx, b, eps, full](Group& g) { IterParams ip = iter_params(ctx, g, b); SMALL(ctx, g, "small", k_mmala_end, ctx->D, ctx->DP, g.ch, ip, eps, full); });
  run_phases(ctx, ph);
}
// record at theta0 (generic kernels; mMALA never uses the fused stepping kernel)
static int mmala_init(rmhmc_ctx* ctx, const double* theta0_host) {
  std::vector<double> th;
  if (!theta0_host) { th.assign((size_t)ctx->n * ctx->D, 0.0); theta0_host = th.data(); }  // BLR_mMALA_Simp.m:175
  const bool fused = ctx->fused;
  ctx->fused = false;
  const int rc = init_chains(ctx, theta0_host);
  ctx->fused = fused;
  return rc;
}

int rmhmc_mmala_transition(rmhmc_ctx* ctx, double* w, const double* z, const double* u_acc, double eps, int32_t* accepted_out,
                           double* ratio_out, double* w_prop_out) {
  NEED_DATA(ctx);
  if (!w || !z || !u_acc || !(eps > 0)) return fail(ctx, RMHMC_ERR_INVALID, "mmala_transition: null pointer or eps <= 0");
  ctx->chains_ready = false;
  ctx->eps = eps;
  RC(mmala_init(ctx, w));
  RC(upload(ctx, ctx->d_z, z, (size_t)ctx->n * ctx->D));
  RC(upload(ctx, ctx->d_uacc, u_acc, ctx->n));
  const IterBase ib{1, 0, 0, nullptr, true, false};
  launch_mmala_step(ctx, ib);
  std::vector<long long> acc(ctx->n);
  RC(download_vec(ctx, w, ctx->ch.cur.w));
  RC(download(ctx, acc.data(), ctx->ch.accepted, ctx->n));
  if (ratio_out) RC(download(ctx, ratio_out, ctx->ch.Hprop, ctx->n));
  if (w_prop_out) RC(download_vec(ctx, w_prop_out, ctx->ch.trj.w));
  RC(sync(ctx));
  if (accepted_out) for (int64_t c = 0; c < ctx->n; ++c) accepted_out[c] = (int32_t)acc[c];
  return RMHMC_OK;
}

int rmhmc_mmala_sample(rmhmc_ctx* ctx, int64_t n_iter, int64_t burn_in, double eps, uint64_t seed, int64_t chain_offset,
                       const double* theta0, double* samples_out, int64_t* accept_out, double* seconds_out) {
  NEED_DATA(ctx);
  if (!samples_out || burn_in < 0 || burn_in >= n_iter || !(eps > 0))
    return fail(ctx, RMHMC_ERR_INVALID, "mmala_sample: need samples_out, 0 <= burn_in < n_iter, eps > 0");
  ctx->chains_ready = false;
  ctx->eps = eps; ctx->seed = seed; ctx->chain_offset = chain_offset;
  const long long S = n_iter - burn_in;
  double* d_samples = nullptr;
  HIPCK(hipMalloc((void**)&d_samples, sizeof(double) * (size_t)ctx->n * S * ctx->D));
  int rc = [&]() -> int {
    RC(mmala_init(ctx, theta0));
    const IterBase ib{n_iter, burn_in, S, d_samples, false, false};
    for (int64_t it = 0; it <= burn_in; ++it) { launch_mmala_step(ctx, ib); flow_tick(ctx); }
    RC(sync(ctx));
    const auto t0 = std::chrono::steady_clock::now();
    for (int64_t it = burn_in + 1; it < n_iter; ++it) { launch_mmala_step(ctx, ib); flow_tick(ctx); }
    RC(sync(ctx));
    if (seconds_out) *seconds_out = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    HIPCK(hipMemcpyAsync(samples_out, d_samples, sizeof(double) * (size_t)ctx->n * S * ctx->D, hipMemcpyDeviceToHost, ctx->stream));
    if (accept_out) {
      static_assert(sizeof(long long) == sizeof(int64_t), "int64");
      RC(download(ctx, (long long*)accept_out, ctx->ch.accepted, ctx->n));
    }
    return sync(ctx);
  }();
  (void)hipFree(d_samples);
  return rc;
}

int rmhmc_chains_init(rmhmc_ctx* ctx, const double* theta0, uint64_t seed, int64_t chain_offset, int32_t L, double eps, int32_t K) {
  NEED_DATA(ctx);
  if (L < 1 || K < 1) return fail(ctx, RMHMC_ERR_INVALID, "chains_init: L >= 1 and K >= 1 required");
  ctx->L = L; ctx->eps = eps; ctx->K = K; ctx->seed = seed; ctx->chain_offset = chain_offset;
  RC(init_chains(ctx, theta0));
  ctx->chains_ready = true;
  return RMHMC_OK;
}

int rmhmc_chains_run(rmhmc_ctx* ctx, int64_t n_steps) {
  NEED_DATA(ctx);
  if (!ctx->chains_ready) return fail(ctx, RMHMC_ERR_INVALID, "chains_run: rmhmc_chains_init has not been called");
  const IterBase ib{(long long)1 << 62, 0, 0, nullptr, false, false};
  {
    Timed t(ctx, "total", ctx->stream);
    if (ctx->fused) {
      launch_fused(ctx, ib, n_steps);
    } else {
      StepGraph sg;
      if (step_graph_usable(ctx, n_steps)) (void)build_step_graph(ctx, ib, sg);
      run_generic_steps(ctx, ib, n_steps, &sg);
      HIPCK(hipStreamSynchronize(ctx->stream));  // the graph is destroyed at the end of this scope
    }
  }
  return sync(ctx);
}

static int chains_state_impl(rmhmc_ctx* ctx, double* w_out, int64_t* iters_out, int64_t* accept_out, bool dev) {
  NEED_DATA(ctx);
  if (!ctx->chains_ready) return fail(ctx, RMHMC_ERR_INVALID, "chains_state: rmhmc_chains_init has not been called");
  const hipMemcpyKind kind = dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  if (w_out)  // strip the padding: [n][DP] -> [n][D]
    HIPCK(hipMemcpy2DAsync(w_out, ctx->D * sizeof(double), ctx->ch.cur.w, ctx->DP * sizeof(double), ctx->D * sizeof(double), ctx->n, kind, ctx->stream));
  static_assert(sizeof(long long) == sizeof(int64_t), "int64");
  if (iters_out) HIPCK(hipMemcpyAsync(iters_out, ctx->ch.iter, sizeof(int64_t) * ctx->n, kind, ctx->stream));
  if (accept_out) HIPCK(hipMemcpyAsync(accept_out, ctx->ch.accepted, sizeof(int64_t) * ctx->n, kind, ctx->stream));
  return sync(ctx);
}
int rmhmc_chains_state(rmhmc_ctx* ctx, double* w_out, int64_t* iters_out, int64_t* accept_out) {
  return chains_state_impl(ctx, w_out, iters_out, accept_out, false);
}
int rmhmc_chains_state_dev(rmhmc_ctx* ctx, double* w_dev, int64_t* iters_dev, int64_t* accept_dev) {
  return chains_state_impl(ctx, w_dev, iters_dev, accept_dev, true);
}

int rmhmc_chains_restore(rmhmc_ctx* ctx, const int64_t* iters, const int64_t* accepted) {
  NEED_DATA(ctx);
  if (!ctx->chains_ready || !iters || !accepted) return fail(ctx, RMHMC_ERR_INVALID, "chains_restore: call rmhmc_chains_init first");
  for (int64_t c = 0; c < ctx->n; ++c)
    if (iters[c] < 0 || accepted[c] < 0) return fail(ctx, RMHMC_ERR_INVALID, "chains_restore: negative counter");
  RC(upload(ctx, ctx->ch.iter, (const long long*)iters, ctx->n));
  RC(upload(ctx, ctx->ch.accepted, (const long long*)accepted, ctx->n));
  return sync(ctx);
}

int rmhmc_set_progress(rmhmc_ctx* ctx, rmhmc_progress_fn fn, int64_t first, int64_t every, void* user) {
  if (!ctx || (fn && (first < 1 || every < 1))) return fail(ctx, RMHMC_ERR_INVALID, "set_progress: first >= 1 and every >= 1 required");
  ctx->progress_fn = fn; ctx->progress_first = first; ctx->progress_every = every; ctx->progress_user = user;
  return RMHMC_OK;
}

int rmhmc_int8_certificate(rmhmc_ctx* ctx, double* bound_out, int32_t* active_out) {
  if (!ctx) return fail(nullptr, RMHMC_ERR_INVALID, "int8_certificate: null context");
  if (bound_out) *bound_out = ctx->i8_bound;
  if (active_out) *active_out = (ctx->i8_requested && ctx->i8) ? 1 : 0;
  return RMHMC_OK;
}

int rmhmc_kernel_time(rmhmc_ctx* ctx, const char* which, double* seconds_out, int64_t* launches_out) {
  if (!ctx || !which) return fail(ctx, RMHMC_ERR_INVALID, "kernel_time: bad argument");
  HIPCK(hipSetDevice(ctx->device));
  const std::string w(which);
  if (seconds_out) *seconds_out = 0.0;
  if (launches_out) *launches_out = 0;
  if (w == "enable") { ctx->timing = true; return RMHMC_OK; }
  if (w == "disable") { ctx->timing = false; return RMHMC_OK; }
  if (w == "reset") {
    RC(sync(ctx));
    for (auto& kv : ctx->events) { for (auto& e : kv.second) ctx->pool.push_back(e); kv.second.clear(); }
    return RMHMC_OK;
  }
  RC(sync(ctx));
  auto it = ctx->events.find(w);
  if (it == ctx->events.end()) return RMHMC_OK;
  double ms = 0.0;
  for (auto& e : it->second) {
    float m = 0.f;
    HIPCK(hipEventElapsedTime(&m, e.a, e.b));
    ms += m;
  }
  if (seconds_out) *seconds_out = ms * 1e-3;
  if (launches_out) *launches_out = (int64_t)it->second.size();
  return RMHMC_OK;
}

}  // extern "C"
