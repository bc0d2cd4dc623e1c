// rocprof_queue_repro - a stand-alone HIP program (no code of this library) that enqueues work the way librmhmc_hip's bulk sampler
// does, to find out what makes `rocprofv3 --kernel-trace` (librocprofiler-sdk 7.2) fault in its queue interceptor.
//   rocprof_queue_repro <mode> <kernel_us> <steps> <nodes_per_step> [inflight]
//     mode direct       : steps x nodes kernel launches + a 4-byte memset every 6th node, no host synchronisation in between
//     mode graph        : the same step captured once into a hipGraph and replayed `steps` times
//     mode graph+direct : graph replays, hipStreamSynchronize, graph destroyed, then steps/4 steps of direct launches (the sampler's
//                         phase schedule: whole-batch graph replays followed by direct launches of the shrinking prefix)
//     inflight          : at most that many steps queued ahead of the device (0 = unbounded, the library's behaviour up to round 2)
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/rocprof_queue_repro tools/rocprof_queue_repro.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

struct BigArgs { double* p[40]; int n[24]; };   // ~420 bytes of by-value arguments, like the library's Chains struct

__global__ void k_spin(BigArgs a, long long ticks, int* sink) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) { }
  if (a.n[0] == 12345 && sink) sink[threadIdx.x] = (int)ticks;
}

static void one_step(hipStream_t st, const BigArgs& a, long long ticks, int nodes, int* d_flag, int* sink) {
  for (int k = 0; k < nodes; ++k) {
    if (k % 6 == 0) (void)hipMemsetAsync(d_flag, 0, sizeof(int), st);
    hipLaunchKernelGGL(k_spin, dim3(64), dim3(64), (k % 5 == 0) ? 4096 : 0, st, a, ticks, sink);
  }
}

int main(int argc, char** argv) {
  if (argc < 5) { fprintf(stderr, "usage: %s direct|graph|graph+direct kernel_us steps nodes [inflight]\n", argv[0]); return 1; }
  const char* mode = argv[1];
  const long long ticks = atoll(argv[2]) * 100;   // wall_clock64 runs at 100 MHz
  const int steps = atoi(argv[3]), nodes = atoi(argv[4]), inflight = argc > 5 ? atoi(argv[5]) : 0;
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  int *d_flag, *sink;
  CK(hipMalloc(&d_flag, 64)); CK(hipMalloc(&sink, 4096));
  BigArgs a{};
  std::vector<hipEvent_t> ev(4);
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  long long ticks_done = 0, since = 0;
  auto flow = [&]() {
    if (inflight <= 0) return;
    const int stride = inflight / 4 > 0 ? inflight / 4 : 1;
    if (++since < stride) return;
    since = 0;
    hipEvent_t e = ev[ticks_done % 4];
    if (ticks_done >= 4) (void)hipEventSynchronize(e);
    (void)hipEventRecord(e, st);
    ++ticks_done;
  };
  const bool graph = !strncmp(mode, "graph", 5), tail = !strcmp(mode, "graph+direct");
  if (graph) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    one_step(st, a, ticks, nodes, d_flag, sink);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int s = 0; s < steps; ++s) { CK(hipGraphLaunch(ge, st)); flow(); }
    CK(hipStreamSynchronize(st));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    fprintf(stderr, "graph phase done\n");
  }
  if (!graph || tail) {
    const int n = graph ? steps / 4 : steps;
    for (int s = 0; s < n; ++s) { one_step(st, a, ticks, nodes, d_flag, sink); flow(); }
    CK(hipStreamSynchronize(st));
    fprintf(stderr, "direct phase done\n");
  }
  CK(hipDeviceSynchronize());
  printf("ok %s: %d steps x %d nodes, inflight %d\n", mode, steps, nodes, inflight);
  return 0;
}
