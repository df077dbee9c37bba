// What does ONE dependent kernel boundary cost on this box, and what does it depend on?
//   hipcc --offload-arch=gfx950 -O3 -o boundary boundary.hip && ./boundary
// A chain of N launches on one stream (every launch depends on the one before it: same stream, in order), issued three ways --
//   eager   : hipLaunchKernelGGL from a C loop
//   capture : hipStreamBeginCapture -> N launches -> hipGraphInstantiate -> hipGraphLaunch
//   explicit: hipGraphAddKernelNode with an explicit dependency on the previous node
// for kernels that differ in ONE property each: grid, block, LDS, kernel-argument bytes, bytes streamed, bytes left dirty.
// Host wall clock around R replays of the chain (sync on both sides) / (R N) = microseconds per node.  A node's figure minus the
// kernel's own body (the `stream` kernels: bytes / 6 TB/s) is the boundary.  MI355X_MICROARCH.md "boundary" row measures 1.45 us
// between trivial 256-workgroup kernels and 1.7-1.9 us between streaming kernels; rocprofv3 inside this repo's replayed training
// step reads 4.7-5.6 us for its trivial nodes (profiles/r05_step_sequence.txt).  This program is the reconciliation.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

struct Big { int v[256]; };           // 1 KiB of kernel arguments by value
struct Mid { int v[64]; };            // 256 B

__global__ void k_empty() {}
__global__ void k_empty_mid(Mid a, int* sink) { if (a.v[threadIdx.x & 63] == 0x7fffffff) sink[0] = 1; }
__global__ void k_empty_big(Big a, int* sink) { if (a.v[threadIdx.x & 255] == 0x7fffffff) sink[0] = 1; }
__global__ void k_lds(int* sink) {
  extern __shared__ int lds[];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  if (lds[(threadIdx.x + 1) % blockDim.x] == 0x7fffffff) sink[0] = 1;
}
// out = in + 1 over n float4: reads n*16 B, writes n*16 B; ping-pong between two buffers makes node i read what node i-1 wrote
__global__ void k_stream(const float4* __restrict__ in, float4* __restrict__ out, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float4 v = in[i];
    v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f;
    out[i] = v;
  }
}
// read-only: nothing dirty at the end
__global__ void k_read(const float4* __restrict__ in, long n, int* sink) {
  float s = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) { float4 v = in[i]; s += v.x + v.y + v.z + v.w; }
  if (s == 1.2345e30f) sink[0] = 1;
}
// a latency chain like the fused launches' prologues: one workgroup per CU, D dependent loads each
__global__ void k_chase(const int* __restrict__ next, int depth, int* sink) {
  int p = blockIdx.x * 64;
  for (int d = 0; d < depth; ++d) p = next[p];
  if (p == 0x7fffffff) sink[0] = 1;
}

// occupies its workgroups for `ticks` of the 100 MHz constant clock (1 tick = 10 ns): a stand-in for a latency-bound kernel
__global__ void k_spin(long ticks, int* sink) {
  const long t0 = (long)__builtin_amdgcn_s_memrealtime();
  while ((long)__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (ticks < 0) sink[0] = 1;
}

using Launch = std::function<void(hipStream_t, int)>;     // (stream, index of the node in the chain)

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static double time_eager(const Launch& f, int N, int R, hipStream_t s) {
  for (int i = 0; i < N; ++i) f(s, i);
  CK(hipStreamSynchronize(s));
  double best = 1e30;
  for (int r = 0; r < R; ++r) {
    double t0 = now_us();
    for (int i = 0; i < N; ++i) f(s, i);
    CK(hipStreamSynchronize(s));
    double t = (now_us() - t0) / N;
    if (t < best) best = t;
  }
  return best;
}

static double time_graph(hipGraphExec_t ex, int N, int R, hipStream_t s, double* ev_us) {
  for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ex, s));
  CK(hipStreamSynchronize(s));
  // (a) R replays back to back, one sync: the per-replay host cost is amortised by the queue
  double t0 = now_us();
  for (int r = 0; r < R; ++r) CK(hipGraphLaunch(ex, s));
  CK(hipStreamSynchronize(s));
  double t = (now_us() - t0) / ((double)R * N);
  // (b) device events around the same
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, s));
  for (int r = 0; r < R; ++r) CK(hipGraphLaunch(ex, s));
  CK(hipEventRecord(e1, s));
  CK(hipStreamSynchronize(s));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  *ev_us = ms * 1e3 / ((double)R * N);
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return t;
}

static hipGraphExec_t capture(const Launch& f, int N, hipStream_t s) {
  hipGraph_t g;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < N; ++i) f(s, i);
  CK(hipStreamEndCapture(s, &g));
  hipGraphExec_t ex;
  CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
  CK(hipGraphDestroy(g));
  return ex;
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 200, R = argc > 2 ? atoi(argv[2]) : 20;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("# device %s, %d CUs; chain of N = %d nodes, R = %d replays; us per node\n", prop.name, prop.multiProcessorCount, N, R);
  const char* envs[] = {"HIP_FORCE_DEV_KERNARG", "DEBUG_CLR_GRAPH_PACKET_CAPTURE", "DEBUG_HIP_GRAPH_BATCH_SIZE", "AMD_OPT_FLUSH", "GPU_MAX_HW_QUEUES", "DEBUG_HIP_FORCE_GRAPH_QUEUES", "AMD_DIRECT_DISPATCH"};
  for (const char* e : envs) if (getenv(e)) printf("# %s=%s\n", e, getenv(e));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const long NB = 64l << 20;                                     // two 64 MiB buffers
  float4 *a, *b; int* sink; int* next;
  CK(hipMalloc(&a, NB)); CK(hipMalloc(&b, NB)); CK(hipMalloc(&sink, 4096)); CK(hipMalloc(&next, 1 << 26));
  CK(hipMemset(a, 0, NB)); CK(hipMemset(b, 0, NB)); CK(hipMemset(sink, 0, 4096));
  {
    std::vector<int> h((1 << 26) / 4);
    unsigned x = 12345;
    for (size_t i = 0; i < h.size(); ++i) { x = x * 1664525u + 1013904223u; h[i] = (int)((x >> 4) % h.size()) & ~15; }
    CK(hipMemcpy(next, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  }
  Big big; memset(&big, 0, sizeof(big));
  Mid mid; memset(&mid, 0, sizeof(mid));
  CK(hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));

  struct Case { std::string name; Launch f; double body_us; };
  std::vector<Case> cases;
  auto add = [&](const std::string& n, Launch f, double body = 0.) { cases.push_back({n, f, body}); };
  for (int g : {1, 256, 1024, 4096})
    add("empty grid " + std::to_string(g) + " x 256", [=](hipStream_t st, int) { hipLaunchKernelGGL(k_empty, dim3(g), dim3(256), 0, st); });
  add("empty grid 256 x 512", [=](hipStream_t st, int) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(512), 0, st); });
  add("empty grid 256 x 1024", [=](hipStream_t st, int) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(1024), 0, st); });
  add("kernarg 256 B, grid 256 x 256", [=](hipStream_t st, int) { hipLaunchKernelGGL(k_empty_mid, dim3(256), dim3(256), 0, st, mid, sink); });
  add("kernarg 1 KiB, grid 256 x 256", [=](hipStream_t st, int) { hipLaunchKernelGGL(k_empty_big, dim3(256), dim3(256), 0, st, big, sink); });
  add("LDS 64 KiB, grid 256 x 512", [=](hipStream_t st, int) { hipLaunchKernelGGL(k_lds, dim3(256), dim3(512), 64 * 1024, st, sink); });
  add("LDS 160 KiB, grid 256 x 512", [=](hipStream_t st, int) { hipLaunchKernelGGL(k_lds, dim3(256), dim3(512), 160 * 1024, st, sink); });
  for (long mb : {1l, 4l, 16l, 64l}) {
    const long n = (mb << 20) / 16;
    const double body = 2. * (double)(mb << 20) / 6.0e6;          // read + write at 6 TB/s, us
    add("stream r+w " + std::to_string(mb) + " MiB each, ping-pong, grid 1024 x 256",
        [=](hipStream_t st, int i) { hipLaunchKernelGGL(k_stream, dim3(1024), dim3(256), 0, st, (i & 1) ? b : a, (i & 1) ? a : b, n); }, body);
  }
  for (long mb : {4l, 64l}) {
    const long n = (mb << 20) / 16;
    add("read-only " + std::to_string(mb) + " MiB, grid 1024 x 256", [=](hipStream_t st, int) { hipLaunchKernelGGL(k_read, dim3(1024), dim3(256), 0, st, a, n, sink); },
        (double)(mb << 20) / 6.0e6);
  }
  for (int d : {1, 4, 16})
    add("pointer chase depth " + std::to_string(d) + " over 64 MiB, grid 256 x 64", [=](hipStream_t st, int) { hipLaunchKernelGGL(k_chase, dim3(256), dim3(64), 0, st, next, d, sink); });
  // alternating kernels, like a real step: no two neighbours are the same code
  add("alternating: empty / stream 4 MiB / LDS 160 KiB / chase 4", [=](hipStream_t st, int i) {
    switch (i & 3) {
      case 0: hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st); break;
      case 1: hipLaunchKernelGGL(k_stream, dim3(1024), dim3(256), 0, st, a, b, (4l << 20) / 16); break;
      case 2: hipLaunchKernelGGL(k_lds, dim3(256), dim3(512), 160 * 1024, st, sink); break;
      default: hipLaunchKernelGGL(k_chase, dim3(256), dim3(64), 0, st, next, 4, sink); break;
    } }, 0.25 * 2. * (4 << 20) / 6.0e6);

  printf("%-72s %9s %9s %9s %9s %9s\n", "kernel", "eager", "capture", "cap(ev)", "explicit", "body@6TB/s");
  for (auto& c : cases) {
    double eager = time_eager(c.f, N, 5, s);
    hipGraphExec_t ex = capture(c.f, N, s);
    double ev = 0., ev2 = 0.;
    double cap = time_graph(ex, N, R, s, &ev);
    CK(hipGraphExecDestroy(ex));
    // explicit graph: the captured graph's nodes re-created by hand is the same object; what differs is a graph made of
    // hipGraphAddKernelNode calls -- only for the argument-free kernel, whose node parameters need no marshalling
    double expl = -1.;
    if (c.name.rfind("empty grid", 0) == 0) {
      unsigned g = 256, blk = 256;
      sscanf(c.name.c_str(), "empty grid %u x %u", &g, &blk);
      hipGraph_t gr; CK(hipGraphCreate(&gr, 0));
      hipGraphNode_t prev = nullptr;
      for (int i = 0; i < N; ++i) {
        hipKernelNodeParams p{};
        p.func = (void*)k_empty; p.gridDim = dim3(g); p.blockDim = dim3(blk); p.sharedMemBytes = 0; p.kernelParams = nullptr; p.extra = nullptr;
        hipGraphNode_t nd;
        CK(hipGraphAddKernelNode(&nd, gr, prev ? &prev : nullptr, prev ? 1 : 0, &p));
        prev = nd;
      }
      hipGraphExec_t ex2; CK(hipGraphInstantiate(&ex2, gr, nullptr, nullptr, 0));
      expl = time_graph(ex2, N, R, s, &ev2);
      CK(hipGraphExecDestroy(ex2)); CK(hipGraphDestroy(gr));
    }
    printf("%-72s %9.2f %9.2f %9.2f %9.2f %9.2f\n", c.name.c_str(), eager, cap, ev, expl, c.body_us);
    fflush(stdout);
  }
  // ---- do two BRANCHES of a captured graph run side by side?  Each node holds `grid` workgroups for 20 us (k_spin).  One chain of
  // 2 n nodes against two forked chains of n nodes each (event fork / join inside the capture): side by side = half the time.
  {
    hipStream_t s2;
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t ef, ej;
    CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    const int n = 20;
    for (int grid : {64, 128, 256}) {
      for (int lds : {0, 100 * 1024}) {
        auto node = [&](hipStream_t st) {
          if (lds) hipLaunchKernelGGL(k_lds, dim3(grid), dim3(256), lds, st, sink);       // residency probe only: trivial body
          hipLaunchKernelGGL(k_spin, dim3(grid), dim3(256), 0, st, 2000l, sink);
        };
        hipGraph_t g1, g2;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 2 * n; ++i) node(s);
        CK(hipStreamEndCapture(s, &g1));
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        CK(hipEventRecord(ef, s)); CK(hipStreamWaitEvent(s2, ef, 0));
        for (int i = 0; i < n; ++i) { node(s); node(s2); }
        CK(hipEventRecord(ej, s2)); CK(hipStreamWaitEvent(s, ej, 0));
        CK(hipStreamEndCapture(s, &g2));
        hipGraphExec_t x1, x2;
        CK(hipGraphInstantiate(&x1, g1, nullptr, nullptr, 0)); CK(hipGraphInstantiate(&x2, g2, nullptr, nullptr, 0));
        double e1, e2;
        const double t1 = time_graph(x1, 1, 10, s, &e1), t2 = time_graph(x2, 1, 10, s, &e2);
        // eager on two streams, for reference
        for (int i = 0; i < n; ++i) { node(s); node(s2); }
        CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2));
        double t0 = now_us();
        for (int i = 0; i < n; ++i) { node(s); node(s2); }
        CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2));
        const double te = now_us() - t0;
        printf("# fork: %d x (20 us spin on %3d workgroups%s): one chain of %d nodes %.1f us; two captured branches of %d %.1f us; two eager streams %.1f us\n",
               2 * n, grid, lds ? " + a 100 KiB-LDS node" : "", 2 * n, t1, n, t2, te);
        CK(hipGraphExecDestroy(x1)); CK(hipGraphExecDestroy(x2)); CK(hipGraphDestroy(g1)); CK(hipGraphDestroy(g2));
      }
    }
  }
  // ---- what does ONE fork / join pair cost a long chain?  200 trivial nodes; in the forked variant nodes 100..109 have a 2-node side
  // branch beside them.  If only the fork's neighbourhood pays, the difference is a few us; if the runtime executes a graph with a
  // branch differently as a whole, every node pays.
  {
    hipStream_t s2;
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t ef, ej;
    CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    for (int body = 0; body < 2; ++body) {
      auto node = [&](hipStream_t st) {
        if (body) hipLaunchKernelGGL(k_stream, dim3(1024), dim3(256), 0, st, a, b, (4l << 20) / 16);
        else hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st);
      };
      auto side_node = [&](hipStream_t st) { hipLaunchKernelGGL(k_read, dim3(256), dim3(256), 0, st, a + (16l << 20) / 16, (1l << 20) / 16, sink); };
      hipGraph_t g1, g2, g3;
      CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      for (int i = 0; i < 200; ++i) node(s);
      side_node(s); side_node(s);
      CK(hipStreamEndCapture(s, &g1));
      CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      for (int i = 0; i < 200; ++i) {
        if (i == 100) { CK(hipEventRecord(ef, s)); CK(hipStreamWaitEvent(s2, ef, 0)); side_node(s2); side_node(s2); CK(hipEventRecord(ej, s2)); }
        if (i == 110) CK(hipStreamWaitEvent(s, ej, 0));
        node(s);
      }
      CK(hipStreamEndCapture(s, &g2));
      // fork at the very start, join at the very end
      CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      CK(hipEventRecord(ef, s)); CK(hipStreamWaitEvent(s2, ef, 0)); side_node(s2); side_node(s2); CK(hipEventRecord(ej, s2));
      for (int i = 0; i < 200; ++i) node(s);
      CK(hipStreamWaitEvent(s, ej, 0));
      CK(hipStreamEndCapture(s, &g3));
      hipGraphExec_t x1, x2, x3;
      CK(hipGraphInstantiate(&x1, g1, nullptr, nullptr, 0)); CK(hipGraphInstantiate(&x2, g2, nullptr, nullptr, 0)); CK(hipGraphInstantiate(&x3, g3, nullptr, nullptr, 0));
      double e1, e2, e3;
      const double t1 = time_graph(x1, 1, 20, s, &e1), t2 = time_graph(x2, 1, 20, s, &e2), t3 = time_graph(x3, 1, 20, s, &e3);
      printf("# one fork in a chain of 200 %s nodes (+ 2 side nodes): all in one chain %.1f us; side nodes as a branch beside nodes 100..109 %.1f us; "
             "branch forked at the start, joined at the end %.1f us\n", body ? "4 MiB-stream" : "trivial", t1, t2, t3);
      CK(hipGraphExecDestroy(x1)); CK(hipGraphExecDestroy(x2)); CK(hipGraphExecDestroy(x3));
      CK(hipGraphDestroy(g1)); CK(hipGraphDestroy(g2)); CK(hipGraphDestroy(g3));
    }
  }
  // one replay at a time with a host sync in between: the per-replay floor (not a per-node cost)
  {
    Launch f = cases[1].f;
    hipGraphExec_t ex = capture(f, N, s);
    for (int r = 0; r < 3; ++r) { CK(hipGraphLaunch(ex, s)); CK(hipStreamSynchronize(s)); }
    double t0 = now_us();
    for (int r = 0; r < R; ++r) { CK(hipGraphLaunch(ex, s)); CK(hipStreamSynchronize(s)); }
    double per = (now_us() - t0) / R;
    printf("# one replay of %d empty 256 x 256 nodes with a host sync after each: %.1f us per replay = %.2f us per node\n", N, per, per / N);
    CK(hipGraphExecDestroy(ex));
  }
  return 0;
}
