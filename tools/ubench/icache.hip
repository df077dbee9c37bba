// Does COLD CODE cost a kernel microseconds?  Every launch of the training step is a different kernel of 20-60 KB whose first
// thousands of instructions are straight-line, fully unrolled prologue code executed ONCE per wave (clock stamps: 15-19 k cycles
// from kernel entry to the end of a LayerNorm prologue that moves ~100 KB, profiles/r05_dec_trace.txt).  This program times chains
// of kernels whose body is N straight-line dependent FMAs (8 B each) executed once:
//   same     : one kernel repeated (its code stays in the instruction cache / L2)
//   rotating : K different kernels of the same size in rotation (each one's code was last touched K launches ago)
// and the rolled equivalent (a loop of the same trip count: a few cache lines of code).  us per node in a captured graph chain.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

template <int N, int ID> __global__ __launch_bounds__(256) void k_straight(float* out, float a, float b) {
  float v = (float)threadIdx.x + ID;
#pragma unroll
  for (int i = 0; i < N; ++i) v = __builtin_fmaf(v, a, b + (float)(i * 7 + ID));      // distinct literal per instruction: no folding, 8-12 B each
  if (v == 1.2345e30f) out[0] = v;
}
template <int ID> __global__ __launch_bounds__(256) void k_rolled(float* out, float a, float b, int n) {
  float v = (float)threadIdx.x + ID;
#pragma unroll 1
  for (int i = 0; i < n; ++i) v = __builtin_fmaf(v, a, b + (float)ID);
  if (v == 1.2345e30f) out[0] = v;
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
typedef void (*launch_t)(hipStream_t, float*, int);
template <int N, int ID> void L(hipStream_t s, float* o, int g) { hipLaunchKernelGGL((k_straight<N, ID>), dim3(g), dim3(256), 0, s, o, 1.0001f, 0.5f); }
template <int ID> void R(hipStream_t s, float* o, int g, int n) { hipLaunchKernelGGL((k_rolled<ID>), dim3(g), dim3(256), 0, s, o, 1.0001f, 0.5f, n); }

static double chain(hipStream_t s, const std::vector<launch_t>& ks, float* o, int grid, int nodes) {
  hipGraph_t g; hipGraphExec_t x;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < nodes; ++i) ks[i % ks.size()](s, o, grid);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&x, g, nullptr, nullptr, 0));
  for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(x, s));
  CK(hipStreamSynchronize(s));
  const int R_ = 10;
  double t0 = now_us();
  for (int r = 0; r < R_; ++r) CK(hipGraphLaunch(x, s));
  CK(hipStreamSynchronize(s));
  double t = (now_us() - t0) / (R_ * (double)nodes);
  CK(hipGraphExecDestroy(x)); CK(hipGraphDestroy(g));
  return t;
}
#define ROT8(N) {L<N, 0>, L<N, 1>, L<N, 2>, L<N, 3>, L<N, 4>, L<N, 5>, L<N, 6>, L<N, 7>}
int main() {
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  float* o; CK(hipMalloc(&o, 4096));
  printf("# straight-line body executed once per wave; grid 256 x 256 threads; us per node (captured chain of 160 nodes)\n");
  printf("%-28s %10s %10s %10s\n", "body", "same", "rotate 8", "rolled");
  {
    std::vector<launch_t> same = {L<500, 0>}, rot = ROT8(500);
    std::vector<launch_t> rl = {[](hipStream_t st, float* oo, int g) { R<0>(st, oo, g, 500); }};
    printf("%-28s %10.2f %10.2f %10.2f\n", "500 fma (~5 KB)", chain(s, same, o, 256, 160), chain(s, rot, o, 256, 160), chain(s, rl, o, 256, 160));
  }
  {
    std::vector<launch_t> same = {L<2000, 0>}, rot = ROT8(2000);
    std::vector<launch_t> rl = {[](hipStream_t st, float* oo, int g) { R<0>(st, oo, g, 2000); }};
    printf("%-28s %10.2f %10.2f %10.2f\n", "2000 fma (~20 KB)", chain(s, same, o, 256, 160), chain(s, rot, o, 256, 160), chain(s, rl, o, 256, 160));
  }
  {
    std::vector<launch_t> same = {L<6000, 0>}, rot = ROT8(6000);
    std::vector<launch_t> rl = {[](hipStream_t st, float* oo, int g) { R<0>(st, oo, g, 6000); }};
    printf("%-28s %10.2f %10.2f %10.2f\n", "6000 fma (~60 KB)", chain(s, same, o, 256, 160), chain(s, rot, o, 256, 160), chain(s, rl, o, 256, 160));
  }
  printf("# dependent fma: 4-8 cycles each; 2000 of them ~ 16 k cycles = 7 us of issue alone when they do not overlap -- compare 'same' with 'rolled' for the\n# pure execution time and 'rotate 8' with 'same' for what cold code adds\n");
  return 0;
}
