// microbenchmark: per-CU streaming bandwidth for a buffer every CU re-reads from L2 (the weight stream of ffn_fused.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define G __attribute__((address_space(1)))

template <int U, int MODE>
__global__ __launch_bounds__(256, 1) void stream_kernel(const uint4* __restrict__ buf, size_t n16, int reps, unsigned* out, int stagger) {
  const int tid = threadIdx.x;
  const size_t per = n16 / 256;             // 16-byte items per thread-slot stream
  unsigned acc = 0;
  size_t start = stagger ? ((size_t)blockIdx.x * 7919) % (per / U) * U : 0;
  for (int r = 0; r < reps; ++r) {
    for (size_t i0 = 0; i0 < per; i0 += U) {
      u32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        size_t i = (start + i0 + u) % per;
        const G u32x4* p = (MODE == 2) ? (const G u32x4*)(buf + (i * 4 + (u & 3)) * 64 + (tid & 63)) : (const G u32x4*)(buf + i * 256 + tid);
        if (MODE == 1) v[u] = __builtin_nontemporal_load(p);
        else v[u] = *p;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
  }
  if (acc == 0x12345678u) out[blockIdx.x] = acc;
}

template <int U, int MODE> float run(const uint4* buf, size_t n16, int blocks, int reps, unsigned* out, int stagger) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((stream_kernel<U, MODE>), dim3(blocks), dim3(256), 0, 0, buf, n16, 1, out, stagger);
  hipEventRecord(e0);
  hipLaunchKernelGGL((stream_kernel<U, MODE>), dim3(blocks), dim3(256), 0, 0, buf, n16, reps, out, stagger);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
  unsigned* out; hipMalloc(&out, 4096 * 4);
  for (size_t mb : {1, 3, 5, 12}) {
    size_t bytes = mb << 20; uint4* buf; hipMalloc(&buf, bytes); hipMemset(buf, 1, bytes);
    size_t n16 = bytes / 16;
    for (int blocks : {249, 498}) for (int stagger : {0, 1}) {
      const int reps = 8;
      float t4 = run<4, 0>(buf, n16, blocks, reps, out, stagger);
      float t8 = run<8, 0>(buf, n16, blocks, reps, out, stagger);
      float t16 = run<16, 0>(buf, n16, blocks, reps, out, stagger);
      float t32 = run<32, 0>(buf, n16, blocks, reps, out, stagger);
      float t16n = run<16, 1>(buf, n16, blocks, reps, out, stagger);
      float t16s = run<16, 2>(buf, n16, blocks, reps, out, stagger);
      double tot = (double)bytes * reps * blocks;
      auto bw = [&](float ms) { return tot / (ms * 1e-3) / 1e12; };
      printf("buf %2zu MB blocks %3d stagger %d : TB/s  U4 %.2f  U8 %.2f  U16 %.2f  U32 %.2f  U16nt %.2f  U16shared(4 waves same addr, counted per wave) %.2f   (per-CU B/clk @2.4GHz U16: %.1f)\n", mb, blocks, stagger,
             bw(t4), bw(t8), bw(t16), bw(t32), bw(t16n), bw(t16s), bw(t16) * 1e12 / (blocks < 256 ? blocks : 256) / 2.4e9);
    }
    hipFree(buf);
  }
  return 0;
}
