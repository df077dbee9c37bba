// microbenchmark: LDS cycles per ds_read_b64_tr_b16 wave-instruction for several lane -> address patterns (8 waves per CU,
// 12 independent reads per iteration like csrc/wgrad256.hip), next to ds_read_b64 / ds_read_b128 on the same bytes.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/trread.hip -o tools/ubench/trread ; run: tools/ubench/trread
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef short v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s lds_v4s;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) u32x2 lds_u2;
typedef __attribute__((address_space(3))) u32x4 lds_u4;

__device__ __forceinline__ int pattern(int pat, int lane) {
  const int g = lane >> 4, p = lane & 15;
  switch (pat) {
    case 0: return lane * 8;                                                         // linear: [16 rows][16 cols] block
    case 1: return (4 * (g >> 1) + (p >> 2)) * 64 + (g & 1) * 32 + (p & 3) * 8;      // wgrad256: [8][32], groups = (row half, col half)
    case 2: return (4 * (g & 1) + (p >> 2)) * 64 + (g >> 1) * 32 + (p & 3) * 8;      // [8][32], group order swapped
    case 3: return (g & 1) * 512 + (4 * (g >> 1) + (p >> 2)) * 32 + (p & 3) * 8;     // two [16][16] blocks side by side
    case 4: return (p >> 2) * 128 + g * 32 + (p & 3) * 8;                            // [4][64]
    case 5: return (4 * (g >> 1) + (p >> 2)) * 64 + ((g & 1) * 32 + (p & 3) * 8 ^ (((p >> 2) & 1) * 32));   // [8][32], odd rows swap halves
    case 6: return (4 * (g >> 1) + (p >> 2)) * 72 + (g & 1) * 32 + (p & 3) * 8;      // [8][32] rows padded to 72 B
    case 7: return (4 * (g >> 1) + (p >> 2)) * 80 + (g & 1) * 32 + (p & 3) * 8;      // rows padded to 80 B
    default: return lane * 8;
  }
}

template <int KIND>   // 0 = tr_b16, 1 = ds_read_b64, 2 = ds_read_b128 (linear 16 B per lane)
__global__ __launch_bounds__(512) void k(int pat, int iters, unsigned long long* out, unsigned* sink) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[65536];
  for (int i = threadIdx.x; i < 65536 / 4; i += 512) reinterpret_cast<unsigned*>(smem)[i] = i * 2654435761u;
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  unsigned addr = base + (KIND == 2 ? lane * 16 : pattern(pat, lane)) + (wid & 3) * 1024;
  unsigned acc = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const unsigned a = addr + ((it & 3) * 16384);
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const unsigned ai = a + (i & 7) * (KIND == 2 ? 1024 : 512) + (i >> 3) * 8192;
      if (KIND == 0) { v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)(size_t)ai); acc ^= (unsigned)r[0] ^ ((unsigned)r[3] << 16); }
      else if (KIND == 1) { u32x2 r = *(lds_u2*)(size_t)ai; acc ^= r.x ^ r.y; }
      else { u32x4 r = *(lds_u4*)(size_t)ai; acc ^= r.x ^ r.w; }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (acc == 0x9e3779b9u) sink[0] = acc;
  if (lane == 0) out[blockIdx.x * 8 + wid] = t1 - t0;
}

int main() {
  unsigned long long* out; unsigned* sink;
  hipMalloc(&out, 256 * 8 * 8); hipMalloc(&sink, 64);
  const int iters = 4000;
  unsigned long long h[2048];
  auto report = [&](const char* name, int pat) {
    hipDeviceSynchronize();
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 2048; ++i) s += (double)h[i];
    s /= 2048;   // cycles one wave spent (s_memtime ticks = shader cycles... readcyclecounter may be a fixed 100 MHz clock)
    printf("%-10s pattern %d: %.1f ticks per iteration of 12 reads per wave -> %.2f ticks per wave-instruction at 8 waves/CU\n", name, pat,
           s / iters, s / iters / 12.0 / 8.0);
  };
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int kind = 0; kind < 3; ++kind)
    for (int pat = 0; pat < (kind == 2 ? 1 : 8); ++pat) {
      hipEventRecord(e0);
      if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, pat, iters, out, sink);
      else if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, pat, iters, out, sink);
      else hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, pat, iters, out, sink);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      // wall time -> ns per wave-instruction per CU: ms / (iters * 12 reads * 8 waves)
      printf("kind %d pat %d: %.3f ms wall = %.3f ns per wave-instruction per CU (x ~2.1 GHz = %.2f clk)\n", kind, pat, ms,
             ms * 1e6 / ((double)iters * 12 * 8), ms * 1e6 / ((double)iters * 12 * 8) * 2.1);
      report(kind == 0 ? "tr_b16" : kind == 1 ? "b64" : "b128", pat);
    }
  return 0;
}
