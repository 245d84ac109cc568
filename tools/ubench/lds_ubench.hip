// LDS micro-benchmarks on gfx950: latency/throughput of the primitives the env kernel leans on.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N 512
__device__ __forceinline__ void wsync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
__global__ void __launch_bounds__(512) k(unsigned long long *out, int mode) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float *L = lds + wave * 4096;
  for (int i = lane; i < 4096; i += 64) L[i] = (float)((i * 7 + 13) & 4095);
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  float acc = 0.f; int idx = lane;
  if (mode == 0) { for (int i = 0; i < N; i++) { idx = (int)L[idx]; } acc = (float)idx; }                     // dependent read chain
  else if (mode == 1) { for (int i = 0; i < N; i++) acc += L[(lane + 64 * i) & 4095]; }                        // independent reads
  else if (mode == 2) { for (int i = 0; i < N; i++) __hip_atomic_fetch_add(&L[(lane + 64 * i) & 4095], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }  // atomics, no conflict
  else if (mode == 3) { for (int i = 0; i < N; i++) __hip_atomic_fetch_add(&L[(lane % 9) + 16 * (i & 7)], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }  // 64 lanes -> 9 addresses
  else if (mode == 4) { for (int i = 0; i < N; i++) { L[lane] = acc; wsync(); acc += L[(lane + 1) & 63]; wsync(); } }   // write / sync / read / sync
  else if (mode == 5) { for (int i = 0; i < N; i++) { float v = L[(lane * 3 + i) & 4095]; L[(lane * 3 + i) & 4095] = v + 1.f; } }  // plain RMW
  else if (mode == 6) { for (int i = 0; i < N; i++) { int t = ((const int *)L)[lane & 31]; acc += L[(t + i) & 4095]; } }   // table read -> data read (2-deep chain)
  else if (mode == 7) { for (int i = 0; i < N; i++) atomicAdd(&L[(lane + 64 * i) & 4095], 1.0f); }          // plain atomicAdd
  unsigned long long t1 = __builtin_readcyclecounter();
  if (acc == 12345.678f) out[1000] = 1;
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}
int main() {
  unsigned long long *d; hipMalloc(&d, 1 << 20);
  const char *names[] = {"dependent ds_read chain", "independent ds_read", "ds_add_f32 no conflict", "ds_add_f32 64 lanes->9 addr", "write+sync+read+sync", "plain RMW", "table->data chain", "atomicAdd(float)"};
  for (int waves = 1; waves <= 8; waves *= 8) for (int blocks = 1; blocks <= 256; blocks *= 256) for (int mode = 0; mode < 8; mode++) {
    hipMemset(d, 0, 1 << 20);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * waves), waves * 4096 * 4, 0, d, mode);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 8); hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; int c = 0; for (int b = 0; b < blocks; b++) for (int w = 0; w < waves; w++) { s += h[b * 8 + w]; c++; }
    printf("waves/WG=%d blocks=%3d  %-30s %8.1f ticks/iter\n", waves, blocks, names[mode], s / c / N);
  }
  return 0;
}
