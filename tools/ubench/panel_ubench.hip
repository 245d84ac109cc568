// Micro-benchmarks behind the dense solve of the body-body contact path (round 6): what one step of the register-resident panel
// factorization and of the back substitution cost on gfx950, for a lone wave and with 8 waves per CU.
//   hipcc --offload-arch=gfx950 -O3 -o panel_ubench panel_ubench.hip && ./panel_ubench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#ifndef REPS
#define REPS 64
#endif
__device__ __forceinline__ float rl(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
__device__ __forceinline__ float rcp_nr(float x) { float r = __builtin_amdgcn_rcpf(x); return r * (2.0f - x * r); }
__global__ void __launch_bounds__(512) k(unsigned long long *out, float *sink, int mode, int lb) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ float lds[8][64 * 17];
  float *L = lds[wave];
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; i++) a[i] = (lane == i ? 20.f : 0.f) + 1.0f / (1 + lane + i);
  unsigned long long t0 = __builtin_readcyclecounter();
  if (mode == 0) {                                           // panel: readlane pivots, rcp + Newton step
    for (int r = 0; r < REPS; r++) {
#pragma unroll
      for (int kk = 0; kk < 16; kk++) {
        const float dk = rl(a[kk], lb + kk);
        const float tl = a[kk] * rcp_nr(dk);
#pragma unroll
        for (int j = kk + 1; j < 16; j++) a[j] -= tl * rl(a[kk], lb + j);
        a[kk] = tl + 20.f;
      }
    }
  } else if (mode == 1) {                                    // the same without the Newton step
    for (int r = 0; r < REPS; r++) {
#pragma unroll
      for (int kk = 0; kk < 16; kk++) {
        const float dk = rl(a[kk], lb + kk);
        const float tl = a[kk] * __builtin_amdgcn_rcpf(dk);
#pragma unroll
        for (int j = kk + 1; j < 16; j++) a[j] -= tl * rl(a[kk], lb + j);
        a[kk] = tl + 20.f;
      }
    }
  } else if (mode == 2) {                                    // pivot column through LDS: the 16 pivot-row lanes write a[kk], everybody reads 16 - kk values (uniform addresses)
    for (int r = 0; r < REPS; r++) {
#pragma unroll
      for (int kk = 0; kk < 16; kk++) {
        if (lane >= lb && lane < lb + 16) L[lane - lb] = a[kk];
        __builtin_amdgcn_wave_barrier();
        const float dk = L[kk];
        const float tl = a[kk] * rcp_nr(dk);
#pragma unroll
        for (int j = kk + 1; j < 16; j++) a[j] -= tl * L[j];
        a[kk] = tl + 20.f;
        __builtin_amdgcn_wave_barrier();
      }
    }
  } else if (mode == 3) {                                    // back substitution step: sub, readlane, select, fma
    float v = a[0], acc = 0, z = 0;
    for (int r = 0; r < REPS; r++) {
#pragma unroll
      for (int i = 15; i >= 0; i--) {
        const float zi = rl(v - acc, i);
        z = (lane & 15) == i ? zi : z;
        acc += a[i] * zi;
      }
    }
    a[0] = z + acc;
  } else if (mode == 4) {                                    // 16 independent readlanes + 16 fmas on them (throughput of the pair)
    for (int r = 0; r < REPS; r++) {
#pragma unroll
      for (int j = 0; j < 16; j++) a[j] += a[(j + 1) & 15] * rl(a[(j + 5) & 15], lb + j);
    }
  } else if (mode == 5) {                                    // rcp chain
    float x = a[0];
    for (int r = 0; r < REPS; r++) {
#pragma unroll
      for (int j = 0; j < 16; j++) x = __builtin_amdgcn_rcpf(x) + 1.5f;
    }
    a[0] = x;
  } else if (mode == 6) {                                    // readlane -> fma chain through the SGPR (each fma feeds the next readlane)
    float x = a[0];
    for (int r = 0; r < REPS; r++) {
#pragma unroll
      for (int j = 0; j < 16; j++) x = x * 0.5f + rl(x, lb + j);
    }
    a[0] = x;
  } else if (mode == 7) {                                    // dependent LDS round trip: write, read neighbour
    float x = a[0];
    for (int r = 0; r < REPS; r++) {
#pragma unroll
      for (int j = 0; j < 16; j++) { L[lane] = x; __builtin_amdgcn_wave_barrier(); x = L[(lane + 1) & 63] + 1.f; __builtin_amdgcn_wave_barrier(); }
    }
    a[0] = x;
  } else if (mode == 8) {                                    // 4 dependent f32 matrix instructions
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 c = {a[0], a[1], a[2], a[3]};
    for (int r = 0; r < REPS; r++) {
#pragma unroll
      for (int j = 0; j < 16; j++) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j & 7], a[8 + (j & 7)], c, 0, 0, 0);
    }
    a[0] = c[0] + c[1] + c[2] + c[3];
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s += a[i];
  sink[blockIdx.x * 512 + threadIdx.x] = s;
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}
int main() {
  unsigned long long *d; float *sink; hipMalloc(&d, 1 << 20); hipMalloc(&sink, 256 * 512 * 4);
  const char *names[] = {"panel 16x16 readlane, rcp+NR", "panel 16x16 readlane, rcp", "panel 16x16 via LDS", "back substitution, 16 steps", "16 x (readlane + fma), independent",
                         "16 x rcp chain", "16 x (readlane -> fma) chain", "16 x LDS write->read round trip", "16 dependent mfma 16x16x4 f32"};
  for (int waves : {1, 8}) for (int mode = 0; mode < 9; mode++) {
    double us = 0;
    for (int rep = 0; rep < 2; rep++) {
      hipMemset(d, 0, 1 << 20);
      hipDeviceSynchronize();
      auto c0 = std::chrono::steady_clock::now();
      hipLaunchKernelGGL(k, dim3(256), dim3(64 * waves), 0, 0, d, sink, mode, 16);
      hipDeviceSynchronize();
      us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - c0).count();
    }
    std::vector<unsigned long long> h(256 * 8); hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; int c = 0; for (int b = 0; b < 256; b++) for (int w = 0; w < waves; w++) { s += h[b * 8 + w]; c++; }
    printf("waves/CU=%d  %-38s %8.1f ticks per block of 16   launch %8.1f us  = %.2f ticks per ns\n", waves, names[mode], s / c / REPS, us, s / c / (us * 1e3));
  }
  return 0;
}
