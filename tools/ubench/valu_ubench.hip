// VALU issue micro-benchmarks on gfx950: cycles per instruction of one wavefront's dependent / independent
// FMA streams, scalar and packed, alone on its SIMD and with 2-3 waves sharing it; plus s_memtime's own rate.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define N 1024
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(1024) k(unsigned long long *out, int mode, float a, float b) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float x0 = lane, x1 = lane + 1, x2 = lane + 2, x3 = lane + 3, x4 = lane + 4, x5 = lane + 5, x6 = lane + 6, x7 = lane + 7;
  f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, pa = {a, a}, pb = {b, b};
  unsigned long long t0 = __builtin_readcyclecounter();
  if (mode == 0) { for (int i = 0; i < N; i++) { x0 = x0 * a + b; x0 = x0 * a + b; x0 = x0 * a + b; x0 = x0 * a + b; x0 = x0 * a + b; x0 = x0 * a + b; x0 = x0 * a + b; x0 = x0 * a + b; } }
  else if (mode == 1) { for (int i = 0; i < N; i++) { x0 = x0 * a + b; x1 = x1 * a + b; x2 = x2 * a + b; x3 = x3 * a + b; x4 = x4 * a + b; x5 = x5 * a + b; x6 = x6 * a + b; x7 = x7 * a + b; } }
  else if (mode == 2) { for (int i = 0; i < N; i++) { p0 = p0 * pa + pb; p0 = p0 * pa + pb; p0 = p0 * pa + pb; p0 = p0 * pa + pb; p0 = p0 * pa + pb; p0 = p0 * pa + pb; p0 = p0 * pa + pb; p0 = p0 * pa + pb; } }
  else if (mode == 3) { for (int i = 0; i < N; i++) { p0 = p0 * pa + pb; p1 = p1 * pa + pb; p2 = p2 * pa + pb; p3 = p3 * pa + pb; p0 = p0 * pa + pb; p1 = p1 * pa + pb; p2 = p2 * pa + pb; p3 = p3 * pa + pb; } }
  else if (mode == 4) { for (int i = 0; i < N; i++) { x0 = x0 * a + b; x1 = x1 * a + b; x0 = x0 * a + b; x1 = x1 * a + b; x0 = x0 * a + b; x1 = x1 * a + b; x0 = x0 * a + b; x1 = x1 * a + b; } }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
  if (s == 12345.678f) out[100000] = 1;
  if (lane == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
}
int main() {
  unsigned long long *d; hipMalloc(&d, 1 << 22);
  const char *names[] = {"dependent v_fma chain", "8 independent v_fma", "dependent v_pk_fma chain", "4 independent v_pk_fma", "2 independent v_fma chains"};
  for (int waves : {1, 4, 8, 12, 16}) for (int mode = 0; mode < 5; mode++) {
    hipMemset(d, 0, 1 << 22);
    auto c0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(k, dim3(256), dim3(64 * waves), 0, 0, d, mode, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - c0).count();
    std::vector<unsigned long long> h(256 * 16); hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; int c = 0; for (int b = 0; b < 256; b++) for (int w = 0; w < waves; w++) { s += h[b * 16 + w]; c++; }
    printf("waves/CU=%2d  %-28s %6.2f ticks per instruction   (launch %.0f us)\n", waves, names[mode], s / c / N / 8, us);
  }
  return 0;
}
