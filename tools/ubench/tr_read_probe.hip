// What ds_read_b64_tr_b16 delivers (gfx950).  LDS holds its own element index (u16 at element e = e); lane l passes the byte address A(l); the four
// 16-bit results of every lane are printed.  Two address patterns: (a) lane l -> 8 * l (every lane its own 4 consecutive elements, linear),
// (b) a [4 rows][16 cols] row-major block per 16-lane group with row stride 64 elements: lane p of a group -> row p / 4, cols 4 (p % 4) ..
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
__global__ void probe(uint16_t *out, int pattern) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int l = threadIdx.x;
  unsigned addr;
  if (pattern == 0) addr = 8u * l;
  else { const int g = l >> 4, p = l & 15; addr = 2u * (g * 1024 + (p >> 2) * 64 + (p & 3) * 4); }
  addr += (unsigned)(uintptr_t)lds;
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[4 * l + 0] = v[0] & 0xffff; out[4 * l + 1] = v[0] >> 16; out[4 * l + 2] = v[1] & 0xffff; out[4 * l + 3] = v[1] >> 16;
}
int main() {
  uint16_t *d, h[256];
  hipMalloc(&d, sizeof(h));
  for (int pattern = 0; pattern < 2; pattern++) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, pattern);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("pattern %d (%s)\n", pattern, pattern == 0 ? "lane l reads at element 4 l" : "[4][16] block per 16-lane group, row stride 64, group base 1024 g");
    for (int l = 0; l < 64; l++) printf("  lane %2d: %5d %5d %5d %5d%s", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3], (l & 3) == 3 ? "\n" : "");
  }
  return 0;
}
