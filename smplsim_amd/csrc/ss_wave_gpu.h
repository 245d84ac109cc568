// ss_wave_gpu.h — the wavefront policy of the gfx950 build: cross-lane operations as DPP / readlane / ballot instructions and
// the wave-level LDS hand-off.  Shared by the two translation units of libsmplsim_hip.so (stepper, motion library).
#pragma once
#include <hip/hip_runtime.h>

namespace {

struct WaveGpu {
  int ln;
  __device__ __forceinline__ int lane() const { return ln; }
  // wave-level LDS hand-off: the DS instructions of one wavefront are issued and executed in program order, so a
  // read that follows a write in the instruction stream sees it without draining lgkmcnt; what is needed is only
  // that the compiler keeps LDS accesses on their side of the hand-off (memory clobber + scheduling barrier)
  __device__ __forceinline__ void sync() const {
#ifdef SS_SYNC_DRAIN
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
    asm volatile("" ::: "memory");
#endif
    __builtin_amdgcn_wave_barrier();
  }
  // cross-lane moves as DPP modifiers of VALU instructions (no LDS-crossbar ds_bpermute round trips):
  // quad_perm [1,0,3,2] = 0xB1, quad_perm [2,3,0,1] = 0x4E, row_half_mirror = 0x141, row_mirror = 0x140
  template <int CTRL> static __device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
  template <int CTRL> static __device__ __forceinline__ float dpp_f(float v) { return __builtin_bit_cast(float, dpp_i<CTRL>(__builtin_bit_cast(int, v))); }
  static __device__ __forceinline__ float rl(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
  // wave-wide sum, result in every lane: 4 DPP steps give every 16-lane row its sum r0..r3, row_bcast:15 adds lane 15 of every row
  // into the next one (row 3 = r3 + r2, row 1 = r1 + r0; rows 0 and 2 are not used again), row_bcast:31 adds lane 31 into rows 2
  // and 3: lane 63 holds (r3 + r2) + (r1 + r0), one readlane (6 fused DPP adds + 1 readlane; four readlanes and three adds were 12
  // instructions: profiles/r03_centred_elimination.md 11)
  __device__ __forceinline__ float sum(float v) const {
    v += dpp_f<0xB1>(v); v += dpp_f<0x4E>(v); v += dpp_f<0x141>(v); v += dpp_f<0x140>(v);
    v += dpp_f<0x142>(v); v += dpp_f<0x143>(v);
    asm volatile("" : "+v"(v));
    return rl(v, 63);
  }
  // (the empty asm pins the last add next to its DPP move: the optimizer otherwise sinks the add into the conditional block that
  // consumes the sum, where it can no longer be fused into one v_add_f32_dpp — 9 extra instructions per tree level)
  __device__ __forceinline__ float sum8(float v) const { v += dpp_f<0xB1>(v); v += dpp_f<0x4E>(v); v += dpp_f<0x141>(v); asm volatile("" : "+v"(v)); return v; }
  // workgroup-level lock on an LDS word (0 = free): the wavefronts of a workgroup share one dense block for the rare coupled sets that do
  // not fit an env's own LDS region (ss_hdr.h).  The holder never waits for anybody, so a waiting wave only spins.  The DS instructions of
  // a wave are issued in order: everything the holder did to the block precedes its release in the LDS queue
  __device__ __forceinline__ void lock_acquire(int *p) const {
    for (;;) {
      int got = 0;
      if (ln == 0) got = atomicCAS(p, 0, 1) == 0;
      if (__builtin_amdgcn_readfirstlane(got)) break;
      __builtin_amdgcn_s_sleep(16);
    }
    asm volatile("" ::: "memory");
  }
  __device__ __forceinline__ void lock_release(int *p) const {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (ln == 0) __hip_atomic_store(p, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  // value of lane `src` (wave-uniform index) in every lane
  __device__ __forceinline__ float bcast(float v, int src) const { return rl(v, src); }
  __device__ __forceinline__ int bcast_i(int v, int src) const { return __builtin_amdgcn_readlane(v, src); }
  // D = A B + C on the matrix core, exact float32 (v_mfma_f32_16x16x4_f32: an fmaf chain over k = 0..3): lane l supplies A[l & 15][l >> 4]
  // and B[l >> 4][l & 15] and holds C/D[4 (l >> 4) + r][l & 15] in c[r]
  __device__ __forceinline__ void mfma16(float a, float b, float *c) const {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 v = {c[0], c[1], c[2], c[3]};
    v = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, v, 0, 0, 0);
    c[0] = v[0]; c[1] = v[1]; c[2] = v[2]; c[3] = v[3];
  }
  __device__ __forceinline__ float quad_xor1(float v) const { return dpp_f<0xB1>(v); }
  __device__ __forceinline__ float quad_xor2(float v) const { return dpp_f<0x4E>(v); }
  __device__ __forceinline__ int quad_xor1_i(int v) const { return dpp_i<0xB1>(v); }
  __device__ __forceinline__ int quad_xor2_i(int v) const { return dpp_i<0x4E>(v); }
  __device__ __forceinline__ void mem_fence() const { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent"); }
  __device__ __forceinline__ unsigned long long clock() const { return __builtin_readcyclecounter(); }
  __device__ __forceinline__ void atomic_add_u64(unsigned long long *p, unsigned long long v) const { atomicAdd(p, v); }
  __device__ __forceinline__ int opaque(int x) const { return __builtin_amdgcn_readfirstlane(x); }   // wave-uniform, optimizer-opaque
  __device__ __forceinline__ int opaque_v(int x) const { asm volatile("" : "+v"(x)); return x; }   // per-lane value, optimizer-opaque
  __device__ __forceinline__ float shfl_xor(float v, int m) const { return __shfl_xor(v, m, 64); }
  __device__ __forceinline__ int shfl_xor_i(int v, int m) const { return __shfl_xor(v, m, 64); }
  __device__ __forceinline__ unsigned long long ballot(int p) const { return __ballot(p); }
  __device__ __forceinline__ bool any(int p) const { return __any(p) != 0; }
  __device__ __forceinline__ unsigned long long bor(unsigned long long v) const {
    int lo = (int)(unsigned)v, hi = (int)(unsigned)(v >> 32);
    lo |= dpp_i<0xB1>(lo); lo |= dpp_i<0x4E>(lo); lo |= dpp_i<0x141>(lo); lo |= dpp_i<0x140>(lo);
    hi |= dpp_i<0xB1>(hi); hi |= dpp_i<0x4E>(hi); hi |= dpp_i<0x141>(hi); hi |= dpp_i<0x140>(hi);
    const unsigned l = (unsigned)(__builtin_amdgcn_readlane(lo, 0) | __builtin_amdgcn_readlane(lo, 16) | __builtin_amdgcn_readlane(lo, 32) | __builtin_amdgcn_readlane(lo, 48));
    const unsigned h = (unsigned)(__builtin_amdgcn_readlane(hi, 0) | __builtin_amdgcn_readlane(hi, 16) | __builtin_amdgcn_readlane(hi, 32) | __builtin_amdgcn_readlane(hi, 48));
    return ((unsigned long long)h << 32) | l;
  }
  __device__ __forceinline__ void atomic_add(float *p, float v) const {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  }
};

}  // namespace
