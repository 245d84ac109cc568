// smplsim_mlp.hip — gfx950 policy-inference kernels + their C ABI (include/smplsim_mlp.h): y = act(x W^T + b) on the matrix
// cores (v_mfma_f32_32x32x16_bf16: bf16 operands, fp32 accumulation), bias and activation fused into the epilogue.
//
// Tiling for 64-wide wavefronts: a workgroup of 8 waves (4 x 2) owns a 128 x BN output tile (BN = 64 / 128 / 192 / 256: the one whose
// tile count fills the 256 CUs in whole rounds), each wave 32 x BN/2 of it = 1 x BN/64 MFMA tiles of 32 x 32 in 16 .. 64 accumulator
// registers (8 waves instead of 4 with twice the tile each: -16 % on the whole MLP — at 4096 rows there are only ~1.5 workgroups
// per CU, and the K loop's barrier and load latency need waves to hide behind; a 16-wave split measured the same as 8).  Both operands are K-contiguous (activations row-major, weights in torch.nn.Linear's [out, in] layout), so a lane's
// MFMA fragment — 8 consecutive k of one row — is one 16-byte LDS read; K advances 64 per LDS tile (four MFMA K-steps), the next
// tile's global loads are in flight while the current one is multiplied (register double buffer, two LDS buffers, one barrier per
// tile).  LDS rows are padded by 8 bf16 (16 B) so that the 32 rows a fragment read touches spread over the banks.
// What bounds it (round 4, profiles/r04_mlp_gemm.txt): not the matrix cores — a 4-wave variant with 64 x 96 wave tiles (0.83 instead of 1.33
// fragment reads per MFMA, accumulators in AGPRs) ran 58 us against 41 us on the 2048 -> 1536 layer and was not faster with its MFMAs
// REMOVED (61.6 vs 61.4 us); without its LDS stores 40.6 us, without the global loads of the loop 47.6 us.  The K loop is a chain of
// global-load -> LDS-store -> barrier -> fragment-read latencies that only resident waves hide, and the 92-111 KB of LDS of a wide tile
// allow one workgroup per CU.  The 8-wave kernel stays; the wide tiles give 616 (was 570) TFLOP/s on that layer = 0.25 of the dense peak.
// The A and B fragments use the same lane -> k assignment (k = k16 + 8 * (lane / 32) + j), which is all the instruction needs for
// the products to pair up; the C layout is col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>
#include <string>

#include "../../include/smplsim_hip.h"
#include "../../include/smplsim_mlp.h"
#include "ss_api.h"
#include "ss_gemm256.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;   // 16-byte staging register (a native vector: HIP's uint4 class kept loop-carried stages in scratch)

constexpr int BM = 128;

__device__ __forceinline__ float activate(float v, int act) {
  if (act == SS_ACT_SILU) return v / (1.f + __expf(-v));
  if (act == SS_ACT_TANH) { const float e = __expf(-2.f * fabsf(v)); const float t = (1.f - e) / (1.f + e); return v < 0.f ? -t : t; }
  if (act == SS_ACT_RELU) return v > 0.f ? v : 0.f;
  return v;
}

// Epilogue arithmetic with the activation chosen ONCE (round 5: the per-element `switch` with an IEEE division behind it was ~75
// instructions per output element, 3600 per wave for a 32 x 96 wave tile — a quarter of the kernel's time on the wide layers);
// reciprocal by v_rcp_f32 (1 ulp: invisible behind the bf16 rounding of the result).
struct ActNone { __device__ __forceinline__ float operator()(float v) const { return v; } };
struct ActSilu { __device__ __forceinline__ float operator()(float v) const { return v * __builtin_amdgcn_rcpf(1.f + __expf(-v)); } };
struct ActTanh { __device__ __forceinline__ float operator()(float v) const { const float e = __expf(-2.f * fabsf(v)); const float t = (1.f - e) * __builtin_amdgcn_rcpf(1.f + e); return v < 0.f ? -t : t; } };
struct ActRelu { __device__ __forceinline__ float operator()(float v) const { return v > 0.f ? v : 0.f; } };
template <class F> __device__ __forceinline__ void with_activation(int act, F &&f) {
  if (act == SS_ACT_SILU) f(ActSilu{}); else if (act == SS_ACT_TANH) f(ActTanh{}); else if (act == SS_ACT_RELU) f(ActRelu{}); else f(ActNone{});
}

template <int BN, int BK, bool F32OUT, int WM, int WN = 2>
__global__ void __launch_bounds__(64 * WM * WN) ss_linear_kernel(const __bf16 *__restrict__ X, const __bf16 *__restrict__ W, const float *__restrict__ bias,
                                                        void *__restrict__ Y, int M, int N, int K, int ldy, int act, int xcd_remap) {
  constexpr int TN = BN / (32 * WN);                         // MFMA tiles per wave along N (WM x WN waves, each (128 / WM) x (BN / WN))
  constexpr int LDS_STRIDE = BK + 8;                         // K per LDS tile (64, or 32 when K is an odd multiple of 32); rows padded by 16 bytes
  constexpr int NT = 64 * WM * WN;
  constexpr int TM = BM / (32 * WM);                         // MFMA tiles per wave along M (2 for 4 waves, 1 for 8)
  constexpr int CPR = BK / 8;                                // 16-byte chunks per tile row
  constexpr int ACH = BM * CPR / NT, BCH = BN * CPR / NT;    // chunks per thread

  // dynamic LDS (the 192- and 256-column tiles need 92 / 111 KB: above the 64 KB of static __shared__): A buffers, then B buffers
  extern __shared__ __attribute__((aligned(16))) __bf16 lds_ab[];
  __bf16(*As)[BM * LDS_STRIDE] = reinterpret_cast<__bf16(*)[BM * LDS_STRIDE]>(lds_ab);
  __bf16(*Bs)[BN * LDS_STRIDE] = reinterpret_cast<__bf16(*)[BN * LDS_STRIDE]>(lds_ab + 2 * BM * LDS_STRIDE);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave / WN, wn = wave % WN;
  // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (id % 8), each with its own 4 MB L2.  Give XCD x the
  // row-blocks [x gy/8, (x+1) gy/8) and walk them column-block-major inside the XCD: its activations (gy/8 x 128 rows) stay in
  // its L2 across all column-blocks and the weights stream through once per XCD — instead of every XCD touching every row-block
  // (the kernel is bound by L2 / fabric traffic: 64 FLOP per byte at 128 x 128 tiles).
  int bx = blockIdx.x, by = blockIdx.y;
  {
    const int gx = gridDim.x, gy = gridDim.y;
    if (xcd_remap && gy % 8 == 0) {
      const int id = by * gx + bx, xcd = id & 7, idx = id >> 3, rows_per = gy >> 3;
      by = xcd * rows_per + idx % rows_per;
      bx = idx / rows_per;
    }
  }
  const int m0 = by * BM, n0 = bx * BN;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
  // global -> registers -> LDS: 16-byte chunk c of a tile = row c / CPR, k offset (c % CPR) * 8; rows beyond the matrix re-read its
  // last row.  The row of chunk tid + 256 i is the row of chunk tid plus 256 i / CPR, the k offset is the same.
  const int crow = tid / CPR, ckc = (tid % CPR) * 8, soff0 = crow * LDS_STRIDE + ckc;
  constexpr int RSTEP = NT / CPR;                           // rows between a thread's consecutive chunks
  // Two register stages: the loads of tile t + 2 are issued while tile t is multiplied and are written to LDS one step later, so a
  // load has two steps to land (one step — ~16 MFMAs per wave — is shorter than the memory latency; with a single stage every
  // step waited for its own load: 1.6 us per step).  The K loop is unrolled by two so that the stages are fixed registers.
  // (the stages are structs of named registers: arrays indexed by unrolled loops were left in scratch by the compiler here)
  struct Stage { u32x4 a0, a1, a2, a3, b0, b1, b2, b3; };
  Stage s0, s1;
  const int nkt = K / BK;
  const __bf16 *xrow[4], *wrow[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int ra = m0 + crow + RSTEP * (i < ACH ? i : 0), rb = n0 + crow + RSTEP * (i < BCH ? i : 0);
    xrow[i] = X + (size_t)(ra < M ? ra : M - 1) * K + ckc;
    wrow[i] = W + (size_t)(rb < N ? rb : N - 1) * K + ckc;
  }
#define SS_LD(p) (*reinterpret_cast<const u32x4 *>(p))
#define SS_LOAD(S, T)                                                                                                          \
  {                                                                                                                            \
    const int ko_ = ((T) < nkt ? (T) : nkt - 1) * BK;                                                                          \
    S.a0 = SS_LD(xrow[0] + ko_); if (ACH > 1) S.a1 = SS_LD(xrow[1] + ko_); if (ACH > 2) S.a2 = SS_LD(xrow[2] + ko_); if (ACH > 3) S.a3 = SS_LD(xrow[3] + ko_); \
    S.b0 = SS_LD(wrow[0] + ko_); if (BCH > 1) S.b1 = SS_LD(wrow[1] + ko_); if (BCH > 2) S.b2 = SS_LD(wrow[2] + ko_); if (BCH > 3) S.b3 = SS_LD(wrow[3] + ko_); \
  }
#define SS_ST(base, i, v) (*reinterpret_cast<u32x4 *>(&base[soff0 + RSTEP * (i) * LDS_STRIDE]) = (v))
#define SS_STORE(S, BUF)                                                                                                       \
  {                                                                                                                            \
    SS_ST(As[BUF], 0, S.a0); if (ACH > 1) SS_ST(As[BUF], 1, S.a1); if (ACH > 2) SS_ST(As[BUF], 2, S.a2); if (ACH > 3) SS_ST(As[BUF], 3, S.a3); \
    SS_ST(Bs[BUF], 0, S.b0); if (BCH > 1) SS_ST(Bs[BUF], 1, S.b1); if (BCH > 2) SS_ST(Bs[BUF], 2, S.b2); if (BCH > 3) SS_ST(Bs[BUF], 3, S.b3); \
  }
#define SS_COMPUTE(BUF)                                                                                                        \
  _Pragma("unroll") for (int k16 = 0; k16 < BK; k16 += 16) {                                                                   \
    const int ko_ = k16 + 8 * (lane >> 5);                                                                                     \
    bf16x8 a[TM], b[TN];                                                                                                       \
    _Pragma("unroll") for (int tm = 0; tm < TM; tm++) a[tm] = *reinterpret_cast<const bf16x8 *>(&As[BUF][(wm * (32 * TM) + tm * 32 + (lane & 31)) * LDS_STRIDE + ko_]); \
    _Pragma("unroll") for (int tn = 0; tn < TN; tn++) b[tn] = *reinterpret_cast<const bf16x8 *>(&Bs[BUF][(wn * (BN / WN) + tn * 32 + (lane & 31)) * LDS_STRIDE + ko_]); \
    _Pragma("unroll") for (int tm = 0; tm < TM; tm++)                                                                          \
      _Pragma("unroll") for (int tn = 0; tn < TN; tn++) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm], b[tn], acc[tm][tn], 0, 0, 0); \
  }
  SS_LOAD(s0, 0)
  SS_STORE(s0, 0)
  SS_LOAD(s1, 1)
  __syncthreads();
  for (int kt = 0; kt < nkt; kt += 2) {
    SS_LOAD(s0, kt + 2)                                       // tile kt + 2 -> stage 0 (in flight for two steps)
    SS_COMPUTE(0)                                            // tile kt
    SS_STORE(s1, 1)                                           // tile kt + 1 (requested a step ago) -> the other LDS buffer
    __syncthreads();
    SS_LOAD(s1, kt + 3)                                       // (past the end the loads re-read the last tile: unconditional)
    if (kt + 1 < nkt) { SS_COMPUTE(1) }
    SS_STORE(s0, 0)
    __syncthreads();
  }
#undef SS_LD
#undef SS_ST
#undef SS_LOAD
#undef SS_STORE
#undef SS_COMPUTE
  // epilogue: bias + activation (chosen once), bf16 (the next layer's operand; staged through LDS so that a lane stores 16 contiguous
  // bytes of a row) or f32 (the head: a few columns, stored directly)
  float bv[TN];
#pragma unroll
  for (int tn = 0; tn < TN; tn++) {
    const int col = n0 + wn * (BN / WN) + tn * 32 + (lane & 31);
    bv[tn] = (bias && col < N) ? bias[col] : 0.f;
  }
  if constexpr (F32OUT) {
    with_activation(act, [&](auto fn) {
#pragma unroll
      for (int tm = 0; tm < TM; tm++)
#pragma unroll
        for (int tn = 0; tn < TN; tn++) {
          const int col = n0 + wn * (BN / WN) + tn * 32 + (lane & 31);
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const int row = m0 + wm * (32 * TM) + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row < M && col < N) reinterpret_cast<float *>(Y)[(size_t)row * ldy + col] = fn(acc[tm][tn][r] + bv[tn]);
          }
        }
    });
  } else {
    constexpr int CS = BN + 8;
    __bf16 *Cs = lds_ab;                                       // (the K loop ended with a barrier: its buffers are free)
    with_activation(act, [&](auto fn) {
#pragma unroll
      for (int tm = 0; tm < TM; tm++)
#pragma unroll
        for (int tn = 0; tn < TN; tn++) {
          const int cl = wn * (BN / WN) + tn * 32 + (lane & 31);
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const int rl = wm * (32 * TM) + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            Cs[rl * CS + cl] = (__bf16)fn(acc[tm][tn][r] + bv[tn]);
          }
        }
    });
    __syncthreads();
    __bf16 *Yb = reinterpret_cast<__bf16 *>(Y);
    const bool vec_ok = (ldy & 7) == 0 && (reinterpret_cast<size_t>(Yb) & 15) == 0;
    constexpr int OCPR = BN / 8;                              // 16-byte chunks per output tile row
#pragma unroll
    for (int i = 0; i < BM * OCPR / NT; i++) {
      const int id = tid + NT * i, rl = id / OCPR, cc = (id % OCPR) * 8, row = m0 + rl, col = n0 + cc;
      if (row >= M || col >= N) continue;
      const u32x4 v = *reinterpret_cast<const u32x4 *>(Cs + rl * CS + cc);
      if (vec_ok && col + 8 <= N) *reinterpret_cast<u32x4 *>(Yb + (size_t)row * ldy + col) = v;
      else {
        const __bf16 *e = reinterpret_cast<const __bf16 *>(&v);
        for (int j = 0; j < 8 && col + j < N; j++) Yb[(size_t)row * ldy + col + j] = e[j];
      }
    }
  }
}

// ---- round 5: the same GEMM with asynchronous global -> LDS copies (global_load_lds_dwordx4) and three K tiles in flight.
// What round 4 measured (profiles/r04_mlp_gemm.txt): the register-staged K loop above is a chain global load -> LDS store -> barrier ->
// fragment read that one workgroup per CU cannot hide.  Here a tile goes from global memory straight into LDS (no staging registers,
// no ds_write pass), is requested two K steps before it is multiplied, and one barrier per K step remains:
//     wait until tile t has landed (s_waitcnt vmcnt(loads of tile t + 1)) -> barrier -> request tile t + 2 -> multiply tile t
// (the barrier proves that every wave has finished multiplying tile t - 1, whose LDS stage tile t + 2 overwrites).
// LDS layout: a tile row is 64 bf16 = eight 16-byte chunks, rows dense (the copy writes wave-base + lane * 16: 8 rows of 8 chunks per
// wave instruction, no padding possible); chunk c of row r sits at position c ^ ((r >> 1) & 7) — two rows fill the 64 banks, so the 16
// rows a quarter of a fragment read touches need 8 distinct positions per row parity — the XOR is applied to the per-lane SOURCE
// address of the copy (within the row's one 128-byte line) and to the fragment reads' addresses, so that the 32 rows a fragment
// read touches spread over all banks (a ds_read_b128 of 64 lanes takes its four cycles, no more).
typedef __attribute__((address_space(1))) const void ss_gvoid;
typedef __attribute__((address_space(3))) void ss_lvoid;

template <int BN, bool F32OUT>
__global__ void __launch_bounds__(512) ss_linear_glds_kernel(const __bf16 *__restrict__ X, const __bf16 *__restrict__ W, const float *__restrict__ bias,
                                                             void *__restrict__ Y, int M, int N, int K, int ldy, int act, int xcd_remap) {
  constexpr int BK = 64, WM = 4, WN = 2, NST = 3;
  constexpr int TN = BN / (32 * WN);                         // MFMA tiles per wave along N (a wave owns 32 x BN/2 of the 128 x BN tile)
  constexpr int STAGE = (BM + BN) * BK;                      // bf16 elements of one stage: A rows, then B rows
  constexpr int NI = (BM + BN) / 64;                         // copy instructions per wave and tile (8 rows each, 8 waves)
  extern __shared__ __attribute__((aligned(16))) __bf16 lds_g[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave / WN, wn = wave % WN;
  int bx = blockIdx.x, by = blockIdx.y;
  {
    const int gx = gridDim.x, gy = gridDim.y;
    if (xcd_remap && gy % 8 == 0) {                          // XCD-aware tile order (see ss_linear_kernel)
      const int id = by * gx + bx, xcd = id & 7, idx = id >> 3, rows_per = gy >> 3;
      by = xcd * rows_per + idx % rows_per;
      bx = idx / rows_per;
    }
  }
  const int m0 = by * BM, n0 = bx * BN;
  f32x16 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; j++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
  // this wave's copy instructions: instruction i covers tile rows 8 (wave + 8 i) .. + 8 of the stacked (A | B) tile; lane -> row
  // lane / 8, chunk position lane % 8, source chunk (lane % 8) ^ (lane / 8)
  const __bf16 *src[NI];
  int dst[NI];
#pragma unroll
  for (int i = 0; i < NI; i++) {
    const int blk = wave + 8 * i, row = 8 * blk + (lane >> 3), chunk = (lane & 7) ^ ((row >> 1) & 7);
    if (row < BM) { const int g = m0 + row; src[i] = X + (size_t)(g < M ? g : M - 1) * K + chunk * 8; }
    else { const int g = n0 + row - BM; src[i] = W + (size_t)(g < N ? g : N - 1) * K + chunk * 8; }
    dst[i] = 8 * blk * BK;                                   // wave-uniform; the lane's 16 bytes follow at lane * 16
  }
  const int nkt = K / BK;
  auto request = [&](int t) {
    __bf16 *stage = lds_g + (t % NST) * STAGE;
#pragma unroll
    for (int i = 0; i < NI; i++) __builtin_amdgcn_global_load_lds((ss_gvoid *)(src[i] + (size_t)t * BK), (ss_lvoid *)(stage + dst[i]), 16, 0, 0);
  };
  request(0);
  if (nkt > 1) request(1);
  const int arow = wm * 32 + (lane & 31), ax = (arow >> 1) & 7, half = lane >> 5;
  for (int kt = 0; kt < nkt; kt++) {
    if (kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");   // tile kt landed (tile kt + 1 may still be on its way)
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                             // ... for every wave's share of it, and tile kt - 1 is multiplied everywhere
#ifndef SS_GX_NOLOAD
    if (kt + 2 < nkt) request(kt + 2);
#endif
    const __bf16 *As = lds_g + (kt % NST) * STAGE, *Bs = As + BM * BK;
    // fragments of k16 step ks + 1 are requested before step ks is multiplied (two register sets): the LDS latency of a step hides
    // behind the previous step's MFMAs instead of in front of its own
    bf16x8 fa[2], fb[2][TN];
    auto frags = [&](int ks, int set) {
      const int c = 2 * ks + half;                           // this lane's chunk of the k16 step: k = 16 ks + 8 (lane / 32) ..
      fa[set] = *reinterpret_cast<const bf16x8 *>(As + arow * BK + ((c ^ ax) << 3));
#pragma unroll
      for (int tn = 0; tn < TN; tn++) {
        const int brow = wn * (BN / WN) + tn * 32 + (lane & 31);
        fb[set][tn] = *reinterpret_cast<const bf16x8 *>(Bs + brow * BK + ((c ^ ((brow >> 1) & 7)) << 3));
      }
    };
    frags(0, 0);
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      if (ks < 3) frags(ks + 1, (ks + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);                      // (the scheduler would sink the reads back in front of their own MFMAs)
#ifdef SS_GX_NOMFMA
#pragma unroll
      for (int tn = 0; tn < TN; tn++) acc[tn][0] += (float)fa[ks & 1][0] * (float)fb[ks & 1][tn][0];
#else
#pragma unroll
      for (int tn = 0; tn < TN; tn++) acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks & 1], fb[ks & 1][tn], acc[tn], 0, 0, 0);
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // (Round 5 also built the deeper pipeline — a whole tile's fragments of tile t + 1 read during tile t's MFMAs, three tiles in flight,
  // the copy instructions spread behind the MFMAs: correct, and not faster (35.1-35.5 us against 34.6 on the 2048 -> 1536 layer).  With
  // the K loop's copies removed the same loop runs 28.3 us: what is left is the L2 -> LDS traffic itself, 335 MB per GEMM at 128 x 192
  // tiles = ~12 TB/s over the eight L2s; fewer bytes per flop need 256-wide tiles, of which this layer has 96-128 for 256 CUs.
  // profiles/r05_mlp_gemm.txt)
  // ---- epilogue: bias + activation, then the tile goes through LDS (the stages are free) so that every lane stores 16 contiguous
  // bytes of an output row (the MFMA's C layout puts one column per lane: 2-byte stores, 48 store instructions per wave before)
  float bv[TN];
#pragma unroll
  for (int tn = 0; tn < TN; tn++) {
    const int col = n0 + wn * (BN / WN) + tn * 32 + (lane & 31);
    bv[tn] = (bias && col < N) ? bias[col] : 0.f;
  }
  if constexpr (F32OUT) {
    with_activation(act, [&](auto fn) {
#pragma unroll
      for (int tn = 0; tn < TN; tn++) {
        const int col = n0 + wn * (BN / WN) + tn * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (row < M && col < N) reinterpret_cast<float *>(Y)[(size_t)row * ldy + col] = fn(acc[tn][r] + bv[tn]);
        }
      }
    });
  } else {
    constexpr int CS = BN + 8;                               // row stride of the staged tile (16 bytes of padding: the column-per-lane writes spread over the banks)
    __bf16 *Cs = lds_g;
    __builtin_amdgcn_s_barrier();                             // every wave has multiplied the last tile: the stages may be overwritten
    with_activation(act, [&](auto fn) {
#pragma unroll
      for (int tn = 0; tn < TN; tn++) {
        const int cl = wn * (BN / WN) + tn * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int rl = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          Cs[rl * CS + cl] = (__bf16)fn(acc[tn][r] + bv[tn]);
        }
      }
    });
    __syncthreads();
    __bf16 *Yb = reinterpret_cast<__bf16 *>(Y);
    const bool vec_ok = (ldy & 7) == 0 && (reinterpret_cast<size_t>(Yb) & 15) == 0;
    constexpr int CPR = BN / 8;                               // 16-byte chunks per tile row
#pragma unroll
    for (int i = 0; i < BM * CPR / 512; i++) {
      const int id = tid + 512 * i, rl = id / CPR, cc = (id % CPR) * 8, row = m0 + rl, col = n0 + cc;
      if (row >= M || col >= N) continue;
      const u32x4 v = *reinterpret_cast<const u32x4 *>(Cs + rl * CS + cc);
      if (vec_ok && col + 8 <= N) *reinterpret_cast<u32x4 *>(Yb + (size_t)row * ldy + col) = v;
      else {
        const __bf16 *e = reinterpret_cast<const __bf16 *>(&v);
        for (int j = 0; j < 8 && col + j < N; j++) Yb[(size_t)row * ldy + col + j] = e[j];
      }
    }
  }
}

// ---- round 6: the GEMM of the PPO update (reference agent_ppo.py:20-83 runs the same three products per layer through
// autograd).  The asynchronous-copy K loop of ss_linear_glds_kernel with (a) a K split over blockIdx.z for products whose output is small and whose
// contraction is the batch (dW = dZ^T h: 2048 x 1536 outputs over K = 53248 rows), partial sums added to the fp32 output by hardware atomics;
// (b) an epilogue that multiplies by a second operand (the stored activation derivative: dZ = (dZ' W) * act'(z)) and / or applies the activation;
// (c) up to three bf16 outputs of the same tile staged through LDS at once: the result, its TRANSPOSE (every product of the backward pass
// contracts over what the forward pass has as rows: with h^T and dZ^T written here, all three products of a layer are the one K-contiguous
// "x W^T" kernel — no transposing loads, no separate transpose launches), and the activation's derivative at the pre-activation.
struct LinearTrainArgs {
  const __bf16 *X, *W;          // [M, K], [N, K] row-major, K a multiple of 64
  const float *bias;            // [N] or null
  const __bf16 *mul;            // [M, ldy] or null: the result is multiplied by it before the activation
  void *Y;                      // [M, ldy] bf16 (or fp32 when f32_atomic: += partial sums; the caller zeroes it) or null
  __bf16 *Yt;                   // [N, ldyt] transposed copy or null
  __bf16 *Dact;                 // [M, ldy] act'(pre-activation) or null
  int M, N, K, ldy, ldyt, act, xcd_remap, ksplit;
  int kper = 0;                 // ss_gemm256_kernel: K tiles per share (even; the host's number, not re-derived from ksplit)
  float *colsum = nullptr;      // ss_gemm256_kernel with `mul`: [N] += column sums of the fp32 result (the bias gradient of the layer below)
};

template <int BN, bool F32ATOMIC>
__global__ void __launch_bounds__(512) ss_linear_train_kernel(const LinearTrainArgs a) {
  constexpr int BK = 64, WM = 4, WN = 2, NST = 3;
  constexpr int TN = BN / (32 * WN);
  constexpr int STAGE = (BM + BN) * BK;
  constexpr int NI = (BM + BN) / 64;
  extern __shared__ __attribute__((aligned(16))) __bf16 lds_g[];
  const __bf16 *__restrict__ X = a.X, *__restrict__ W = a.W;
  const int M = a.M, N = a.N, K = a.K;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave / WN, wn = wave % WN;
  int bx = blockIdx.x, by = blockIdx.y;
  {
    const int gx = gridDim.x, gy = gridDim.y;
    if (a.xcd_remap && gy % 8 == 0) {
      const int id = by * gx + bx, xcd = id & 7, idx = id >> 3, rows_per = gy >> 3;
      by = xcd * rows_per + idx % rows_per;
      bx = idx / rows_per;
    }
  }
  const int m0 = by * BM, n0 = bx * BN;
  f32x16 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; j++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
  const __bf16 *src[NI];
  int dst[NI];
#pragma unroll
  for (int i = 0; i < NI; i++) {
    const int blk = wave + 8 * i, row = 8 * blk + (lane >> 3), chunk = (lane & 7) ^ ((row >> 1) & 7);
    if (row < BM) { const int g = m0 + row; src[i] = X + (size_t)(g < M ? g : M - 1) * K + chunk * 8; }
    else { const int g = n0 + row - BM; src[i] = W + (size_t)(g < N ? g : N - 1) * K + chunk * 8; }
    dst[i] = 8 * blk * BK;
  }
  // this workgroup's share of the K tiles
  const int nkt_all = K / BK, per = (nkt_all + a.ksplit - 1) / a.ksplit, kt0 = (int)blockIdx.z * per, kt1 = kt0 + per < nkt_all ? kt0 + per : nkt_all;
  const int nkt = kt1 - kt0;
  if (nkt <= 0) return;
  auto request = [&](int t) {
    __bf16 *stage = lds_g + (t % NST) * STAGE;
#pragma unroll
    for (int i = 0; i < NI; i++) __builtin_amdgcn_global_load_lds((ss_gvoid *)(src[i] + (size_t)(kt0 + t) * BK), (ss_lvoid *)(stage + dst[i]), 16, 0, 0);
  };
  request(0);
  if (nkt > 1) request(1);
  const int arow = wm * 32 + (lane & 31), ax = (arow >> 1) & 7, half = lane >> 5;
  for (int kt = 0; kt < nkt; kt++) {
    if (kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 2 < nkt) request(kt + 2);
    const __bf16 *As = lds_g + (kt % NST) * STAGE, *Bs = As + BM * BK;
    bf16x8 fa[2], fb[2][TN];
    auto frags = [&](int ks, int set) {
      const int c = 2 * ks + half;
      fa[set] = *reinterpret_cast<const bf16x8 *>(As + arow * BK + ((c ^ ax) << 3));
#pragma unroll
      for (int tn = 0; tn < TN; tn++) {
        const int brow = wn * (BN / WN) + tn * 32 + (lane & 31);
        fb[set][tn] = *reinterpret_cast<const bf16x8 *>(Bs + brow * BK + ((c ^ ((brow >> 1) & 7)) << 3));
      }
    };
    frags(0, 0);
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      if (ks < 3) frags(ks + 1, (ks + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int tn = 0; tn < TN; tn++) acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks & 1], fb[ks & 1][tn], acc[tn], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if constexpr (F32ATOMIC) {
    // partial sums of this K share: hardware fp32 atomics, 32 adjacent columns per wave instruction (the bias, if any, is added by share 0)
    float *Yf = reinterpret_cast<float *>(a.Y);
#pragma unroll
    for (int tn = 0; tn < TN; tn++) {
      const int col = n0 + wn * (BN / WN) + tn * 32 + (lane & 31);
      const float bv = (a.bias && col < N && blockIdx.z == 0) ? a.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < M && col < N) unsafeAtomicAdd(Yf + (size_t)row * a.ldy + col, acc[tn][r] + bv);
      }
    }
  } else {
    // up to three bf16 images of the tile, one after the other through the same LDS (rows | derivative rows | transposed): each pass
    // re-evaluates the epilogue arithmetic from the accumulators (~10 instructions per element against a K loop of thousands of cycles)
    constexpr int CS = BN + 8, CST = BM + 8;
    __bf16 *Cs = lds_g;
    const bool want_d = a.Dact != nullptr, want_t = a.Yt != nullptr;
    constexpr int CPR = BN / 8;
    auto value = [&](int tn, int r, float &v) {              // bias, multiplying operand; returns through v the pre-activation
      const int cl = wn * (BN / WN) + tn * 32 + (lane & 31), col = n0 + cl;
      const int rl = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), row = m0 + rl;
      v = acc[tn][r] + ((a.bias && col < N) ? a.bias[col] : 0.f);
      if (a.mul) v *= (row < M && col < N) ? (float)a.mul[(size_t)row * a.ldy + col] : 0.f;
    };
    auto rows_out = [&](__bf16 *Yb) {
      const bool vec_ok = (a.ldy & 7) == 0 && (reinterpret_cast<size_t>(Yb) & 15) == 0;
#pragma unroll
      for (int i = 0; i < BM * CPR / 512; i++) {
        const int id = tid + 512 * i, rl = id / CPR, cc = (id % CPR) * 8, row = m0 + rl, col = n0 + cc;
        if (row >= M || col >= N) continue;
        const u32x4 v = *reinterpret_cast<const u32x4 *>(Cs + rl * CS + cc);
        if (vec_ok && col + 8 <= N) *reinterpret_cast<u32x4 *>(Yb + (size_t)row * a.ldy + col) = v;
        else {
          const __bf16 *e = reinterpret_cast<const __bf16 *>(&v);
          for (int j = 0; j < 8 && col + j < N; j++) Yb[(size_t)row * a.ldy + col + j] = e[j];
        }
      }
    };
    // the multiplying operand is read once: fold it (and the bias) into the accumulators
#pragma unroll
    for (int tn = 0; tn < TN; tn++)
#pragma unroll
      for (int r = 0; r < 16; r++) { float v; value(tn, r, v); acc[tn][r] = v; }
    if (a.Y) {
      __builtin_amdgcn_s_barrier();
      with_activation(a.act, [&](auto fn) {
#pragma unroll
        for (int tn = 0; tn < TN; tn++)
#pragma unroll
          for (int r = 0; r < 16; r++)
            Cs[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * CS + wn * (BN / WN) + tn * 32 + (lane & 31)] = (__bf16)fn(acc[tn][r]);
      });
      __syncthreads();
      rows_out(reinterpret_cast<__bf16 *>(a.Y));
    }
    if (want_d) {
      __syncthreads();
#pragma unroll
      for (int tn = 0; tn < TN; tn++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const float v = acc[tn][r];
          float d = 1.f;
          if (a.act == SS_ACT_SILU) { const float sg = __builtin_amdgcn_rcpf(1.f + __expf(-v)); d = sg * (1.f + v * (1.f - sg)); }
          else if (a.act == SS_ACT_TANH) { const float e = __expf(-2.f * fabsf(v)); const float t = (1.f - e) * __builtin_amdgcn_rcpf(1.f + e); d = 1.f - t * t; }
          else if (a.act == SS_ACT_RELU) d = v > 0.f ? 1.f : 0.f;
          Cs[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * CS + wn * (BN / WN) + tn * 32 + (lane & 31)] = (__bf16)d;
        }
      __syncthreads();
      rows_out(a.Dact);
    }
    if (want_t) {
      __bf16 *Ct = lds_g;
      __syncthreads();
      with_activation(a.act, [&](auto fn) {
#pragma unroll
        for (int tn = 0; tn < TN; tn++)
#pragma unroll
          for (int r = 0; r < 16; r++)
            Ct[(wn * (BN / WN) + tn * 32 + (lane & 31)) * CST + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)] = (__bf16)fn(acc[tn][r]);
      });
      __syncthreads();
      constexpr int RPC = BM / 8;                             // 16-byte chunks per transposed tile row (a column of the result: 128 rows)
      const bool vec_ok = (a.ldyt & 7) == 0 && (reinterpret_cast<size_t>(a.Yt) & 15) == 0;
#pragma unroll
      for (int i = 0; i < BN * RPC / 512; i++) {
        const int id = tid + 512 * i, cl = id / RPC, rc = (id % RPC) * 8, col = n0 + cl, row = m0 + rc;
        if (col >= N || row >= M) continue;
        const u32x4 v = *reinterpret_cast<const u32x4 *>(Ct + cl * CST + rc);
        if (vec_ok && row + 8 <= M) *reinterpret_cast<u32x4 *>(a.Yt + (size_t)col * a.ldyt + row) = v;
        else {
          const __bf16 *e = reinterpret_cast<const __bf16 *>(&v);
          for (int j = 0; j < 8 && row + j < M; j++) a.Yt[(size_t)col * a.ldyt + row + j] = e[j];
        }
      }
    }
  }
}

// ---- round 6, second GEMM of the update: 256 x 256 macro-tiles, the two wave rows one barrier apart (ss_gemm256.h has the K loop and the
// ordering argument).  Arguments as ss_linear_train_kernel; the set of outputs is a template parameter:
//     G256_ACCUM   fp32 partial sums of a K share added to Y by atomics (weight gradients)
//     G256_PLAIN   Y = act(x W^T + b)
//     G256_FWD     Y, Y^T and act'(pre-activation)                  (a hidden layer's forward pass)
//     G256_DX      Y = (x W^T) * mul and Y^T                        (dZ of the layer below)
//     G256_FWDN / G256_DXN   the same two without Y^T: since ss_wgrad_bf16 contracts over the ROWS of dZ and h, nothing needs a transposed copy
// Why compile-time: the K loop alone runs the 53 248 x 1536 x 2048 product in 256 us = 1.31 PFLOP/s; the first epilogue (run-time `if (Y)`, `if (Dact)` per
// element, one dependent exp -> rcp chain after the other between the branches, every 16-byte chunk of the output parked in scratch because its edge
// path indexed it dynamically) cost 90 us for ONE image and 200 us for three (profiles/r06_gemm256.txt).
enum { G256_ACCUM = 0, G256_PLAIN = 1, G256_FWD = 2, G256_DX = 3, G256_FWDN = 4, G256_DXN = 5 };   // ..N: without the transposed image (ss_wgrad_bf16 reads the operands as they lie)

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  union { __bf16 h[2]; unsigned u; } c;
  c.h[0] = (__bf16)lo; c.h[1] = (__bf16)hi;
  return c.u;
}
// the first n (< 8) elements of a 16-byte chunk, 2 bytes at a time (static indices: a dynamically indexed chunk lives in scratch)
__device__ __forceinline__ void store_chunk_edge(__bf16 *dst, const u32x4 &v, int n) {
#pragma unroll
  for (int j = 0; j < 8; j++)
    if (j < n) reinterpret_cast<unsigned short *>(dst)[j] = (unsigned short)((j & 1) ? (v[j >> 1] >> 16) : (v[j >> 1] & 0xffffu));
}
__device__ __forceinline__ u32x4 load_chunk_edge(const __bf16 *src, int n) {
  u32x4 v = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int j = 0; j < 8; j++)
    if (j < n) v[j >> 1] |= (unsigned)reinterpret_cast<const unsigned short *>(src)[j] << (16 * (j & 1));
  return v;
}

template <int MODE>
__global__ void __launch_bounds__(512) ss_gemm256_kernel(const LinearTrainArgs a) {
  extern __shared__ __attribute__((aligned(16))) __bf16 lds_g[];
  constexpr int T = gemm256::TILE;
  const int M = a.M, N = a.N, K = a.K;
  int bx = blockIdx.x, by = blockIdx.y;
  {
    const int gx = gridDim.x, gy = gridDim.y;
    if ((a.xcd_remap & 1) && gy % 8 == 0) {
      const int id = by * gx + bx, xcd = id & 7, idx = id >> 3, rows_per = gy >> 3;
      by = xcd * rows_per + idx % rows_per;
      bx = idx / rows_per;
    }
  }
  const int m0 = by * T, n0 = bx * T;
  const int nkt_all = K / 64, per = a.kper > 0 ? a.kper : nkt_all, kt0 = (int)blockIdx.z * per, kt1 = kt0 + per < nkt_all ? kt0 + per : nkt_all;
  const int nkt = kt1 - kt0;
  if (nkt < 2 || (nkt & 1)) return;                           // (the host cuts K into shares of an even number of tiles)
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
  gemm256::Loop L;
  L.init(a.X, a.W, M, N, K, m0, n0, kt0, reinterpret_cast<char *>(lds_g));
  L.run(acc, nkt);
#ifdef SS_GEMM256_ABLATE
  if (a.xcd_remap & 2) {                                      // K loop only: one store per lane keeps the accumulators alive
    float sacc = 0.f;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 2; j++) for (int r = 0; r < 16; r++) sacc += acc[i][j][r];
    if (sacc == 12345.f) reinterpret_cast<__bf16 *>(a.Y)[threadIdx.x] = (__bf16)sacc;
    return;
  }
#endif
  const int tid = threadIdx.x, lane = L.lane, wr = L.wr, wc = L.wc;
  const int col_l = wc * 64 + (lane & 31), row_l = wr * 128 + 4 * (lane >> 5);   // + tn * 32 resp. + tm * 32 + (r & 3) + 8 * (r >> 2)
  if constexpr (MODE == G256_ACCUM) {
    float *Yf = reinterpret_cast<float *>(a.Y);
#pragma unroll
    for (int tn = 0; tn < 2; tn++) {
      const int col = n0 + col_l + tn * 32;
      const float bv = (a.bias && col < N && blockIdx.z == 0) ? a.bias[col] : 0.f;
#pragma unroll
      for (int tm = 0; tm < 4; tm++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int row = m0 + row_l + tm * 32 + (r & 3) + 8 * (r >> 2);
          if (row < M && col < N) unsafeAtomicAdd(Yf + (size_t)row * a.ldy + col, acc[tm][tn][r] + bv);
        }
    }
  } else {
    constexpr bool HAS_MUL = MODE == G256_DX || MODE == G256_DXN, HAS_D = MODE == G256_FWD || MODE == G256_FWDN, HAS_T = MODE == G256_FWD || MODE == G256_DX;
    constexpr int CS = T + 8, CPR = T / 8;
    constexpr int HALF_IMG = 128 * CS;                        // elements of one half image (128 rows)
    __bf16 *Cs = lds_g;
    __bf16 *Yb = reinterpret_cast<__bf16 *>(a.Y);
    const bool rows_vec = (a.ldy & 7) == 0 && (reinterpret_cast<size_t>(Yb) & 15) == 0 && (!HAS_D || (reinterpret_cast<size_t>(a.Dact) & 15) == 0) &&
                          (!HAS_MUL || (reinterpret_cast<size_t>(a.mul) & 15) == 0);
    // ---- the multiplying operand: the tile by 16-byte row loads into LDS, from there into the accumulators
    if constexpr (HAS_MUL) {
      if (m0 + T <= M && n0 + T <= N && rows_vec) {
        // a tile inside the matrix: eight loads in flight, NO control flow between them.  With the bounds tests around every load the compiler put each load in
        // its own branch region and waited (vmcnt(0)) before entering the next: 16 HBM round trips one after the other, 30 us per tile, +110 us on a 53 248 x 1024
        // product (profiles/r06_gemm256.txt)
        const __bf16 *src = a.mul + (size_t)(m0 + tid / CPR) * a.ldy + n0 + (tid % CPR) * 8;
        __bf16 *dst = Cs + (tid / CPR) * CS + (tid % CPR) * 8;
        const size_t rstep = (size_t)(512 / CPR) * a.ldy;     // 512 threads cover 16 rows per step
#pragma unroll
        for (int i0 = 0; i0 < T * CPR / 512; i0 += 8) {
          u32x4 v[8];
#pragma unroll
          for (int i = 0; i < 8; i++) v[i] = *reinterpret_cast<const u32x4 *>(src + (size_t)(i0 + i) * rstep);
#pragma unroll
          for (int i = 0; i < 8; i++) *reinterpret_cast<u32x4 *>(dst + (i0 + i) * (512 / CPR) * CS) = v[i];
        }
      } else {
#pragma unroll 1
        for (int i = 0; i < T * CPR / 512; i++) {
          const int id = tid + 512 * i, rl = id / CPR, cc = (id % CPR) * 8, row = m0 + rl, col = n0 + cc;
          u32x4 v = {0u, 0u, 0u, 0u};
          if (row < M && col < N) {
            const __bf16 *src = a.mul + (size_t)row * a.ldy + col;
            if (rows_vec && col + 8 <= N) v = *reinterpret_cast<const u32x4 *>(src);
            else v = load_chunk_edge(src, N - col);
          }
          *reinterpret_cast<u32x4 *>(Cs + rl * CS + cc) = v;
        }
      }
      __syncthreads();
    }
#pragma unroll
    for (int tn = 0; tn < 2; tn++) {
      const int col = n0 + col_l + tn * 32;
      const float bv = (a.bias && col < N) ? a.bias[col] : 0.f;
#pragma unroll
      for (int tm = 0; tm < 4; tm++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          float v = acc[tm][tn][r] + bv;
          if constexpr (HAS_MUL) v *= (float)Cs[(row_l + tm * 32 + (r & 3) + 8 * (r >> 2)) * CS + col_l + tn * 32];
          acc[tm][tn][r] = v;
        }
    }
    if constexpr (HAS_MUL) {
      if (a.colsum) {
        // bias gradient of the layer below: the column sums of dZ, from the fp32 values in the accumulators (a lane holds 64 rows of each of its two columns;
        // its partner 32 lanes on holds the other 64 of this wave's 128), one atomic per column and wave
#pragma unroll
        for (int tn = 0; tn < 2; tn++) {
          float sum = 0.f;
#pragma unroll
          for (int tm = 0; tm < 4; tm++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
              const int row = m0 + row_l + tm * 32 + (r & 3) + 8 * (r >> 2);
              sum += row < M ? acc[tm][tn][r] : 0.f;
            }
          sum += __shfl_xor(sum, 32, 64);
          const int col = n0 + col_l + tn * 32;
          if (lane < 32 && col < N) unsafeAtomicAdd(a.colsum + col, sum);
        }
      }
      __syncthreads();
    }
    // ---- result (and derivative) in two halves of the tile — each wave's upper 64 rows, then its lower 64 — so that a half's two images sit in
    // LDS side by side and the exponential is evaluated ONCE per element; the accumulators keep the activated values for the transposed image
    auto half_rows = [&](__bf16 *dst, const __bf16 *img, int half) {
      u32x4 v[8];
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const int id = tid + 512 * i, lr = id / CPR, cc = (id % CPR) * 8;
        v[i] = *reinterpret_cast<const u32x4 *>(img + lr * CS + cc);
      }
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const int id = tid + 512 * i, lr = id / CPR, cc = (id % CPR) * 8, row = m0 + (lr >> 6) * 128 + half * 64 + (lr & 63), col = n0 + cc;
        if (row >= M || col >= N) continue;
#ifdef SS_GEMM256_ABLATE
        if ((a.xcd_remap & 4) && v[i][0] != 0x12345678u) continue;
#endif
        if (rows_vec && col + 8 <= N) *reinterpret_cast<u32x4 *>(dst + (size_t)row * a.ldy + col) = v[i];
        else store_chunk_edge(dst + (size_t)row * a.ldy + col, v[i], N - col);
      }
    };
    auto halves = [&](auto fn2) {
#pragma unroll
      for (int half = 0; half < 2; half++) {
#pragma unroll
        for (int tml = 0; tml < 2; tml++)
#pragma unroll
          for (int tn = 0; tn < 2; tn++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
              float h, d;
              fn2(acc[2 * half + tml][tn][r], h, d);
              acc[2 * half + tml][tn][r] = h;
              const int at = (wr * 64 + tml * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * CS + col_l + tn * 32;
              Cs[at] = (__bf16)h;
              if constexpr (HAS_D) Cs[HALF_IMG + at] = (__bf16)d;
            }
        __syncthreads();
        half_rows(Yb, Cs, half);
        if constexpr (HAS_D) half_rows(a.Dact, Cs + HALF_IMG, half);
        __syncthreads();
      }
    };
    if (a.act == SS_ACT_SILU) halves([](float v, float &h, float &d) { const float sg = __builtin_amdgcn_rcpf(1.f + __expf(-v)); h = v * sg; d = sg * (1.f + v * (1.f - sg)); });
    else if (a.act == SS_ACT_TANH) halves([](float v, float &h, float &d) { const float e = __expf(-2.f * fabsf(v)); const float t = (1.f - e) * __builtin_amdgcn_rcpf(1.f + e); h = v < 0.f ? -t : t; d = 1.f - t * t; });
    else if (a.act == SS_ACT_RELU) halves([](float v, float &h, float &d) { h = v > 0.f ? v : 0.f; d = v > 0.f ? 1.f : 0.f; });
    else halves([](float v, float &h, float &d) { h = v; d = 1.f; });
    // ---- transposed image: an accumulator's registers r .. r + 3 are four consecutive rows of one column = 8 contiguous bytes of it
    if constexpr (HAS_T) {
      typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
#pragma unroll
      for (int tm = 0; tm < 4; tm++)
#pragma unroll
        for (int tn = 0; tn < 2; tn++)
#pragma unroll
          for (int r = 0; r < 16; r += 4) {
            u32x2 v;
            v[0] = pack_bf16x2(acc[tm][tn][r], acc[tm][tn][r + 1]); v[1] = pack_bf16x2(acc[tm][tn][r + 2], acc[tm][tn][r + 3]);
            *reinterpret_cast<u32x2 *>(Cs + (col_l + tn * 32) * CS + row_l + tm * 32 + 8 * (r >> 2)) = v;
          }
      __syncthreads();
      const bool cols_vec = (a.ldyt & 7) == 0 && (reinterpret_cast<size_t>(a.Yt) & 15) == 0;
#pragma unroll
      for (int i0 = 0; i0 < T * CPR / 512; i0 += 8) {
        u32x4 v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const int id = tid + 512 * (i0 + i), cl = id / CPR, rc = (id % CPR) * 8;
          v[i] = *reinterpret_cast<const u32x4 *>(Cs + cl * CS + rc);
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const int id = tid + 512 * (i0 + i), cl = id / CPR, rc = (id % CPR) * 8, col = n0 + cl, row = m0 + rc;
          if (col >= N || row >= M) continue;
          if (cols_vec && row + 8 <= M) *reinterpret_cast<u32x4 *>(a.Yt + (size_t)col * a.ldyt + row) = v[i];
          else store_chunk_edge(a.Yt + (size_t)col * a.ldyt + row, v[i], M - row);
        }
      }
    }
  }
}

// ---- round 6: the weight gradient from dZ and the layer's input AS THEY LIE (both [batch, features] row-major): dW[i][j] += sum_m dZ[m][i] h[m][j].  The forward and dX
// products of the update are bound by what they WRITE (profiles/r06_gemm256.txt); with this kernel they need not write transposed copies any more.  K loop:
// gemm256::LoopTN (fragments by ds_read_b64_tr_b16); K split over one round of the CUs, fp32 atomics, as ss_gemm256_kernel<G256_ACCUM>.
struct WgradArgs {
  const __bf16 *Z, *H;          // [Mb, ldz], [Mb, ldh]
  float *dW;                    // [NI, ldw] += ; NI = columns of Z used, NJ = columns of H used
  int NI, NJ, ldz, ldh, ldw, nkt, kper;
};

__global__ void __launch_bounds__(512) ss_wgrad_tn_kernel(const WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) __bf16 lds_g[];
  const int i0 = blockIdx.y * 256, j0 = blockIdx.x * 256;
  const int kt0 = (int)blockIdx.z * a.kper, kt1 = kt0 + a.kper < a.nkt ? kt0 + a.kper : a.nkt, nkt = kt1 - kt0;
  if (nkt < 2 || (nkt & 1)) return;
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
  gemm256::LoopTN L;
  L.init(a.Z, a.H, a.ldz, a.ldh, a.NI, a.NJ, i0, j0, kt0, reinterpret_cast<char *>(lds_g));
  L.run(acc, nkt);
  const int lane = L.lane;
#pragma unroll
  for (int tn = 0; tn < 2; tn++) {
    const int col = j0 + L.wc * 64 + tn * 32 + (lane & 31);
#pragma unroll
    for (int tm = 0; tm < 4; tm++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = i0 + L.wr * 128 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < a.NI && col < a.NJ) unsafeAtomicAdd(a.dW + (size_t)row * a.ldw + col, acc[tm][tn][r]);
      }
  }
}

// torch.clamp semantics: a NaN stays a NaN (fminf / fmaxf would return the bound and hide a diverged policy or observation from the env)
__device__ __forceinline__ float clamp_keep_nan(float v, float lo, float hi) { return v != v ? v : fminf(fmaxf(v, lo), hi); }

__global__ void __launch_bounds__(256) ss_obs_to_bf16_kernel(const float *obs, int M, int dim, int stride, const float *mean, const float *sd,
                                                             const long long *n, float lo, float hi, float clip, __bf16 *out, int kpad) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)M * kpad) return;
  const int row = (int)(idx / kpad), c = (int)(idx % kpad);
  float v = 0.f;
  if (c < dim) {
    v = clamp_keep_nan(obs[(size_t)row * stride + c], lo, hi);
    if (mean && sd && n && *n > 0) v = clamp_keep_nan((v - mean[c]) / (sd[c] + 1e-8f), -clip, clip);
  }
  out[idx] = (__bf16)v;
}

// Gaussian policy head of the sampler, one wavefront per env row: a = mean + exp(log_std) * noise (the product and the sum rounded
// separately, like the torch expression it replaces), its clipped copy for the env, and the log-density of the draw.
__global__ void __launch_bounds__(256) ss_gaussian_sample_kernel(const float *mean, const float *noise, const float *log_std, int M, int dim,
                                                                 float *action, int lda, float *action_env, int lde, float lo, float hi, float *logp) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  float acc = 0.f;
  for (int j = lane; j < dim; j += 64) {
    const float ls = log_std[j], z = noise[(size_t)row * dim + j];
    const float a = __fadd_rn(mean[(size_t)row * dim + j], __fmul_rn(__expf(ls), z));
    action[(size_t)row * lda + j] = a;
    if (action_env) action_env[(size_t)row * lde + j] = clamp_keep_nan(a, lo, hi);
    acc += -0.5f * z * z - 0.91893853320467274f - ls;          // - log sqrt(2 pi)
  }
  if (logp) {
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
    if (lane == 0) logp[row] = acc;
  }
}

int fail(int code, const char *msg) { ss::last_error() = msg; return code; }

}  // namespace

extern "C" {

int ss_linear_bf16(const void *x, const void *w, const float *bias, void *y, int32_t M, int32_t N, int32_t K, int32_t ldy, int32_t act,
                   int32_t y_is_f32, void *stream) {
  if (!x || !w || !y) return fail(SS_ERR_INVALID, "null argument");
  if (M < 1 || N < 1 || K < 32 || K % 32 || ldy < N) return fail(SS_ERR_INVALID, "ss_linear_bf16: K must be a positive multiple of 32, ldy >= N");
  if (act < SS_ACT_NONE || act > SS_ACT_RELU) return fail(SS_ERR_INVALID, "unknown activation");
  const __bf16 *X = static_cast<const __bf16 *>(x), *Wt = static_cast<const __bf16 *>(w);
  hipStream_t st = (hipStream_t)stream;
  // thousands of rows (the update's and GAE's passes over the whole rollout): the 256 x 256 kernel (ss_gemm256.h), 1.19 PFLOP/s on 53 248 x 1536 x 2048
  // against 0.60 for the 128-row tiles below; at the sampler's 4096 rows it has at most 128 tiles for 256 CUs and stays out
  {
    static const bool no256 = getenv("SS_MLP_NO256") != nullptr;
    const long long tiles = (long long)((M + 255) / 256) * ((N + 255) / 256);
    if (!no256 && !y_is_f32 && tiles >= 230 && N % 256 == 0 && K >= 256 && K % 128 == 0 && (long long)M * K < (1ll << 32) && (long long)N * K < (1ll << 32)) {
      LinearTrainArgs a{X, Wt, bias, nullptr, y, nullptr, nullptr, M, N, K, ldy, 0, act, 1, 1};
      auto kern_ = ss_gemm256_kernel<G256_PLAIN>;
      const size_t epi_ = (size_t)gemm256::TILE * (gemm256::TILE + 8) * sizeof(__bf16);
      if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern_), hipFuncAttributeMaxDynamicSharedMemorySize, (int)epi_) != hipSuccess) return fail(SS_ERR_HIP, "cannot size the GEMM's LDS");
      hipLaunchKernelGGL(kern_, dim3((N + 255) / 256, (M + 255) / 256, 1), dim3(512), epi_, st, a);
      hipError_t e = hipGetLastError();
      return e == hipSuccess ? SS_OK : fail(SS_ERR_HIP, hipGetErrorString(e));
    }
  }
  const int gm = (M + BM - 1) / BM;
  // Tile width: the one whose tile count fills the chip's 256 CUs in whole rounds (4096 x 2048 -> 256 columns, x 1536 -> 192, x 1024 ->
  // 128, x 512 -> 64: exactly one 128-row tile per CU each; round 3 used 128 x 128 for the two widest layers = 1.5 rounds of
  // workgroups, a third of the chip idle in the second), ties to the wider tile (more MFMAs per LDS byte and per barrier)
  static const char *force_bn = getenv("SS_MLP_BN");            // A/B switches (tools/gpu_mlp.py)
  static const bool force32 = getenv("SS_MLP_BK32") != nullptr;
  static const int remap = getenv("SS_MLP_NOREMAP") ? 0 : 1;
  static const bool waves8 = getenv("SS_MLP_WAVES4") == nullptr;   // 8 waves per workgroup (32 x BN/2 each): twice the waves per SIMD
  const bool k64 = K % 64 == 0 && !force32;
  int bn = 64;
  {
    double best = -1;
    const int cand[4] = {256, 192, 128, 64};
    for (int c = 0; c < 4; c++) {
      if ((!(k64 && waves8) || K < 512) && cand[c] > 128) continue;   // the wide tiles: 8-wave, K-tile-of-64 flavour only, and only for a deep K
                                                                      // (289 -> 2048 has 5 K tiles: the 256-column tile's longer prologue / epilogue made it 24 vs 18 us)
      if (cand[c] > 64 && N < cand[c]) continue;
      const long long tiles = (long long)((N + cand[c] - 1) / cand[c]) * gm, rounds = (tiles + 255) / 256;
      const double fill = (double)tiles / (double)(rounds * 256) * ((double)N / (double)(((N + cand[c] - 1) / cand[c]) * cand[c]));
      if (fill > best + 1e-9) { best = fill; bn = cand[c]; }
    }
    if (force_bn && atoi(force_bn) > 0) bn = atoi(force_bn);
  }
  static const bool glds = getenv("SS_MLP_NOGLDS") == nullptr;   // round 5: asynchronous global -> LDS copies, three K tiles in flight
#define SS_LAUNCH(BN_, BK_, F32_)                                                                                              \
  do {                                                                                                                         \
    dim3 grid((N + BN_ - 1) / BN_, gm);                                                                                        \
    const size_t loop_ = (size_t)2 * (BM + BN_) * (BK_ + 8) * sizeof(__bf16), tile_ = (size_t)BM * (BN_ + 8) * sizeof(__bf16);   \
    const size_t lds_ = loop_ > tile_ ? loop_ : tile_;   /* K loop's two stages; the epilogue stages the output tile in the same memory */ \
    if (waves8 && BK_ == 64 && glds && K >= 512) {   /* (a shallow K keeps the two-stage kernel: two workgroups per CU hide its short loop better) */                                                                                         \
      auto kern_ = ss_linear_glds_kernel<BN_, F32_>;                                                                           \
      const size_t l3_ = (size_t)3 * (BM + BN_) * 64 * sizeof(__bf16);                                                         \
      if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern_), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l3_) != hipSuccess) return fail(SS_ERR_HIP, "cannot size the GEMM's LDS"); \
      hipLaunchKernelGGL(kern_, grid, dim3(512), l3_, st, X, Wt, bias, y, M, N, K, ldy, act, remap);                            \
    } else if (waves8 && BK_ == 64) {                                                                                          \
      auto kern_ = ss_linear_kernel<BN_, BK_, F32_, 4>;                                                                        \
      /* per launch: the attribute belongs to the (kernel, device) pair and the call is a table write (a process-wide flag left */ \
      /* the wide tiles of a second GPU at the 64 KB default: ADVICE r4) */                                                     \
      if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern_), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_) != hipSuccess) return fail(SS_ERR_HIP, "cannot size the GEMM's LDS"); \
      hipLaunchKernelGGL(kern_, grid, dim3(512), lds_, st, X, Wt, bias, y, M, N, K, ldy, act, remap);                           \
    } else {                                                                                                                   \
      auto kern_ = ss_linear_kernel<(BN_ > 128 ? 128 : BN_), BK_, F32_, 2>;                                                     \
      if ((size_t)2 * (BM + (BN_ > 128 ? 128 : BN_)) * (BK_ + 8) * sizeof(__bf16) > 64 * 1024 &&                                  \
          hipFuncSetAttribute(reinterpret_cast<const void *>(kern_), hipFuncAttributeMaxDynamicSharedMemorySize,                \
                              (int)((size_t)2 * (BM + (BN_ > 128 ? 128 : BN_)) * (BK_ + 8) * sizeof(__bf16))) != hipSuccess) return fail(SS_ERR_HIP, "cannot size the GEMM's LDS"); \
      const size_t l4a_ = (size_t)2 * (BM + (BN_ > 128 ? 128 : BN_)) * (BK_ + 8) * sizeof(__bf16), l4b_ = (size_t)BM * ((BN_ > 128 ? 128 : BN_) + 8) * sizeof(__bf16); \
      hipLaunchKernelGGL(kern_, dim3((N + (BN_ > 128 ? 128 : BN_) - 1) / (BN_ > 128 ? 128 : BN_), gm), dim3(256),              \
                         l4a_ > l4b_ ? l4a_ : l4b_, st, X, Wt, bias, y, M, N, K, ldy, act, remap);                              \
    }                                                                                                                          \
  } while (0)
#define SS_PICK(BN_)                                                                                                           \
  do {                                                                                                                         \
    if (k64) { if (y_is_f32) SS_LAUNCH(BN_, 64, true); else SS_LAUNCH(BN_, 64, false); }                                       \
    else { if (y_is_f32) SS_LAUNCH(BN_, 32, true); else SS_LAUNCH(BN_, 32, false); }                                           \
  } while (0)
  if (bn == 256) SS_PICK(256);
  else if (bn == 192) SS_PICK(192);
  else if (bn == 128) SS_PICK(128);
  else SS_PICK(64);
#undef SS_PICK
#undef SS_LAUNCH
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SS_OK : fail(SS_ERR_HIP, hipGetErrorString(e));
}

static int linear_train_impl(const void *x, const void *w, const float *bias, const void *mul, void *y, void *yt, void *dact, int32_t M, int32_t N, int32_t K,
                             int32_t ldy, int32_t ldyt, int32_t act, int32_t y_is_f32_accumulate, float *colsum, void *stream) {
  if (!x || !w || (!y && !yt)) return fail(SS_ERR_INVALID, "null argument");
  if (M < 1 || N < 1 || K < 64 || K % 64 || (y && ldy < N) || (yt && ldyt < M)) return fail(SS_ERR_INVALID, "ss_linear_bf16_train: K must be a positive multiple of 64, ldy >= N, ldyt >= M");
  if (act < SS_ACT_NONE || act > SS_ACT_RELU) return fail(SS_ERR_INVALID, "unknown activation");
  if (y_is_f32_accumulate && (yt || dact || mul || act != SS_ACT_NONE || !y)) return fail(SS_ERR_INVALID, "ss_linear_bf16_train: the accumulating fp32 form has no other outputs, operand or activation");
  if ((mul || dact) && !y) return fail(SS_ERR_INVALID, "ss_linear_bf16_train: mul / dact share y's row stride: y must be given");
  hipStream_t st = (hipStream_t)stream;
  const int gm = (M + BM - 1) / BM;
  static const int remap = getenv("SS_MLP_NOREMAP") ? 0 : 1;
  // tile width as in ss_linear_bf16 (whole rounds of the 256 CUs), then the K split: products whose output has fewer tiles than CUs and a deep
  // contraction (the weight gradients: K = the batch) are cut along K into as many shares as fill about two rounds
  int bn = 64;
  {
    double best = -1;
    const int cand[4] = {256, 192, 128, 64};
    for (int c = 0; c < 4; c++) {
      if (cand[c] > 64 && N < cand[c]) continue;
      if (cand[c] > 128 && K < 512) continue;
      const long long tiles = (long long)((N + cand[c] - 1) / cand[c]) * gm, rounds = (tiles + 255) / 256;
      // (the accumulating form fills the chip with its K split: only the columns wasted by the last tile count)
      // ... and so does a batch of thousands of rows: there the widest tile wins on every product measured (profiles/r06_train_gemm_sweep.txt)
      const double fill = ((y_is_f32_accumulate || gm >= 64) ? 1.0 : (double)tiles / (double)(rounds * 256)) * ((double)N / (double)(((N + cand[c] - 1) / cand[c]) * cand[c]));
      if (fill > best + 1e-9) { best = fill; bn = cand[c]; }
    }
    static const char *force_bn = getenv("SS_MLP_TRAIN_BN");
    if (force_bn && atoi(force_bn) > 0) bn = atoi(force_bn);
    { const char *live = getenv("SS_MLP_TRAIN_BN_LIVE"); if (live && atoi(live) > 0) bn = atoi(live); }   // (tools/gpu_train_gemm_sweep.py: re-read per call)
  }
  int ksplit = 1;
  if (y_is_f32_accumulate) {
    const long long tiles = (long long)((N + bn - 1) / bn) * gm;
    const int nkt = K / 64;
    ksplit = (int)((512 + tiles - 1) / tiles);
    if (ksplit > nkt / 8) ksplit = nkt / 8 > 0 ? nkt / 8 : 1;   // at least 8 K tiles per share
    if (ksplit < 1) ksplit = 1;
    static const char *force_ks = getenv("SS_MLP_TRAIN_KSPLIT");
    if (force_ks && atoi(force_ks) > 0) ksplit = atoi(force_ks);
  }
  LinearTrainArgs a{static_cast<const __bf16 *>(x), static_cast<const __bf16 *>(w), bias, static_cast<const __bf16 *>(mul), y, static_cast<__bf16 *>(yt),
                    static_cast<__bf16 *>(dact), M, N, K, ldy, ldyt, act, remap, ksplit};
  // the 256 x 256 kernel (ss_gemm256.h): products with thousands of rows or a K split to fill the chip with; its source offsets are 32-bit
  // (measured, profiles/r06_gemm256.txt: thousands of rows -> 1.5-2 x the 128-row kernel; weight gradients with the batch as K -> 1.05-2.2 x once the K split fills ONE round)
  // the set of outputs it is built for (the other combinations keep the 128-row kernel)
  const int mode256 = y_is_f32_accumulate ? G256_ACCUM : (y && !mul && !yt && !dact) ? G256_PLAIN : (y && !mul && yt && dact) ? G256_FWD : (y && mul && yt && !dact) ? G256_DX :
                      (y && !mul && !yt && dact) ? G256_FWDN : (y && mul && !yt && !dact) ? G256_DXN : -1;
  bool big = mode256 >= 0 && N >= 256 && M >= 256 && K >= 128 && K % 128 == 0 && (long long)M * K < (1ll << 32) && (long long)N * K < (1ll << 32) &&
             (y_is_f32_accumulate ? K >= 8192 : M >= 2048);
  { const char *live = getenv("SS_MLP_TRAIN_256"); if (live) big = atoi(live) != 0 && mode256 >= 0 && K >= 128 && K % 128 == 0 && (long long)M * K < (1ll << 32) && (long long)N * K < (1ll << 32); }
  if (colsum && !(big && (mode256 == G256_DX || mode256 == G256_DXN)))
    return fail(SS_ERR_INVALID, "ss_linear_bf16_dx: column sums come from the 256 x 256 kernel only (M >= 2048, N >= 256, K a multiple of 128, mul and y given)");
  if (big) {
    a.colsum = colsum;
    constexpr int T = gemm256::TILE;
    const int gx = (N + T - 1) / T, gy = (M + T - 1) / T, nkt = K / 64;
    int ks = 1;
    if (y_is_f32_accumulate) {
      // whole rounds of the 256 CUs (one workgroup per CU): as many shares as fill ONE round — 28 tiles x 19 shares = 532 workgroups ran as three rounds
      int rounds = 1;
      { const char *rd = getenv("SS_MLP_TRAIN_ROUNDS"); if (rd && atoi(rd) > 0) rounds = atoi(rd); }
      ks = (int)(((long long)256 * rounds) / ((long long)gx * gy));
      if (ks < 1) ks = 1;
      if (ks > nkt / 8) ks = nkt / 8 > 0 ? nkt / 8 : 1;
      static const char *force_ks = getenv("SS_MLP_TRAIN_KSPLIT");
      if (force_ks && atoi(force_ks) > 0) ks = atoi(force_ks);
      if (ks > nkt / 2) ks = nkt / 2;
      int per = (nkt + ks - 1) / ks;
      per += per & 1;                                         // shares of an even number of K tiles (nkt is even: the last share too)
      ks = (nkt + per - 1) / per;
      a.kper = per;
    }
    a.ksplit = ks;
#ifdef SS_GEMM256_ABLATE
    { const char *ab = getenv("SS_GEMM256_KLOOP_ONLY"); if (ab && atoi(ab)) a.xcd_remap |= 2 * atoi(ab); }
#endif
    dim3 grid(gx, gy, ks);
    const size_t loop_ = gemm256::LOOP_LDS_BYTES, epi_ = (size_t)T * (T + 8) * sizeof(__bf16);
#define SS_G256(MODE_, LDS_)                                                                                                   \
    do {                                                                                                                       \
      auto kern_ = ss_gemm256_kernel<MODE_>;                                                                                   \
      if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern_), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS_)) != hipSuccess) return fail(SS_ERR_HIP, "cannot size the GEMM's LDS"); \
      hipLaunchKernelGGL(kern_, grid, dim3(512), LDS_, st, a);                                                                 \
    } while (0)
    if (y_is_f32_accumulate) SS_G256(G256_ACCUM, loop_);
    else if (mode256 == G256_PLAIN) SS_G256(G256_PLAIN, epi_);
    else if (mode256 == G256_FWD) SS_G256(G256_FWD, epi_);
    else if (mode256 == G256_DX) SS_G256(G256_DX, epi_);
    else if (mode256 == G256_FWDN) SS_G256(G256_FWDN, epi_);
    else SS_G256(G256_DXN, epi_);
#undef SS_G256
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SS_OK : fail(SS_ERR_HIP, hipGetErrorString(e));
  }
#define SS_TRAIN(BN_)                                                                                                          \
  do {                                                                                                                         \
    dim3 grid((N + BN_ - 1) / BN_, gm, ksplit);                                                                                \
    const size_t loop_ = (size_t)3 * (BM + BN_) * 64 * sizeof(__bf16);                                                         \
    const size_t epi_ = ((size_t)BM * (BN_ + 8) > (size_t)BN_ * (BM + 8) ? (size_t)BM * (BN_ + 8) : (size_t)BN_ * (BM + 8)) * sizeof(__bf16); \
    const size_t lds_ = loop_ > epi_ ? loop_ : epi_;                                                                           \
    if (y_is_f32_accumulate) {                                                                                                 \
      auto kern_ = ss_linear_train_kernel<BN_, true>;                                                                          \
      if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern_), hipFuncAttributeMaxDynamicSharedMemorySize, (int)loop_) != hipSuccess) return fail(SS_ERR_HIP, "cannot size the GEMM's LDS"); \
      hipLaunchKernelGGL(kern_, grid, dim3(512), loop_, st, a);                                                                \
    } else {                                                                                                                   \
      auto kern_ = ss_linear_train_kernel<BN_, false>;                                                                         \
      if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern_), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_) != hipSuccess) return fail(SS_ERR_HIP, "cannot size the GEMM's LDS"); \
      hipLaunchKernelGGL(kern_, grid, dim3(512), lds_, st, a);                                                                 \
    }                                                                                                                          \
  } while (0)
  if (bn == 256) SS_TRAIN(256);
  else if (bn == 192) SS_TRAIN(192);
  else if (bn == 128) SS_TRAIN(128);
  else SS_TRAIN(64);
#undef SS_TRAIN
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SS_OK : fail(SS_ERR_HIP, hipGetErrorString(e));
}

int ss_linear_bf16_train(const void *x, const void *w, const float *bias, const void *mul, void *y, void *yt, void *dact, int32_t M, int32_t N, int32_t K,
                         int32_t ldy, int32_t ldyt, int32_t act, int32_t y_is_f32_accumulate, void *stream) {
  return linear_train_impl(x, w, bias, mul, y, yt, dact, M, N, K, ldy, ldyt, act, y_is_f32_accumulate, nullptr, stream);
}

int ss_linear_bf16_dx(const void *x, const void *w, const void *mul, void *y, float *colsum, int32_t M, int32_t N, int32_t K, int32_t ldy, void *stream) {
  if (!mul || !y || !colsum) return fail(SS_ERR_INVALID, "null argument");
  return linear_train_impl(x, w, nullptr, mul, y, nullptr, nullptr, M, N, K, ldy, 0, SS_ACT_NONE, 0, colsum, stream);
}

int ss_wgrad_bf16(const void *dz, const void *h, float *dw, int32_t Mb, int32_t n_out, int32_t n_in, int32_t ldz, int32_t ldh, int32_t ldw, void *stream) {
  if (!dz || !h || !dw) return fail(SS_ERR_INVALID, "null argument");
  if (Mb < 128 || Mb % 128 || n_out < 1 || n_in < 1 || ldz < n_out || ldh < n_in || ldw < n_in || (ldz & 7) || (ldh & 7) || (n_out & 7) || (n_in & 7) ||
      (reinterpret_cast<size_t>(dz) & 15) || (reinterpret_cast<size_t>(h) & 15))
    return fail(SS_ERR_INVALID, "ss_wgrad_bf16: the batch a multiple of 128 rows; n_out, n_in, ldz, ldh multiples of 8; 16-byte aligned operands; ldw >= n_in");
  if ((long long)Mb * ldz >= (1ll << 32) || (long long)Mb * ldh >= (1ll << 32)) return fail(SS_ERR_INVALID, "ss_wgrad_bf16: operands beyond 2^32 elements");
  const int gx = (n_in + 255) / 256, gy = (n_out + 255) / 256, nkt = Mb / 64;
  int ks = (int)(256ll / ((long long)gx * gy));               // K shares that fill one round of the CUs (one workgroup per CU)
  if (ks < 1) ks = 1;
  if (ks > nkt / 8) ks = nkt / 8 > 0 ? nkt / 8 : 1;
  int per = (nkt + ks - 1) / ks;
  per += per & 1;                                             // shares of an even number of K tiles (nkt is even)
  ks = (nkt + per - 1) / per;
  WgradArgs a{static_cast<const __bf16 *>(dz), static_cast<const __bf16 *>(h), dw, n_out, n_in, ldz, ldh, ldw, nkt, per};
  const size_t lds = gemm256::LoopTN::LDS_BYTES;
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(ss_wgrad_tn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return fail(SS_ERR_HIP, "cannot size the GEMM's LDS");
  hipLaunchKernelGGL(ss_wgrad_tn_kernel, dim3(gx, gy, ks), dim3(512), lds, (hipStream_t)stream, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SS_OK : fail(SS_ERR_HIP, hipGetErrorString(e));
}

int ss_gaussian_sample(const float *mean, const float *noise, const float *log_std, int32_t M, int32_t dim, float *action, int32_t lda,
                       float *action_env, int32_t lde, float clip_lo, float clip_hi, float *logp, void *stream) {
  if (!mean || !noise || !log_std || !action) return fail(SS_ERR_INVALID, "null argument");
  if (M < 1 || dim < 1 || lda < dim || (action_env && lde < dim)) return fail(SS_ERR_INVALID, "ss_gaussian_sample: row strides must be >= dim");
  hipLaunchKernelGGL(ss_gaussian_sample_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream, mean, noise, log_std, M, dim, action, lda,
                     action_env, lde, clip_lo, clip_hi, logp);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SS_OK : fail(SS_ERR_HIP, hipGetErrorString(e));
}

int ss_obs_to_bf16(const float *obs, int32_t M, int32_t dim, int32_t obs_stride, const float *mean, const float *sd, const int64_t *n,
                   float lo, float hi, float clip, void *out, int32_t kpad, void *stream) {
  if (!obs || !out) return fail(SS_ERR_INVALID, "null argument");
  if (M < 1 || dim < 1 || kpad < dim || kpad % 32 || obs_stride < dim) return fail(SS_ERR_INVALID, "ss_obs_to_bf16: kpad must be a multiple of 32 and >= dim");
  const long long total = (long long)M * kpad;
  hipLaunchKernelGGL(ss_obs_to_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, obs, M, dim, obs_stride, mean, sd,
                     reinterpret_cast<const long long *>(n), lo, hi, clip, static_cast<__bf16 *>(out), kpad);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SS_OK : fail(SS_ERR_HIP, hipGetErrorString(e));
}

}
