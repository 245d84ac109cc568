// ss_gemm256.h — K loop of the 256 x 256 macro-tile bf16 GEMM (y = x W^T, both operands K-contiguous) for gfx950, round 6.
//
// Why another K loop: the 128 x BN kernels of smplsim_mlp.hip give a wave a 32 x BN/2 tile — 5 fragment reads per 4 matrix instructions, 87 FLOP
// per byte copied L2 -> LDS — and every wave of the workgroup reads, waits and multiplies in lockstep, so the matrix pipe idles while the
// fragments arrive: 0.36-0.60 PFLOP/s on the PPO update's shapes (53 248 rows), half of hipBLASLt (profiles/r06_train_gemm_sweep.txt).
// Here:
//   * 8 waves as 2 (M) x 4 (N), a wave owns 128 x 64 = 4 x 2 tiles of v_mfma_f32_32x32x16_bf16 (128 accumulator registers): 24 fragment reads
//     per 32 matrix instructions, 128 FLOP per copied byte.
//   * a K tile (64 deep) is four HALF-TILES of 128 rows x 128 bytes — A rows of the waves' upper / lower 64 rows, B rows of their left / right
//     32 columns — each copied by two global_load_lds_dwordx4 per wave; two LDS buffers of four half-tiles = 128 KB.
//   * a K tile is four PHASES, one quadrant (64 x 32 per wave, 8 matrix instructions) each:
//         P1 reads a0, b0 -> a0 b0     P2 reads b1 -> a0 b1     P3 reads a1 -> a1 b1     P4 reads nothing -> a1 b0
//     and every phase requests ONE half-tile, six phases before the phase that reads it; `s_waitcnt vmcnt(8)` after the request = "the
//     half-tile requested four phases ago has landed", never a drain.  A half-tile's LDS region is requested again two phases after its last
//     fragment read (half-tile s goes where s - 8 was; s - 8 is last read in phase <= s - 8, s is requested in phase s - 6).
//   * the two wave rows run ONE BARRIER apart: phase = { fragment reads, request, counted wait | barrier | matrix instructions | barrier },
//     so while waves 0-3 multiply, waves 4-7 (the other wave of each SIMD) read and request, and the other way round after the next barrier.
//     Ordering rules this relies on (cdna_hip_programming.md §5, "8-phase"): a copy is visible to a reader once the REQUESTING wave has waited for
//     it and a barrier both have passed follows — with the stagger that is the phase after the wait (phase g reads half-tiles <= g + 1, the wait of
//     phase g - 1 retires half-tiles <= g + 1); a region may be overwritten once every wave's reads of it are complete (lgkmcnt(0) at the head of
//     the phase's multiply part) and a barrier follows — two phases later covers the late wave row.
//   * LDS image of a half-tile: row r = 8 chunks of 16 bytes, chunk c at position c ^ ((r >> 1) & 7) — applied to the per-lane SOURCE address of
//     the copy (the copy writes wave-base + lane * 16, the image itself is linear) and to the fragment reads (same scheme as ss_linear_glds_kernel:
//     a 64-lane ds_read_b128 takes its four cycles).
#pragma once

namespace gemm256 {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(1))) const void gvoid;
typedef __attribute__((address_space(3))) void lvoid;

constexpr int TILE = 256, BK = 64;
constexpr int HALF_BYTES = 128 * BK * 2;                     // 16 KB
constexpr int BUF_BYTES = 4 * HALF_BYTES;                    // A0 | B0 | B1 | A1 of one K tile
constexpr int LOOP_LDS_BYTES = 2 * BUF_BYTES;                // 128 KB

template <int N> struct I { static constexpr int value = N; };

// half-tile kinds in request order
enum { KA0 = 0, KB0 = 1, KB1 = 2, KA1 = 3 };

struct Loop {
  // per lane: source element offsets (from X resp. W, K tile 0) of its two copy instructions of each half-tile kind
  unsigned soff[4][2];
  const __bf16 *X, *W;
  char *lds;                                                  // the workgroup's LDS (LOOP_LDS_BYTES)
  int wave, lane, wr, wc;
  unsigned fa_off, fb_off;                                    // this lane's fragment-read byte offsets inside an A resp. B half-tile (k16 step 0)
  bf16x8 fa[2][4], fb[2][4];                                  // [row tile of the half | column half][k16 step]
  int kt0;                                                    // first K tile of this workgroup's share

  __device__ __forceinline__ void init(const __bf16 *X_, const __bf16 *W_, int M, int N, int K, int m0, int n0, int kt0_, char *lds_) {
    X = X_; W = W_; lds = lds_; kt0 = kt0_;
    const int tid = threadIdx.x;
    lane = tid & 63;
    wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    wr = wave >> 2; wc = wave & 3;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int lr = 8 * (wave + 8 * i) + (lane >> 3);       // local row of the half-tile
      const int chunk = (lane & 7) ^ ((lr >> 1) & 7);
#pragma unroll
      for (int h = 0; h < 2; h++) {
        int ga = m0 + (lr >> 6) * 128 + h * 64 + (lr & 63);   // A half h: rows wr' * 128 + h * 64 + i
        ga = ga < M ? ga : M - 1;
        int gb = n0 + (lr >> 5) * 64 + h * 32 + (lr & 31);    // B half h: rows (output columns) wc' * 64 + h * 32 + j
        gb = gb < N ? gb : N - 1;
        soff[h ? KA1 : KA0][i] = (unsigned)ga * (unsigned)K + chunk * 8;
        soff[h ? KB1 : KB0][i] = (unsigned)gb * (unsigned)K + chunk * 8;
      }
    }
    const int l31 = lane & 31, half = lane >> 5, ax = (l31 >> 1) & 7;
    fa_off = (unsigned)((wr * 64 + l31) * 128 + ((half ^ ax) << 4));   // k16 step ks: ^ (ks << 5); row tile tm: + tm * 32 * 128
    fb_off = (unsigned)((wc * 32 + l31) * 128 + ((half ^ ax) << 4));
  }

  // request half-tile s (sequence number over the K tiles of this workgroup: tile s / 4, kind s % 4)
  template <int KIND, int BUF> __device__ __forceinline__ void request(int tile) {
    const __bf16 *base = (KIND == KA0 || KIND == KA1) ? X : W;
    const size_t k0 = (size_t)(kt0 + tile) * BK;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      char *dst = lds + BUF * BUF_BYTES + KIND * HALF_BYTES + (wave + 8 * i) * 1024;
      __builtin_amdgcn_global_load_lds((gvoid *)(base + soff[KIND][i] + k0), (lvoid *)dst, 16, 0, 0);
    }
  }

  template <int BUF, int H> __device__ __forceinline__ void read_a() {
    const char *r = lds + BUF * BUF_BYTES + (H ? KA1 : KA0) * HALF_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ks++)
#pragma unroll
      for (int tm = 0; tm < 2; tm++) fa[tm][ks] = *reinterpret_cast<const bf16x8 *>(r + (fa_off ^ (ks << 5)) + tm * 4096);
  }
  template <int BUF, int H> __device__ __forceinline__ void read_b() {
    const char *r = lds + BUF * BUF_BYTES + (H ? KB1 : KB0) * HALF_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ks++) fb[H][ks] = *reinterpret_cast<const bf16x8 *>(r + (fb_off ^ (ks << 5)));
  }
  template <int HA, int HB> __device__ __forceinline__ void multiply(f32x16 (&acc)[4][2]) {
#pragma unroll
    for (int ks = 0; ks < 4; ks++)
#pragma unroll
      for (int tm = 0; tm < 2; tm++)
        acc[2 * HA + tm][HB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm][ks], fb[HB][ks], acc[2 * HA + tm][HB], 0, 0, 0);
  }

  // one phase.  J = phase of the K tile (0..3), BUF = the tile's buffer, REQ = request a half-tile (tile index rt, into buffer RBUF), VM = the counted wait
  template <int J, int BUF, bool REQ, int VM> __device__ __forceinline__ void phase(f32x16 (&acc)[4][2], int rt) {
    if constexpr (J == 0) { read_a<BUF, 0>(); read_b<BUF, 0>(); }
    if constexpr (J == 1) read_b<BUF, 1>();
    if constexpr (J == 2) read_a<BUF, 1>();
    if constexpr (REQ) {
      // phase g requests half-tile g + 6: kinds B1, A1 of the next tile (other buffer) in phases 0, 1; A0, B0 of the tile after it (this buffer) in 2, 3
      if constexpr (J == 0) request<KB1, BUF ^ 1>(rt);
      if constexpr (J == 1) request<KA1, BUF ^ 1>(rt);
      if constexpr (J == 2) request<KA0, BUF>(rt);
      if constexpr (J == 3) request<KB0, BUF>(rt);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    if constexpr (J == 0) multiply<0, 0>(acc);
    if constexpr (J == 1) multiply<0, 1>(acc);
    if constexpr (J == 2) multiply<1, 1>(acc);
    if constexpr (J == 3) multiply<1, 0>(acc);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  }

  // one K tile (t, buffer BUF) of nkt.  The last six phases of the loop have no half-tile left to request; they request the LAST tile's again
  // (valid memory, L2 hits) into the regions the schedule would use — regions nobody reads any more, by the same two-phase rule — so that
  // there is ONE copy of the phase code with ONE wait count (a peeled tail with its own counts spilled 300 registers: the loop body
  // sits at the 256-register limit): 6 of 4 nkt half-tiles copied in vain.
  template <int BUF> __device__ __forceinline__ void tile(f32x16 (&acc)[4][2], int t, int last) {
    const int t1 = t + 1 < last ? t + 1 : last, t2 = t + 2 < last ? t + 2 : last;
    phase<0, BUF, true, 8>(acc, t1); phase<1, BUF, true, 8>(acc, t1); phase<2, BUF, true, 8>(acc, t2); phase<3, BUF, true, 8>(acc, t2);
  }

  // the whole loop over an EVEN number nkt >= 2 of K tiles; on return every wave has passed the same number of barriers and no copy is in flight
  __device__ __forceinline__ void run(f32x16 (&acc)[4][2], int nkt) {
    // half-tiles 0..5: tile 0 whole, A0 and B0 of tile 1
    request<KA0, 0>(0); request<KB0, 0>(0); request<KB1, 0>(0); request<KA1, 0>(0);
    request<KA0, 1>(1); request<KB0, 1>(1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");          // half-tiles 0, 1 landed (this wave's share)
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();                // the lower wave row runs one barrier behind
    for (int t = 0; t < nkt; t += 2) { tile<0>(acc, t, nkt - 1); tile<1>(acc, t + 1, nkt - 1); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the copies made in vain
    if (wr == 0) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();
  }
};

// ---- the same loop for a product that contracts over the ROWS of both operands: dW[i][j] = sum_m Z[m][i] H[m][j], Z [Mb, ldz] and H [Mb, ldh] row-major
// (the weight gradient of a layer from dZ and the layer's input as they lie in memory: no transposed copies).  What changes against Loop:
//   * a half-tile is 32 m x 256 columns of one operand, stored as 16 subtiles [32 m][16 columns] of 1 KB (one copy instruction each: lane -> row lane / 2, 16-byte
//     half lane % 2), 1152 bytes apart so that the even and odd subtiles of a fragment read fall into different halves of the 256-byte bank row;
//   * a fragment (lane = column, 8 consecutive m) is two ds_read_b64_tr_b16: per 16-lane group lane p passes the address of row p / 4, column quad p % 4 of a
//     [4 m][16 columns] block and receives column p of it (profiles/r06_tr_read_probe.txt);
//   * a phase is one k16 step of the whole 128 x 64 wave tile (12 reads, 8 matrix instructions); phases 0, 1 read the m-low half-tiles, 2, 3 the m-high ones;
//     request order per K tile Z-low, H-low, Z-high, H-high, half-tile s requested in phase s - 5 (its region's last reader is phase s - 7 at the latest: two
//     phases before), `s_waitcnt vmcnt(6)` = three half-tiles in flight.
struct LoopTN {
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
  static constexpr int SUB = 1152, HALF = 16 * SUB, BUF = 4 * HALF;   // bytes: subtile stride, half-tile, one K tile's four half-tiles
  static constexpr int LDS_BYTES = 2 * BUF;                            // 147 456
  unsigned zoff[2], hoff[2];                                  // per lane: element offsets (K tile 0, m-low) of its two copies of a Z resp. H half-tile
  const __bf16 *Z, *H;
  unsigned ldz, ldh;
  unsigned lds;                                               // LDS byte address of the workgroup's block
  int wave, lane, wr, wc, kt0;
  unsigned fa_addr, fb_addr;                                  // this lane's fragment-read byte offsets inside a Z resp. H half-tile (row tile 0, k16 step 0, read 0)
  bf16x8 fa[4], fb[2];

  __device__ __forceinline__ void init(const __bf16 *Z_, const __bf16 *H_, int ldz_, int ldh_, int NI, int NJ, int i0, int j0, int kt0_, char *lds_) {
    Z = Z_; H = H_; ldz = (unsigned)ldz_; ldh = (unsigned)ldh_; kt0 = kt0_;
    lds = (unsigned)(size_t)lds_;
    const int tid = threadIdx.x;
    lane = tid & 63;
    wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    wr = wave >> 2; wc = wave & 3;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int nblk = wave + 8 * i, ml = lane >> 1, hh = lane & 1;
      int ci = i0 + 16 * nblk + 8 * hh, cj = j0 + 16 * nblk + 8 * hh;
      ci = ci + 8 <= NI ? ci : 0;                             // (columns beyond the matrix: any readable chunk, their results are never stored)
      cj = cj + 8 <= NJ ? cj : 0;
      zoff[i] = (unsigned)ml * ldz + (unsigned)ci;
      hoff[i] = (unsigned)ml * ldh + (unsigned)cj;
    }
    const int g = lane >> 4, p = lane & 15;
    const unsigned inner = (unsigned)((8 * (g >> 1) + (p >> 2)) * 32 + (p & 3) * 8);
    fa_addr = (unsigned)((2 * (wr * 4) + (g & 1)) * SUB) + inner;     // row tile tm: + 2 tm SUB; k16 step ks: half-tile ks / 2, + 512 (ks % 2); second read: + 128
    fb_addr = (unsigned)((2 * (wc * 2) + (g & 1)) * SUB) + inner;
  }

  template <int KIND, int BUFI> __device__ __forceinline__ void request(int tile) {   // KIND 0 Z-low, 1 H-low, 2 Z-high, 3 H-high
    constexpr bool isz = (KIND & 1) == 0;
    const __bf16 *base = isz ? Z : H;
    const size_t m = (size_t)(kt0 + tile) * 64 + (KIND >> 1) * 32;
    const size_t adv = m * (isz ? ldz : ldh);
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const unsigned dst = lds + BUFI * BUF + KIND * HALF + (wave + 8 * i) * SUB;
      __builtin_amdgcn_global_load_lds((gvoid *)(base + (isz ? zoff[i] : hoff[i]) + adv), (lvoid *)(size_t)dst, 16, 0, 0);
    }
  }

  template <int OFF> __device__ __forceinline__ bf16x8 frag(unsigned addr) {
    u32x2 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(addr), "n"(OFF));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(OFF + 128));
    union { struct { u32x2 a, b; } s; bf16x8 v; } c;
    c.s.a = lo; c.s.b = hi;
    return c.v;
  }

  // one phase = k16 step J of K tile (buffer BUFI): fragment reads, request of half-tile (phase + 5), counted wait | barrier | 8 matrix instructions | barrier
  template <int J, int BUFI> __device__ __forceinline__ void phase(f32x16 (&acc)[4][2], int rt) {
    constexpr int zh = (J >> 1) ? 2 : 0, hh = (J >> 1) ? 3 : 1;                  // the half-tiles this step reads
    constexpr int ko = 512 * (J & 1);
    const unsigned ab = lds + BUFI * BUF + zh * HALF + fa_addr, bb = lds + BUFI * BUF + hh * HALF + fb_addr;
    fa[0] = frag<ko>(ab); fa[1] = frag<ko + 2 * SUB>(ab); fa[2] = frag<ko + 4 * SUB>(ab); fa[3] = frag<ko + 6 * SUB>(ab);
    fb[0] = frag<ko>(bb); fb[1] = frag<ko + 2 * SUB>(bb);
    // phase g requests half-tile g + 5: H-low, Z-high, H-high of the next tile (other buffer) in steps 0, 1, 2; Z-low of the tile after it (this buffer) in step 3
    if constexpr (J == 0) request<1, BUFI ^ 1>(rt);
    if constexpr (J == 1) request<2, BUFI ^ 1>(rt);
    if constexpr (J == 2) request<3, BUFI ^ 1>(rt);
    if constexpr (J == 3) request<0, BUFI>(rt);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int tn = 0; tn < 2; tn++)
#pragma unroll
      for (int tm = 0; tm < 4; tm++) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm], fb[tn], acc[tm][tn], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  }

  template <int BUFI> __device__ __forceinline__ void tile(f32x16 (&acc)[4][2], int t, int last) {
    const int t1 = t + 1 < last ? t + 1 : last, t2 = t + 2 < last ? t + 2 : last;
    phase<0, BUFI>(acc, t1); phase<1, BUFI>(acc, t1); phase<2, BUFI>(acc, t1); phase<3, BUFI>(acc, t2);
  }

  // an EVEN number nkt >= 2 of K tiles (64 rows of Z and H each)
  __device__ __forceinline__ void run(f32x16 (&acc)[4][2], int nkt) {
    request<0, 0>(0); request<1, 0>(0); request<2, 0>(0); request<3, 0>(0);   // half-tiles 0..4: tile 0 whole, Z-low of tile 1
    request<0, 1>(1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");          // half-tiles 0, 1 landed (this wave's share)
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();
    for (int t = 0; t < nkt; t += 2) { tile<0>(acc, t, nkt - 1); tile<1>(acc, t + 1, nkt - 1); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (wr == 0) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();
  }
};

}  // namespace gemm256
