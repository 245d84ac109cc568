// emu.cpp — 64-fiber wavefront emulator: runs the *same* kernel source (smplsim_amd/csrc/ss_kernel.h)
// on the CPU so the float32 kernel logic can be unit-tested against the oracle without a GPU.
//
// UNIT-TEST INFRASTRUCTURE ONLY.  The product (smplsim_amd/) never loads this library; the product
// path is libsmplsim_hip.so and fails loudly when it is missing.
//
// Every lane of the wavefront is a cooperative fiber (hand-rolled x86-64 context switch).  A wave
// sync / collective is a round-robin switch through all 64 fibers, so lanes only ever observe each
// other's LDS writes across sync points — a stricter model than lock-step SIMT, which makes missing
// syncs show up as wrong answers here.  Each collective carries a site id that is checked for
// convergence (all lanes must arrive at the same site).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <algorithm>
#include <vector>

#include "../../smplsim_amd/csrc/ss_api.h"
#include "../../smplsim_amd/csrc/ss_kernel.h"
#include "../../smplsim_amd/csrc/ss_motion_api.h"

extern "C" void emu_switch(void **save_sp, void *load_sp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
)");

namespace {

using ss::real;
constexpr int kLanes = 64;
constexpr size_t kStack = 256 * 1024;

struct WaveEmu;
struct Fiber { void *sp; char *stack; };

struct Machine {
  Fiber fib[kLanes];
  void *main_sp = nullptr;
  int cur = -1;
  int done[kLanes];
  // collective scratch (double holds both builds' values exactly)
  double fx[kLanes], fy[kLanes];
  unsigned long long ux[kLanes];
  int siteh[4][kLanes];
  unsigned narr[kLanes];
  void (*entry)(int lane, void *arg) = nullptr;
  void *arg = nullptr;
  Machine() { for (int i = 0; i < kLanes; i++) fib[i].stack = (char *)aligned_alloc(64, kStack); }
  ~Machine() { for (int i = 0; i < kLanes; i++) free(fib[i].stack); }
};

thread_local Machine *g_m = nullptr;

extern "C" void emu_trampoline() {
  Machine *m = g_m;
  int lane = m->cur;
  m->entry(lane, m->arg);
  m->done[lane] = 1;
  // return to main; never resumed
  emu_switch(&m->fib[lane].sp, m->main_sp);
  abort();
}

void yield_to_next(Machine *m) {
  // round robin among unfinished fibers; when wrapping past the last lane control passes through main
  int lane = m->cur;
  emu_switch(&m->fib[lane].sp, m->main_sp);
}

void run_wave(Machine *m, void (*entry)(int, void *), void *arg) {
  g_m = m;
  m->entry = entry; m->arg = arg;
  for (int i = 0; i < kLanes; i++) {
    m->done[i] = 0; m->narr[i] = 0;
    char *top = m->fib[i].stack + kStack;
    top = (char *)((uintptr_t)top & ~(uintptr_t)15);
    void **sp = (void **)(top - 64);
    // layout (low -> high): r15 r14 r13 r12 rbx rbp ret pad
    for (int k = 0; k < 6; k++) sp[k] = nullptr;
    sp[6] = (void *)&emu_trampoline;
    sp[7] = nullptr;
    m->fib[i].sp = sp;
  }
  // scheduler: repeatedly sweep lanes 0..63, resuming each unfinished fiber until it yields or finishes
  for (;;) {
    int alive = 0;
    for (int i = 0; i < kLanes; i++) {
      if (m->done[i]) continue;
      alive++;
      m->cur = i;
      emu_switch(&m->main_sp, m->fib[i].sp);
    }
    if (!alive) break;
  }
}

struct WaveEmu {
  Machine *m;
  int ln;
  int lane() const { return ln; }
  void arrive(int id) {
    // lane 0 is always the first to reach arrival #n (it runs first in every sweep): compare with it
    unsigned n = ++m->narr[ln];
    m->siteh[n & 3][ln] = id;
    if (ln > 0 && !m->done[0] && m->siteh[n & 3][0] != id) {
      fprintf(stderr, "wave_emu: divergent collective #%u (lane %d at site %d, lane 0 at site %d)\n", n, ln, id, m->siteh[n & 3][0]);
      abort();
    }
    yield_to_next(m);
  }
  void sync() { arrive(1); }
  real sum(real v) {
    m->fx[ln] = v;
    arrive(2);
    real t[kLanes];
    for (int i = 0; i < kLanes; i++) t[i] = m->fx[i];
    for (int mask = 32; mask >= 1; mask >>= 1) {            // same butterfly order as the GPU's __shfl_xor ladder
      real n[kLanes];
      for (int i = 0; i < kLanes; i++) n[i] = t[i] + t[i ^ mask];
      memcpy(t, n, sizeof t);
    }
    real r = t[ln];
    arrive(3);                                              // nobody overwrites fx before everyone has read it
    return r;
  }
  unsigned long long ballot(int p) {
    m->ux[ln] = p ? 1ull : 0ull;
    arrive(4);
    unsigned long long r = 0;
    for (int i = 0; i < kLanes; i++) r |= m->ux[i] << i;
    arrive(5);
    return r;
  }
  unsigned long long bor(unsigned long long v) {
    m->ux[ln] = v;
    arrive(6);
    unsigned long long r = 0;
    for (int i = 0; i < kLanes; i++) r |= m->ux[i];
    arrive(7);
    return r;
  }
  real shfl_xor(real v, int mask) {
    m->fx[ln] = v;
    arrive(8);
    real r = m->fx[ln ^ mask];
    arrive(9);
    return r;
  }
  int shfl_xor_i(int v, int mask) {
    m->ux[ln] = (unsigned long long)(unsigned)v;
    arrive(10);
    int r = (int)(unsigned)m->ux[ln ^ mask];
    arrive(11);
    return r;
  }
  real sum8(real v) {                                        // sum over the lane's aligned group of 8
    m->fx[ln] = v;
    arrive(12);
    real r = 0;
    for (int i = 0; i < 8; i++) r += m->fx[(ln & ~7) + i];
    arrive(13);
    return r;
  }
  // the emulator runs one wavefront at a time: the shared dense block is always free
  void lock_acquire(int *p) { if (ln == 0) { if (*p != 0) { fprintf(stderr, "wave_emu: shared block already taken\n"); abort(); } *p = 1; } sync(); }
  void lock_release(int *p) { sync(); if (ln == 0) *p = 0; }
  real bcast(real v, int src) {
    m->fx[ln] = v;
    arrive(14);
    real r = m->fx[src];
    arrive(15);
    return r;
  }
  int bcast_i(int v, int src) {
    m->ux[ln] = (unsigned long long)(unsigned)v;
    arrive(16);
    int r = (int)(unsigned)m->ux[src];
    arrive(17);
    return r;
  }
  // the 16x16x4 matrix instruction of the GPU build (ss_wave_gpu.h::mfma16), same operand layout, an fma chain over k
  void mfma16(real a, real b, real *c) {
    m->fx[ln] = a; m->fy[ln] = b;
    arrive(18);
    const int col = ln & 15, q = ln >> 4;
    for (int r = 0; r < 4; r++) {
      real acc = c[r];
      for (int kk = 0; kk < 4; kk++) acc = std::fma((real)m->fx[(4 * q + r) + 16 * kk], (real)m->fy[col + 16 * kk], acc);
      c[r] = acc;
    }
    arrive(19);
  }
  real quad_xor1(real v) { return shfl_xor(v, 1); }
  real quad_xor2(real v) { return shfl_xor(v, 2); }
  int quad_xor1_i(int v) { return shfl_xor_i(v, 1); }
  int quad_xor2_i(int v) { return shfl_xor_i(v, 2); }
  bool any(int p) { return ballot(p) != 0ull; }
  int opaque(int x) { volatile int y = x; return y; }
  int opaque_v(int x) { volatile int y = x; return y; }
  void mem_fence() { sync(); }                               // all lanes' stores done before anyone reads them back
  unsigned long long clock() { return 0; }
  void atomic_add_u64(unsigned long long *p, unsigned long long v) { *p += v; }
  void atomic_add(real *p, real v) { *p += v; }
};

struct LaunchCtx { const ss::KArgs *k; const uint32_t *T; ss::real *L; int env; Machine *m; ss::real *pool; };

template <int DOFP, int CANDP, int SLOTP, int NPASS, bool SHAPED, class HT = ss::HdrRuntime, bool SELFCOL = false>
void lane_entry(int lane, void *arg) {
  LaunchCtx *c = (LaunchCtx *)arg;
  WaveEmu w{c->m, lane};
  int mode = c->k->mode;
#ifndef SS_F64
  if (c->k->im) {                                            // the IMIT instantiation of the GPU kernel (smplsim_hip.hip)
    const ss::mo::ImFused *f = static_cast<const ss::mo::ImFused *>(c->k->im);
    ss::run_env<WaveEmu, DOFP, CANDP, SLOTP, NPASS, true, SHAPED, HT, SELFCOL>(&w, c->k, c->T, c->L, c->env, mode, c->pool);
    w.sync();
    if (ss::mo::fused_after_step(&w, f, c->k->im_rand, c->env)) {
      ss::run_env<WaveEmu, DOFP, CANDP, SLOTP, NPASS, true, SHAPED, HT, SELFCOL>(&w, c->k, c->T, c->L, c->env, ss::MODE_RESET, c->pool);
      w.sync();
      ss::mo::fused_after_reset(&w, f, c->env);
    }
    return;
  }
#endif
  for (int rep = 0; rep < 2; rep++) {
    const bool again = ss::run_env<WaveEmu, DOFP, CANDP, SLOTP, NPASS, true, SHAPED, HT, SELFCOL>(&w, c->k, c->T, c->L, c->env, mode, c->pool);
    w.sync();
    if (!again) break;
    mode = ss::MODE_RESET;
  }
}

struct EmuBackend {
  static void *alloc(size_t n) { return calloc(1, n); }
  static void free_(void *p) { free(p); }
  static bool upload(void *dst, const void *src, size_t n) { memcpy(dst, src, n); return true; }
  static bool set_device(int) { return true; }
  static bool download(void *dst, const void *src, size_t n) { memcpy(dst, src, n); return true; }
  static bool copy_d2d(void *dst, const void *src, size_t n, void *) { memmove(dst, src, n); return true; }
  static int lds_capacity() { return 160 * 1024 * (int)(sizeof(ss::real) / 4); }   // the float64 triage build's slices are twice as large
  static int kernel_regs() { return 0; }
  static int max_waves(int, int) { return 16; }
  static const char *order_by_key(const ss_state &st, int nv, int32_t *key, int32_t *order, void *) {   // same key as ss_key_kernel
    const int n = st.num_envs;
    const ss::real *qv = reinterpret_cast<const ss::real *>(st.qvel), *qa = reinterpret_cast<const ss::real *>(st.qacc_warm);
    for (int e = 0; e < n; e++) {
      float vm = 0.f, am = 0.f;
      for (int i = 0; i < nv; i++) {
        const float v = fabsf((float)qv[(size_t)e * nv + i]), a = fabsf((float)qa[(size_t)e * nv + i]);
        vm = fmaxf(vm, v == v ? v : 1e6f); am = fmaxf(am, a == a ? a : 1e12f);
      }
      const int tc = __builtin_popcount((unsigned)st.touch[2 * e]) + __builtin_popcount((unsigned)st.touch[2 * e + 1]);
      const float k = (float)st.solver_iters[e] + 6.f * (float)tc + 8.f * log1pf(fminf(am, 1e12f)) + 8.f * log1pf(fminf(vm, 1e6f));
      key[e] = (int32_t)fminf(fmaxf(k, 0.f), 1023.f);
    }
    std::vector<int> idx(n);
    for (int i = 0; i < n; i++) idx[i] = i;
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return key[a] > key[b]; });
    for (int i = 0; i < n; i++) order[i] = idx[i];
    return nullptr;
  }
  static const char *gae(const float *rew, const float *nd, const float *ndead, const float *val, const float *boot, int T, int N,
                         float gamma, float tau, float *adv, float *ret, void *) {
    for (int n = 0; n < N; n++) {                            // same float32 recursion as the device kernel
      float next_v = boot ? boot[n] : 0.f, next_a = 0.f;
      for (int t = T - 1; t >= 0; --t) {
        const size_t i = (size_t)t * N + n;
        const float v = val[i];
        const float delta = rew[i] + gamma * next_v * ndead[i] - v;
        const float a = delta + gamma * tau * next_a * nd[i];
        adv[i] = a; ret[i] = v + a;
        next_v = v; next_a = a;
      }
    }
    return nullptr;
  }
  // ---- motion library: the element functions run as plain loops, the two wave functions on the 64-fiber machine
  struct FixCtx { const ss::mo::CookArgs *a; int m; Machine *mach; };
  static void fix_entry(int lane, void *arg) { FixCtx *c = (FixCtx *)arg; WaveEmu w{c->mach, lane}; ss::mo::dof_fix_clip(&w, *c->a, c->m); }
  struct FkCtx { const ss::mo::CookArgs *a; int wave; Machine *mach; float *xf; };
  template <int LPE> static void fk_entry(int lane, void *arg) { FkCtx *c = (FkCtx *)arg; WaveEmu w{c->mach, lane}; ss::mo::fk_wave<WaveEmu, LPE>(&w, *c->a, c->wave, c->xf); }
  struct VelCtx { const ss::mo::CookArgs *a; int wave; Machine *mach; float *raw; };
  static void vel_entry(int lane, void *arg) { VelCtx *c = (VelCtx *)arg; WaveEmu w{c->mach, lane}; ss::mo::vel_wave(&w, *c->a, c->wave, c->raw); }
  struct ImCtx { const ss::mo::ImArgs *a; int wave; Machine *mach; };
  template <int LPE> static void im_entry(int lane, void *arg) { ImCtx *c = (ImCtx *)arg; WaveEmu w{c->mach, lane}; ss::mo::imitation_wave<WaveEmu, LPE>(&w, *c->a, c->wave); }
  static Machine *machine() { static thread_local Machine *m = new Machine(); return m; }
  static const char *motion_cook(const ss::mo::CookArgs &a, void *) {
    float xf[64 * ss::mo::kXformStride];
    const int lpe = a.sk.nb <= 32 ? 32 : 64, per = 64 / lpe;
    for (int wv = 0; wv * per < a.d.num_frames; wv++) { FkCtx c{&a, wv, machine(), xf}; run_wave(machine(), lpe == 32 ? fk_entry<32> : fk_entry<64>, &c); }
    for (int m = 0; m < a.d.num_motions; m++) { FixCtx c{&a, m, machine()}; run_wave(machine(), fix_entry, &c); }
    std::vector<float> raw((size_t)(ss::mo::kVelTile + 2 * ss::mo::kGaussRadius) * a.sk.nb * 6);
    for (int wv = 0; wv * ss::mo::kVelTile < a.d.num_frames; wv++) {
      for (auto &x : raw) x = __builtin_nanf("");            // reads of slots that were never written would show
      VelCtx c{&a, wv, machine(), raw.data()}; run_wave(machine(), vel_entry, &c);
    }
    return nullptr;
  }
  static const char *motion_state(const ss::mo::StateArgs &a, void *) {
    for (int n = 0; n < a.N; n++) for (int j = 0; j < a.d.nbody; j++) ss::mo::state_elem(a, n, j);
    return nullptr;
  }
  static const char *motion_resample(const ss::mo::ResampleArgs &a, void *) {
    for (int n = 0; n < a.N; n++) ss::mo::resample_elem(a, n);
    return nullptr;
  }
  static const char *imitation(const ss::mo::ImArgs &a, void *) {
    const int lpe = a.d.nbody <= 32 ? 32 : 64, per = 64 / lpe;
    for (int wv = 0; wv * per < a.N; wv++) { ImCtx c{&a, wv, machine()}; run_wave(machine(), lpe == 32 ? im_entry<32> : im_entry<64>, &c); }
    return nullptr;
  }
  static const char *launch(const ss::KArgs &k, int nenv, int envs_per_wg, size_t lds_bytes, void *, int, int) {
    (void)envs_per_wg; (void)lds_bytes;
    static thread_local Machine *m = new Machine();
    std::vector<ss::real> L(ss::env_slice_floats(k));
    // the workgroup's shared dense block (lock word first); NaNs behind it: a read past the block's end shows up in the results instead of going unnoticed
    const size_t pool_n = k.cfg.self_collision ? (size_t)ss::ss_pool_floats(k.sc) + 4 : 4;
    std::vector<ss::real> pool(pool_n + 2048, std::numeric_limits<ss::real>::quiet_NaN());
    std::fill(pool.begin(), pool.begin() + pool_n, ss::real(0));
    if (*k.work_counter != 0) return "work counter not zero at launch";
    *k.work_counter_next = 0;
    for (int env = 0; env < nenv; env++) {
      // poison LDS so that reads of never-written locations are visible: NaN by default; SS_EMU_POISON = neg | pos | rand puts
      // finite garbage there instead (a NaN hides a read that only feeds a comparison: the comparison is just false)
      {
        static const char *pz = getenv("SS_EMU_POISON");
        static unsigned long long lcg = 88172645463325252ull;
        for (auto &x : L) {
          if (!pz) x = __builtin_nanf("");
          else if (pz[0] == 'n') x = (ss::real)-1e30;
          else if (pz[0] == 'p') x = (ss::real)1e30;
          else { lcg = lcg * 6364136223846793005ull + 1442695040888963407ull; x = (ss::real)((double)(long long)(lcg >> 11) * 1e-12 - 4e3); }
        }
      }
      LaunchCtx c{&k, k.shared_g, L.data(), k.order ? k.order[env] : env, m, pool.data()};
      void (*entry)(int, void *) = nullptr;
      const int variant = ss::kernel_variant(k.h);
      if (k.cfg.self_collision) {
        if (variant == 0) entry = k.st.shape_id ? lane_entry<2, 2, 1, 1, true, ss::HdrRuntime, true> : lane_entry<2, 2, 1, 1, false, ss::HdrRuntime, true>;
        else if (variant == 1) entry = k.st.shape_id ? lane_entry<3, 3, 2, 2, true, ss::HdrRuntime, true> : lane_entry<3, 3, 2, 2, false, ss::HdrRuntime, true>;
        else return "no kernel variant for this model size";
        run_wave(m, entry, &c);
        continue;
      }
      // like the GPU launcher: the SMPL-sized model without per-env shapes runs the compile-time-layout instantiation
      if (variant == 0 && !k.st.shape_id && ss::HdrSmplFixed::matches(k.h, k.hc) && !getenv("SS_EMU_GENERIC")) entry = lane_entry<2, 2, 1, 1, false, ss::HdrSmplFixed>;
      else if (variant == 0) entry = k.st.shape_id ? lane_entry<2, 2, 1, 1, true> : lane_entry<2, 2, 1, 1, false>;
      else if (variant == 1 && !k.st.shape_id && ss::HdrSmplxFixed::matches(k.h, k.hc) && !getenv("SS_EMU_GENERIC")) entry = lane_entry<3, 3, 2, 2, false, ss::HdrSmplxFixed>;
      else if (variant == 1) entry = k.st.shape_id ? lane_entry<3, 3, 2, 2, true> : lane_entry<3, 3, 2, 2, false>;
      else return "no kernel variant for this model size";
      run_wave(m, entry, &c);
    }
    return nullptr;
  }
};

}  // namespace

// test hook: the kernel's pair functions (ss_selfcol.h) on raw geometry, same calling convention as the oracle's
// om_narrow_phase (oracle/oracle.c); always float64 in / out
extern "C" int ss_emu_narrow_phase(int kind, const double *in, double *out) {
  using namespace ss;
  real v[32];
  for (int i = 0; i < 31; i++) v[i] = (real)in[i];
  real c[sc::kBoxBoxWork]; int n = 0;
  if (kind == 0) n = sc::capsule_capsule(v, v + 3, v[6], v[7], v + 8, v + 11, v[14], v[15], v[16], c);
  else if (kind == 1) n = sc::capsule_box(v, v + 3, v[6], v[7], v + 8, v + 11, v + 20, v[23], c);
  else n = sc::box_box(v, v + 3, v + 12, v + 15, v + 18, v + 27, v[30], c);
  out[0] = n;
  for (int i = 0; i < n; i++) for (int k = 0; k < 7; k++) out[1 + 7 * i + k] = c[sc::kConOut * i + k];
  return n;
}

SS_DEFINE_C_API(EmuBackend)
SS_DEFINE_MOTION_API(EmuBackend)
