// ss_env_kernel.h — the gfx950 step kernel template, shared by the translation units that instantiate it (smplsim_hip.hip: the
// plain / body-output / shaped flavours and the C ABI; smplsim_hip_sc.hip: body-body contacts; smplsim_hip_im.hip: the imitation
// step).  Three units so that they compile side by side: every instantiation of the state machine costs ~15-25 s of hipcc.
#pragma once
#include <hip/hip_runtime.h>

#include "ss_api.h"
#include "ss_kernel.h"
#include "ss_wave_gpu.h"

// launch bounds per kernel variant = the number of envs whose LDS slices fit one CU, rounded up to whole waves per
// SIMD.  SMPL: 12 envs -> 3 waves/SIMD -> 168-VGPR cap (16 spilled dwords at -O3, 47 at the shipped -Os, which is faster all the same).  SMPL-X: 7 envs (aliased LDS layout + lean tables, round 5; 5 on the plain one) -> 2 waves/SIMD, 256 VGPRs,
// no spills.  (History of the trade-off: profiles/r01i_ab_launch_bounds.txt.)
#ifndef SS_MAX_THREADS
#define SS_MAX_THREADS 768
#endif
#ifndef SS_MAX_THREADS_X
#define SS_MAX_THREADS_X 448
#endif
// self-collision instantiation of the SMPL size: 8 envs at most share a CU -> 2 waves/SIMD, 256 VGPRs
#ifndef SS_MAX_THREADS_SC
#define SS_MAX_THREADS_SC 512
#endif

namespace ss {
typedef void (*kern_t)(const KArgs);
// instantiations that live in the other translation units (nullptr = not compiled for this size class)
kern_t pick_kernel_selfcol(int variant, bool shaped, bool imit, const Hdr &h, const HdrC &hc);
kern_t pick_kernel_imitation(int variant, bool shaped, const Hdr &h, const HdrC &hc);
kern_t pick_kernel_imitation_selfcol(int variant, bool shaped, const Hdr &h, const HdrC &hc);
kern_t pick_kernel_x(int flavour, const Hdr &h, const HdrC &hc);                      // smplsim_hip_x.hip: the SMPL-X/H size class
}  // namespace ss

namespace {

// IMIT: the instantiation of ss_imitation_step_fused — after the step pass the wave runs the imitation task of its env and, if the
// env finished, the reference-state re-initialisation (ss_imfused.h) around the stepper's own reset pass.
template <int DOFP, int CANDP, int SLOTP, int NPASS, int MAXT, bool BODYOUT, bool SHAPED, class HT = ss::HdrRuntime, bool SELFCOL = false, bool IMIT = false>
__global__ void __launch_bounds__(MAXT) ss_env_kernel(const ss::KArgs k) {
  extern __shared__ __align__(16) uint32_t lds[];
  if (blockIdx.x == 0 && threadIdx.x == 0) *k.work_counter_next = 0;   // the next launch's counter (this launch uses the other one)
  for (int i = threadIdx.x; i < k.h.shared_words; i += blockDim.x) lds[i] = k.shared_g[i];
  if constexpr (SELFCOL)                                     // lock word of the shared dense block (ss_hdr.h), free
    if (threadIdx.x == 0 && ss::ss_pool_floats(k.sc) > 0)
      *reinterpret_cast<int *>(reinterpret_cast<float *>(lds + ((k.h.shared_words + 3) & ~3)) + (size_t)(blockDim.x >> 6) * k.sc.env_floats) = 0;
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform -> LDS bases stay in SGPRs
  const int slice = SELFCOL ? k.sc.env_floats : HT::view(k.h).env_floats;
  float *L = reinterpret_cast<float *>(lds + ((k.h.shared_words + 3) & ~3)) + (size_t)wave * slice;
  float *pool = nullptr;                                     // SELFCOL: the workgroup's shared dense block behind the env slices (ss_hdr.h)
  if constexpr (SELFCOL) {
    if (ss::ss_pool_floats(k.sc) > 0) {
      pool = reinterpret_cast<float *>(lds + ((k.h.shared_words + 3) & ~3)) + (size_t)(blockDim.x >> 6) * slice;
    }
  }
  WaveGpu w{(int)(threadIdx.x & 63)};
  // persistent wavefronts: env-steps have heavy-tailed cost (Newton iterations), so every wave pulls the
  // next env id from a device counter instead of owning a fixed slice of the batch.  The first env of every wave is
  // its own global wave index (no atomic: thousands of waves hitting one counter at launch serialise in L2).
  const int total_waves = (int)(gridDim.x * (blockDim.x >> 6));
  bool first = true;
  for (;;) {
    int env = wave * (int)gridDim.x + (int)blockIdx.x;       // consecutive (= similarly heavy) envs go to different CUs
    if (!first) {
      if (w.ln == 0) env = atomicAdd(k.work_counter, 1) + total_waves;
      env = __builtin_amdgcn_readfirstlane(env);
    }
    first = false;
    if (env >= k.st.num_envs) break;
    if (k.order) env = __builtin_amdgcn_readfirstlane(k.order[env]);   // longest-processing-time-first hand-out
    int mode = k.mode;
    if constexpr (IMIT) {
      const ss::mo::ImFused *f = static_cast<const ss::mo::ImFused *>(k.im);
      ss::run_env<WaveGpu, DOFP, CANDP, SLOTP, NPASS, BODYOUT, SHAPED, HT, SELFCOL>(&w, &k, lds, L, env, mode, pool);
      w.sync();
      if (ss::mo::fused_after_step(&w, f, k.im_rand, env)) {
        ss::run_env<WaveGpu, DOFP, CANDP, SLOTP, NPASS, BODYOUT, SHAPED, HT, SELFCOL>(&w, &k, lds, L, env, ss::MODE_RESET, pool);
        w.sync();
        ss::mo::fused_after_reset(&w, f, env);
      }
      continue;
    }
    for (int rep = 0; rep < 2; rep++) {                       // second trip = fused Default reset of an env whose episode ended
      const bool again = ss::run_env<WaveGpu, DOFP, CANDP, SLOTP, NPASS, BODYOUT, SHAPED, HT, SELFCOL>(&w, &k, lds, L, env, mode, pool);
      w.sync();
      if (!again) break;
      mode = ss::MODE_RESET;
    }
  }
}


typedef ss::HdrSmplFixed HdrSmpl;       // the packaged fixtures' compile-time layouts (ss_hdr.h)
typedef ss::HdrSmplxFixed HdrSmplx;

}  // namespace
