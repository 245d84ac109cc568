// smplsim_hip_sc.hip — the step kernel's instantiations with body-body contacts (ss_env_cfg.self_collision; SELFCOL = true): a
// translation unit of its own so that it compiles next to smplsim_hip.hip instead of after it (ss_env_kernel.h).
#include "ss_env_kernel.h"

namespace ss {

kern_t pick_kernel_selfcol(int variant, bool shaped, bool imit, const Hdr &h, const HdrC &hc) {
  if (imit) return pick_kernel_imitation_selfcol(variant, shaped, h, hc);   // smplsim_hip_im.hip
  if (shaped) {                                              // one geom table per body shape
    if (variant == 0) return ss_env_kernel<2, 2, 1, 1, SS_MAX_THREADS_SC, true, true, HdrRuntime, true>;
    if (variant == 1) return ss_env_kernel<3, 3, 2, 2, SS_MAX_THREADS_X, true, true, HdrRuntime, true>;
    return nullptr;
  }
#ifndef SS_NO_FIXED_LAYOUT
  // the shipped SMPL humanoid with its layout as compile-time constants (as the plain headline kernel: ss_env_kernel.h)
  if (variant == 0 && HdrSmpl::matches(h, hc)) return ss_env_kernel<2, 2, 1, 1, SS_MAX_THREADS_SC, true, false, HdrSmpl, true>;
  if (variant == 1 && ss::HdrSmplxFixedSC::matches(h, hc)) return ss_env_kernel<3, 3, 2, 2, SS_MAX_THREADS_X, true, false, ss::HdrSmplxFixedSC, true>;   // (the plain, non-aliased layout: ss_hdr.h)
#endif
  if (variant == 0) return ss_env_kernel<2, 2, 1, 1, SS_MAX_THREADS_SC, true, false, HdrRuntime, true>;   // (also writes the body frames)
  if (variant == 1) return ss_env_kernel<3, 3, 2, 2, SS_MAX_THREADS_X, true, false, HdrRuntime, true>;
  return nullptr;
}

}  // namespace ss
