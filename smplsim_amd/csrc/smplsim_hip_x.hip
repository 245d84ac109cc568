// smplsim_hip_x.hip — the step kernel's instantiations for the SMPL-X/H size class (52 bodies, nv 159: variant 1 of
// ss::kernel_variant; plain, body-output and per-env-shape flavours).  A translation unit of its own since round 5: this size
// class wants -O2 (smplsim_amd/_lib.py X_OPT: 2.942 / 2.925 ms per 4096-env step against 2.979 / 2.965 at the SMPL kernels' -Os and
// 2.956 / 2.949 at -O3, two seeds, same box: profiles/r05_smplx_flags_ab.txt), and the unit compiles next to the others.
#include "ss_env_kernel.h"

namespace ss {

kern_t pick_kernel_x(int flavour, const Hdr &h, const HdrC &hc) {
#ifndef SS_NO_FIXED_LAYOUT
  // the packaged SMPL-X fixture with its (aliased) LDS layout and its tree as compile-time constants
  if (flavour == 0 && HdrSmplx::matches(h, hc)) return ss_env_kernel<3, 3, 2, 2, SS_MAX_THREADS_X, false, false, HdrSmplx>;
#endif
  if (flavour == 0) return ss_env_kernel<3, 3, 2, 2, SS_MAX_THREADS_X, false, false>;
  if (flavour == 1) return ss_env_kernel<3, 3, 2, 2, SS_MAX_THREADS_X, true, false>;
  return ss_env_kernel<3, 3, 2, 2, SS_MAX_THREADS_X, true, true>;   // per-env body shapes
}

}  // namespace ss
