// smplsim_motion.hip — gfx950 build of the motion-library / imitation kernels + their C ABI (include/smplsim_motion.h).
// A translation unit of its own so that it can be compiled with -O3 while the stepper (smplsim_hip.hip) is compiled with
// -Os: the stepper's big state machine is faster optimised for size (2 % SMPL, 8 % SMPL-X), these small arithmetic kernels
// are twice as slow that way (smplsim_amd/_lib.py).
#include <hip/hip_runtime.h>

#include "ss_motion_api.h"
#include "ss_wave_gpu.h"

namespace {

// ---- motion library (include/smplsim_motion.h; element / wave functions in ss_motion.h).  All HBM-bound gathers: the
// grids are one lane per (frame, body) for FK (tree levels in sequence through a small LDS tile), one wave per clip (the
// sequential Euler-angle fix), one thread per (frame, body) and per (env, body).
template <int LPE>
__global__ void __launch_bounds__(256) ss_motion_fk_kernel(const ss::mo::CookArgs a) {
  __shared__ float xf[4 * 64 * ss::mo::kXformStride];
  WaveGpu w{(int)(threadIdx.x & 63)};
  const int wave = threadIdx.x >> 6;
  ss::mo::fk_wave<WaveGpu, LPE>(&w, a, (int)(blockIdx.x * 4 + wave), xf + wave * 64 * ss::mo::kXformStride);
}
__global__ void __launch_bounds__(64) ss_motion_fix_kernel(const ss::mo::CookArgs a) {
  WaveGpu w{(int)threadIdx.x};
  ss::mo::dof_fix_clip(&w, a, (int)blockIdx.x);
}
__global__ void __launch_bounds__(128) ss_motion_vel_kernel(const ss::mo::CookArgs a) {
  extern __shared__ float raw[];                             // per wave: (tile + 16) frames x J bodies x 6 floats
  WaveGpu w{(int)(threadIdx.x & 63)};
  const int wave = threadIdx.x >> 6, per = (ss::mo::kVelTile + 2 * ss::mo::kGaussRadius) * a.sk.nb * 6;
  const int wave_id = (int)(blockIdx.x * (blockDim.x >> 6)) + wave;
  if (wave_id * ss::mo::kVelTile < a.d.num_frames) ss::mo::vel_wave(&w, a, wave_id, raw + (size_t)wave * per);
}
__global__ void __launch_bounds__(256) ss_motion_state_kernel(const ss::mo::StateArgs a) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const int J = a.d.nbody;
  if (idx < (long long)a.N * J) ss::mo::state_elem(a, (int)(idx / J), (int)(idx % J));
}
__global__ void __launch_bounds__(256) ss_motion_resample_kernel(const ss::mo::ResampleArgs a) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n < a.N) ss::mo::resample_elem(a, n);
}
template <int LPE>
__global__ void __launch_bounds__(256) ss_imitation_kernel(const ss::mo::ImArgs a) {
  WaveGpu w{(int)(threadIdx.x & 63)};
  ss::mo::imitation_wave<WaveGpu, LPE>(&w, a, (int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
}

struct MotionHipBackend {
  static const char *hip_err() { hipError_t e = hipGetLastError(); return e == hipSuccess ? nullptr : hipGetErrorString(e); }
  static const char *motion_cook(const ss::mo::CookArgs &a, void *stream) {
    const int F = a.d.num_frames, J = a.sk.nb;
    if (J <= 32) hipLaunchKernelGGL(ss_motion_fk_kernel<32>, dim3((F + 7) / 8), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(ss_motion_fk_kernel<64>, dim3((F + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(ss_motion_fix_kernel, dim3(a.d.num_motions), dim3(64), 0, (hipStream_t)stream, a);
    const int vwaves = J <= 32 ? 2 : 1, tiles = (F + ss::mo::kVelTile - 1) / ss::mo::kVelTile;      // <= 40 KiB of LDS per workgroup
    const size_t vlds = (size_t)vwaves * (ss::mo::kVelTile + 2 * ss::mo::kGaussRadius) * J * 6 * sizeof(float);
    hipLaunchKernelGGL(ss_motion_vel_kernel, dim3((tiles + vwaves - 1) / vwaves), dim3(64 * vwaves), vlds, (hipStream_t)stream, a);
    return hip_err();
  }
  static const char *motion_state(const ss::mo::StateArgs &a, void *stream) {
    hipLaunchKernelGGL(ss_motion_state_kernel, dim3((unsigned)(((long long)a.N * a.d.nbody + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return hip_err();
  }
  static const char *motion_resample(const ss::mo::ResampleArgs &a, void *stream) {
    hipLaunchKernelGGL(ss_motion_resample_kernel, dim3((a.N + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
    return hip_err();
  }
  static const char *imitation(const ss::mo::ImArgs &a, void *stream) {
    if (a.d.nbody <= 32) hipLaunchKernelGGL(ss_imitation_kernel<32>, dim3((a.N + 7) / 8), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(ss_imitation_kernel<64>, dim3((a.N + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
    return hip_err();
  }
};

}  // namespace

SS_DEFINE_MOTION_API(MotionHipBackend)
