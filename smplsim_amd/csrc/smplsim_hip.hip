// smplsim_hip.hip — gfx950 (MI355X) build of the wavefront env stepper + the C ABI (include/smplsim_hip.h).
//
// Launch geometry: one workgroup per CU = E persistent wavefronts, each stepping one environment at a time (E chosen
// so that the shared index tables + E per-env LDS slices fit the CU's 160 KiB LDS: 12 for SMPL, 7 for SMPL-X); no
// workgroup barrier after the table copy — the wavefronts of a workgroup never talk to each other.  4096 envs =
// 3072 resident wave slots: the first env of a wave is static, the rest come from a device counter.
#include <hip/hip_runtime.h>

#include "ss_env_kernel.h"

namespace {


// GAE: one lane per env column, time loop backwards; every [t, :] row access is coalesced across the wave.  HBM bound
// and tiny (5 arrays of T*N floats), it only exists so that the rollout never leaves the device.
__global__ void ss_gae_kernel(const float *rew, const float *nd, const float *ndead, const float *val, const float *boot,
                              int T, int N, float gamma, float tau, float *adv, float *ret) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
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

// hand-out key of an env for the coming step (ss_schedule_longest_first): a predictor of its Newton-iteration count from what the
// previous step left in HBM.  Last step's count alone finds 45 % of the 256 heaviest envs of the coming step among its first 768;
//     key = iters + 6 (bodies touching the floor) + 8 ln(1 + max |qacc|) + 8 ln(1 + max |qvel|)
// finds 87 % (fitted on the benchmark's own distribution, the same weights hold for the getup, SMPL-X and body-body-contact
// workloads: profiles/r03_lpt_features.txt; a perfect order would make the launch 13 % shorter than the count alone).
// One wavefront per env: two coalesced row reads and a wave maximum.
__global__ void __launch_bounds__(256) ss_key_kernel(const float *qvel, const float *qacc, const int32_t *touch, const int32_t *iters, int nv, int n,
                                                     int32_t *key) {
  const int env = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (env >= n) return;
  float vm = 0.f, am = 0.f;
  for (int i = lane; i < nv; i += 64) {
    const float v = fabsf(qvel[(size_t)env * nv + i]), a = fabsf(qacc[(size_t)env * nv + i]);
    vm = fmaxf(vm, v == v ? v : 1e6f); am = fmaxf(am, a == a ? a : 1e12f);   // NaN counts as huge
  }
  for (int m = 32; m >= 1; m >>= 1) { vm = fmaxf(vm, __shfl_xor(vm, m, 64)); am = fmaxf(am, __shfl_xor(am, m, 64)); }
  if (lane == 0) {
    const int tc = __popc((unsigned)touch[2 * env]) + __popc((unsigned)touch[2 * env + 1]);
    const float k = (float)iters[env] + 6.f * (float)tc + 8.f * log1pf(fminf(am, 1e12f)) + 8.f * log1pf(fminf(vm, 1e6f));
    key[env] = (int32_t)fminf(fmaxf(k, 0.f), 1023.f);
  }
}

// hand-out order = envs by decreasing key: one-workgroup counting sort (1024 buckets)
__global__ void __launch_bounds__(1024) ss_order_kernel(const int32_t *keys, int32_t *order, int n) {
  __shared__ int hist[1024], offs[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) hist[i] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x) { int key = keys[i]; key = key < 0 ? 0 : (key > 1023 ? 1023 : key); atomicAdd(&hist[1023 - key], 1); }
  __syncthreads();
  if (threadIdx.x < 32) {                                     // exclusive prefix over the 1024 buckets: 32 lanes x 32 buckets each
    int acc = 0;
    for (int i = 0; i < 32; i++) { const int h_ = hist[threadIdx.x * 32 + i]; offs[threadIdx.x * 32 + i] = acc; acc += h_; }
    int base = 0;
    for (int l = 0; l < 32; l++) { const int t = __shfl(acc, l, 64); if (l < (int)threadIdx.x) base += t; }
    for (int i = 0; i < 32; i++) offs[threadIdx.x * 32 + i] += base;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x) { int key = keys[i]; key = key < 0 ? 0 : (key > 1023 ? 1023 : key); order[atomicAdd(&offs[1023 - key], 1)] = i; }
}


typedef ss::kern_t kern_t;
// instantiations per model size: plain (the headline), +body-frame outputs, +per-env body shapes (which includes the outputs)
// The two fixtures' sizes get instantiations with compile-time dimensions and LDS layout (HdrFixedT, ss_hdr.h): -2.4% per step
// launch on the SMPL headline (profiles/r02b_variants.txt); any other model of a variant's size class runs the generic one.
kern_t pick_kernel(int variant, int flavour, const ss::Hdr &h, const ss::HdrC &hc) {
#ifndef SS_NO_FIXED_LAYOUT
  if (variant == 0 && flavour == 0 && HdrSmpl::matches(h, hc)) return ss_env_kernel<2, 2, 1, 1, SS_MAX_THREADS, false, false, HdrSmpl>;
#ifndef SS_ONLY_HEADLINE
  if (variant == 0 && flavour == 1 && HdrSmpl::matches(h, hc)) return ss_env_kernel<2, 2, 1, 1, SS_MAX_THREADS, true, false, HdrSmpl>;
  if (variant == 0 && flavour == 2 && HdrSmpl::matches(h, hc)) return ss_env_kernel<2, 2, 1, 1, SS_MAX_THREADS, true, true, HdrSmpl>;   // per-env body shapes
#endif
#endif
  if (variant == 0 && flavour == 0) return ss_env_kernel<2, 2, 1, 1, SS_MAX_THREADS, false, false>;
#ifndef SS_ONLY_HEADLINE                                     // (experiment builds, tools/build_variant.sh, keep the headline kernel alone)
  if (flavour == 3 || flavour == 5 || flavour == 7) return ss::pick_kernel_selfcol(variant, flavour == 5, flavour == 7, h, hc);   // smplsim_hip_sc.hip
  if (flavour == 4 || flavour == 6) return ss::pick_kernel_imitation(variant, flavour == 6, h, hc);                          // smplsim_hip_im.hip
  if (variant == 0) {                                        // SMPL layout (24 bodies)
    if (flavour == 1) return ss_env_kernel<2, 2, 1, 1, SS_MAX_THREADS, true, false>;
    return ss_env_kernel<2, 2, 1, 1, SS_MAX_THREADS, true, true>;
  }
  if (variant == 1) return ss::pick_kernel_x(flavour, h, hc);   // SMPL-X/H layout (52 bodies): smplsim_hip_x.hip
#endif
  return nullptr;
}

struct HipBackend {
  static void *alloc(size_t n) { void *p = nullptr; return hipMalloc(&p, n) == hipSuccess ? p : nullptr; }
  static void free_(void *p) { if (p) (void)hipFree(p); }
  static bool upload(void *dst, const void *src, size_t n) { return hipMemcpy(dst, src, n, hipMemcpyHostToDevice) == hipSuccess; }
  static bool set_device(int d) { return hipSetDevice(d) == hipSuccess; }
  static bool download(void *dst, const void *src, size_t n) { return hipMemcpy(dst, src, n, hipMemcpyDeviceToHost) == hipSuccess; }
  static bool copy_d2d(void *dst, const void *src, size_t n, void *stream) { return hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, (hipStream_t)stream) == hipSuccess; }
  static int lds_capacity() { return 160 * 1024; }
  static int num_cus() { int d = 0, n = 256; if (hipGetDevice(&d) == hipSuccess) { hipDeviceProp_t p; if (hipGetDeviceProperties(&p, d) == hipSuccess) n = p.multiProcessorCount; } return n; }
  static int max_waves(int variant, int selfcol) { return (variant == 0 ? (selfcol ? SS_MAX_THREADS_SC : SS_MAX_THREADS) : SS_MAX_THREADS_X) / 64; }
  static const char *order_by_key(const ss_state &st, int nv, int32_t *key, int32_t *order, void *stream) {
    const int n = st.num_envs;
    hipLaunchKernelGGL(ss_key_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, st.qvel, st.qacc_warm, st.touch, st.solver_iters, nv, n, key);
    hipLaunchKernelGGL(ss_order_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, key, order, n);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? nullptr : hipGetErrorString(e);
  }
  static const char *gae(const float *rew, const float *nd, const float *ndead, const float *val, const float *boot, int T, int N,
                         float gamma, float tau, float *adv, float *ret, void *stream) {
    hipLaunchKernelGGL(ss_gae_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, rew, nd, ndead, val, boot, T, N, gamma, tau, adv, ret);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? nullptr : hipGetErrorString(e);
  }
  static int &regs_ref() { static thread_local int r = 0; return r; }
  static int kernel_regs() { return regs_ref(); }
  static const char *launch(const ss::KArgs &k, int nenv, int envs_per_wg, size_t lds_bytes, void *stream, int fixed_epw, int max_wgs) {
    const bool bodyout = (k.out0 || k.power) && (k.mode == ss::MODE_STEP || k.mode == ss::MODE_RESET);
    const int flavour = k.im ? (k.cfg.self_collision ? 7 : (k.st.shape_id ? 6 : 4)) : (k.cfg.self_collision ? (k.st.shape_id ? 5 : 3) : (k.st.shape_id ? 2 : (bodyout ? 1 : 0)));
    kern_t kern = pick_kernel(ss::kernel_variant(k.h), flavour, k.h, k.hc);
    if (!kern) return "no kernel variant for this model size";
    static thread_local kern_t configured[32] = {};
    static thread_local size_t configured_lds[32] = {};
    static thread_local int regs_by_slot[32] = {};            // VGPRs per instantiation: launch_info reports the LAUNCHED kernel's
    const int slot = 16 * ss::kernel_variant(k.h) + flavour;
    if (configured[slot] != kern || configured_lds[slot] < lds_bytes) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
      if (e != hipSuccess) return hipGetErrorString(e);
      hipFuncAttributes fa;
      if (hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kern)) == hipSuccess) regs_by_slot[slot] = fa.numRegs;
      configured[slot] = kern; configured_lds[slot] = lds_bytes;
    }
    regs_ref() = regs_by_slot[slot];                          // of this launch (ss_api::run copies it into the batch)
    static thread_local int cus = num_cus();
    // small batches: spread the envs over all CUs instead of filling a third of them with full workgroups — a wave that
    // shares its CU with 3 others runs ~13% faster than one of 12 (1024 envs: 1.27 -> 1.13 ms per step launch)
    const int per_cu = (nenv + cus - 1) / cus;
    if (fixed_epw > 0) {                                      // fixed by the caller (ss_set_launch_geometry)
      envs_per_wg = fixed_epw;
      lds_bytes = ss::launch_lds_bytes(k, envs_per_wg);
    } else if (per_cu < envs_per_wg) {
      envs_per_wg = per_cu < 1 ? 1 : per_cu;
      lds_bytes = ss::launch_lds_bytes(k, envs_per_wg);
    }
    int wgs = (nenv + envs_per_wg - 1) / envs_per_wg;
    const int resident = cus * (int)(lds_capacity() / lds_bytes > 0 ? lds_capacity() / lds_bytes : 1);
    if (wgs > resident) wgs = resident;                      // persistent: one resident set of workgroups
    if (max_wgs > 0 && wgs > max_wgs) wgs = max_wgs;         // the caller shares the GPU between concurrent batches
    dim3 grid(wgs), block(64 * envs_per_wg);
    hipLaunchKernelGGL(kern, grid, block, lds_bytes, (hipStream_t)stream, k);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? nullptr : hipGetErrorString(e);
  }
};

}  // namespace

SS_DEFINE_C_API(HipBackend)
