// ss_api.h — host side of the C ABI (include/smplsim_hip.h), written once and instantiated with a
// backend: the HIP backend in smplsim_hip.hip (the product) and the wavefront-emulator backend in
// tests/wave_emu/emu.cpp (unit-test infrastructure).  A backend provides:
//   static void *alloc(size_t);  static void free_(void *);  static bool upload(void *dst, const void *src, size_t);
//   static int lds_capacity();   static int max_waves(int variant);   static const char *launch(const ss::KArgs &, int nenv, int envs_per_wg, size_t lds_bytes, void *stream, int fixed_envs_per_wg, int max_wgs);
//   static bool set_device(int);
#pragma once
#include <cstdlib>
#include <new>
#include <string>

#include "ss_imfused.h"
#include "ss_mjcf.h"
#include "ss_tables.h"

namespace ss {
inline std::string &last_error() { static thread_local std::string e; return e; }
inline std::string *&error_slot() { static thread_local std::string *p = nullptr; return p; }   // the handle the running entry works on
struct HandleScope {                      // entries that take a handle also record their error message in it (ss_batch_last_error)
  explicit HandleScope(std::string *slot) { error_slot() = slot; }
  ~HandleScope() { error_slot() = nullptr; }
};
}  // namespace ss

struct ss_model {
  ss::HostModel hm;
  int device = 0;
  uint32_t *d_shared = nullptr;
  ss::real *d_bodyc = nullptr, *d_candc = nullptr;
  int32_t *d_candb = nullptr;
  int32_t *d_pairs = nullptr;             // body-body candidate pairs, geom table (self_collision batches)
  ss::real *d_geomc = nullptr;
  int num_shapes = 1;                     // ss_model_create_shapes: d_bodyc / d_candc hold num_shapes consecutive tables (ss_hdr.h)
  mutable std::string err;                // message of the last failed call that took this handle
};
struct ss_batch {
  const ss_model *m = nullptr;
  ss_env_cfg cfg{};
  ss_state st{};
  int envs_per_wg = 1;
  size_t lds_bytes = 0;
  int obs_size = 0;
  int32_t *d_counter = nullptr;
  unsigned long long *d_prof = nullptr;   // SS_PROFILE builds only
  const int32_t *order = nullptr;         // caller-owned device array or null
  int32_t *d_sched = nullptr;             // library-owned [2N]: hand-out order written by ss_schedule_longest_first, then its keys
  float *body_xpos = nullptr, *body_xmat = nullptr;   // caller-owned, optional: written by every step / reset (ss_set_body_outputs)
  static ss::real *R(float *p) { return reinterpret_cast<ss::real *>(p); }               // C-ABI arrays as the kernel's scalar type
  static const ss::real *R(const float *p) { return reinterpret_cast<const ss::real *>(p); }
  float *dbg_self = nullptr;              // caller-owned, optional (ss_debug_self_contacts)
  int32_t *self_trunc = nullptr;          // caller-owned, optional [N] (ss_debug_self_truncation)
  float *power = nullptr;                 // caller-owned, optional [N, control_freq_inv, nv - 6] (ss_set_power_output)
  int fixed_envs_per_wg = 0, max_wgs = 0; // ss_set_launch_geometry: 0 = automatic (small batches are spread over all CUs)
  mutable std::string err;                // message of the last failed call that took this handle
  mutable int kernel_regs = 0;            // VGPRs of the kernel instantiation this batch launched last (ss_launch_info)
  mutable int parity = 0;                 // which of the two work counters the next launch uses (it clears the other one)
  float *d_kin = nullptr;                 // [N, nb, 12] scratch of ss_get_state(XPOS / XMAT)
  const float *fall_actions = nullptr;    // caller-owned [N,3,nu] draws of the in-launch Fall reset (ss_set_fall_actions)
  void *d_im = nullptr;                   // ss::mo::ImFused on the device (ss_imitation_bind)
  ss_imitation_io im_io{};
};

inline bool same_pair_ids(const std::vector<int32_t> &a, const std::vector<int32_t> &b) {
  if (a.size() != b.size()) return false;
  for (size_t i = 0; i < a.size(); i += 2) if (a[i] != b[i]) return false;
  return true;
}

template <class BE>
struct ss_api {
  static int fail(int code, const std::string &msg) {
    ss::last_error() = msg;
    if (ss::error_slot()) *ss::error_slot() = msg;
    return code;
  }

  static int model_create(const ss_model_desc *d, int device, ss_model **out) {
    if (!d || !out) return fail(SS_ERR_INVALID, "null argument");
    ss_model *m = new (std::nothrow) ss_model();
    if (!m) return fail(SS_ERR_NOMEM, "out of host memory");
    if (!ss::build_host_model(*d, m->hm)) { std::string e = m->hm.error; delete m; return fail(SS_ERR_INVALID, e); }
    m->device = device;
    if (!BE::set_device(device)) { delete m; return fail(SS_ERR_HIP, "cannot select device"); }
    auto up = [&](const void *src, size_t bytes) -> void * {
      void *p = BE::alloc(bytes ? bytes : 4);
      if (p && bytes && !BE::upload(p, src, bytes)) { BE::free_(p); p = nullptr; }
      return p;
    };
    m->d_shared = (uint32_t *)up(m->hm.shared.data(), m->hm.shared.size() * 4);
    m->d_bodyc = (ss::real *)up(m->hm.bodyc.data(), m->hm.bodyc.size() * sizeof(ss::real));
    m->d_candc = (ss::real *)up(m->hm.candc.data(), m->hm.candc.size() * sizeof(ss::real));
    m->d_candb = (int32_t *)up(m->hm.candb.data(), m->hm.candb.size() * 4);
    {
      std::vector<int32_t> pt(m->hm.pairs);                  // pair table, then the per-body elimination-tree table of the SELFCOL kernels
      m->hm.sc.o_sctab = (int)pt.size();
      pt.insert(pt.end(), m->hm.sctab.begin(), m->hm.sctab.end());
      m->d_pairs = (int32_t *)up(pt.data(), pt.size() * 4);
    }
    m->d_geomc = (ss::real *)up(m->hm.geomc.data(), m->hm.geomc.size() * sizeof(ss::real));
    if (!m->d_shared || !m->d_bodyc || !m->d_candc || !m->d_candb || !m->d_pairs || !m->d_geomc) { model_destroy(m); return fail(SS_ERR_HIP, "device table upload failed"); }
    *out = m;
    return SS_OK;
  }
  // S descriptions of the same humanoid with different body shapes -> one model whose geometry tables have S entries
  static int model_create_shapes(const ss_model_desc *d, int num_shapes, int device, ss_model **out) {
    if (!d || !out || num_shapes < 1) return fail(SS_ERR_INVALID, "null argument or num_shapes < 1");
    if (num_shapes == 1) return model_create(d, device, out);
    ss_model *m = new (std::nothrow) ss_model();
    if (!m) return fail(SS_ERR_NOMEM, "out of host memory");
    std::vector<ss::real> bodyc, candc, geomc;
    std::vector<int32_t> pairs;                              // per shape: the ids are the same, the bounding-sphere reaches are not
    for (int s = 0; s < num_shapes; s++) {
      ss::HostModel hm;
      if (!ss::build_host_model(d[s], hm)) { std::string e = "shape " + std::to_string(s) + ": " + hm.error; delete m; return fail(SS_ERR_INVALID, e); }
      const ss::Hdr &h = hm.h;
      std::vector<ss::real> iw(h.nv);
      ss::real *sf = ss::shared_reals(hm);
      for (int i = 0; i < h.nv; i++) { iw[i] = sf[h.o_dofc + i * ss::kDofC + 4]; sf[h.o_dofc + i * ss::kDofC + 4] = 0; }
      for (int i = 0; i < 3 * h.nb; i++) sf[h.o_boff + i] = 0;              // the two shape-dependent parts of the shared blob
      if (s == 0) m->hm = hm;
      else {
        const ss::Hdr &g = m->hm.h;
        // everything but the geometry must agree: tree, joints, limits, gains, actuators, geom types, contact set, options
        const bool same = h.nb == g.nb && h.nv == g.nv && h.nu == g.nu && h.ncand == g.ncand && h.nbox == g.nbox && h.nslot == g.nslot &&
                          h.shared_words == g.shared_words && h.env_floats == g.env_floats && hm.shared == m->hm.shared &&
                          hm.candb == m->hm.candb && same_pair_ids(hm.pairs, m->hm.pairs) && hm.illegal_mask == m->hm.illegal_mask && h.dt == g.dt && h.grav == g.grav &&
                          h.margin == g.margin && h.mu == g.mu && h.K == g.K && h.B == g.B;
        if (!same) { delete m; return fail(SS_ERR_INVALID, "shape " + std::to_string(s) + " differs from shape 0 in more than its geometry"); }
      }
      bodyc.insert(bodyc.end(), hm.bodyc.begin(), hm.bodyc.end());
      iw.resize((h.nv + 3) & ~3, ss::real(0));
      bodyc.insert(bodyc.end(), iw.begin(), iw.end());        // block = body constants, then the dof inverse weights
      candc.insert(candc.end(), hm.candc.begin(), hm.candc.end());
      geomc.insert(geomc.end(), hm.geomc.begin(), hm.geomc.end());   // pair functions (self_collision batches): geoms of this shape
      pairs.insert(pairs.end(), hm.pairs.begin(), hm.pairs.end());
    }
    if (12 * m->hm.h_sc.nb > m->hm.h_sc.l_Wst - m->hm.h_sc.l_IA) { delete m; return fail(SS_ERR_LDS, "no room for the per-env body offsets"); }   // (the aliased layout has the 12 nb by construction)
    m->device = device; m->num_shapes = num_shapes;
    if (!BE::set_device(device)) { delete m; return fail(SS_ERR_HIP, "cannot select device"); }
    auto up = [&](const void *src, size_t bytes) -> void * {
      void *p = BE::alloc(bytes ? bytes : 4);
      if (p && bytes && !BE::upload(p, src, bytes)) { BE::free_(p); p = nullptr; }
      return p;
    };
    m->d_shared = (uint32_t *)up(m->hm.shared.data(), m->hm.shared.size() * 4);
    m->d_bodyc = (ss::real *)up(bodyc.data(), bodyc.size() * sizeof(ss::real));
    m->d_candc = (ss::real *)up(candc.data(), candc.size() * sizeof(ss::real));
    m->d_candb = (int32_t *)up(m->hm.candb.data(), m->hm.candb.size() * 4);
    m->hm.sc.o_sctab = (int)pairs.size();
    pairs.insert(pairs.end(), m->hm.sctab.begin(), m->hm.sctab.end());
    m->d_pairs = (int32_t *)up(pairs.data(), pairs.size() * 4);
    m->d_geomc = (ss::real *)up(geomc.data(), geomc.size() * sizeof(ss::real));
    if (!m->d_shared || !m->d_bodyc || !m->d_candc || !m->d_candb || !m->d_pairs || !m->d_geomc) { model_destroy(m); return fail(SS_ERR_HIP, "device table upload failed"); }
    *out = m;
    return SS_OK;
  }
  static void model_destroy(ss_model *m) {
    if (!m) return;
    BE::free_(m->d_shared); BE::free_(m->d_bodyc); BE::free_(m->d_candc); BE::free_(m->d_candb); BE::free_(m->d_pairs); BE::free_(m->d_geomc);
    delete m;
  }
  static int batch_create(const ss_model *m, const ss_env_cfg *cfg, const ss_state *st, ss_batch **out) {
    if (!m || !cfg || !st || !out) return fail(SS_ERR_INVALID, "null argument");
    if (st->num_envs < 1) return fail(SS_ERR_INVALID, "num_envs must be positive");
    if (!st->qpos || !st->qvel || !st->qpos_prev || !st->qvel_prev || !st->qacc_warm || !st->body_vel || !st->touch ||
        !st->cur_t || !st->task || !st->nwarn || !st->solver_iters)
      return fail(SS_ERR_INVALID, "every ss_state buffer must be provided");
    if (cfg->control_mode < SS_CTRL_UHC_PD || cfg->control_mode > SS_CTRL_DEFAULT) return fail(SS_ERR_INVALID, "unknown control_mode");
    if (cfg->control_mode == SS_CTRL_SIMPLE_PID && (!st->pid_integral || !st->pid_last_error || !st->pid_started))
      return fail(SS_ERR_INVALID, "control_mode simple_pid needs the pid_* state buffers");
    if (cfg->self_obs_v != 1 && cfg->self_obs_v != 2) return fail(SS_ERR_INVALID, "self_obs_v must be 1 or 2");
    if (cfg->control_freq_inv < 1) return fail(SS_ERR_INVALID, "control_freq_inv must be >= 1");
    if (m->num_shapes > 1 && !st->shape_id) return fail(SS_ERR_INVALID, "a model with several shapes needs ss_state.shape_id");
    if (m->num_shapes == 1 && st->shape_id) return fail(SS_ERR_INVALID, "ss_state.shape_id given for a single-shape model");
    if (cfg->self_collision && !m->d_pairs) return fail(SS_ERR_INVALID, "model has no pair table");
    if (cfg->self_collision && !m->hm.self_collision_unavailable.empty()) return fail(SS_ERR_INVALID, m->hm.self_collision_unavailable.c_str());
    if (cfg->task == SS_TASK_REACH && (cfg->reach_body < 0 || cfg->reach_body >= m->hm.h.nb)) return fail(SS_ERR_INVALID, "reach_body out of range");
    const ss::Hdr &h = m->hm.h;
    if (ss::kernel_variant(h) < 0) return fail(SS_ERR_INVALID, "model too large for the compiled kernel variants");
    ss_batch *b = new (std::nothrow) ss_batch();
    if (!b) return fail(SS_ERR_NOMEM, "out of host memory");
    b->m = m; b->cfg = *cfg; b->st = *st;
    // mjOption.iterations / tolerance (MuJoCo's defaults: the reference MJCF sets neither).  The kernel gets the tolerance in
    // cost units: tolerance / scale, scale = 1 / (meaninertia * max(1, nv)) as in mj_solPrimal (shape 0's meaninertia for
    // a model with several body shapes)
    if (b->cfg.newton_iters <= 0) b->cfg.newton_iters = 100;
    b->cfg.solver_tolerance = (float)((cfg->solver_tolerance > 0.f ? (double)cfg->solver_tolerance : 1e-8) * m->hm.meaninertia * (h.nv > 1 ? h.nv : 1));
    b->obs_size = ss::obs_size(h, *cfg);
    size_t shared_b = (size_t)((h.shared_words + 3) & ~3) * 4;
    const size_t env_b = (size_t)(cfg->self_collision ? m->hm.sc.env_floats : h.env_floats) * sizeof(ss::real);
    int cap = BE::lds_capacity();
    const size_t pool_b = cfg->self_collision ? (size_t)ss::ss_pool_floats(m->hm.sc) * sizeof(ss::real) : 0;   // the workgroup's shared dense block
    int e = (int)((cap - (long)shared_b - (long)pool_b) / (long)env_b);
    if (e < 1) { delete b; return fail(SS_ERR_LDS, "model does not fit in LDS"); }
    { const int mw = BE::max_waves(ss::kernel_variant(h), cfg->self_collision); if (e > mw) e = mw; }   // launch bound of the kernel variant
    { const char *cap = getenv("SS_ENVS_PER_WG"); if (cap && atoi(cap) > 0 && atoi(cap) < e) e = atoi(cap); }
    b->envs_per_wg = e;
    b->lds_bytes = shared_b + (size_t)e * env_b + pool_b;
#ifdef SS_PROFILE
    b->d_prof = (unsigned long long *)BE::alloc(64 * sizeof(unsigned long long));
    if (b->d_prof) { unsigned long long z[64] = {0}; BE::upload(b->d_prof, z, sizeof z); }
#endif
    if (!BE::set_device(m->device)) { delete b; return fail(SS_ERR_HIP, "cannot select device"); }   // the counter must live on the batch's GPU
    b->d_counter = (int32_t *)BE::alloc(2 * sizeof(int32_t));
    const int32_t zeros[2] = {0, 0};
    if (!b->d_counter || !BE::upload(b->d_counter, zeros, sizeof zeros)) { BE::free_(b->d_counter); delete b; return fail(SS_ERR_NOMEM, "device allocation failed"); }
    *out = b;
    return SS_OK;
  }
  static ss::KArgs base_args(const ss_batch *b, int mode) {
    ss::KArgs k{};
    const ss_model *m = b->m;
    k.h = b->cfg.self_collision ? m->hm.h_sc : m->hm.h; k.cfg = b->cfg; k.st = b->st;   // (body-body-contact batches: the plain LDS layout, ss_tables.h)
    k.shared_g = m->d_shared; k.bodyc = m->d_bodyc; k.candc = m->d_candc; k.candb = m->d_candb;
    k.illegal_mask = m->hm.illegal_mask;
    k.hc = m->hm.hc;
    k.sc = m->hm.sc; k.pairs = m->d_pairs; k.geomc = m->d_geomc; k.dbg_self = ss_batch::R(b->dbg_self);
    k.mode = mode; k.nsub = b->cfg.control_freq_inv; k.obs_size = b->obs_size;
    k.work_counter = b->d_counter + b->parity; k.work_counter_next = b->d_counter + (b->parity ^ 1);   // run() flips the parity
    k.obs_stride = b->obs_size;
    k.prof = b->d_prof;
    k.order = b->order;
    if (mode == ss::MODE_STEP || mode == ss::MODE_RESET) { k.out0 = ss_batch::R(b->body_xpos); k.out1 = ss_batch::R(b->body_xmat); }
    if (mode == ss::MODE_STEP) { k.power = ss_batch::R(b->power); k.self_trunc = b->self_trunc; }
    return k;
  }
  static int run(const ss_batch *b, const ss::KArgs &k, void *stream) {
    if (!BE::set_device(b->m->device)) return fail(SS_ERR_HIP, "cannot select device");
    const char *err = BE::launch(k, b->st.num_envs, b->envs_per_wg, b->lds_bytes, stream, b->fixed_envs_per_wg, b->max_wgs);
    if (err) return fail(SS_ERR_HIP, err);
    b->kernel_regs = BE::kernel_regs();
    // only a launch that ran has zeroed the other work counter: a failed one leaves the pair as it was (the counter it would
    // have used is still zero), so the next launch starts from a clean counter either way
    b->parity ^= 1;
    return SS_OK;
  }
  static int reset(ss_batch *b, const uint8_t *mask, const float *fall_actions, const float *task_rand, float *obs, void *stream) {
    if (!b || !obs) return fail(SS_ERR_INVALID, "null argument");
    if (b->cfg.state_init == SS_INIT_FALL && !fall_actions) return fail(SS_ERR_INVALID, "StateInit.Fall needs fall_actions");
    ss::KArgs k = base_args(b, ss::MODE_RESET);
    k.mask = mask; k.fall_actions = ss_batch::R(fall_actions); k.task_rand = ss_batch::R(task_rand); k.obs = ss_batch::R(obs);
    return run(b, k, stream);
  }
  static int step(ss_batch *b, const float *actions, const float *task_rand, float *obs, float *reward, uint8_t *term,
                  uint8_t *trunc, void *stream) {
    if (!b || !actions || !obs || !reward || !term || !trunc) return fail(SS_ERR_INVALID, "null argument");
    ss::KArgs k = base_args(b, ss::MODE_STEP);
    k.actions = ss_batch::R(actions); k.task_rand = ss_batch::R(task_rand); k.obs = ss_batch::R(obs); k.reward = ss_batch::R(reward);
    k.terminated = term; k.truncated = trunc;
    return run(b, k, stream);
  }
  static int step_autoreset(ss_batch *b, const float *actions, const float *task_rand, const float *reset_task_rand, float *obs,
                            float *obs_next, float *reward, uint8_t *term, uint8_t *trunc, void *stream) {
    if (!b || !actions || !obs || !obs_next || !reward || !term || !trunc) return fail(SS_ERR_INVALID, "null argument");
    if (b->cfg.state_init == SS_INIT_EXTERNAL) return fail(SS_ERR_INVALID, "in-launch autoreset: StateInit External has no reset state of its own");
    if (b->cfg.state_init == SS_INIT_FALL && !b->fall_actions) return fail(SS_ERR_INVALID, "in-launch Fall reset needs ss_set_fall_actions");
    ss::KArgs k = base_args(b, ss::MODE_STEP);
    k.actions = ss_batch::R(actions); k.task_rand = ss_batch::R(task_rand); k.obs = ss_batch::R(obs); k.reward = ss_batch::R(reward);
    k.terminated = term; k.truncated = trunc;
    k.fused_reset = 1; k.obs2 = ss_batch::R(obs_next); k.task_rand2 = ss_batch::R(reset_task_rand);
    if (b->cfg.state_init == SS_INIT_FALL) k.fall_actions = ss_batch::R(b->fall_actions);
    return run(b, k, stream);
  }
  // ---- imitation folded into the step launch (include/smplsim_motion.h)
  static int imitation_bind(ss_batch *b, const ss_imitation_io *io) {
    if (!b || !io) return fail(SS_ERR_INVALID, "null argument");
    if (sizeof(ss::real) != sizeof(float)) return fail(SS_ERR_INVALID, "not available in the float64 build");
    if (const char *e = ss::mo::check_data(io->data, false)) return fail(SS_ERR_INVALID, e);
    const ss::Hdr &h = b->m->hm.h;
    if (io->data->nbody != h.nb) return fail(SS_ERR_INVALID, "motion data and model disagree on nbody");
    if (b->cfg.state_init != SS_INIT_EXTERNAL || b->cfg.task != SS_TASK_BASE)
      return fail(SS_ERR_INVALID, "the fused imitation step needs a batch with task base and StateInit External");
    if (b->cfg.self_collision && (b->st.shape_id || ss::kernel_variant(h) != 0))
      return fail(SS_ERR_INVALID, "fused imitation step with self_collision: SMPL-sized single-shape models only (use the separate launches)");
    if (!b->body_xpos || !b->body_xmat) return fail(SS_ERR_INVALID, "call ss_set_body_outputs first");
    if (!io->motion_ids || !io->start_times || !io->obs_final || !io->obs_next || !io->reward || !io->terminated || !io->truncated)
      return fail(SS_ERR_INVALID, "null buffer in ss_imitation_io");
    if (io->obs_stride < b->obs_size + 24 * h.nb) return fail(SS_ERR_INVALID, "obs_stride smaller than self + task observation");
    if (io->truncate_time < 0.f) return fail(SS_ERR_INVALID, "truncate_time must be >= 0");
    ss::mo::ImFused f{};
    f.im = ss::mo::ImArgs{*io->data, io->cfg, io->motion_ids, io->start_times, b->st.cur_t, io->offset, nullptr, b->st.num_envs,
                          b->body_xpos, b->body_xmat, b->st.body_vel, io->obs_final + b->obs_size, io->obs_stride, io->reward,
                          io->reward_parts, io->terminated, io->truncated};
    f.ids = io->motion_ids; f.start_times = io->start_times; f.cdf = io->sampling_cdf; f.truncate = io->truncate_time;
    f.random_start = io->random_start; f.obs_final = io->obs_final; f.obs_next = io->obs_next;
    f.obs_floats = b->obs_size + 24 * h.nb; f.qpos = b->st.qpos; f.qvel = b->st.qvel; f.nq = h.nq; f.nv = h.nv;
    if (!BE::set_device(b->m->device)) return fail(SS_ERR_HIP, "cannot select device");
    if (!b->d_im) b->d_im = BE::alloc(sizeof f);
    if (!b->d_im || !BE::upload(b->d_im, &f, sizeof f)) return fail(SS_ERR_NOMEM, "device allocation failed");
    b->im_io = *io;
    return SS_OK;
  }
  static int imitation_step_fused(ss_batch *b, const float *actions, const float *rand, void *stream) {
    if (!b || !actions) return fail(SS_ERR_INVALID, "null argument");
    if (!b->d_im) return fail(SS_ERR_INVALID, "ss_imitation_bind first");
    if (rand && !b->im_io.sampling_cdf) return fail(SS_ERR_INVALID, "re-initialisation needs ss_imitation_io.sampling_cdf");
    ss::KArgs k = base_args(b, ss::MODE_STEP);
    k.actions = ss_batch::R(actions);
    k.obs = ss_batch::R(b->im_io.obs_final); k.obs2 = ss_batch::R(b->im_io.obs_next); k.obs_stride = b->im_io.obs_stride;
    k.im = b->d_im; k.im_rand = rand;                        // reward / flags come from the imitation part
    return run(b, k, stream);
  }
  // ---- state by field (device-to-device copies between the bound buffers and the caller's)
  static int state_field(ss_batch *b, int field, void *buf, bool set, void *stream) {
    if (!b || !buf) return fail(SS_ERR_INVALID, "null argument");
    if (sizeof(ss::real) != sizeof(float)) return fail(SS_ERR_INVALID, "not available in the float64 build");
    const ss::Hdr &h = b->m->hm.h;
    const size_t N = (size_t)b->st.num_envs;
    if (!BE::set_device(b->m->device)) return fail(SS_ERR_HIP, "cannot select device");
    auto copy = [&](void *mine, size_t bytes) { return (set ? BE::copy_d2d(mine, buf, bytes, stream) : BE::copy_d2d(buf, mine, bytes, stream)) ? SS_OK : fail(SS_ERR_HIP, "device copy failed"); };
    switch (field) {
      case SS_FIELD_QPOS: {
        if (int rc = copy(b->st.qpos, N * h.nq * 4)) return rc;
        return set ? (BE::copy_d2d(b->st.qpos_prev, buf, N * h.nq * 4, stream) ? SS_OK : fail(SS_ERR_HIP, "device copy failed")) : SS_OK;
      }
      case SS_FIELD_QVEL: {
        if (int rc = copy(b->st.qvel, N * h.nv * 4)) return rc;
        return set ? (BE::copy_d2d(b->st.qvel_prev, buf, N * h.nv * 4, stream) ? SS_OK : fail(SS_ERR_HIP, "device copy failed")) : SS_OK;
      }
      case SS_FIELD_QACC_WARM: return copy(b->st.qacc_warm, N * h.nv * 4);
      case SS_FIELD_CUR_T: return copy(b->st.cur_t, N * 4);
      default: break;
    }
    if (set) return fail(SS_ERR_INVALID, "ss_set_state: field is read-only (QPOS, QVEL, QACC_WARM, CUR_T can be set)");
    switch (field) {
      case SS_FIELD_BODY_VEL: return copy(b->st.body_vel, N * h.nb * 6 * 4);
      case SS_FIELD_TOUCH: return copy(b->st.touch, N * 2 * 4);
      case SS_FIELD_XPOS: case SS_FIELD_XMAT: {
        if (!b->d_kin) b->d_kin = (float *)BE::alloc(N * h.nb * 12 * 4);
        if (!b->d_kin) return fail(SS_ERR_NOMEM, "device allocation failed");
        float *xp = b->d_kin, *xm = b->d_kin + N * h.nb * 3;
        if (int rc = kinematics(b, xp, xm, stream)) return rc;
        return field == SS_FIELD_XPOS ? copy(xp, N * h.nb * 3 * 4) : copy(xm, N * h.nb * 9 * 4);
      }
      default: return fail(SS_ERR_INVALID, "unknown state field");
    }
  }
  static int substep(ss_batch *b, const float *actions, int n, void *stream) {
    if (!b || !actions || n < 1) return fail(SS_ERR_INVALID, "bad argument");
    ss::KArgs k = base_args(b, ss::MODE_SUBSTEP);
    k.actions = ss_batch::R(actions); k.nsub = n;
    return run(b, k, stream);
  }
  static int kinematics(ss_batch *b, float *xpos, float *xmat, void *stream) {
    if (!b || !xpos || !xmat) return fail(SS_ERR_INVALID, "null argument");
    ss::KArgs k = base_args(b, ss::MODE_KINEMATICS);
    k.out0 = ss_batch::R(xpos); k.out1 = ss_batch::R(xmat);
    return run(b, k, stream);
  }
  static int debug_forward(ss_batch *b, const float *torques, float *M, float *bias, float *qacc, void *stream) {
    if (!b || !M || !bias || !qacc) return fail(SS_ERR_INVALID, "null argument");
    ss::KArgs k = base_args(b, ss::MODE_DEBUG_FORWARD);
    k.actions = ss_batch::R(torques); k.out0 = ss_batch::R(M); k.out1 = ss_batch::R(bias); k.out2 = ss_batch::R(qacc);
    return run(b, k, stream);
  }
};

// The extern "C" surface, identical for every backend.
#define SS_DEFINE_C_API(BE)                                                                                          \
  extern "C" {                                                                                                       \
  int ss_model_create(const ss_model_desc *d, int dev, ss_model **out) { return ss_api<BE>::model_create(d, dev, out); } \
  int ss_model_create_shapes(const ss_model_desc *d, int32_t n, int dev, ss_model **out) { return ss_api<BE>::model_create_shapes(d, n, dev, out); } \
  void ss_model_destroy(ss_model *m) { ss_api<BE>::model_destroy(m); }                                               \
  int ss_model_dims(const ss_model *m, int32_t *nq, int32_t *nv, int32_t *nu, int32_t *nb) {                          \
    ss::HandleScope hs_(m ? &m->err : nullptr);                                                                      \
    if (!m) return ss_api<BE>::fail(SS_ERR_INVALID, "null model");                                                   \
    if (nq) *nq = m->hm.h.nq;                                                                                        \
    if (nv) *nv = m->hm.h.nv;                                                                                        \
    if (nu) *nu = m->hm.h.nu;                                                                                        \
    if (nb) *nb = m->hm.h.nb;                                                                                        \
    return SS_OK;                                                                                                    \
  }                                                                                                                  \
  int ss_model_elimination_tree(const ss_model *m, int32_t *root, int32_t *levels, int32_t *widest, int32_t *widths, int32_t *most_children) { \
    ss::HandleScope hs_(m ? &m->err : nullptr);                                                                      \
    if (!m) return ss_api<BE>::fail(SS_ERR_INVALID, "null model");                                                   \
    const ss::HdrC &c = m->hm.hc;                                                                                    \
    if (root) *root = c.root;                                                                                        \
    if (levels) *levels = c.nlev;                                                                                    \
    if (widest) *widest = m->hm.h.maxlev;                                                                            \
    if (widths) {                                                                                                    \
      widths[0] = c.pel_level;                                                                                       \
      for (int L = 1; L <= c.nlev && L <= 32; L++) widths[L] = (int)((c.nkpack[(L - 1) >> 4] >> (4 * ((L - 1) & 15))) & 15ull) + 1; \
    }                                                                                                                \
    if (most_children) most_children[0] = (int)((c.chain & 0xffffull) | ((c.neg & 0x7fffull) << 16));                                            \
    if (most_children) for (int L = 1; L <= c.nlev && L <= 32; L++) most_children[L] = (L <= 21 && c.cpack != ~0ull) ? (int)((c.cpack >> (3 * (L - 1))) & 7ull) : -1; \
    return SS_OK;                                                                                                    \
  }                                                                                                                  \
  int ss_obs_size(const ss_model *m, const ss_env_cfg *c) { return (m && c) ? ss::obs_size(m->hm.h, *c) : SS_ERR_INVALID; } \
  int ss_batch_create(const ss_model *m, const ss_env_cfg *c, const ss_state *s, ss_batch **o) { ss::HandleScope hs_(m ? &m->err : nullptr); return ss_api<BE>::batch_create(m, c, s, o); } \
  void ss_batch_destroy(ss_batch *b) { if (b) { BE::free_(b->d_kin); BE::free_(b->d_im); BE::free_(b->d_counter); BE::free_(b->d_prof); BE::free_(b->d_sched); delete b; } }          \
  int ss_set_order(ss_batch *b, const int32_t *order) { ss::HandleScope hs_(b ? &b->err : nullptr);                                                              \
    if (!b) return ss_api<BE>::fail(SS_ERR_INVALID, "null batch");                                                  \
    b->order = order; return SS_OK;                                                                                  \
  }                                                                                                                  \
  int ss_set_launch_geometry(ss_batch *b, int32_t envs_per_wg, int32_t max_workgroups) { ss::HandleScope hs_(b ? &b->err : nullptr);                              \
    if (!b || envs_per_wg < 0 || envs_per_wg > b->envs_per_wg || max_workgroups < 0)                                   \
      return ss_api<BE>::fail(SS_ERR_INVALID, "envs per workgroup must be in [0, ss_launch_info's value], max_workgroups >= 0"); \
    b->fixed_envs_per_wg = envs_per_wg; b->max_wgs = max_workgroups; return SS_OK;                                     \
  }                                                                                                                  \
  int ss_set_body_outputs(ss_batch *b, float *xpos, float *xmat) { ss::HandleScope hs_(b ? &b->err : nullptr);                                                    \
    if (!b || (!xpos) != (!xmat)) return ss_api<BE>::fail(SS_ERR_INVALID, "pass both buffers or neither");           \
    b->body_xpos = xpos; b->body_xmat = xmat; return SS_OK;                                                          \
  }                                                                                                                  \
  int ss_set_power_output(ss_batch *b, float *power) { ss::HandleScope hs_(b ? &b->err : nullptr);                                                               \
    if (!b) return ss_api<BE>::fail(SS_ERR_INVALID, "null batch");                                                  \
    b->power = power; return SS_OK;                                                                                  \
  }                                                                                                                  \
  int ss_debug_self_truncation(ss_batch *b, int32_t *counts) { ss::HandleScope hs_(b ? &b->err : nullptr);                                                       \
    if (!b) return ss_api<BE>::fail(SS_ERR_INVALID, "null batch");                                                  \
    b->self_trunc = counts; return SS_OK;                                                                            \
  }                                                                                                                  \
  int ss_debug_self_contacts(ss_batch *b, float *records) { ss::HandleScope hs_(b ? &b->err : nullptr);                                                           \
    if (!b) return ss_api<BE>::fail(SS_ERR_INVALID, "null batch");                                                  \
    b->dbg_self = records; return SS_OK;                                                                             \
  }                                                                                                                  \
  int ss_schedule_longest_first(ss_batch *b, void *stream) { ss::HandleScope hs_(b ? &b->err : nullptr);                                                        \
    if (!b) return ss_api<BE>::fail(SS_ERR_INVALID, "null batch");                                                  \
    if (!BE::set_device(b->m->device)) return ss_api<BE>::fail(SS_ERR_HIP, "cannot select device");                  \
    if (!b->d_sched) b->d_sched = (int32_t *)BE::alloc(2 * sizeof(int32_t) * (size_t)b->st.num_envs);                  \
    if (!b->d_sched) return ss_api<BE>::fail(SS_ERR_NOMEM, "device allocation failed");                              \
    const char *err = BE::order_by_key(b->st, b->m->hm.h.nv, b->d_sched + b->st.num_envs, b->d_sched, stream);         \
    if (err) return ss_api<BE>::fail(SS_ERR_HIP, err);                                                               \
    b->order = b->d_sched; return SS_OK;                                                                             \
  }                                                                                                                  \
  int ss_gae(const float *rew, const float *nd, const float *ndead, const float *val, const float *boot, int32_t T,   \
             int32_t N, float gamma, float tau, float *adv, float *ret, void *stream) {                             \
    if (!rew || !nd || !ndead || !val || !adv || !ret) return ss_api<BE>::fail(SS_ERR_INVALID, "null argument");     \
    if (T < 1 || N < 1) return ss_api<BE>::fail(SS_ERR_INVALID, "T and N must be positive");                         \
    const char *err = BE::gae(rew, nd, ndead, val, boot, T, N, gamma, tau, adv, ret, stream);                         \
    return err ? ss_api<BE>::fail(SS_ERR_HIP, err) : SS_OK;                                                          \
  }                                                                                                                  \
  int ss_debug_prof(ss_batch *b, unsigned long long *out, int n) { ss::HandleScope hs_(b ? &b->err : nullptr);                                                   \
    if (!b || !out || !b->d_prof) return ss_api<BE>::fail(SS_ERR_INVALID, "not a profiling build");                 \
    return BE::download(out, b->d_prof, (size_t)n * 8) ? SS_OK : SS_ERR_HIP;                                         \
  }                                                                   \
  int ss_reset(ss_batch *b, const uint8_t *mask, const float *fa, const float *tr, float *obs, void *st) { ss::HandleScope hs_(b ? &b->err : nullptr); return ss_api<BE>::reset(b, mask, fa, tr, obs, st); } \
  int ss_step(ss_batch *b, const float *a, const float *tr, float *obs, float *rew, uint8_t *te, uint8_t *tu, void *st) { ss::HandleScope hs_(b ? &b->err : nullptr); return ss_api<BE>::step(b, a, tr, obs, rew, te, tu, st); } \
  int ss_step_autoreset(ss_batch *b, const float *a, const float *tr, const float *tr2, float *obs, float *obs_next, float *rew, \
                        uint8_t *te, uint8_t *tu, void *st) { ss::HandleScope hs_(b ? &b->err : nullptr); return ss_api<BE>::step_autoreset(b, a, tr, tr2, obs, obs_next, rew, te, tu, st); } \
  int ss_set_fall_actions(ss_batch *b, const float *fa) {                                                             \
    if (!b) return ss_api<BE>::fail(SS_ERR_INVALID, "null batch");                                                  \
    b->fall_actions = fa; return SS_OK;                                                                              \
  }                                                                                                                  \
  int ss_get_state(ss_batch *b, int32_t field, void *buf, void *st) { ss::HandleScope hs_(b ? &b->err : nullptr); return ss_api<BE>::state_field(b, field, buf, false, st); } \
  int ss_set_state(ss_batch *b, int32_t field, const void *buf, void *st) { ss::HandleScope hs_(b ? &b->err : nullptr); return ss_api<BE>::state_field(b, field, const_cast<void *>(buf), true, st); } \
  int ss_imitation_bind(ss_batch *b, const ss_imitation_io *io) { ss::HandleScope hs_(b ? &b->err : nullptr); return ss_api<BE>::imitation_bind(b, io); } \
  int ss_imitation_step_fused(ss_batch *b, const float *a, const float *rand, void *st) { ss::HandleScope hs_(b ? &b->err : nullptr); return ss_api<BE>::imitation_step_fused(b, a, rand, st); } \
  int ss_substep(ss_batch *b, const float *a, int n, void *st) { ss::HandleScope hs_(b ? &b->err : nullptr); return ss_api<BE>::substep(b, a, n, st); }          \
  int ss_kinematics(ss_batch *b, float *xpos, float *xmat, void *st) { ss::HandleScope hs_(b ? &b->err : nullptr); return ss_api<BE>::kinematics(b, xpos, xmat, st); } \
  int ss_debug_forward(ss_batch *b, const float *tq, float *M, float *bias, float *qacc, void *st) { ss::HandleScope hs_(b ? &b->err : nullptr); return ss_api<BE>::debug_forward(b, tq, M, bias, qacc, st); } \
  int ss_launch_info(const ss_batch *b, int32_t *epw, int32_t *lds, int32_t *regs) { ss::HandleScope hs_(b ? &b->err : nullptr);                                 \
    if (!b) return ss_api<BE>::fail(SS_ERR_INVALID, "null batch");                                                   \
    if (epw) *epw = b->envs_per_wg;                                                                                  \
    if (lds) *lds = (int32_t)b->lds_bytes;                                                                           \
    if (regs) *regs = b->kernel_regs;                                                                                \
    return SS_OK;                                                                                                    \
  }                                                                                                                  \
  const char *ss_last_error(void) { return ss::last_error().c_str(); }                                               \
  const char *ss_model_last_error(const ss_model *m) { return m ? m->err.c_str() : ""; }                               \
  const char *ss_batch_last_error(const ss_batch *b) { return b ? b->err.c_str() : ""; }                               \
  int ss_model_create_from_mjcf(const char *xml, size_t len, const ss_mjcf_options *opt, int dev, ss_model **out) {   \
    if (!xml || !out) return ss_api<BE>::fail(SS_ERR_INVALID, "null argument");                                      \
    ss::mjcf::Compiled c; std::string e;                                                                             \
    if (!ss::mjcf::compile(xml, len ? len : std::strlen(xml), opt, c, e)) return ss_api<BE>::fail(SS_ERR_INVALID, "MJCF: " + e); \
    return ss_api<BE>::model_create(&c.desc, dev, out);                                                              \
  }                                                                                                                  \
  }
