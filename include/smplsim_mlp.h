/* smplsim_mlp.h — C ABI of the policy-inference kernels of libsmplsim_hip.so (the sampler's caller side).
 *
 * The reference's sampler evaluates its Gaussian policy once per env step on the CPU worker that owns the env
 * (PolicyGaussian.select_action -> MLP.forward, smpl_sim/learning/policy_gaussian.py:14-41, mlp.py:36-60, with the
 * observation normalised by RunningNorm, running_norm.py:5-42).  With thousands of envs per GPU that forward pass is
 * GEMM-shaped (4096 x 289 -> 2048 -> 1536 -> 1024 -> 1024 -> 512 -> 512 -> 69: 59 GFLOP per env step) and sits in the
 * sampling loop next to ss_step, so it runs on the matrix cores: bf16 operands, fp32 accumulation
 * (v_mfma_f32_32x32x16_bf16), bias + activation fused into the GEMM's epilogue, activations kept in bf16 between layers.
 * The PPO update (backward pass) stays with the caller's autograd framework.
 *
 * Conventions as in smplsim_hip.h: device pointers, int status + ss_last_error(), work enqueued on the caller's stream.
 */
#ifndef SMPLSIM_MLP_H
#define SMPLSIM_MLP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { SS_ACT_NONE = 0, SS_ACT_SILU = 1, SS_ACT_TANH = 2, SS_ACT_RELU = 3 };   /* mlp.py:13-21 (the ones the reference's cfgs use) */

/* y = act(x W^T + b): x [M, K] bf16 row-major (K a multiple of 32, zero padded), W [N, K] bf16 row-major (torch.nn.Linear's
 * layout), b [N] f32 or NULL, y [M, ldy] bf16 (y_is_f32 = 0) or f32; columns >= N of y are not written. */
int ss_linear_bf16(const void *x, const void *w, const float *bias, void *y, int32_t M, int32_t N, int32_t K, int32_t ldy,
                   int32_t activation, int32_t y_is_f32, void *stream);

/* The product of the PPO update's forward and backward passes (round 6; the reference's update, agents/agent_ppo.py:20-83, runs them through
 * autograd): the same K-contiguous y = x W^T on the matrix cores, K a multiple of 64, with what a training step needs of it
 *   bf16 form (y_is_f32_accumulate = 0):  v = x W^T + bias;  v *= mul (if given: [M, ldy] bf16, the stored activation derivative of the layer
 *       below: dZ = (dZ' W) * act'(z));  y [M, ldy] = act(v);  yt [N, ldyt] = y^T (every product of the backward pass contracts over what is a row
 *       here: written transposed, dW = dZ^T h and dX = dZ W are this same kernel again);  dact [M, ldy] = act'(v).  Any of y / yt / dact may be
 *       NULL (mul and dact need y).
 *   accumulating fp32 form (y_is_f32_accumulate = 1):  y [M, ldy] fp32 += x W^T (+ bias), the contraction split over several workgroups whose
 *       partial sums meet in y by hardware atomics — for products with few outputs and a deep K (the weight gradients: K = the batch).
 *       The caller zeroes y. */
int ss_linear_bf16_train(const void *x, const void *w, const float *bias, const void *mul, void *y, void *yt, void *dact, int32_t M, int32_t N,
                         int32_t K, int32_t ldy, int32_t ldyt, int32_t activation, int32_t y_is_f32_accumulate, void *stream);

/* Observation -> first layer input: y = clamp(obs, clip_lo, clip_hi) (AgentPPO's clip_obs), then, when *norm_n > 0,
 * clamp((y - mean) / (std + 1e-8), -norm_clip, norm_clip) (RunningNorm.forward in eval mode), rounded to bf16 into
 * out [M, kpad] with the columns >= dim zeroed.  norm_* may be NULL (no normalisation). */
int ss_obs_to_bf16(const float *obs, int32_t M, int32_t dim, int32_t obs_stride, const float *norm_mean, const float *norm_std,
                   const int64_t *norm_n, float clip_lo, float clip_hi, float norm_clip, void *out, int32_t kpad, void *stream);

/* dZ of the layer below in one launch (the `grad_input = dZ @ W` of torch.nn.Linear's backward followed by the activation's backward, agents/agent_ppo.py:20-83):
 *   y = (x W^T) * mul   [M, ldy] bf16        x = dZ [M, K], w = W^T [N, K] (both K-contiguous), mul = act'(z_below) [M, ldy] bf16
 *   colsum[j] += sum_m of the fp32 result    [N] fp32: the bias gradient of the layer below (the caller zeroes it; partial sums by fp32 atomics)
 * Served by the 256 x 256 kernel only: M >= 2048, N >= 256, K a multiple of 128 (SS_ERR_INVALID otherwise: use ss_linear_bf16_train and sum the columns yourself). */
int ss_linear_bf16_dx(const void *x, const void *w, const void *mul, void *y, float *colsum, int32_t M, int32_t N, int32_t K, int32_t ldy, void *stream);

/* Weight gradient of a linear layer from the two tensors as autograd holds them (replaces `grad_W = dZ^T @ h` of torch.nn.Linear's backward inside the
 * reference's update_policy / update_value, agents/agent_ppo.py:20-83):
 *   dw[i, j] += sum_m dz[m, i] * h[m, j]      dz [Mb, ldz] bf16 (columns 0 .. n_out - 1 used), h [Mb, ldh] bf16 (columns 0 .. n_in - 1), dw [n_out, ldw] fp32
 * Both operands are read untransposed (contraction over their ROWS); the caller zeroes dw; partial sums of a K split meet by fp32 atomics.
 * Mb a multiple of 128 (pad rows zero), n_out, n_in, ldz, ldh multiples of 8, dz and h 16-byte aligned, ldw >= n_in. */
int ss_wgrad_bf16(const void *dz, const void *h, float *dw, int32_t Mb, int32_t n_out, int32_t n_in, int32_t ldz, int32_t ldh, int32_t ldw, void *stream);

/* The Gaussian head of the sampler in one launch (PolicyGaussian.select_action, policy_gaussian.py:25-41 -> DiagGaussian.sample;
 * Agent.preprocess_actions with clip_actions, agents/agent.py:153-161; normal_log_density of get_log_prob): per row
 *   action = mean + exp(log_std) * noise          [M, dim], row stride lda (the rollout's action row: the UNCLIPPED draw is what is stored)
 *   action_env = clamp(action, clip_lo, clip_hi)  [M, dim], row stride lde, or NULL (what the env is stepped with)
 *   logp = sum_j -noise^2 / 2 - log sqrt(2 pi) - log_std_j   [M] or NULL (the behaviour policy's log-density of the draw)
 * mean, noise [M, dim] dense f32; log_std [dim].  The caller draws the noise (its generator, its stream order). */
int ss_gaussian_sample(const float *mean, const float *noise, const float *log_std, int32_t M, int32_t dim, float *action, int32_t lda,
                       float *action_env, int32_t lde, float clip_lo, float clip_hi, float *logp, void *stream);

#ifdef __cplusplus
}
#endif
#endif
