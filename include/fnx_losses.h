/*
 * fnx_losses.h -- C ABI of the fused image loss (L1 + SSIM) for gfx950.
 *
 * Replaces, for the hot loop, the un-fused PyTorch chain of the reference
 *   FluidDynamics/utils/loss_utils.py: l1_loss :9-10, ssim/_ssim :33-64 (11x11 Gaussian window,
 *   sigma 1.5, zero padding 5, five depthwise conv2d, C1 = 0.01^2, C2 = 0.03^2, mean),
 * and, with grey = 1, the grey-mean + 3x replication that precedes it in the physical stage
 *   (entries_fluid_nexus/train_physical_particle.py:356-363).
 * The window is applied separably (row pass then column pass through LDS); values agree with the
 * reference's 2-D window to fp32 rounding (pinned by tests/golden/loss_utils.npz).
 *
 * forward : partials[b] = (sum |x - y|, sum ssim_map) of tile b; dmaps = the three per-pixel
 *           partial derivatives d map / d mu1, d map / d E[x^2], d map / d E[xy] for the backward.
 * backward: dL_dimg = g_l1 * sign(x - y) / n + g_ssim / n * (window (*) dmaps combined with x, y),
 *           g_l1 / g_ssim are DEVICE scalars (the upstream gradients of the two means).
 * n = Ce*H*W with Ce = 1 if grey else C.  All pointers are device pointers; work goes to `stream`.
 */
#ifndef FNX_LOSSES_H
#define FNX_LOSSES_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef void *fnx_stream_t;
int fnx_losses_abi_version(void);
const char *fnx_losses_last_error(void);
/* `grey`: 0 = the loss per channel; 1 = on the grey means of the two 3-channel images (train_physical_particle.py:356-360:
 * torch.mean over the channels, repeated three times -- the three copies give the same value, so one is evaluated);
 * 2 = as 1, but `gt` already holds the target's grey mean, ONE plane per image ([N,1,H,W], formed as ((r + g) + b) * (1/3)
 * in fp32 like the kernel does): the target is constant across a frame's iterations, a third of the loads goes. */
/* number of workgroup tiles (32x32 pixels per channel plane) = length of `partials` / 2 per image */
int fnx_l1_ssim_tiles(int C, int H, int W, int grey);
int fnx_l1_ssim_forward(const float *img, const float *gt, int C, int H, int W, int grey, float *partials,
                        float *dmaps /* [3, Ce, H, W] */, fnx_stream_t stream);
int fnx_l1_ssim_backward(const float *img, const float *gt, int C, int H, int W, int grey, const float *dmaps,
                         const float *g_l1, const float *g_ssim, float *dL_dimg /* [C, H, W] */, fnx_stream_t stream);
/* The same over a batch of N image pairs [N,C,H,W] in one launch (the views of a training batch):
 * partials [N, tiles, 2], dmaps [N, 3, Ce, H, W], g_l1 / g_ssim DEVICE arrays of N, dL_dimg [N,C,H,W]. */
int fnx_l1_ssim_forward_batch(const float *img, const float *gt, int N, int C, int H, int W, int grey,
                              float *partials, float *dmaps, fnx_stream_t stream);
int fnx_l1_ssim_backward_batch(const float *img, const float *gt, int N, int C, int H, int W, int grey,
                               const float *dmaps, const float *g_l1, const float *g_ssim, float *dL_dimg,
                               fnx_stream_t stream);
/* The image term of a training batch as one scalar (train_physical_particle.py:356-366, summed over the
 * views): loss = sum_n ( w_l1 * L1_n + w_dssim * (1 - SSIM_n) ) with w_l1 = (1 - lambda_dssim) * lambda_image,
 * w_dssim = lambda_dssim * lambda_image.  forward also returns per_image [N,2] = (L1_n, SSIM_n);
 * backward takes the upstream gradient of the scalar as a DEVICE scalar g_loss. */
int fnx_image_loss_forward(const float *img, const float *gt, int N, int C, int H, int W, int grey, float w_l1,
                           float w_dssim, float *partials, float *dmaps, float *per_image, float *loss,
                           fnx_stream_t stream);
int fnx_image_loss_backward(const float *img, const float *gt, int N, int C, int H, int W, int grey, float w_l1,
                            float w_dssim, const float *dmaps, const float *g_loss, float *dL_dimg,
                            fnx_stream_t stream);
/* fnx_image_loss_forward + fnx_image_loss_backward as two launches: the reduction to per_image / loss (consumed by
 * logging only) rides inside the backward launch.  g_loss: device scalar (upstream gradient, usually 1).  N <= 64. */
int fnx_image_loss_value_and_grad(const float *img, const float *gt, int N, int C, int H, int W, int grey, float w_l1,
                                  float w_dssim, float *partials, float *dmaps, float *per_image, float *loss,
                                  const float *g_loss, float *dL_dimg, fnx_stream_t stream);
/* Visual-particle stage (train_visual_particle.py:133-222): the leaves are the raw attributes of the n fluid Gaussians.
 * fnx_level2_activate writes their activations (gm_dynamics.py getters: colour repeated to 3 channels, sigmoid(opacity),
 * exp(scales), normalize(rotation)) into the first n rows of the arrays the rasteriser reads (pipe_dynamics.py:88-148). */
int fnx_level2_activate(const float *raw_color, const float *raw_opacity, const float *raw_scales, const float *raw_rotation,
                        int n, float *colors /* [.,3] */, float *opacity /* [.,1] */, float *scales /* [.,3] */,
                        float *rotations /* [.,4] */, fnx_stream_t stream);
/* d[k] = scale * ( d(rasteriser term)/d raw[k]  (from g[k], the gradient with respect to the activated rows)
 *                + reg_count * ( lambdas[k] * d mse(raw[k][:n_prev], prev[k]) / d raw[k]            (tvp:161-186)
 *                              + [k = scales] lambda_reg * d mean_i relu(max_i / min_i - reg_threshold) / d raw ) ),   (tvp:188-194)
 * k = 0 colour [n,1] (g[0]: [n,3]), 1 opacity [n,1], 2 scales [n,3], 3 rotation [n,4]; d[k] NULL skips an attribute that
 * is not fitted.  reg_count = the number of local views (every view's loss carries the view-independent terms once),
 * scale = 1 / batch (gm_dynamics.py:494-503).  The four arrays of pointers and lambdas are read on the host. */
int fnx_level2_backward(const float *const raw[4], const float *const prev[4], const float *const g[4], float *const d[4],
                        int n, int n_prev, const float lambdas[4], float lambda_reg, float reg_threshold, float reg_count,
                        float scale, fnx_stream_t stream);
#ifdef __cplusplus
}
#endif
#endif
