/*
 * fnx_physics.h -- C ABI of the fused particle-physics kernels (gfx950).
 *
 * The reference evaluates its physics-informed losses as un-fused PyTorch op chains on an edge
 * list from torch_cluster.radius / radius_graph (FluidDynamics/gaussian_splatting/gm_dynamics.py):
 *   poly6                                   :188-191
 *   get_gas_constraints_from_exyz_nn        :1269-1294  (density ratio, "gas constraint")
 *   get_gas_constraints_from_vel_nn_guess   :1296-1320  (same on the one-tick-advected positions)
 *   get_visual_xyz_from_nn                  :1453-1498  (hidden -> visual velocity interpolation)
 * Each entry point below is the forward or the analytic backward of one of those, with the
 * neighbour search (uniform hash grid, cell = H) fused in; no edge list is materialised.
 *
 * Neighbour rule: j is a neighbour of i iff |x_i - x_j|^2 < H^2 in fp32 (this is poly6's own
 * mask, :190; self included, as radius_graph(loop=True)).  The default entry points take ALL such
 * pairs: results equal the reference's whenever no query has more than KNN_K neighbours.  The
 * `_kcap` entry points (round 4) reproduce torch_cluster's max_num_neighbors truncation as its CUDA
 * kernel performs it -- a query keeps the K smallest INDICES among its neighbours (fnx_knn_cut);
 * torch_cluster (1.6.3, not vendored by the reference) is absent here, so that rule is restated from
 * its published kernel and parity of the capped mode is pinned only against this repository's
 * brute-force restatement (oracle/physics_oracle.py, `knn_k`), not against the library itself.
 *
 * All pointers are device pointers (fp32 / opaque bytes); work is enqueued on `stream`.
 * Returns 0 or an FNX_ERR_* code from fnx_raster.h; fnx_physics_last_error() gives the text.
 */
#ifndef FNX_PHYSICS_H
#define FNX_PHYSICS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *fnx_stream_t;

/* 2: fnx_adam_step gained `arrived` (before `stream`); compare with fnx_physics_abi_version() before anything else. */
#define FNX_PHYSICS_ABI_VERSION 3
int fnx_physics_abi_version(void);
const char *fnx_physics_last_error(void);

/* Uniform hash grid over N points, cell edge = `cell`.  The blob is opaque; it also holds per-call
 * scratch (per-slot payloads) that the visual_interp entry points rewrite, so one grid must not be
 * used from two streams at once. */
size_t fnx_grid_bytes(int N);
int fnx_grid_build(const float *xyz, int N, float cell, char *grid, fnx_stream_t stream);

/* p_ratio[i] = ( sum_{j: r2_ij < H^2} term1 (H^2 - r2_ij)^3 ) / imass[i] / p0, term1 = 315/(64 pi H^9).
 * `grid` must have been built over `xyz` with cell = H. */
int fnx_density_forward(const float *xyz, int N, const float *imass, float H, float p0, const char *grid,
                        float *p_ratio, fnx_stream_t stream);
/* dL_dxyz[i] = sum_j (g_i/(imass_i p0) + g_j/(imass_j p0)) * dW/dr2(r2_ij) * 2 (x_i - x_j), g = dL_dp_ratio. */
int fnx_density_backward(const float *xyz, int N, const float *imass, float H, float p0, const char *grid,
                         const float *dL_dp_ratio, float *dL_dxyz, fnx_stream_t stream);

/* ---- max_num_neighbors mode (gm_dynamics.py:1276, 1302, 1463: radius / radius_graph(..., max_num_neighbors = KNN_K)).
 * cut[q] = the K-th smallest index among the points of `points_grid` (built over N_points points, cell = H) within H
 * of query q, or 0xFFFFFFFF when q has at most K of them: "q keeps point j" <=> j <= cut[q]. */
int fnx_knn_cut(const float *queries, int Nq, int N_points, float H, int K, const char *points_grid, uint32_t *cut,
                fnx_stream_t stream);
/* fnx_density_forward / _backward on the capped edge set (cut = fnx_knn_cut(xyz, N, N, H, K, grid)): the edge
 * "query q keeps neighbour i" adds poly6 at i (radius_graph's source_to_target flow + index_add_ on row, :1277-1288),
 * p_i = sum_{q: i <= cut[q]} poly6(r2_iq); backward with the edge set frozen (it is piecewise constant). */
int fnx_density_forward_kcap(const float *xyz, int N, const float *imass, float H, float p0, const char *grid,
                             const uint32_t *cut, float *p_ratio, fnx_stream_t stream);
int fnx_density_backward_kcap(const float *xyz, int N, const float *imass, float H, float p0, const char *grid,
                              const uint32_t *cut, const float *dL_dp_ratio, float *dL_dxyz, fnx_stream_t stream);
/* fnx_visual_interp_forward / _backward where visual particle v keeps the hidden particles j <= cutv[v]
 * (cutv = fnx_knn_cut(visual, V, N, H, K, hidden_grid)). */
int fnx_visual_interp_forward_kcap(const float *visual, int V, const float *hidden, const float *hidden_prev, int N,
                                   float H, float secs, float eps, const char *hidden_grid, const uint32_t *cutv,
                                   float *out, float *sum_w, float *wvel, fnx_stream_t stream);
int fnx_visual_interp_backward_kcap(const float *visual, int V, const float *hidden, const float *hidden_prev, int N,
                                    float H, float secs, float eps, const char *visual_grid, const uint32_t *cutv,
                                    const float *sum_w, const float *wvel, const float *dL_dout, float *dL_dhidden,
                                    fnx_stream_t stream);

/* out[v] = visual[v] + secs * sum_j w_vj u_j / max(sum_j w_vj, eps),  w_vj = poly6(|visual_v - hidden_j|^2),
 * u_j = (hidden_j - hidden_prev_j) / secs.  Also returns sum_w [V] (unclamped) and wvel [V,3] = sum_j w_vj u_j
 * for the backward.  `hidden_grid` is built over `hidden` with cell = H. */
int fnx_visual_interp_forward(const float *visual, int V, const float *hidden, const float *hidden_prev, int N,
                              float H, float secs, float eps, const char *hidden_grid, float *out, float *sum_w,
                              float *wvel, fnx_stream_t stream);
/* Work items of a built grid for the cell-by-cell kernels: `items` (fnx_grid_cell_items_bytes(N) bytes, device)
 * receives a count word and, per non-empty bucket, ceil(count/64) (first slot, slots) pairs in arbitrary order.
 * Built once per grid. */
size_t fnx_grid_cell_items_bytes(int N);
int fnx_grid_cell_items(const char *grid, int N, char *items, fnx_stream_t stream);
/* fnx_visual_interp_forward with the visual particles walked cell by cell: `visual_grid` is a grid built over
 * `visual` with cell = H and `visual_items` its work items (both static within a frame, so built once); the hidden
 * neighbourhood of a cell is staged in LDS once per cell instead of being re-read per particle.  Same results up
 * to fp32 summation order. */
int fnx_visual_interp_forward_cells(const float *visual, int V, const float *hidden, const float *hidden_prev, int N,
                                    float H, float secs, float eps, const char *hidden_grid, const char *visual_grid,
                                    const char *visual_items, float *out, float *sum_w, float *wvel,
                                    fnx_stream_t stream);
/* ... and also writes out / divisor to `out_div` [V,3] (NULL: not written): the advected positions in render units
 * (gm_dynamics.py:1498 + pipe_dynamics.py:40, `/ scale_factor`), evaluated as torch evaluates tensor / scalar
 * (x * fp32(1 / divisor)), without a second pass. */
int fnx_visual_interp_forward_cells_div(const float *visual, int V, const float *hidden, const float *hidden_prev, int N,
                                        float H, float secs, float eps, const char *hidden_grid, const char *visual_grid,
                                        const char *visual_items, float *out, float *sum_w, float *wvel, float *out_div,
                                        float divisor, fnx_stream_t stream);
/* dL_dhidden[j] = sum_v [ w_vj/S_v * g_v + dL/dw_vj * dW/dr2 * 2 (hidden_j - visual_v) ], S_v = max(sum_w, eps),
 * dL/dw_vj = secs * (g_v . u_j)/S_v - [sum_w_v > eps] * secs * (g_v . wvel_v)/S_v^2.
 * `visual_grid` is built over `visual` with cell = H. */
int fnx_visual_interp_backward(const float *visual, int V, const float *hidden, const float *hidden_prev, int N,
                               float H, float secs, float eps, const char *visual_grid, const float *sum_w,
                               const float *wvel, const float *dL_dout, float *dL_dhidden, fnx_stream_t stream);
/* The same with the hidden particles walked cell by cell: `hidden_grid` is the grid built over `hidden` (cell = H)
 * and `hidden_items` its work items (fnx_grid_cell_items, rebuilt whenever the grid is); the visual neighbourhood of
 * a cell is read once per cell instead of once per hidden particle.  Same results up to fp32 summation order.
 * `hidden` itself is only checked for NULL: positions come from the grid records (bit-identical copies). */
int fnx_visual_interp_backward_cells(const float *visual, int V, const float *hidden, const float *hidden_prev, int N,
                                     float H, float secs, float eps, const char *visual_grid, const char *hidden_grid,
                                     const char *hidden_items, const float *sum_w, const float *wvel,
                                     const float *dL_dout, float *dL_dhidden, fnx_stream_t stream);
/* ... for the upstream gradient dL_dout + scale2 * dL_dout2 (dL_dout2 NULL: dL_dout alone): a second term that
 * arrives from elsewhere (the distance loss on the rendered positions, train_physical_particle.py:365-366) is added
 * while the per-particle payload is formed instead of by a pass of its own. */
int fnx_visual_interp_backward_cells_sum(const float *visual, int V, const float *hidden, const float *hidden_prev, int N,
                                         float H, float secs, float eps, const char *visual_grid,
                                         const char *hidden_grid, const char *hidden_items, const float *sum_w,
                                         const float *wvel, const float *dL_dout, const float *dL_dout2, float scale2,
                                         float *dL_dhidden, fnx_stream_t stream);

/* The three physics terms of the physical-particle stage and their gradient in one call
 * (entries_fluid_nexus/train_physical_particle.py:368-404 as one launch sequence of ~12 kernels):
 *   x  = x_nn * scale_factor                                   (hidden particles, scaled units)
 *   x' = x + secs * ((x - x_prev) / secs + b' secs + secs force), b' = buoyancy (1 - x_nn.y / buoyancy_max_y)
 *        when buoyancy_max_y > 0, else buoyancy                  (gm_dynamics.py:1014-1030)
 *   loss = lam_e mean((x - x_est)^2) + lam_g mean((p_ratio(x) - 1)^2) + lam_n mean((p_ratio(x') - 1)^2)
 * terms[3] receive the three unweighted SUMS of squares, *loss the weighted loss, grad [N,3] = d loss / d x_nn.
 * est_grid: hash grid over x with cell = H -- built here when build_est_grid != 0, otherwise it must already
 * be up to date (callers that just ran fnx_visual_interp_forward on the same x own one); guess_grid: blob of
 * fnx_grid_bytes(N), always rebuilt (over x'); scratch: 15 N + 64 floats.  A term with lambda <= 0 is skipped. */
int fnx_physical_stage(const float *x_nn, int N, float scale_factor, const float *x_est, const float *x_prev,
                       const float *imass, const float *buoyancy, const float *force, float buoyancy_max_y, float H,
                       float p0, float secs, float lam_e, float lam_g, float lam_n, char *est_grid, int build_est_grid,
                       char *guess_grid, float *scratch, float *terms, float *loss, float *grad, fnx_stream_t stream);

/*
 * Position-based-fluids predictor / solver of the per-frame step.  Each call is the fused,
 * GPU-resident form of one method of the reference's GaussianModel (gm_dynamics.py); all arrays are [N,3] / [N]
 * fp32 device arrays updated in place as the reference updates its attributes.
 *   fnx_pbf_predict          guess_hidden_particles :978-1012 (no wind): buoyancy <- gravity * alpha [* decay],
 *                            velocity += (buoyancy (1 - y / scale_max_y)) secs + secs force, force <- 0,
 *                            estimate <- xyz + secs velocity, counts <- 0.  gravity: HOST float[3];
 *                            scale_max_y = buoyancy_max_y * scale_factor, <= 0 disables the height term;
 *                            alpha / secs are the caller's (stable: -1.0 / 0.01).
 *   fnx_pbf_neighbor_counts  remove_invalid_particles :1040-1045: number of OTHER particles within H (builds `grid`).
 *   fnx_pbf_project          project_gas_constraints :1075-1160: rebuilds `grid` over estimate_xyz, node pass
 *                            (density ratio, lambda, neighbour count, force += velocity (1 - p_ratio)(-k)), then the
 *                            Jacobi position pass (lambda_i + lambda_j + lamb_corr) spiky_grad / p0 /
 *                            (neighbours + counts).  scratch: 5 N floats.
 *   fnx_pbf_confirm          confirm_guess_hidden_particles(_wo_velocity) :1323-1350.
 *   fnx_visual_advect        update_visual_particles :1353-1398: visual += secs sum_j w_vj velocity_j /
 *                            max(sum_j w_vj, eps), in place (builds `hidden_grid` over `hidden`); scratch: 4 V floats.
 * Neighbour rule as above (r^2 < H^2, no max_num_neighbors truncation).
 */
int fnx_pbf_predict(const float *xyz, float *velocity, float *buoyancy, float *force, float *estimate_xyz, float *counts,
                    int N, const float *gravity, float alpha, float secs, float scale_max_y, float decay_rate,
                    fnx_stream_t stream);
int fnx_pbf_neighbor_counts(const float *xyz, int N, float H, char *grid, int *counts, fnx_stream_t stream);
int fnx_pbf_project(float *estimate_xyz, const float *velocity, float *force, const float *imass, const float *counts,
                    int N, float H, float p0, float k, float relaxation, float K_P, float E_P, float DQ_P, float eps,
                    char *grid, float *scratch, fnx_stream_t stream);
int fnx_pbf_confirm(float *xyz, const float *estimate_xyz, float *velocity, int N, float secs, float eps,
                    fnx_stream_t stream);
int fnx_visual_advect(float *visual, int V, const float *hidden, const float *velocity, int N, float H, float secs,
                      float eps, char *hidden_grid, float *scratch, fnx_stream_t stream);

/* simple-knn's distCUDA2 (submodules/simple-knn/simple_knn.cu:134-203): mean_dist2[i] = mean of the
 * squared distances from point i to its 3 nearest OTHER points (exact; FLT_MAX terms when N < 4, as the reference).
 * `cell` is the edge of the search grid built here in `grid` (fnx_grid_bytes(N)): any positive value is correct,
 * ~(volume / N)^(1/3) is fast. */
int fnx_knn_mean_dist2(const float *xyz, int N, float cell, char *grid, float *mean_dist2, fnx_stream_t stream);

/* Pairwise distance loss of FluidDynamics/utils/loss_utils.py:98-121 (distance_loss(positions, threshold), called
 * per view at entries_fluid_nexus/train_physical_particle.py:141-144,365-366), radius-limited instead of a dense
 * N x N torch.cdist:  loss = sum over ordered pairs i != j with |x_i - x_j| < threshold of (threshold - |x_i - x_j|)^2.
 * `grid`: fnx_grid_bytes(N) bytes of scratch (a hash grid with cell = 2 threshold is built in it);
 * partials [fnx_distance_loss_partials(N)]: per-workgroup sums, loss = their sum (deterministic);
 * grad [N,3] (may be NULL): d loss / d xyz, zero for coincident points like torch.cdist's backward. */
int fnx_distance_loss_partials(int N);
int fnx_distance_loss(const float *xyz, int N, float threshold, char *grid, float *partials, float *grad,
                      fnx_stream_t stream);
/* The same loss on per-bucket linked lists instead of a counted / scanned / filled grid: two launches (build, search) and
 * one thread per point -- an eighth of the waves, which is what the branch costs the rasteriser's blend forward it runs
 * beside (csrc/physics.hip).  `table`: fnx_distance_table_bytes(N) bytes owned by the caller, PERSISTENT across calls
 * with the same N and ZERO-FILLED ONCE (entries carry the number of the call that wrote them, so the table is never
 * cleared); one call at a time per table.  loss_out[0] = the loss (partial sums added in workgroup order by the
 * workgroup that arrives last); grad as above (may be NULL). */
/* Scheduling aid: one sleeping wave on `stream` for about `microseconds` (a side branch of a captured graph can only fork
 * at a kernel boundary of the main chain; this moves its start INTO the kernel that follows the fork point). */
int fnx_stream_delay(float microseconds, fnx_stream_t stream);
/* K-cap watch.  The reference's three neighbour searches keep at most KNN_K neighbours per query (torch_cluster
 * radius_graph(loop = True) / radius with max_num_neighbors = KNN_K, gm_dynamics.py:1276,1302,1463); the fused stage and the
 * cell-by-cell interpolation take every pair within H -- identical while no list is longer than K.  With a watch armed
 * (per host thread, until fnx_knn_watch(NULL, 0)) fnx_physical_stage and fnx_visual_interp_forward_cells* count every
 * query's neighbours on their way and OR into flags[0]: 1 = a hidden-particle list at the optimised positions exceeds K,
 * 2 = one at the guessed positions, 4 = a visual particle's list of hidden particles.  No host synchronisation: the
 * caller owns the word (zero it once) and reads it when it likes; launches recorded into a hipGraph keep the pointer. */
int fnx_knn_watch(uint32_t *flags, int K);
size_t fnx_distance_table_bytes(int N);
int fnx_distance_loss_lists(const float *xyz, int N, float threshold, char *table, float *grad, float *loss_out,
                            fnx_stream_t stream);
/* The same loss with VERLET pair lists (round 5): consecutive calls of the optimisation loop see positions that differ by
 * ~lr, so the pairs closer than `threshold` are kept from call to call.  `state`: fnx_distance_verlet_bytes(N, K) bytes
 * owned by the caller, PERSISTENT and ZERO-FILLED ONCE, one call at a time: per point the <= K indices of the points within
 * threshold + skin of it when the lists were built, and its position then.  Every call checks on the device that no point
 * has moved further than skin / 2 since (then no pair outside the lists can be closer than `threshold`); otherwise -- or
 * when N, threshold, skin or K differ from the build's, or a list overflowed K -- the same launch sequence rebuilds the
 * lists through the bucket table (`table`: as above, fnx_distance_table_bytes(N), cell = 2 threshold, 27 cells per point)
 * and evaluates the loss on its way: no host decision, graph-capturable, always the exact pair set.
 * 0 < skin <= threshold, 1 <= K <= 64.  The first 16 words of `state` (after 256-byte alignment) are counters a caller
 * may read: valid, -, calls, rebuilds, points whose list overflowed at the last rebuild.  Same loss / grad contract. */
size_t fnx_distance_verlet_bytes(int N, int K);
int fnx_distance_loss_verlet(const float *xyz, int N, float threshold, float skin, char *table, char *state, int K,
                             float *grad, float *loss_out, fnx_stream_t stream);

/* Gradient mean + optimiser step of the particle positions in one launch (gm_dynamics.py:461-472 followed by
 * torch.optim.Adam.step with amsgrad = False, weight_decay = 0, maximize = False):
 *   g = ((g0 s0 + g1 s1) + g2 s2) * inv_batch            (NULL terms are skipped; n = number of floats)
 *   exp_avg += (1 - beta1)(g - exp_avg); exp_avg_sq = beta2 exp_avg_sq + (1 - beta2) g^2; t = *step + 1
 *   x -= lr / (1 - beta1^t) * exp_avg / (sqrt(exp_avg_sq) / sqrt(1 - beta2^t) + eps);  *step = t
 * exp_avg / exp_avg_sq / step are the optimiser's own state tensors (step: DEVICE fp32 scalar), so the state
 * stays loadable by torch.  The betas are doubles so that 1 - beta is rounded from the double difference like
 * torch's (1 - 0.999f differs from float(0.001) by 1.3e-5 relative).  grad_out (optional) receives g; scaled_out
 * (optional) receives the updated x * scale (the positions in simulation units, gm_dynamics.py:1256).
 * `arrived`: one zero-initialised device word owned by this optimiser (the last workgroup to arrive advances *step
 * and resets the word); steps of different optimisers may overlap, two steps of the same one may not. */
int fnx_adam_step(float *x, int n, const float *g0, float s0, const float *g1, float s1, const float *g2, float s2,
                  float inv_batch, float *exp_avg, float *exp_avg_sq, float *step, float lr, double beta1, double beta2,
                  float eps, float *grad_out, float *scaled_out, float scale, unsigned int *arrived,
                  fnx_stream_t stream);

/* fnx_adam_step on N particles (x, exp_avg, exp_avg_sq, g*, grad_out, scaled_out: [N,3]) FOLLOWED BY the build of the
 * hash grid (cell edge `cell`) over the updated positions in simulation units, scaled_out = x * scale -- what the next
 * iteration's neighbour searches start with (gm_dynamics.py:1256, 1269-1294, 1453-1498) -- in two launches instead of
 * six: [step + bucket count + scan by the last workgroup to arrive] and [slot fill + per-slot velocity + clearing of the
 * counts].  Same arithmetic per element as fnx_adam_step; the grid equals fnx_grid_build(scaled_out, N, cell, grid) up to
 * the order of the records inside a bucket (both fill by atomic cursor).  `prev` [N,3] (optional): the per-slot velocity
 * (scaled_out - prev) / secs is left in the grid's payload, where fnx_visual_interp_forward_cells_vel(...,
 * velocity_ready = 1, ...) finds it.
 * CONTRACT: the bucket counts of `grid` (a blob of fnx_grid_bytes(N)) are zero on entry -- a zero-filled blob the first
 * time; the call leaves them zero.  fnx_grid_build on the same blob breaks that (zero the blob again before reuse). */
int fnx_adam_step_grid(float *x, int N, const float *g0, float s0, const float *g1, float s1, const float *g2, float s2,
                       float inv_batch, float *exp_avg, float *exp_avg_sq, float *step, float lr, double beta1,
                       double beta2, float eps, float *grad_out, float *scaled_out, float scale, unsigned int *arrived,
                       float cell, char *grid, const float *prev, float secs, fnx_stream_t stream);
/* fnx_visual_interp_forward_cells_div; velocity_ready != 0: the hidden grid's per-slot velocities were left by
 * fnx_adam_step_grid for the same hidden_prev / secs (the slot-velocity launch is skipped). */
int fnx_visual_interp_forward_cells_vel(const float *visual, int V, const float *hidden, const float *hidden_prev, int N,
                                        float H, float secs, float eps, const char *hidden_grid, const char *visual_grid,
                                        const char *visual_items, float *out, float *sum_w, float *wvel, float *out_div,
                                        float divisor, int velocity_ready, fnx_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FNX_PHYSICS_H */
