/*
 * fnx_raster.h -- C ABI of the MI355X-native (gfx950) differentiable 3D-Gaussian rasteriser.
 *
 * Drop-in boundary for the torch-free C++ API the reference binds through pybind11:
 *   CudaRasterizer::Rasterizer::{forward, backward, markVisible}
 *   FluidDynamics/submodules/gaussian_rasterization_ch3/cuda_rasterizer/rasterizer.h:18-84
 *   (identical in gaussian_rasterization_ch1; the two differ only in NUM_CHANNELS, config.h:15,
 *   which is the `channels` argument here).
 *
 * Conventions (reference behaviour in brackets):
 *  - every pointer is a DEVICE pointer to contiguous fp32/int32 data unless stated otherwise;
 *    NULL means "not provided" [0-element tensors surface as nullptr, rasterize_points.cu:95-101].
 *  - all work is enqueued on `stream` (a hipStream_t) [reference: legacy default stream].
 *  - outputs and scratch are caller-allocated.  The three scratch blobs (geometry, binning,
 *    image) are opaque, produced by forward and consumed by backward [__init__.py:77-79];
 *    their sizes come from fnx_*_bytes() [reference: std::function<char*(size_t)> resize
 *    callbacks, rasterizer.h:31-33; fnx_rasterize_forward keeps that callback form].
 *  - gradient outputs must be zero-filled by the caller [torch::zeros, rasterize_points.cu:150-158].
 *  - every entry point returns FNX_OK or an error code; fnx_last_error() gives the text
 *    [reference: AT_ERROR / std::runtime_error, no CUDA error checks].
 *  - P == 0 is a no-op that leaves the (zero-filled) outputs untouched [rasterize_points.cu:81,160].
 */
#ifndef FNX_RASTER_H
#define FNX_RASTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FNX_OK 0
#define FNX_ERR_INVALID_ARG 1
#define FNX_ERR_NON_RGB_NEEDS_COLORS 2 /* rasterizer_impl.cu:226-228 */
#define FNX_ERR_HIP 3
#define FNX_ERR_CAPACITY 4 /* binning buffer smaller than num_rendered */
#define FNX_ERR_UNSUPPORTED 5 /* e.g. more than 16384 tiles (image larger than 2048x2048) */
#define FNX_ERR_SORT_SPAN 6 /* sort_mode FNX_SORT_NARROW and a view's depth keys span 2^27 ulps or more: results invalid */

typedef void *fnx_stream_t; /* hipStream_t */

/* Resizable-buffer callback: must return a device pointer to at least `bytes` bytes
 * [std::function<char*(size_t)>, rasterizer.h:31-33 / resizeFunctional, rasterize_points.cu:27-33]. */
typedef char *(*fnx_alloc_fn)(size_t bytes, void *user);

/* Version of this interface.  Bumped whenever a scratch-blob layout, a layout struct or an argument list changes
 * (2: binning blob carries the forward -> backward hand-over, fnx_*_layout_t grew, fnx_set_blend_math added; 3: the image
 * blob's n_contrib array doubled -- its second half is the backward's per-pixel walking limit, fnx_request_gradient_limit;
 * 4: per-call options fnx_raster_opts_t and the *_opts entry points, the process-wide setters and one-shot requests are
 * deprecated shims; culled splats keep their depth in the sort keys; image header grew to 16 words with the walked-entry
 * counters; 5: fnx_raster_opts_t.dual, fnx_binning_bytes_dual; 6: fnx_raster_opts_t.segment_scratch; 7: the last three words
 * of a splat's 64-byte blend record (geometry blob, static blob) carry its world position -- the positions-only backward
 * reads them -- and fnx_set_backward_form / fnx_get_backward_form); a caller compares fnx_abi_version() with the
 * FNX_ABI_VERSION it was compiled against before anything else. */
#define FNX_ABI_VERSION 7
int fnx_abi_version(void);
const char *fnx_last_error(void);


/*
 * Per-call options (ABI 4).  The reference's interface is a set of stateless static functions (rasterizer.h:18-84); the
 * tuning knobs this library adds are therefore ARGUMENTS of a call, not process state: two rasteriser instances with
 * different options may be driven from different streams or host threads of one process.  Pass a pointer to the *_opts
 * entry points; NULL means "all defaults" and is what the entry points without the suffix pass.
 *
 * Integer fields take FNX_OPT_DEFAULT to defer to the deprecated process-wide setter of the same name (kept for one
 * round for ABI-3 callers: fnx_set_blend_math, fnx_set_lean_geometry, fnx_set_sort_narrow, fnx_set_deep_kernel,
 * fnx_set_deep_threshold); a zero-initialised struct with `size` set selects the library defaults (exact arithmetic,
 * full geometry state, four sort passes, no deep kernel, no gradient limit, no extras).
 */
#define FNX_OPT_DEFAULT (-2147483647 - 1)
#define FNX_SORT_FULL 0     /* 9-bit LSD radix passes; the fourth runs only when the key span needs it            */
#define FNX_SORT_NARROW 1   /* the fourth pass is not launched; a view that needed it reports FNX_ERR_SORT_SPAN */
#define FNX_SORT_COHERENT 2 /* one launch: repair of the previous call's order held in `sort_state` (see below) */
/*
 * Dual mode of the static-split view batch (ABI 5): a SECOND, single-channel image of the per-call splats alone, blended --
 * and differentiated -- in the same pass over the same tile lists as the 3-channel image of all splats.  It is what a scene
 * that runs the ch1 rasteriser over the fluid and the ch3 rasteriser over fluid + background per view (BASELINE configs[4])
 * would otherwise pay a second preprocess, depth sort, emission and pair of blend launches for: the per-call splats, their
 * depth order and every alpha are the same in both renders.  The second image's value per splat is CHANNEL 0 of the
 * splat's colour; its pixels equal a 1-channel render of the per-call splats (bit for bit in the exact arithmetic).
 * Stage 2 (channels = 3, static blobs given, grad_splat_limit = P_dyn or unset) writes out_color1 / out_depth1 and the
 * second image's per-pixel state into image_buffers1; the backward (geometry_only = 3, grad_splat_limit = P_dyn) reads
 * dL_dpix1 and adds both images' gradients into dL_dmean3D.  The binning blobs must be fnx_binning_bytes_dual() large.
 */
typedef struct fnx_raster_dual {
    char *image_buffers1;     /* V * fnx_image_bytes(W, H) bytes, laid out like image_buffers: the second image's state  */
    const float *background1; /* [1]                                                                                      */
    float *out_color1;        /* stage 2: [V,1,H,W]                                                                       */
    float *out_depth1;        /* stage 2: [V,1,H,W]                                                                       */
    const float *dL_dpix1;    /* backward: [V,1,H,W]                                                                      */
} fnx_raster_dual_t;
size_t fnx_binning_bytes_dual(int64_t capacity, int64_t R_static_capacity);

typedef struct fnx_raster_opts {
    uint32_t size;            /* sizeof(fnx_raster_opts_t) of the caller (checked)                                  */
    int32_t blend_math;       /* 0 exact / 1 fast (see fnx_set_blend_math); forward and backward must agree          */
    int32_t lean_geometry;    /* see fnx_set_lean_geometry; forward and backward must agree                          */
    int32_t sort_mode;        /* FNX_SORT_*; stage 1                                                                 */
    int32_t deep_kernel;      /* see fnx_set_deep_kernel; stage 2                                                    */
    int32_t grad_splat_limit; /* stage 2: the backward will differentiate ids < limit only (< 0: all), see
                                 fnx_request_gradient_limit                                                          */
    uint32_t deep_threshold;  /* 0: default (1024), see fnx_set_deep_threshold; stage 1                              */
    uint32_t reserved0;
    float *zero3;             /* stage 1: [3 (P_dyn + P_static)] floats zero-filled on the way (fnx_request_zero3)    */
    char *sort_state;         /* stage 1: V * fnx_sort_state_bytes(P_dyn) bytes owned by the caller and kept across the
                                 calls of one (camera batch, splat count); ZERO-FILLED ONCE before its first use.  With
                                 FNX_SORT_COHERENT the depth order of the previous call is repaired instead of sorting
                                 from scratch (csrc/raster_binning.hip): exact by construction -- the result is verified
                                 as a strictly increasing (depth bits, id) sequence on the device and a view that fails
                                 (new frame, large move, unseeded state) is sorted from scratch inside the same launch.
                                 With the other modes the radix sort leaves the state seeded.  NULL: no state.       */
    const fnx_raster_dual_t *dual; /* stage 2 / backward of a static-split view batch: see fnx_raster_dual_t; NULL: off */
    char *segment_scratch;    /* stage 1 + stage 2 (ABI 6), fast arithmetic, no dual image: V * fnx_segment_scratch_bytes(W, H)
                                 bytes, ZERO-FILLED ONCE, the same pointer in both stages of a call.  The blend forward then
                                 cuts the lists of DEEP tiles (thousands of contributing entries per pixel: the launch ends
                                 when the longest sequential walk ends) into segments that independent workgroups blend at
                                 the same time, each from transmittance 1, and puts them together per pixel; a segment whose
                                 walk depended on the transmittance in front of it (the stop rule fires in it, the median-depth
                                 entry lies in it) is blended again from the true state.  Pixels differ from the one-list walk
                                 by the association of that product (the fast arithmetic's stated tolerance).  NULL: off. */
} fnx_raster_opts_t;
size_t fnx_sort_state_bytes(int P);
size_t fnx_segment_scratch_bytes(int width, int height);
/* Host read-back (blocking) of a view's counters in a segment scratch: out[0] = work items of the last blend launch,
 * out[1] = of those, segments, out[2] = tiles cut into segments, out[3] = running total of tiles whose list was blended
 * on as one list behind some segment, out[4] = running total of the batches that took, out[5 .. 15] = reserved. */
int fnx_segment_scratch_read(const char *scratch, int width, int height, int view, fnx_stream_t stream, uint32_t out[16]);
/* Host read-back (blocking) of a view's counters in a sort state: out[0] = calls in coherent mode, out[1] = of those,
 * calls that fell back to the in-launch full sort, out[2] = why they did, OR-ed over the calls (1: a record not written
 * by the call's preprocess, 2: a sample-sort bucket overflowed, 4: a chunk not strictly increasing, 8: a chunk boundary out
 * of order = an element moved further than the window margin, 16: unseeded state). */
int fnx_sort_state_read(const char *sort_state, int P, int view, fnx_stream_t stream, uint32_t out[3]);
/* Running total of the splats a view's repair calls took as OUTLIERS: splats whose depth left the neighbourhood of their
 * previous rank (about 400 ranks either way, judged by the previous order's sampled keys) travel in a side list of at most
 * 256 per call and are merged in by every repair workgroup, so that a handful of far travellers per call -- fringe
 * particles whose interpolated velocity is noise -- does not cost the in-launch full sort.  (Candidates beyond the 256 stay
 * in place: the repair window reaches 1 024 ranks, and a call it does not suffice for takes the full sort as before.) */
int fnx_sort_state_outliers(const char *sort_state, int P, int view, fnx_stream_t stream, uint32_t *out);

/* Scratch sizes [required<GeometryState>(P), required<ImageState>(W*H), required<BinningState>(R),
 * rasterizer_impl.cu:210,222,266]. */
size_t fnx_geom_bytes(int P, int width, int height); /* also holds per-(splat block, tile) counters */
size_t fnx_image_bytes(int width, int height);
size_t fnx_binning_bytes(int64_t num_rendered);

/*
 * Rasterizer::forward, one call, callback allocation (rasterizer.h:30-54).  Synchronises `stream`
 * once to read num_rendered (the reference's blocking cudaMemcpy, rasterizer_impl.cu:264).
 * Returns the error code; *num_rendered receives the reference's return value.
 */
int fnx_rasterize_forward(int channels, fnx_alloc_fn geometryBuffer, void *geom_user, fnx_alloc_fn binningBuffer,
                          void *binning_user, fnx_alloc_fn imageBuffer, void *image_user, int P, int D, int M,
                          const float *background, int width, int height, const float *means3D, const float *shs,
                          const float *colors_precomp, const float *opacities, const float *scales,
                          float scale_modifier, const float *rotations, const float *cov3D_precomp,
                          const float *viewmatrix, const float *projmatrix, const float *cam_pos, float tan_fovx,
                          float tan_fovy, int prefiltered, float *out_color, float *out_depth, int *radii,
                          fnx_stream_t stream, int *num_rendered);

/*
 * The same forward split at the reference's host sync so that a caller can avoid it:
 *   stage 1 = per-Gaussian preprocess, depth sort of the splats, per-tile instance counts and tile
 *             ranges (num_rendered stays on the device, inside image_buffer);
 *   fnx_read_num_rendered = the optional blocking read-back;
 *   stage 2 = instance emission (straight into depth order) and alpha blending, with a
 *             caller-chosen binning capacity.  If num_rendered > capacity nothing is rendered and
 *             fnx_read_status reports FNX_ERR_CAPACITY.
 */
int fnx_forward_stage1(int channels, char *geom_buffer, char *image_buffer, int P, int D, int M, int width, int height,
                       const float *means3D, const float *shs, const float *colors_precomp, const float *opacities,
                       const float *scales, float scale_modifier, const float *rotations, const float *cov3D_precomp,
                       const float *viewmatrix, const float *projmatrix, const float *cam_pos, float tan_fovx,
                       float tan_fovy, int prefiltered, int *radii, fnx_stream_t stream);
int fnx_read_num_rendered(const char *image_buffer, int width, int height, fnx_stream_t stream, int *num_rendered);
int fnx_forward_stage2(int channels, char *geom_buffer, char *binning_buffer, int64_t binning_capacity,
                       char *image_buffer, int P, int width, int height, const float *background,
                       const float *colors_precomp, const int *radii, float *out_color, float *out_depth,
                       fnx_stream_t stream);
/* Blocking: FNX_OK, or what the last forward / backward that used image_buffer left in the view's status word:
 * FNX_ERR_CAPACITY, FNX_ERR_SORT_SPAN, FNX_ERR_INVALID_ARG (a backward beyond its forward's gradient limit). */
int fnx_read_status(const char *image_buffer, int width, int height, fnx_stream_t stream);

/*
 * Rasterizer::backward (rasterizer.h:56-83).  R = the forward's return value (num_rendered) as in the reference; with
 * the two-stage forward it is the binning capacity stage 2 ran with (the binning blob's layout depends on it; the
 * instance count itself lives in image_buffer).  dL_dmean2D [P,3], dL_dconic [P,4], dL_dopacity [P],
 * dL_dcolor [P,C], dL_dmean3D [P,3], dL_dcov3D [P,6], dL_dsh [P,M,3], dL_dscale [P,3], dL_drot [P,4].
 */
int fnx_rasterize_backward(int channels, int P, int D, int M, int R, const float *background, int width, int height,
                           const float *means3D, const float *shs, const float *colors_precomp, const float *scales,
                           float scale_modifier, const float *rotations, const float *cov3D_precomp,
                           const float *viewmatrix, const float *projmatrix, const float *campos, float tan_fovx,
                           float tan_fovy, const int *radii, char *geom_buffer, char *binning_buffer,
                           char *image_buffer, const float *dL_dpix, float *dL_dmean2D, float *dL_dconic,
                           float *dL_dopacity, float *dL_dcolor, float *dL_dmean3D, float *dL_dcov3D, float *dL_dsh,
                           float *dL_dscale, float *dL_drot, fnx_stream_t stream);

/*
 * Extension of fnx_rasterize_backward for callers that discard part of the result (the reference
 * always computes everything): only splats with id < grad_splat_limit receive gradients (-1 = all;
 * the others still take part in the blending recurrences); with geometry_only == 1 the
 * opacity / colour gradients are not produced (dL_dopacity, dL_dcolor stay as passed in); with
 * geometry_only == 2 ("fixed positions": the visual-particle stage, train_visual_particle.py:133-222, optimises
 * appearance and shape only) the gradient with respect to the 2D means is not accumulated: dL_dmean2D stays as
 * passed in and dL_dmean3D holds the covariance path's share only -- the caller must not use either; with
 * geometry_only == 3 ("positions only": the physical-particle and first-frame stages, train_physical_particle.py,
 * optimise positions alone; colours_precomp, no SH) ONLY dL_dmean3D is produced: the blend backward carries its sums
 * through the per-(splat, view) geometry backward itself and ADDS to dL_dmean3D, which the caller passes in zeroed;
 * every other gradient pointer may be NULL.
 */
int fnx_rasterize_backward_ex(int channels, int P, int D, int M, int R, const float *background, int width, int height,
                              const float *means3D, const float *shs, const float *colors_precomp, const float *scales,
                              float scale_modifier, const float *rotations, const float *cov3D_precomp,
                              const float *viewmatrix, const float *projmatrix, const float *campos, float tan_fovx,
                              float tan_fovy, const int *radii, char *geom_buffer, char *binning_buffer,
                              char *image_buffer, const float *dL_dpix, float *dL_dmean2D, float *dL_dconic,
                              float *dL_dopacity, float *dL_dcolor, float *dL_dmean3D, float *dL_dcov3D, float *dL_dsh,
                              float *dL_dscale, float *dL_drot, int grad_splat_limit, int geometry_only,
                              fnx_stream_t stream);

/*
 * View-batched extension.  The reference renders the views of a training batch one forward/backward
 * call at a time (train_physical_particle.py:338-405); here V cameras looking at the SAME Gaussians
 * go through ONE launch sequence, the view being the second grid dimension of every kernel, so that a
 * small image still fills the 256 compute units and the tail of one view's deep tiles overlaps the
 * other views' work.  Each view's slice of every output and scratch blob is bit-identical to what
 * the single-view calls produce for that camera.
 *
 * Per-view arrays are the single-view arrays repeated V times (V <= FNX_MAX_VIEWS, equal image size):
 *   viewmatrices [V,16], projmatrices [V,16], cam_pos [V,3]  (device);
 *   tan_fovx, tan_fovy: HOST arrays of V floats;
 *   radii [V,P], out_color [V,C,H,W], out_depth [V,1,H,W], dL_dpix [V,C,H,W];
 *   geom_buffers = V * fnx_geom_bytes(P,W,H) bytes, view v at byte offset v * fnx_geom_bytes(P,W,H);
 *   image_buffers likewise with fnx_image_bytes; binning_buffers = V * fnx_binning_bytes(capacity).
 * background, scale_modifier, D, M and `prefiltered` are shared by the views.
 *
 * Backward: dL_dmean2D [V,P,3] (the per-view screen-space gradients, returned), dL_dconic [V,P,4],
 * dL_dopacity_views [V,P] and dL_dcolor_views [V,P,C] are zero-filled per-view accumulators; the
 * remaining outputs are SUMS over the views, written once per splat in view order (no atomics):
 * dL_dopacity [P], dL_dcolor [P,C] (only when colors_precomp is given; may be NULL otherwise),
 * dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot.  With V == 1 dL_dopacity / dL_dcolor may alias
 * their *_views arrays (that is how the single-view entry points are implemented).
 */
#define FNX_MAX_VIEWS 16
int fnx_forward_stage1_views(int channels, int V, char *geom_buffers, char *image_buffers, int P, int D, int M,
                             int width, int height, const float *means3D, const float *shs,
                             const float *colors_precomp, const float *opacities, const float *scales,
                             float scale_modifier, const float *rotations, const float *cov3D_precomp,
                             const float *viewmatrices, const float *projmatrices, const float *cam_pos,
                             const float *tan_fovx, const float *tan_fovy, int prefiltered, int *radii,
                             fnx_stream_t stream);
int fnx_forward_stage2_views(int channels, int V, char *geom_buffers, char *binning_buffers, int64_t binning_capacity,
                             char *image_buffers, int P, int width, int height, const float *background,
                             const int *radii, float *out_color, float *out_depth, fnx_stream_t stream);
/* The same; additionally the last kernel copies the 8 header words of every view (instance count, status, needed
 * capacity, ...) to status_out[8 v ..] (device memory, may be NULL) for a deferred fnx_read_status-like check
 * without a copy per call. */
int fnx_forward_stage2_views_status(int channels, int V, char *geom_buffers, char *binning_buffers,
                                    int64_t binning_capacity, char *image_buffers, int P, int width, int height,
                                    const float *background, const int *radii, float *out_color, float *out_depth,
                                    uint32_t *status_out, fnx_stream_t stream);
int fnx_rasterize_backward_views(int channels, int V, int P, int D, int M, const float *background, int width,
                                 int height, const float *means3D, const float *shs, const float *colors_precomp,
                                 const float *scales, float scale_modifier, const float *rotations,
                                 const float *cov3D_precomp, const float *viewmatrices, const float *projmatrices,
                                 const float *campos, const float *tan_fovx, const float *tan_fovy, const int *radii,
                                 char *geom_buffers, char *binning_buffers, int64_t binning_capacity,
                                 char *image_buffers, const float *dL_dpix, float *dL_dmean2D, float *dL_dconic,
                                 float *dL_dopacity_views, float *dL_dcolor_views, float *dL_dopacity,
                                 float *dL_dcolor, float *dL_dmean3D, float *dL_dcov3D, float *dL_dsh,
                                 float *dL_dscale, float *dL_drot, int grad_splat_limit, int geometry_only,
                                 fnx_stream_t stream);

/*
 * Static-split extension.  In the reference's dynamics stages most Gaussians are frozen: render_dynamics
 * concatenates the optimised fluid Gaussians with the static background ones (renderer/pipe_dynamics.py:46-57),
 * and the cameras of a frame do not move, so within a frame the background's per-tile, depth-ordered instance
 * lists are the same in every iteration (rasterizer_impl.cu:259-296 rebuilds them every call).  Here the caller
 * may declare the LAST P_static splats of its arrays static:
 *   once per frame   fnx_forward_stage1_views over the static subset alone (pointers offset to it) followed by
 *                    fnx_static_finalize_views, which emits its instances and packs what later calls need --
 *                    per-tile (depth bits, id) lists, tile starts, radii, blend records -- into V "static blobs";
 *   every iteration  the *_split entry points below preprocess / sort / emit only the P_dyn leading splats and
 *                    the blend kernel merges the two depth-ordered streams of a tile while it stages them, batch
 *                    by batch, as far as the tile's pixels need them (ties in depth: lower id first, as in the
 *                    reference's stable sort; static ids are the larger ones).
 * Results are bit-identical to the unsplit calls over all P_dyn + P_static splats: colour, depth, radii, final_T,
 * n_contrib, and the prefix of point_list that the blend consumed (everything the backward reads); with
 * materialize_all != 0 the whole merged point_list is written, for parity tests.  Static splats receive no
 * gradients (grad_splat_limit is clamped to P_dyn); a NULL static_blobs selects the unsplit behaviour.
 *
 * Sizes: per-iteration geometry blobs are fnx_geom_bytes(P_dyn, W, H); binning blobs
 * fnx_binning_bytes_split(capacity, R_static_capacity) where capacity bounds the DYNAMIC instances of a view and
 * R_static_capacity is the value given to fnx_static_finalize_views (>= every view's static instance count);
 * static blobs fnx_static_bytes(P_static, W, H, R_static_capacity), view v at v times that.  radii is [V, P_dyn +
 * P_static] (the static part is copied from the blob), the gradient arrays keep their [.., P_dyn + P_static, ..]
 * shapes.  The header words 0..2 of the image blob count DYNAMIC instances (capacity check); word 3 holds the
 * view's static instance count.
 */
size_t fnx_static_bytes(int P_static, int width, int height, int64_t R_static_capacity);
size_t fnx_binning_bytes_split(int64_t capacity, int64_t R_static_capacity);
/* geom_buffers / image_buffers: the blobs of a completed fnx_forward_stage1_views over the static subset;
 * binning_scratch: V * fnx_binning_bytes(R_static_capacity) bytes of scratch; id_offset: id of the first static
 * splat in the caller's full arrays (= P_dyn).  Fails with FNX_ERR_CAPACITY status if a view has more static
 * instances than R_static_capacity (read them with fnx_read_num_rendered first). */
int fnx_static_finalize_views(int V, char *geom_buffers, char *binning_scratch, char *image_buffers, int P_static,
                              int width, int height, int id_offset, int64_t R_static_capacity,
                              const int *radii /* [V, P_static] as written by that stage 1 */, char *static_blobs,
                              fnx_stream_t stream);
int fnx_forward_stage1_views_split(int channels, int V, char *geom_buffers, char *image_buffers, int P_dyn, int D, int M,
                                   int width, int height, const float *means3D, const float *shs,
                                   const float *colors_precomp, const float *opacities, const float *scales,
                                   float scale_modifier, const float *rotations, const float *cov3D_precomp,
                                   const float *viewmatrices, const float *projmatrices, const float *cam_pos,
                                   const float *tan_fovx, const float *tan_fovy, int prefiltered, int *radii,
                                   const char *static_blobs, int P_static, int64_t R_static_capacity,
                                   uint32_t *depth_hint, fnx_stream_t stream);
int fnx_forward_stage2_views_split(int channels, int V, char *geom_buffers, char *binning_buffers,
                                   int64_t binning_capacity, char *image_buffers, int P_dyn, int width, int height,
                                   const float *background, float *out_color, float *out_depth, uint32_t *status_out,
                                   const char *static_blobs, int P_static, int64_t R_static_capacity,
                                   int materialize_all, uint32_t *depth_hint, fnx_stream_t stream);
/* The same three calls with per-call options (opts may be NULL = defaults). */
int fnx_forward_stage1_views_split_opts(int channels, int V, char *geom_buffers, char *image_buffers, int P_dyn, int D,
                                        int M, int width, int height, const float *means3D, const float *shs,
                                        const float *colors_precomp, const float *opacities, const float *scales,
                                        float scale_modifier, const float *rotations, const float *cov3D_precomp,
                                        const float *viewmatrices, const float *projmatrices, const float *cam_pos,
                                        const float *tan_fovx, const float *tan_fovy, int prefiltered, int *radii,
                                        const char *static_blobs, int P_static, int64_t R_static_capacity,
                                        uint32_t *depth_hint, const fnx_raster_opts_t *opts, fnx_stream_t stream);
int fnx_forward_stage2_views_split_opts(int channels, int V, char *geom_buffers, char *binning_buffers,
                                        int64_t binning_capacity, char *image_buffers, int P_dyn, int width, int height,
                                        const float *background, float *out_color, float *out_depth,
                                        uint32_t *status_out, const char *static_blobs, int P_static,
                                        int64_t R_static_capacity, int materialize_all, uint32_t *depth_hint,
                                        const fnx_raster_opts_t *opts, fnx_stream_t stream);
/* depth_hint (may be NULL): u32[V, T] owned by the caller and kept across calls with the same cameras.  Stage 2 records
 * in it how deep (list position of the last contributor) every tile of every view went; stage 1 of the NEXT call reads
 * it (and clears it for that call's stage 2): the tiles that went at least fnx_set_deep_threshold() deep (default 1024)
 * -- lists that do not saturate, thousands of contributing entries per pixel -- are given to the first workgroups of
 * the blend forward, at raised wave priority, because the launch ends when the longest sequential walk ends
 * (csrc/raster_forward.hip).  Purely a scheduling hint: results are bit-identical with any tile order. Zero-fill it once. */
int fnx_set_deep_threshold(unsigned int min_depth);
/* DEPRECATED process-wide default of fnx_raster_opts_t.blend_math (used when opts is NULL or the field is
 * FNX_OPT_DEFAULT).  Arithmetic of the two blend kernels (forward and backward of one render must run in the same mode).
 *   0 (default): bit-reproducible -- every fp32 expression of forward.cu:319-345 as written, no contraction, exp() as one
 *      fixed instruction sequence; pixels / depth / n_contrib equal the CPU oracle bit for bit.
 *   1: fast, stated tolerance -- fused multiply-adds, log2(e) and log2(opacity) folded into per-entry coefficients,
 *      exp as one v_exp_f32 (<= 1 ulp), T <- T - alpha T.  Same lists, same tests (power > 0, alpha < 1/255, T < 1e-4).
 *      |pixel - exact| <= 2e-5 except where a rounding moves an alpha across 1/255 or a T across 1e-4 (a counted
 *      <= 1e-4 fraction of pixels), gradients within 1e-3 relative + 2e-5 of the largest entry; tile / bin indices,
 *      radii and ranges are computed before the blend and stay bit-exact.  This is the mode an nvcc build of the
 *      reference (contraction on, libdevice expf) is closest to; tests/test_fast_math_gpu.py. */
int fnx_set_blend_math(int mode);
int fnx_get_blend_math(void);
/* Fast mode only: tiles that went at least fnx_set_deep_threshold() deep in the previous forward of their view (depth_hint)
 * are blended by a second kernel that takes 1024 list entries at a time -- four 256-entry sub-batches walked
 * concurrently, each from the transmittance a cheap pre-pass computed for it -- instead of one serial walk per tile
 * (csrc/raster_forward.hip, blend_forward_deep_kernel), on a helper stream beside the per-tile kernel.  A deep
 * workgroup holds a whole compute unit to cut the tile's latency: it pays when a launch is bound by its longest walks
 * (one or two views per launch: a rank's share of a sharded batch) and costs throughput otherwise.
 * mode 0: never; 1: launches of at most two views; 2: always.
 * Measured: a one-view forward alone 259 -> 171 us, but inside the replayed iteration the 1024-thread
 * workgroups wait for an empty compute unit behind the per-tile kernel's workgroups and the iteration gets slower
 * (config 3, 2 of 5 views: 1251 -> 1216 it/s), so it is opt-in.
 * Modes 3 .. 5 (round 6), both arithmetics, one image: the blend forward with STAGING WAVES (csrc/raster_forward_ws.h,
 * blend_forward_ws_kernel) -- 512 threads per tile, four waves walk a batch while four others stage the next one into the
 * other half of double-buffered LDS arrays; per pixel the per-tile kernel's arithmetic in its order, outputs bit-equal to
 * its (tests/test_staging_waves_gpu.py).  3: the tiles that went deep in the previous forward, on a helper stream beside
 * the per-tile kernel; 4: every tile, instead of the per-tile kernel; 5 (DEFAULT): by the number of views in the launch --
 * 4 for one or two views, 3 for three, the per-tile kernel for more (a launch of few views is bound by its deepest tiles'
 * chains, a launch of many by the compute units' instruction throughput: DESIGN.md 4.11). */
int fnx_set_deep_kernel(int mode);
/* Which form of the blend BACKWARD the next calls run (process-wide; both are parity-tested against the oracle, same work
 * items, same staging, same gradients within the backward's stated fp32 bound -- the sums are associated differently):
 *   0  a lane = a PIXEL: every 16-lane row walks the list of one 4x4 block entry by entry (rounds 1-5);
 *   1  a lane = a LIST ENTRY (default, round 6): a wave takes one block with 4 pixel rows x 16 entries, the transmittance in
 *      front of an entry is a 16-lane product scan, the per-entry sums stay in the lane (csrc/raster_backward_lanes.h).
 * The dual mode (fnx_raster_dual_t) always runs form 0.  FNX_BWD_FORM in the environment sets the initial value. */
int fnx_set_backward_form(int form);
int fnx_get_backward_form(void);
/* View-batched entry points (V > 1): 1 = the per-view copies of the reference's GeometryState that this library never
 * reads back (means2D, depths, conic_opacity, tiles_touched: 32 of the 136 bytes a visible splat writes per view) are
 * not written, and the world covariance -- identical for every view -- is written for view 0 only and read at stride 0
 * by the backward.  Forward and backward of one render must run under the same setting.  Default 0 (everything written:
 * fnx_geom_layout offsets stay meaningful for tools and tests). */
int fnx_set_lean_geometry(int on);
/* DEPRECATED (fnx_raster_opts_t.zero3).  One-shot, per host thread, consumed -- used or not -- by the next stage-1 entry on
 * that thread whatever it returns: the NEXT fnx_forward_stage1* call also zero-fills rows3[3 * (P_dyn + P_static)]
 * floats (its per-splat kernel writes the zeros on the way).  For the positions-only backward (geometry_only = 3), whose
 * dL_dmean3D accumulator must come in zeroed: the fill otherwise is a launch of its own on the critical path between
 * the image loss and the backward.  NULL cancels. */
int fnx_request_zero3(float *rows3);
/* DEPRECATED (fnx_raster_opts_t.grad_splat_limit).  One-shot, per host thread, consumed by the next stage-2 entry on that
 * thread whatever it returns: the NEXT stage 2 (fnx_forward_stage2*, fnx_rasterize_forward) is told that only splats with id <
 * grad_splat_limit will be differentiated (< 0: all; the same value the backward entry points take).  The forward then
 * records, per pixel, the list position of the last such splat at or in front of the pixel's last contributor (second
 * half of the image blob's n_contrib array) and lays down backward work items only for the batches up to it: a backward
 * pass needs nothing from the entries behind -- what lies behind an entry enters its gradient only through the final
 * colour and transmittance the forward stores -- so with a frozen background BEHIND the optimised splats it skips the
 * tail of every list (benchmark frame: 28-39 % of the walked (pixel, entry) pairs, 41-52 % of the batches).  Gradients
 * are unchanged bit for bit.  In static-split mode the limit is implied (static splats never take gradients).  A
 * backward call with a LARGER limit than its forward's is refused (FNX_ERR_INVALID_ARG in the view's status word, no
 * gradients). */
int fnx_request_gradient_limit(int grad_splat_limit);
/* The depth sort runs 9-bit passes over keys taken relative to the view's nearest visible splat: three passes order any
 * view whose depths span less than 2^27 ulps (far / near < ~2^4 at equal exponent ... 2^16 across exponents); the three
 * kernels of the fourth pass are launched all the same and return at once when it is not needed (~14 us of launches on
 * the critical path).  1 = the caller has seen (status word 5, bits 24-31 = bit length of the span; the Python layer
 * keeps the maximum) that three passes suffice: the fourth is not launched, and a view that would have needed it sets
 * FNX_ERR_SORT_SPAN in its status word instead of rendering.  Default 0. */
int fnx_set_sort_narrow(int on);
int fnx_rasterize_backward_views_split(int channels, int V, int P_dyn, int D, int M, const float *background, int width,
                                       int height, const float *means3D, const float *shs,
                                       const float *colors_precomp, const float *scales, float scale_modifier,
                                       const float *rotations, const float *cov3D_precomp,
                                       const float *viewmatrices, const float *projmatrices, const float *campos,
                                       const float *tan_fovx, const float *tan_fovy, const int *radii,
                                       char *geom_buffers, char *binning_buffers, int64_t binning_capacity,
                                       char *image_buffers, const float *dL_dpix, float *dL_dmean2D, float *dL_dconic,
                                       float *dL_dopacity_views, float *dL_dcolor_views, float *dL_dopacity,
                                       float *dL_dcolor, float *dL_dmean3D, float *dL_dcov3D, float *dL_dsh,
                                       float *dL_dscale, float *dL_drot, int grad_splat_limit, int geometry_only,
                                       const char *static_blobs, int P_static, int64_t R_static_capacity,
                                       fnx_stream_t stream);

/* status_out (may be NULL): as in stage 2 -- the header words of every view are copied there behind the backward, so that a
 * refused backward (FNX_ERR_INVALID_ARG: gradient limit beyond the forward's; FNX_ERR_CAPACITY: blob of another capacity)
 * reaches a caller that checks its status rows instead of returning zero gradients silently. */
int fnx_rasterize_backward_views_split_opts(int channels, int V, int P_dyn, int D, int M, const float *background,
                                            int width, int height, const float *means3D, const float *shs,
                                            const float *colors_precomp, const float *scales, float scale_modifier,
                                            const float *rotations, const float *cov3D_precomp,
                                            const float *viewmatrices, const float *projmatrices, const float *campos,
                                            const float *tan_fovx, const float *tan_fovy, const int *radii,
                                            char *geom_buffers, char *binning_buffers, int64_t binning_capacity,
                                            char *image_buffers, const float *dL_dpix, float *dL_dmean2D,
                                            float *dL_dconic, float *dL_dopacity_views, float *dL_dcolor_views,
                                            float *dL_dopacity, float *dL_dcolor, float *dL_dmean3D, float *dL_dcov3D,
                                            float *dL_dsh, float *dL_dscale, float *dL_drot, int grad_splat_limit,
                                            int geometry_only, const char *static_blobs, int P_static,
                                            int64_t R_static_capacity, uint32_t *status_out,
                                            const fnx_raster_opts_t *opts, fnx_stream_t stream);

/* Rasterizer::markVisible (rasterizer.h:20-25): present[i] = view-space z > 0.2 (auxiliary.h:138). */
int fnx_mark_visible(int P, const float *means3D, const float *viewmatrix, const float *projmatrix, uint8_t *present,
                     fnx_stream_t stream);

/*
 * Optional kernel timing for benchmarks: when enabled, HIP events are recorded on the caller's
 * stream around one kernel class per launch (0 blend forward, 1 blend backward, 2 depth sort +
 * instance counting + scans, 3 preprocess, 4 instance emission; 5 / 6 blend forward / backward of channels == 1).  fnx_profile_read blocks on those events
 * and returns the summed duration and the number of launches since fnx_profile_enable(1).
 */
int fnx_profile_enable(int on);
int fnx_profile_read(int which, double *total_ms, int *launches);

/*
 * Byte offsets of the named arrays inside the opaque scratch blobs, for parity tests and tools
 * (the reference re-derives the same pointers with fromChunk, rasterizer_impl.cu:144-180).
 */
typedef struct {
    size_t depths;        /* f32[P]   view-space z                                  */
    size_t clamped;       /* u8[3P]   SH clamp flags                                */
    size_t radii;         /* i32[P]   internal radii (used when radii == NULL)      */
    size_t means2D;       /* f32[2P]  pixel-space centres                           */
    size_t cov3D;         /* f32[6P]                                                */
    size_t conic_opacity; /* f32[4P]                                                */
    size_t rgb;           /* f32[3P]  SH colours                                    */
    size_t tiles_touched; /* u32[P]                                                 */
    size_t sort_key0;     /* u32[P]   depth bits (culled splats: bit 31 | their depth bits) as written by the preprocess; with
                             sort_key1 one (key - nearest key, id) pair buffer uint2[P] of the depth sort */
    size_t sort_key1;     /* u32[P]   second half of that pair buffer                */
    size_t sort_val0;     /* u32[P]   with sort_val1 the other pair buffer uint2[P]; it holds the pairs in
                             (depth bits, id) order after three passes, the first one after four        */
    size_t sort_val1;     /* u32[P]                                                 */
    size_t rect;          /* u32[2P]  tile rectangle of each splat: x0 | x1 << 16, y0 | y1 << 16 (0, 0 if invisible) */
    size_t rect_sorted;   /* u32[2P]  the same in depth-rank order                  */
    size_t krec;          /* u32[4P]  FNX_SORT_COHERENT: (depth bits, id, rectangle as 4 bytes, call stamp) of every splat,
                             written by the preprocess at the splat's PREVIOUS depth rank                            */
    size_t sort_hist;     /* u32[...] sort scratch: 512-digit chunk histograms + prefixes, digit totals, control
                             words (nearest key, fourth-pass flag), key minima / maxima per block, instances per
                             rank block, emission work items (csrc/fnx_state.h: sort_scratch)                  */
    size_t blk_hist;      /* u16[ceil(P/1024) * T] splats of depth-rank block b touching tile t */
    size_t blk_rel;       /* u32[ceil(P/1024) * T] exclusive prefix over blocks     */
    size_t blend_rec;     /* f32[16P] packed per-splat record read by the blend kernels:
                             x y a b | c o thr depth | ex ey col0 col1 | col2 - - -   */
    size_t total;
} fnx_geom_layout_t;
typedef struct {
    size_t header;      /* u32[16]: [0] num_rendered, [1] status, [2] capacity seen, [3] static instances (split),
                           [4] backward work items, [5] tiles scheduled first ("deep") | bit length of the depth-key
                           span << 24, [6] binning capacity stage 2 ran with, [7] / [8] backward ticket / arrival
                           counters, [9] the forward's gradient limit, [10] list entries the blend forward staged
                           (batches x 256 clipped to the lists), [11] list entries the blend backward walked (work items
                           x their batch length), [12..15] reserved                                                  */
    size_t final_T;     /* f32[H*W]                                                 */
    size_t n_contrib;   /* u32[2*H*W]: last contributor per pixel | the limited backward's walking limit per pixel
                           (fnx_request_gradient_limit; equal to the first half without a limit) */
    size_t ranges;      /* u32[2T] per-tile [start,end) in point_list               */
    size_t tile_count;  /* u32[T]   instances emitted by this call (split: the dynamic ones) */
    size_t dyn_start;   /* u32[T]   split mode: start of the tile's dynamic (key, id) pairs */
    size_t acc_final;   /* f32[3 H W] (C planes used) colour accumulated by the blend before the background term */
    size_t tile_order;  /* u32[T]   tile of every workgroup of the blend forward: the tiles that went deep in the previous
                           forward of the view first (their count: header word 5), then the XCD-aware order          */
    size_t tile_deep;   /* u8[T]    1 for those tiles                                */
    size_t total;
} fnx_image_layout_t;
typedef struct {
    size_t point_list; /* u32[R (+ R_static)] Gaussian ids sorted by (tile, depth bits, id) */
    size_t pairs;      /* split mode: u32[2R] (depth bits, id) of the dynamic instances, per tile in depth order */
    size_t bstate;     /* f32[(R / 256 + 2) * 256 * 4] per-pixel (T, colour) in front of every 256-entry batch b >= 1 the
                          forward blended: slot (tile's first list position) / 256 + b - 1                          */
    size_t bwd_items;  /* u32[...] backward work items (tile | batch << 14) written by the forward; their count is
                          header word 4                                                                             */
    size_t block_masks; /* u16[R (+ R_static)] per list entry the forward blended: which 4x4 blocks of its tile it can
                          reach (bit 4 q + b: block b of quadrant q); the backward builds its lists from them        */
    size_t total;
} fnx_binning_layout_t;
typedef struct {
    size_t header;     /* u32[8]: [0] static instances of the view, [1] P_static, [2] id of the first static splat */
    size_t starts;     /* u32[T+1] exclusive prefix of the per-tile static instance counts */
    size_t radii;      /* i32[P_static]                                             */
    size_t blend_rec;  /* f32[16 P_static] packed blend records (fnx_geom_layout_t) */
    size_t pairs;      /* u32[2 R_static] (depth bits, global id), per tile in (depth bits, id) order */
    size_t total;
} fnx_static_layout_t;
void fnx_geom_layout(int P, int width, int height, fnx_geom_layout_t *out);
void fnx_image_layout(int width, int height, fnx_image_layout_t *out);
void fnx_binning_layout(int64_t num_rendered, fnx_binning_layout_t *out);
void fnx_binning_layout_split(int64_t capacity, int64_t R_static_capacity, fnx_binning_layout_t *out);
void fnx_static_layout(int P_static, int width, int height, int64_t R_static_capacity, fnx_static_layout_t *out);

#ifdef __cplusplus
}
#endif
#endif /* FNX_RASTER_H */
