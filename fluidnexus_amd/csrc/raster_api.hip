// C ABI of the rasteriser (include/fnx_raster.h): argument checks, scratch-blob carving and the
// launch sequence.  No torch types, no allocation, no host synchronisation except where the
// header says so.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <utility>
#include <vector>

#include "fnx_state.h"

namespace fnx {
// launchers defined in raster_forward.hip / raster_binning.hip / raster_backward.hip; V = number of views
// (grid dimension y), vb = per-view blob strides and camera intrinsics
void launch_preprocess(int C, hipStream_t s, int P, int D, int M, const float *means3D, const float *scales,
                       float scale_modifier, const float *rotations, const float *opacities, const float *shs,
                       uint8_t *clamped, const float *cov3D_precomp, const float *colors_precomp, const float *view,
                       const float *proj, const float *campos, int W, int H, int *radii, float2 *means2D,
                       float *depths, float *cov3Ds, float *rgb, float4 *conic_opacity, uint32_t *tiles_touched,
                       uint32_t *sort_key, uint32_t *key_min_blk, uint2 *rect, float4 *blend_rec, int prefiltered, int V,
                       const ViewBatch &vb, const StaticRef &st, int lean, float *zero3, const CohRef &coh, int fast_sh);
void launch_tile_scan(hipStream_t s, int T, const uint32_t *tile_count, uint32_t *ranges, uint32_t *dyn_start,
                      uint32_t *header, int P, int H, uint32_t *sort_scratch_words, uint32_t *depth_hint,
                      uint32_t deep_min, uint32_t *tile_order, uint8_t *tile_deep, int V, const ViewBatch &vb,
                      const StaticRef &st, const SegRef &sg);
void launch_tile_colscan(hipStream_t s, int T, int P, const uint16_t *blk_hist, uint32_t *blk_rel,
                         uint32_t *tile_count, int V, const ViewBatch &vb);
void launch_depth_sort(hipStream_t s, int P, const uint32_t *raw_keys, uint2 *pairs_a, uint2 *pairs_b,
                       uint32_t *scratch, const uint2 *rect, uint2 *rect_sorted, int V, const ViewBatch &vb, int narrow,
                       char *coh_state, int coherent, const uint4 *krec, int W, int H, uint16_t *blk_hist,
                       uint32_t *blk_total, int *hist_done);
void launch_rank_hist(hipStream_t s, int P, int W, int H, const uint2 *rect_sorted, uint16_t *blk_hist,
                      uint32_t *blk_total, int V, const ViewBatch &vb);
void launch_emit(hipStream_t s, int P, int W, int H, const uint2 *sorted3, const uint2 *sorted4,
                 const uint32_t *sort_ctl, const uint2 *rect_sorted, const uint32_t *starts, const uint32_t *blk_rel,
                 uint32_t *emit_ctl, const uint32_t *emit_items, uint32_t *point_list, uint32_t *header,
                 uint32_t capacity, int pairs, int V, const ViewBatch &vb);
void launch_static_pack(hipStream_t s, int P, int W, int H, const uint32_t *point_list, const uint32_t *dyn_start,
                        const uint32_t *header, const int *radii, const float4 *blend_rec, char *blob,
                        size_t blob_stride, const fnx_static_layout_t &L, uint32_t id0, uint32_t r_capacity, int V,
                        const ViewBatch &vb);
void launch_blend_forward(int C, hipStream_t s, int W, int H, const uint32_t *ranges, uint32_t *point_list,
                          const float4 *blend_rec, const float *bg, float *final_T, uint32_t *n_contrib,
                          float *out_color, float *out_depth, uint32_t *header, uint32_t capacity,
                          uint32_t *status_out, const uint32_t *tile_count, const uint32_t *dyn_start,
                          float *acc_final, const uint32_t *tile_order, const uint8_t *tile_deep, uint32_t *depth_hint,
                          const StaticRef &st, int materialize_all, int V, const ViewBatch &vb, int fast, int deep,
                          uint32_t dyn_limit, const InvUpdate &iu, const DualRef &du, const SegRef &sg);
void launch_mark_visible(hipStream_t s, int P, const float *means3D, const float *view, uint8_t *present);
void set_backward_form(int form);
int get_backward_form();
void launch_blend_backward(int C, int mode, hipStream_t s, int P, int W, int H, const uint32_t *ranges,
                           const uint32_t *point_list, const float *bg, const float4 *blend_rec, const float *final_Ts,
                           const uint32_t *n_contrib, const float *acc_final, const float *dL_dpixels,
                           float *dL_dmean2D, float *dL_dconic, float *dL_dopacity, float *dL_dcolors,
                           const uint32_t *header, uint32_t capacity, uint32_t grad_limit, int V, const ViewBatch &vb,
                           const StaticRef &st, const float *means3D, const float *cov3Ds, size_t cov3D_stride,
                           const float *viewmatrix, const float *projmatrix, float *dL_dmean3D, int fast,
                           uint32_t *status_out, const DualRef &du);
void launch_geom_backward(int C, hipStream_t s, int P, int D, int M, const float *means3D, const int *radii,
                          const float *shs, const uint8_t *clamped, const float *scales, const float *rotations,
                          float scale_modifier, const float *cov3Ds, size_t cov3D_stride, const float *view,
                          const float *proj, const float *campos, const float *dL_dmean2D, const float *dL_dconic,
                          const float *dL_dopacity_views, const float *dL_dcolor_views, float *dL_dopacity,
                          float *dL_dcolor, float *dL_dmean3D, float *dL_dcov3D, float *dL_dsh, float *dL_dscale,
                          float *dL_drot, int grad_limit, int V, int sum_appearance, const ViewBatch &vb);
}  // namespace fnx

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int hip_check(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FNX_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    return FNX_OK;
}

char *aligned(char *p) { return (char *)(((uintptr_t)p + fnx::kAlign - 1) / fnx::kAlign * fnx::kAlign); }
// the segment scratch of a view batch (fnx_raster_opts_t.segment_scratch); off when the tile grid does not fit a work item
fnx::SegRef make_seg_ref(char *scratch, int width, int height) {
    fnx::SegRef sg;
    memset(&sg, 0, sizeof(sg));
    const int T = fnx::tiles_x(width) * fnx::tiles_y(height);
    if (!scratch || T > fnx::kMaxTiles) return sg;
    sg.L = fnx::seg_layout(T);
    sg.stride = sg.L.total;
    sg.base = aligned(scratch);
    return sg;
}
const char *aligned(const char *p) { return aligned(const_cast<char *>(p)); }

struct Geom {
    float *depths;
    uint8_t *clamped;
    int *radii;
    float2 *means2D;
    float *cov3D;
    float4 *conic_opacity;
    float *rgb;
    uint32_t *tiles_touched;
    uint32_t *sort_key0, *sort_key1, *sort_val0, *sort_val1, *sort_hist;
    uint2 *rect, *rect_sorted;
    uint4 *krec;
    uint16_t *blk_hist;
    uint32_t *blk_rel;
    float4 *blend_rec;
};
Geom carve_geom(char *blob, int P, int W, int H) {
    fnx_geom_layout_t L;
    fnx::geom_layout(P, W, H, &L);
    char *b = aligned(blob);
    Geom g;
    g.depths = (float *)(b + L.depths);
    g.clamped = (uint8_t *)(b + L.clamped);
    g.radii = (int *)(b + L.radii);
    g.means2D = (float2 *)(b + L.means2D);
    g.cov3D = (float *)(b + L.cov3D);
    g.conic_opacity = (float4 *)(b + L.conic_opacity);
    g.rgb = (float *)(b + L.rgb);
    g.tiles_touched = (uint32_t *)(b + L.tiles_touched);
    g.sort_key0 = (uint32_t *)(b + L.sort_key0);
    g.sort_key1 = (uint32_t *)(b + L.sort_key1);
    g.sort_val0 = (uint32_t *)(b + L.sort_val0);
    g.sort_val1 = (uint32_t *)(b + L.sort_val1);
    g.sort_hist = (uint32_t *)(b + L.sort_hist);
    g.rect = (uint2 *)(b + L.rect);
    g.rect_sorted = (uint2 *)(b + L.rect_sorted);
    g.krec = (uint4 *)(b + L.krec);
    g.blk_hist = (uint16_t *)(b + L.blk_hist);
    g.blk_rel = (uint32_t *)(b + L.blk_rel);
    g.blend_rec = (float4 *)(b + L.blend_rec);
    return g;
}
struct Img {
    uint32_t *header;
    float *final_T;
    uint32_t *n_contrib;
    uint32_t *ranges;
    uint32_t *tile_count;
    uint32_t *dyn_start;
    float *acc_final;
    uint32_t *tile_order;
    uint8_t *tile_deep;
};
Img carve_img(char *blob, int W, int H) {
    fnx_image_layout_t L;
    fnx::image_layout(W, H, &L);
    char *b = aligned(blob);
    Img i;
    i.header = (uint32_t *)(b + L.header);
    i.final_T = (float *)(b + L.final_T);
    i.n_contrib = (uint32_t *)(b + L.n_contrib);
    i.ranges = (uint32_t *)(b + L.ranges);
    i.tile_count = (uint32_t *)(b + L.tile_count);
    i.dyn_start = (uint32_t *)(b + L.dyn_start);
    i.acc_final = (float *)(b + L.acc_final);
    i.tile_order = (uint32_t *)(b + L.tile_order);
    i.tile_deep = (uint8_t *)(b + L.tile_deep);
    return i;
}
struct Bin {
    uint32_t *point_list;
};
Bin carve_bin(char *blob, int64_t R) {
    fnx_binning_layout_t L;
    fnx::binning_layout(R, 0, false, &L);
    char *b = aligned(blob);
    Bin o;
    o.point_list = (uint32_t *)(b + L.point_list);
    return o;
}

bool channels_ok(int c) { return c == 1 || c == 3; }

// Strides and intrinsics of a batch of V views (focal lengths as rasterizer_impl.cu:207-208).
// P = splats this call preprocesses; with a static set (P_static > 0 and `split`) the binning blobs use the split layout
int make_view_batch(int V, int P, int W, int H, int64_t capacity, const float *tan_fovx, const float *tan_fovy,
                    fnx::ViewBatch *vb, bool split = false, int P_static = 0, int64_t R_static = 0, bool dual = false) {
    if (V < 1 || V > fnx::kMaxViews) return fail(FNX_ERR_INVALID_ARG, "V must be in [1, %d] (got %d)", fnx::kMaxViews, V);
    if (!tan_fovx || !tan_fovy) return fail(FNX_ERR_INVALID_ARG, "tan_fovx / tan_fovy is NULL");
    memset(vb, 0, sizeof(*vb));
    vb->geom = fnx_geom_bytes(P, W, H);
    vb->img = fnx_image_bytes(W, H);
    fnx_binning_layout_t BL;
    fnx::binning_layout(capacity, R_static, split, &BL);
    vb->bin = dual ? fnx::binning_dual_total(capacity, R_static, split) : BL.total;
    vb->bin_pairs = BL.pairs;
    vb->bin_bstate = BL.bstate;
    vb->bin_items = BL.bwd_items;
    vb->bin_masks = BL.block_masks;
    vb->radii_stride = (size_t)P + (size_t)(split ? P_static : 0);
    for (int v = 0; v < V; v++) {
        vb->tan_fovx[v] = tan_fovx[v];
        vb->tan_fovy[v] = tan_fovy[v];
        vb->focal_y[v] = H / (2.0f * tan_fovy[v]);
        vb->focal_x[v] = W / (2.0f * tan_fovx[v]);
    }
    return FNX_OK;
}

// static_blobs == NULL: no static set (every field zero)
int make_static_ref(const char *static_blobs, int P_dyn, int P_static, int W, int H, int64_t R_static,
                    fnx::StaticRef *st) {
    memset(st, 0, sizeof(*st));
    if (!static_blobs) return FNX_OK;
    if (P_static < 0 || R_static < 0 || R_static > 0xFFFFFFFFll) return fail(FNX_ERR_INVALID_ARG, "bad static set size");
    fnx_static_layout_t L;
    fnx::static_layout(P_static, W, H, R_static, &L);
    st->base = aligned(static_blobs);
    st->stride = L.total;
    st->id0 = (uint32_t)P_dyn;
    st->P = P_static;
    st->starts = L.starts;
    st->radii = L.radii;
    st->rec = L.blend_rec;
    st->pairs = L.pairs;
    return FNX_OK;
}

// Dual mode (fnx_raster_dual_t): the kernels' view of it; `d` == NULL: off (every field zero)
int make_dual_ref(const fnx_raster_dual_t *d, int channels, bool split, int width, int height, int64_t capacity,
                  int64_t R_static, bool backward, fnx::DualRef *du) {
    memset(du, 0, sizeof(*du));
    if (!d) return FNX_OK;
    if (channels != 3 || !split)
        return fail(FNX_ERR_INVALID_ARG, "fnx_raster_dual_t needs channels = 3 and a static-split view batch");
    if (!d->image_buffers1 || !d->background1 || (backward ? !d->dL_dpix1 : (!d->out_color1 || !d->out_depth1)))
        return fail(FNX_ERR_INVALID_ARG, "fnx_raster_dual_t: a required pointer is NULL");
    fnx_image_layout_t L;
    fnx::image_layout(width, height, &L);
    du->img1 = aligned(d->image_buffers1);
    du->final_T = L.final_T;
    du->n_contrib = L.n_contrib;
    du->acc_final = L.acc_final;
    du->bg1 = d->background1;
    du->out_color1 = d->out_color1;
    du->out_depth1 = d->out_depth1;
    du->dL_dpix1 = d->dL_dpix1;
    du->bin_bstate1 = fnx::binning_dual_offset(capacity, R_static, true);
    return FNX_OK;
}

// Optional in-library kernel timing (bench.py roofline): HIP events recorded on the caller's
// stream around one kernel class; elapsed times are summed when read.
constexpr int kProfClasses = 7;  // 0 blend_forward, 1 blend_backward, 2 sort + counts + scans, 3 preprocess, 4 emit (5, 6: blend forward / backward of the 1-channel rasteriser)
struct ProfClass {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
    size_t used = 0;
};
// Process-wide DEFAULTS of the per-call options (deprecated setters, include/fnx_raster.h): read when a call passes no
// fnx_raster_opts_t or leaves a field at FNX_OPT_DEFAULT.
uint32_t g_deep_min = 1024;  // list depth from which a tile is scheduled first in the next blend forward
int g_sort_narrow = 0;       // the fourth depth-sort pass is not launched (fnx_set_sort_narrow)
int g_lean_geometry = 0;     // view batches: skip the unread GeometryState copies, one world covariance for all views
int g_deep_kernel = 5;       // which blend forward kernel takes which tiles (fnx_set_deep_kernel); 5: by the launch's view count
int g_blend_math = 0;        // 0: bit-reproducible arithmetic (fixed exp sequence, no contraction), 1: fast (fnx_set_blend_math)
// Deprecated one-shot requests: per host thread, and TAKEN (cleared) at the top of the entry point they are meant for,
// whatever that call then returns -- an early return can no longer leave a stale pointer or limit armed for an
// unrelated later call.
thread_local float *t_zero_request = nullptr;                  // fnx_request_zero3 -> next stage 1
thread_local uint32_t t_grad_limit_request = 0xFFFFFFFFu;      // fnx_request_gradient_limit -> next stage 2

// What a call runs with: its fnx_raster_opts_t, field by field, over the process-wide defaults.
struct Opts {
    int blend_math, lean_geometry, sort_mode, deep_kernel;
    uint32_t grad_limit;  // 0xFFFFFFFF: none
    uint32_t deep_min;
    float *zero3;
    char *sort_state;
    const fnx_raster_dual_t *dual;
    char *segment_scratch;
};
int resolve_opts(const fnx_raster_opts_t *o, float *pending_zero3, uint32_t pending_limit, Opts *out) {
    out->blend_math = g_blend_math;
    out->lean_geometry = g_lean_geometry;
    out->sort_mode = g_sort_narrow ? FNX_SORT_NARROW : FNX_SORT_FULL;
    out->deep_kernel = g_deep_kernel;
    out->grad_limit = pending_limit;
    out->deep_min = g_deep_min;
    out->zero3 = pending_zero3;
    out->sort_state = nullptr;
    out->dual = nullptr;
    out->segment_scratch = nullptr;
    if (!o) return FNX_OK;
    if (o->size != sizeof(fnx_raster_opts_t))
        return fail(FNX_ERR_INVALID_ARG, "fnx_raster_opts_t.size is %u, this library's is %u (ABI %d)", o->size,
                    (unsigned)sizeof(fnx_raster_opts_t), FNX_ABI_VERSION);
    if (o->blend_math != FNX_OPT_DEFAULT) out->blend_math = o->blend_math;
    if (o->lean_geometry != FNX_OPT_DEFAULT) out->lean_geometry = o->lean_geometry ? 1 : 0;
    if (o->sort_mode != FNX_OPT_DEFAULT) out->sort_mode = o->sort_mode;
    if (o->deep_kernel != FNX_OPT_DEFAULT) out->deep_kernel = o->deep_kernel;
    // an options struct speaks for the whole call: pending one-shot requests are dropped, not merged
    out->grad_limit = (o->grad_splat_limit == FNX_OPT_DEFAULT || o->grad_splat_limit < 0) ? 0xFFFFFFFFu
                                                                                           : (uint32_t)o->grad_splat_limit;
    if (o->deep_threshold) out->deep_min = o->deep_threshold;
    out->zero3 = o->zero3;
    out->sort_state = o->sort_state;
    out->dual = o->dual;
    out->segment_scratch = o->segment_scratch;
    if (out->blend_math != 0 && out->blend_math != 1) return fail(FNX_ERR_INVALID_ARG, "blend_math must be 0 (exact) or 1 (fast)");
    if (out->sort_mode < FNX_SORT_FULL || out->sort_mode > FNX_SORT_COHERENT) return fail(FNX_ERR_INVALID_ARG, "bad sort_mode");
    if (out->sort_mode == FNX_SORT_COHERENT && !out->sort_state)
        return fail(FNX_ERR_INVALID_ARG, "sort_mode FNX_SORT_COHERENT needs fnx_raster_opts_t.sort_state");
    if (out->deep_kernel < 0 || out->deep_kernel > 5) return fail(FNX_ERR_INVALID_ARG, "deep_kernel must be 0 .. 5");
    return FNX_OK;
}
bool g_prof_on = false;
ProfClass g_prof[kProfClasses];

struct ProfScope {
    hipEvent_t stop = nullptr;
    hipStream_t s;
    ProfScope(int cls, hipStream_t stream) : s(stream) {
        if (!g_prof_on) return;
        ProfClass &c = g_prof[cls];
        if (c.used == c.pool.size()) {
            hipEvent_t a, b;
            (void)hipEventCreate(&a);
            (void)hipEventCreate(&b);
            c.pool.emplace_back(a, b);
        }
        auto &pr = c.pool[c.used++];
        (void)hipEventRecord(pr.first, s);
        stop = pr.second;
    }
    ~ProfScope() {
        if (stop) (void)hipEventRecord(stop, s);
    }
};

}  // namespace

extern "C" {

int fnx_abi_version(void) { return FNX_ABI_VERSION; }
const char *fnx_last_error(void) { return g_err; }

size_t fnx_geom_bytes(int P, int W, int H) {
    fnx_geom_layout_t L;
    fnx::geom_layout(P, W, H, &L);
    return L.total;
}
size_t fnx_image_bytes(int W, int H) {
    fnx_image_layout_t L;
    fnx::image_layout(W, H, &L);
    return L.total;
}
size_t fnx_binning_bytes(int64_t R) {
    fnx_binning_layout_t L;
    fnx::binning_layout(R, 0, false, &L);
    return L.total;
}
size_t fnx_binning_bytes_split(int64_t capacity, int64_t R_static_capacity) {
    fnx_binning_layout_t L;
    fnx::binning_layout(capacity, R_static_capacity, true, &L);
    return L.total;
}
size_t fnx_binning_bytes_dual(int64_t capacity, int64_t R_static_capacity) {
    return fnx::binning_dual_total(capacity, R_static_capacity, true);
}
size_t fnx_static_bytes(int P_static, int W, int H, int64_t R_static_capacity) {
    fnx_static_layout_t L;
    fnx::static_layout(P_static, W, H, R_static_capacity, &L);
    return L.total;
}
void fnx_geom_layout(int P, int W, int H, fnx_geom_layout_t *out) { fnx::geom_layout(P, W, H, out); }
void fnx_image_layout(int W, int H, fnx_image_layout_t *out) { fnx::image_layout(W, H, out); }
void fnx_binning_layout(int64_t R, fnx_binning_layout_t *out) { fnx::binning_layout(R, 0, false, out); }
void fnx_binning_layout_split(int64_t capacity, int64_t R_static_capacity, fnx_binning_layout_t *out) {
    fnx::binning_layout(capacity, R_static_capacity, true, out);
}
void fnx_static_layout(int P_static, int W, int H, int64_t R_static_capacity, fnx_static_layout_t *out) {
    fnx::static_layout(P_static, W, H, R_static_capacity, out);
}

int fnx_forward_stage1_views_split_opts(int channels, int V, char *geom_buffer, char *image_buffer, int P, int D, int M,
                                        int width, int height, const float *means3D, const float *shs,
                                        const float *colors_precomp, const float *opacities, const float *scales,
                                        float scale_modifier, const float *rotations, const float *cov3D_precomp,
                                        const float *viewmatrix, const float *projmatrix, const float *cam_pos,
                                        const float *tan_fovx, const float *tan_fovy, int prefiltered, int *radii,
                                        const char *static_blobs, int P_static, int64_t R_static_capacity,
                                        uint32_t *depth_hint, const fnx_raster_opts_t *opts, fnx_stream_t stream) {
    float *pending_zero3 = t_zero_request;
    t_zero_request = nullptr;  // taken, whatever this call returns
    Opts op;
    if (int rc = resolve_opts(opts, pending_zero3, 0xFFFFFFFFu, &op)) return rc;
    if (!channels_ok(channels)) return fail(FNX_ERR_INVALID_ARG, "channels must be 1 or 3 (got %d)", channels);
    if (P < 0 || width <= 0 || height <= 0) return fail(FNX_ERR_INVALID_ARG, "bad P/width/height");
    if (!image_buffer) return fail(FNX_ERR_INVALID_ARG, "image_buffer is NULL");
    fnx::ViewBatch vb;
    fnx::StaticRef st;
    if (int rc = make_static_ref(static_blobs, P, P_static, width, height, R_static_capacity, &st)) return rc;
    if (int rc = make_view_batch(V, P, width, height, 0, tan_fovx, tan_fovy, &vb, st.base != nullptr, P_static,
                                 R_static_capacity))
        return rc;
    if (st.base && (P == 0 || !radii)) return fail(FNX_ERR_INVALID_ARG, "static split needs P_dyn > 0 and radii [V, P_dyn + P_static]");
    if (V > 1 && P > 0 && !radii) return fail(FNX_ERR_INVALID_ARG, "radii [V,P] is required for V > 1");
    hipStream_t s = (hipStream_t)stream;
    Img img = carve_img(image_buffer, width, height);
    const int T = fnx::tiles_x(width) * fnx::tiles_y(height);
    if (T > fnx::kMaxTiles)
        return fail(FNX_ERR_UNSUPPORTED, "%d tiles > %d (image larger than 2048x2048)", T, fnx::kMaxTiles);
    if (P == 0) {  // no kernels: zero instance count / status and empty ranges
        for (int v = 0; v < V; v++) {
            (void)hipMemsetAsync((char *)img.header + v * vb.img, 0, 64, s);  // all 16 header words
            (void)hipMemsetAsync((char *)img.ranges + v * vb.img, 0, (size_t)T * 8, s);
        }
        return hip_check("stage1(P=0)");
    }
    if (!geom_buffer || !means3D || !opacities || !viewmatrix || !projmatrix)
        return fail(FNX_ERR_INVALID_ARG, "a required pointer is NULL");
    if (channels != 3 && colors_precomp == nullptr)  // rasterizer_impl.cu:226-228
        return fail(FNX_ERR_NON_RGB_NEEDS_COLORS, "For non-RGB, provide precomputed Gaussian colors!");
    if (colors_precomp == nullptr && (shs == nullptr || cam_pos == nullptr))
        return fail(FNX_ERR_INVALID_ARG, "neither colors_precomp nor shs+cam_pos given");
    if (cov3D_precomp == nullptr && (scales == nullptr || rotations == nullptr))
        return fail(FNX_ERR_INVALID_ARG, "neither cov3D_precomp nor scales+rotations given");
    Geom g = carve_geom(geom_buffer, P, width, height);
    int *rad = radii ? radii : g.radii;  // rasterizer_impl.cu:214-216
    // temporal-coherence sort: only where a tile rectangle fits its packed record (else the radix passes, state left seeded)
    char *sort_state = op.sort_state ? aligned(op.sort_state) : nullptr;
    const bool coherent = op.sort_mode == FNX_SORT_COHERENT && fnx::coherent_sort_supported(width, height);
    fnx::CohRef coh;
    memset(&coh, 0, sizeof(coh));
    if (coherent) {
        const fnx::SortStateLayout SL = fnx::sort_state_layout(P);
        coh.krec = g.krec;
        coh.state = sort_state;
        coh.stride = SL.total;
        coh.hdr = SL.hdr;
        coh.inv = SL.inv;
        coh.samples = SL.samples;
        coh.holes = SL.holes;
        coh.olist = SL.olist;
    }
    {
    ProfScope ps(3, s);
    fnx::launch_preprocess(channels, s, P, D, M, means3D, scales, scale_modifier, rotations, opacities, shs, g.clamped,
                           cov3D_precomp, colors_precomp, viewmatrix, projmatrix, cam_pos, width, height, rad,
                           g.means2D, g.depths, g.cov3D, g.rgb, g.conic_opacity, g.tiles_touched, g.sort_key0,
                           g.sort_hist + fnx::sort_scratch(P).kmin_blk, g.rect, g.blend_rec, prefiltered, V, vb, st,
                           (op.lean_geometry && V > 1 && !cov3D_precomp) ? 1 : 0, op.zero3, coh, op.blend_math);
    }
    {
    ProfScope ps(2, s);
    // (key, id) pair buffers: sort_key0|sort_key1 and sort_val0|sort_val1 are adjacent P-word arrays
    int hist_done = 0;  // the coherent sort counts the instances per (rank block, tile) on its way when the tiles fit its LDS
    fnx::launch_depth_sort(s, P, g.sort_key0, (uint2 *)g.sort_key0, (uint2 *)g.sort_val0, g.sort_hist, g.rect,
                           g.rect_sorted, V, vb, op.sort_mode == FNX_SORT_NARROW ? 1 : 0, sort_state, coherent ? 1 : 0,
                           g.krec, width, height, g.blk_hist, g.sort_hist + fnx::sort_scratch(P).blk_total, &hist_done);
    if (!hist_done)
        fnx::launch_rank_hist(s, P, width, height, g.rect_sorted, g.blk_hist, g.sort_hist + fnx::sort_scratch(P).blk_total,
                              V, vb);
    fnx::launch_tile_colscan(s, T, P, g.blk_hist, g.blk_rel, img.tile_count, V, vb);
    fnx::launch_tile_scan(s, T, img.tile_count, img.ranges, img.dyn_start, img.header, P, height, g.sort_hist, depth_hint,
                          op.deep_min, img.tile_order, img.tile_deep, V, vb, st,
                          make_seg_ref(op.segment_scratch, width, height));
    }
    return hip_check("stage1");
}

int fnx_forward_stage1_views_split(int channels, int V, char *geom_buffer, char *image_buffer, int P, int D, int M,
                                   int width, int height, const float *means3D, const float *shs,
                                   const float *colors_precomp, const float *opacities, const float *scales,
                                   float scale_modifier, const float *rotations, const float *cov3D_precomp,
                                   const float *viewmatrix, const float *projmatrix, const float *cam_pos,
                                   const float *tan_fovx, const float *tan_fovy, int prefiltered, int *radii,
                                   const char *static_blobs, int P_static, int64_t R_static_capacity,
                                   uint32_t *depth_hint, fnx_stream_t stream) {
    return fnx_forward_stage1_views_split_opts(channels, V, geom_buffer, image_buffer, P, D, M, width, height, means3D,
                                               shs, colors_precomp, opacities, scales, scale_modifier, rotations,
                                               cov3D_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy,
                                               prefiltered, radii, static_blobs, P_static, R_static_capacity,
                                               depth_hint, nullptr, stream);
}

int fnx_forward_stage1_views(int channels, int V, char *geom_buffer, char *image_buffer, int P, int D, int M, int width,
                             int height, const float *means3D, const float *shs, const float *colors_precomp,
                             const float *opacities, const float *scales, float scale_modifier, const float *rotations,
                             const float *cov3D_precomp, const float *viewmatrix, const float *projmatrix,
                             const float *cam_pos, const float *tan_fovx, const float *tan_fovy, int prefiltered,
                             int *radii, fnx_stream_t stream) {
    return fnx_forward_stage1_views_split(channels, V, geom_buffer, image_buffer, P, D, M, width, height, means3D, shs,
                                          colors_precomp, opacities, scales, scale_modifier, rotations, cov3D_precomp,
                                          viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, prefiltered, radii,
                                          nullptr, 0, 0, nullptr, stream);
}

int fnx_forward_stage1(int channels, char *geom_buffer, char *image_buffer, int P, int D, int M, int width, int height,
                       const float *means3D, const float *shs, const float *colors_precomp, const float *opacities,
                       const float *scales, float scale_modifier, const float *rotations, const float *cov3D_precomp,
                       const float *viewmatrix, const float *projmatrix, const float *cam_pos, float tan_fovx,
                       float tan_fovy, int prefiltered, int *radii, fnx_stream_t stream) {
    return fnx_forward_stage1_views(channels, 1, geom_buffer, image_buffer, P, D, M, width, height, means3D, shs,
                                    colors_precomp, opacities, scales, scale_modifier, rotations, cov3D_precomp,
                                    viewmatrix, projmatrix, cam_pos, &tan_fovx, &tan_fovy, prefiltered, radii, stream);
}

int fnx_read_num_rendered(const char *image_buffer, int width, int height, fnx_stream_t stream, int *num_rendered) {
    if (!image_buffer || !num_rendered) return fail(FNX_ERR_INVALID_ARG, "NULL argument");
    Img img = carve_img(const_cast<char *>(image_buffer), width, height);
    uint32_t v = 0;
    hipError_t e = hipMemcpyAsync(&v, img.header + fnx::HDR_NUM_RENDERED, 4, hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return fail(FNX_ERR_HIP, "read_num_rendered: %s", hipGetErrorString(e));
    *num_rendered = (int)v;
    return FNX_OK;
}

int fnx_read_status(const char *image_buffer, int width, int height, fnx_stream_t stream) {
    if (!image_buffer) return fail(FNX_ERR_INVALID_ARG, "NULL argument");
    Img img = carve_img(const_cast<char *>(image_buffer), width, height);
    uint32_t h[3] = {0, 0, 0};
    hipError_t e = hipMemcpyAsync(h, img.header, 12, hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return fail(FNX_ERR_HIP, "read_status: %s", hipGetErrorString(e));
    if (h[fnx::HDR_STATUS] == FNX_ERR_CAPACITY)
        return fail(FNX_ERR_CAPACITY, "binning capacity %u < num_rendered %u", h[fnx::HDR_CAPACITY],
                    h[fnx::HDR_NUM_RENDERED]);
    if (h[fnx::HDR_STATUS] == FNX_ERR_INVALID_ARG)
        return fail(FNX_ERR_INVALID_ARG, "a backward call asked for gradients beyond its forward's gradient limit "
                                         "(fnx_request_gradient_limit)");
    if (h[fnx::HDR_STATUS] == FNX_ERR_SORT_SPAN)
        return fail(FNX_ERR_SORT_SPAN, "the view's depth keys need the fourth sort pass (fnx_set_sort_narrow)");
    return FNX_OK;
}

int fnx_forward_stage2_views_split_opts(int channels, int V, char *geom_buffer, char *binning_buffer,
                                        int64_t binning_capacity, char *image_buffer, int P, int width, int height,
                                        const float *background, float *out_color, float *out_depth,
                                        uint32_t *status_out, const char *static_blobs, int P_static,
                                        int64_t R_static_capacity, int materialize_all, uint32_t *depth_hint,
                                        const fnx_raster_opts_t *opts, fnx_stream_t stream) {
    const uint32_t pending_limit = t_grad_limit_request;
    t_grad_limit_request = 0xFFFFFFFFu;  // taken, whatever this call returns
    Opts op;
    if (int rc = resolve_opts(opts, nullptr, pending_limit, &op)) return rc;
    if (!channels_ok(channels)) return fail(FNX_ERR_INVALID_ARG, "channels must be 1 or 3 (got %d)", channels);
    if (P == 0) return FNX_OK;  // outputs stay as the caller zero-filled them (rasterize_points.cu:81)
    if (!geom_buffer || !image_buffer || !background || !out_color || !out_depth)
        return fail(FNX_ERR_INVALID_ARG, "a required pointer is NULL");
    if (binning_capacity < 0 || binning_capacity > 0xFFFFFFFFll) return fail(FNX_ERR_INVALID_ARG, "bad capacity");
    if ((binning_capacity > 0 || static_blobs) && !binning_buffer) return fail(FNX_ERR_INVALID_ARG, "binning_buffer is NULL");
    if ((uint64_t)binning_capacity + (uint64_t)(static_blobs ? R_static_capacity : 0) > fnx::kMaxTileList)
        return fail(FNX_ERR_UNSUPPORTED, "%lld instances per view > %llu: a tile list may not exceed 2^26 entries "
                    "(backward work items hold the batch index in 18 bits)",
                    (long long)(binning_capacity + (static_blobs ? R_static_capacity : 0)), (unsigned long long)fnx::kMaxTileList);
    fnx::ViewBatch vb;
    fnx::StaticRef st;
    if (int rc = make_static_ref(static_blobs, P, P_static, width, height, R_static_capacity, &st)) return rc;
    const float unused[fnx::kMaxViews] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
    if (int rc = make_view_batch(V, P, width, height, binning_capacity, unused, unused, &vb, st.base != nullptr, P_static,
                                 R_static_capacity, op.dual != nullptr))
        return rc;
    fnx::DualRef du;
    if (int rc = make_dual_ref(op.dual, channels, st.base != nullptr, width, height, binning_capacity, R_static_capacity,
                               false, &du))
        return rc;
    if (du.img1 && op.grad_limit != 0xFFFFFFFFu && op.grad_limit != (uint32_t)P)
        return fail(FNX_ERR_INVALID_ARG, "fnx_raster_dual_t: grad_splat_limit must be P_dyn (the second image is the per-call splats')");
    hipStream_t s = (hipStream_t)stream;
    Geom g = carve_geom(geom_buffer, P, width, height);
    Img img = carve_img(image_buffer, width, height);
    Bin bin = carve_bin(binning_buffer, binning_capacity);  // point_list sits at offset 0 in both layouts
    const uint32_t cap = (uint32_t)binning_capacity;
    {
    ProfScope ps(4, s);
    const fnx::SortScratch L = fnx::sort_scratch(P);
    fnx::launch_emit(s, P, width, height, (const uint2 *)g.sort_val0, (const uint2 *)g.sort_key0, g.sort_hist + L.ctl,
                     g.rect_sorted, img.dyn_start, g.blk_rel, g.sort_hist + L.emit_ctl, g.sort_hist + L.emit_items,
                     bin.point_list, img.header, cap, st.base ? 1 : 0, V, vb);
    }
    fnx::InvUpdate iu;
    memset(&iu, 0, sizeof(iu));
    if (op.sort_mode == FNX_SORT_COHERENT && op.sort_state && fnx::coherent_sort_supported(width, height)) {
        const fnx::SortStateLayout SL = fnx::sort_state_layout(P);
        iu.pairs = (const uint2 *)g.sort_val0;  // where the coherent sort leaves its (depth bits, id) pairs
        iu.state = aligned(op.sort_state);
        iu.stride = SL.total;
        iu.inv = SL.inv;
        iu.P = P;
    }
    {
        ProfScope ps(channels == 3 ? 0 : 5, s);
        fnx::launch_blend_forward(channels, s, width, height, img.ranges, bin.point_list, g.blend_rec, background,
                                  img.final_T, img.n_contrib, out_color, out_depth, img.header, cap, status_out,
                                  img.tile_count, img.dyn_start, img.acc_final, img.tile_order, img.tile_deep, depth_hint,
                                  st, materialize_all, V, vb, op.blend_math, op.deep_kernel, op.grad_limit, iu, du,
                                  make_seg_ref(op.segment_scratch, width, height));
    }
    return hip_check("stage2");
}

int fnx_forward_stage2_views_split(int channels, int V, char *geom_buffer, char *binning_buffer,
                                   int64_t binning_capacity, char *image_buffer, int P, int width, int height,
                                   const float *background, float *out_color, float *out_depth, uint32_t *status_out,
                                   const char *static_blobs, int P_static, int64_t R_static_capacity,
                                   int materialize_all, uint32_t *depth_hint, fnx_stream_t stream) {
    return fnx_forward_stage2_views_split_opts(channels, V, geom_buffer, binning_buffer, binning_capacity, image_buffer, P,
                                               width, height, background, out_color, out_depth, status_out, static_blobs,
                                               P_static, R_static_capacity, materialize_all, depth_hint, nullptr, stream);
}

int fnx_forward_stage2_views_status(int channels, int V, char *geom_buffer, char *binning_buffer,
                                    int64_t binning_capacity, char *image_buffer, int P, int width, int height,
                                    const float *background, const int *radii, float *out_color, float *out_depth,
                                    uint32_t *status_out, fnx_stream_t stream) {
    if (V > 1 && P > 0 && !radii) return fail(FNX_ERR_INVALID_ARG, "radii [V,P] is required for V > 1");
    return fnx_forward_stage2_views_split(channels, V, geom_buffer, binning_buffer, binning_capacity, image_buffer, P,
                                          width, height, background, out_color, out_depth, status_out, nullptr, 0, 0, 0,
                                          nullptr, stream);
}

// Once per frame: instances of the static subset (already through fnx_forward_stage1_views as a splat set of its own)
// -> the views' static blobs.
int fnx_static_finalize_views(int V, char *geom_buffer, char *binning_scratch, char *image_buffer, int P_static,
                              int width, int height, int id_offset, int64_t R_static_capacity, const int *radii,
                              char *static_blobs, fnx_stream_t stream) {
    if (P_static <= 0 || width <= 0 || height <= 0 || id_offset < 0)
        return fail(FNX_ERR_INVALID_ARG, "bad P_static/width/height/id_offset");
    if (!geom_buffer || !image_buffer || !static_blobs || !radii) return fail(FNX_ERR_INVALID_ARG, "a required pointer is NULL");
    if (R_static_capacity < 0 || R_static_capacity > 0xFFFFFFFFll) return fail(FNX_ERR_INVALID_ARG, "bad capacity");
    if (R_static_capacity > 0 && !binning_scratch) return fail(FNX_ERR_INVALID_ARG, "binning_scratch is NULL");
    fnx::ViewBatch vb;
    const float unused[fnx::kMaxViews] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
    if (int rc = make_view_batch(V, P_static, width, height, R_static_capacity, unused, unused, &vb)) return rc;
    hipStream_t s = (hipStream_t)stream;
    Geom g = carve_geom(geom_buffer, P_static, width, height);
    Img img = carve_img(image_buffer, width, height);
    Bin bin = carve_bin(binning_scratch, R_static_capacity);
    const uint32_t cap = (uint32_t)R_static_capacity;
    const fnx::SortScratch L = fnx::sort_scratch(P_static);
    fnx::launch_emit(s, P_static, width, height, (const uint2 *)g.sort_val0, (const uint2 *)g.sort_key0, g.sort_hist + L.ctl,
                     g.rect_sorted, img.dyn_start, g.blk_rel, g.sort_hist + L.emit_ctl, g.sort_hist + L.emit_items,
                     bin.point_list, img.header, cap, 0, V, vb);
    fnx_static_layout_t SL;
    fnx::static_layout(P_static, width, height, R_static_capacity, &SL);
    fnx::launch_static_pack(s, P_static, width, height, bin.point_list, img.dyn_start, img.header, radii, g.blend_rec,
                            aligned(static_blobs), SL.total, SL, (uint32_t)id_offset, cap, V, vb);
    return hip_check("static_finalize");
}

int fnx_forward_stage2_views(int channels, int V, char *geom_buffer, char *binning_buffer, int64_t binning_capacity,
                             char *image_buffer, int P, int width, int height, const float *background,
                             const int *radii, float *out_color, float *out_depth, fnx_stream_t stream) {
    return fnx_forward_stage2_views_status(channels, V, geom_buffer, binning_buffer, binning_capacity, image_buffer, P,
                                           width, height, background, radii, out_color, out_depth, nullptr, stream);
}

int fnx_forward_stage2(int channels, char *geom_buffer, char *binning_buffer, int64_t binning_capacity,
                       char *image_buffer, int P, int width, int height, const float *background,
                       const float *colors_precomp, const int *radii, float *out_color, float *out_depth,
                       fnx_stream_t stream) {
    (void)colors_precomp;  // colours were packed into the blend records by stage 1 (rasterizer_impl.cu:299)
    return fnx_forward_stage2_views(channels, 1, geom_buffer, binning_buffer, binning_capacity, image_buffer, P, width,
                                    height, background, radii, out_color, out_depth, stream);
}

int fnx_rasterize_forward(int channels, fnx_alloc_fn geometryBuffer, void *geom_user, fnx_alloc_fn binningBuffer,
                          void *binning_user, fnx_alloc_fn imageBuffer, void *image_user, int P, int D, int M,
                          const float *background, int width, int height, const float *means3D, const float *shs,
                          const float *colors_precomp, const float *opacities, const float *scales,
                          float scale_modifier, const float *rotations, const float *cov3D_precomp,
                          const float *viewmatrix, const float *projmatrix, const float *cam_pos, float tan_fovx,
                          float tan_fovy, int prefiltered, float *out_color, float *out_depth, int *radii,
                          fnx_stream_t stream, int *num_rendered) {
    if (!geometryBuffer || !binningBuffer || !imageBuffer) return fail(FNX_ERR_INVALID_ARG, "NULL allocator");
    if (num_rendered) *num_rendered = 0;
    char *geom = geometryBuffer(fnx_geom_bytes(P, width, height), geom_user);
    char *img = imageBuffer(fnx_image_bytes(width, height), image_user);
    int rc = fnx_forward_stage1(channels, geom, img, P, D, M, width, height, means3D, shs, colors_precomp, opacities,
                                scales, scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix, cam_pos,
                                tan_fovx, tan_fovy, prefiltered, radii, stream);
    if (rc != FNX_OK) return rc;
    int R = 0;
    rc = fnx_read_num_rendered(img, width, height, stream, &R);  // the reference's blocking D2H copy
    if (rc != FNX_OK) return rc;
    if (num_rendered) *num_rendered = R;
    char *bin = binningBuffer(fnx_binning_bytes(R), binning_user);
    return fnx_forward_stage2(channels, geom, bin, R, img, P, width, height, background, colors_precomp, radii,
                              out_color, out_depth, stream);
}

int fnx_rasterize_backward_views_split_opts(int channels, int V, int P, int D, int M, const float *background,
                                            int width, int height, const float *means3D, const float *shs,
                                            const float *colors_precomp, const float *scales, float scale_modifier,
                                            const float *rotations, const float *cov3D_precomp, const float *viewmatrix,
                                            const float *projmatrix, const float *campos, const float *tan_fovx,
                                            const float *tan_fovy, const int *radii, char *geom_buffer,
                                            char *binning_buffer, int64_t binning_capacity, char *image_buffer,
                                            const float *dL_dpix, float *dL_dmean2D, float *dL_dconic,
                                            float *dL_dopacity_views, float *dL_dcolor_views, float *dL_dopacity,
                                            float *dL_dcolor, float *dL_dmean3D, float *dL_dcov3D, float *dL_dsh,
                                            float *dL_dscale, float *dL_drot, int grad_splat_limit, int geometry_only,
                                            const char *static_blobs, int P_static, int64_t R_static_capacity,
                                            uint32_t *status_out, const fnx_raster_opts_t *opts, fnx_stream_t stream) {
    Opts op;
    if (int rc = resolve_opts(opts, nullptr, 0xFFFFFFFFu, &op)) return rc;
    if (!channels_ok(channels)) return fail(FNX_ERR_INVALID_ARG, "channels must be 1 or 3 (got %d)", channels);
    if (P == 0) return FNX_OK;  // rasterize_points.cu:160
    if (geometry_only < 0 || geometry_only > 3) return fail(FNX_ERR_INVALID_ARG, "geometry_only must be 0, 1, 2 or 3");
    const bool positions_only = geometry_only == 3;  // only dL_dmean3D is produced (and must come in zeroed)
    if (!geom_buffer || !image_buffer || !background || !means3D || !viewmatrix || !projmatrix || !dL_dpix || !dL_dmean3D)
        return fail(FNX_ERR_INVALID_ARG, "a required pointer is NULL");
    if (!positions_only && (!dL_dmean2D || !dL_dconic || !dL_dopacity_views || !dL_dcolor_views || !dL_dopacity || !dL_dcov3D))
        return fail(FNX_ERR_INVALID_ARG, "a required pointer is NULL");
    if (shs && !positions_only && (!dL_dsh || !campos)) return fail(FNX_ERR_INVALID_ARG, "shs given but dL_dsh/campos NULL");
    if (scales && !positions_only && (!rotations || !dL_dscale || !dL_drot))
        return fail(FNX_ERR_INVALID_ARG, "scales given but rotations/dL_dscale/dL_drot NULL");
    if ((geometry_only == 1 || positions_only) && shs)
        return fail(FNX_ERR_INVALID_ARG, "geometry_only = 1 / 3 cannot be combined with SH colours");
    if (binning_capacity < 0 || binning_capacity > 0xFFFFFFFFll) return fail(FNX_ERR_INVALID_ARG, "bad capacity");
    fnx::ViewBatch vb;
    fnx::StaticRef st;
    if (int rc = make_static_ref(static_blobs, P, P_static, width, height, R_static_capacity, &st)) return rc;
    if (int rc = make_view_batch(V, P, width, height, binning_capacity, tan_fovx, tan_fovy, &vb, st.base != nullptr,
                                 P_static, R_static_capacity, op.dual != nullptr))
        return rc;
    fnx::DualRef du;
    if (int rc = make_dual_ref(op.dual, channels, st.base != nullptr, width, height, binning_capacity, R_static_capacity,
                               true, &du))
        return rc;
    if (du.img1 && !positions_only)
        return fail(FNX_ERR_INVALID_ARG, "fnx_raster_dual_t: the backward supports geometry_only = 3 (positions only)");
    if ((V > 1 || st.base) && !radii) return fail(FNX_ERR_INVALID_ARG, "radii [V,P] is required for V > 1");
    // static splats take no gradients: the limit stays within the per-call splats; the gradient arrays span all splats
    const int limit = (grad_splat_limit < 0 || grad_splat_limit > P) ? P : grad_splat_limit;
    const int P_all = P + (st.base ? P_static : 0);
    hipStream_t s = (hipStream_t)stream;
    Geom g = carve_geom(geom_buffer, P, width, height);
    Img img = carve_img(image_buffer, width, height);
    // point_list sits at offset 0 of each view's binning blob
    Bin bin = carve_bin(binning_buffer, 0);
    const int *rad = radii ? radii : g.radii;
    const float *cov3D_ptr = cov3D_precomp ? cov3D_precomp : g.cov3D;      // rasterizer_impl.cu:390
    // lean geometry (fnx_set_lean_geometry; must match the forward's setting): view 0's blob holds the one covariance array
    const size_t cov3D_stride = (cov3D_precomp || (op.lean_geometry && V > 1)) ? 0 : vb.geom;
    {
        ProfScope ps(channels == 3 ? 1 : 6, s);
        fnx::launch_blend_backward(channels, geometry_only, s, P_all, width, height, img.ranges, bin.point_list,
                                   background, g.blend_rec, img.final_T, img.n_contrib, img.acc_final, dL_dpix,
                                   dL_dmean2D, dL_dconic, dL_dopacity_views, dL_dcolor_views, img.header,
                                   (uint32_t)binning_capacity,
                                   (uint32_t)limit, V, vb, st, means3D, cov3D_ptr, cov3D_stride, viewmatrix, projmatrix,
                                   dL_dmean3D, op.blend_math, status_out, du);
    }
    if (positions_only) return hip_check("backward");  // the blend backward's flush went through the geometry itself
    const int sum_appearance = (V > 1 && geometry_only != 1) ? 1 : 0;
    fnx::launch_geom_backward(channels, s, P_all, D, M, means3D, rad, shs, g.clamped, scales, rotations, scale_modifier,
                              cov3D_ptr, cov3D_stride, viewmatrix, projmatrix, campos, dL_dmean2D, dL_dconic,
                              dL_dopacity_views, dL_dcolor_views, dL_dopacity, shs ? nullptr : dL_dcolor, dL_dmean3D,
                              dL_dcov3D, dL_dsh, dL_dscale, dL_drot, limit, V, sum_appearance, vb);
    return hip_check("backward");
}

int fnx_rasterize_backward_views_split(int channels, int V, int P, int D, int M, const float *background, int width,
                                       int height, const float *means3D, const float *shs,
                                       const float *colors_precomp, const float *scales, float scale_modifier,
                                       const float *rotations, const float *cov3D_precomp, const float *viewmatrix,
                                       const float *projmatrix, const float *campos, const float *tan_fovx,
                                       const float *tan_fovy, const int *radii, char *geom_buffer,
                                       char *binning_buffer, int64_t binning_capacity, char *image_buffer,
                                       const float *dL_dpix, float *dL_dmean2D, float *dL_dconic,
                                       float *dL_dopacity_views, float *dL_dcolor_views, float *dL_dopacity,
                                       float *dL_dcolor, float *dL_dmean3D, float *dL_dcov3D, float *dL_dsh,
                                       float *dL_dscale, float *dL_drot, int grad_splat_limit, int geometry_only,
                                       const char *static_blobs, int P_static, int64_t R_static_capacity,
                                       fnx_stream_t stream) {
    return fnx_rasterize_backward_views_split_opts(
        channels, V, P, D, M, background, width, height, means3D, shs, colors_precomp, scales, scale_modifier, rotations,
        cov3D_precomp, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, radii, geom_buffer, binning_buffer,
        binning_capacity, image_buffer, dL_dpix, dL_dmean2D, dL_dconic, dL_dopacity_views, dL_dcolor_views, dL_dopacity,
        dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, grad_splat_limit, geometry_only, static_blobs,
        P_static, R_static_capacity, nullptr, nullptr, stream);
}

int fnx_rasterize_backward_views(int channels, int V, int P, int D, int M, const float *background, int width,
                                 int height, const float *means3D, const float *shs, const float *colors_precomp,
                                 const float *scales, float scale_modifier, const float *rotations,
                                 const float *cov3D_precomp, const float *viewmatrix, const float *projmatrix,
                                 const float *campos, const float *tan_fovx, const float *tan_fovy, const int *radii,
                                 char *geom_buffer, char *binning_buffer, int64_t binning_capacity, char *image_buffer,
                                 const float *dL_dpix, float *dL_dmean2D, float *dL_dconic, float *dL_dopacity_views,
                                 float *dL_dcolor_views, float *dL_dopacity, float *dL_dcolor, float *dL_dmean3D,
                                 float *dL_dcov3D, float *dL_dsh, float *dL_dscale, float *dL_drot,
                                 int grad_splat_limit, int geometry_only, fnx_stream_t stream) {
    return fnx_rasterize_backward_views_split(channels, V, P, D, M, background, width, height, means3D, shs,
                                              colors_precomp, scales, scale_modifier, rotations, cov3D_precomp,
                                              viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, radii, geom_buffer,
                                              binning_buffer, binning_capacity, image_buffer, dL_dpix, dL_dmean2D,
                                              dL_dconic, dL_dopacity_views, dL_dcolor_views, dL_dopacity, dL_dcolor,
                                              dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, grad_splat_limit,
                                              geometry_only, nullptr, 0, 0, stream);
}

int fnx_rasterize_backward_ex(int channels, int P, int D, int M, int R, const float *background, int width, int height,
                           const float *means3D, const float *shs, const float *colors_precomp, const float *scales,
                           float scale_modifier, const float *rotations, const float *cov3D_precomp,
                           const float *viewmatrix, const float *projmatrix, const float *campos, float tan_fovx,
                           float tan_fovy, const int *radii, char *geom_buffer, char *binning_buffer,
                           char *image_buffer, const float *dL_dpix, float *dL_dmean2D, float *dL_dconic,
                           float *dL_dopacity, float *dL_dcolor, float *dL_dmean3D, float *dL_dcov3D, float *dL_dsh,
                           float *dL_dscale, float *dL_drot, int grad_splat_limit, int geometry_only,
                              fnx_stream_t stream) {
    // R = the binning capacity the forward ran with (num_rendered in the callback form): the binning blob's layout
    // depends on it
    if (P != 0 && !dL_dcolor) return fail(FNX_ERR_INVALID_ARG, "a required pointer is NULL");
    if (R < 0) return fail(FNX_ERR_INVALID_ARG, "R must be the forward's binning capacity");
    return fnx_rasterize_backward_views(channels, 1, P, D, M, background, width, height, means3D, shs, colors_precomp,
                                        scales, scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix,
                                        campos, &tan_fovx, &tan_fovy, radii, geom_buffer, binning_buffer, R,
                                        image_buffer, dL_dpix, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor,
                                        dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot,
                                        grad_splat_limit, geometry_only, stream);
}

int fnx_rasterize_backward(int channels, int P, int D, int M, int R, const float *background, int width, int height,
                           const float *means3D, const float *shs, const float *colors_precomp, const float *scales,
                           float scale_modifier, const float *rotations, const float *cov3D_precomp,
                           const float *viewmatrix, const float *projmatrix, const float *campos, float tan_fovx,
                           float tan_fovy, const int *radii, char *geom_buffer, char *binning_buffer,
                           char *image_buffer, const float *dL_dpix, float *dL_dmean2D, float *dL_dconic,
                           float *dL_dopacity, float *dL_dcolor, float *dL_dmean3D, float *dL_dcov3D, float *dL_dsh,
                           float *dL_dscale, float *dL_drot, fnx_stream_t stream) {
    return fnx_rasterize_backward_ex(channels, P, D, M, R, background, width, height, means3D, shs, colors_precomp,
                                     scales, scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix, campos,
                                     tan_fovx, tan_fovy, radii, geom_buffer, binning_buffer, image_buffer, dL_dpix,
                                     dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh,
                                     dL_dscale, dL_drot, -1, 0, stream);
}

int fnx_set_deep_threshold(unsigned int min_depth) {
    g_deep_min = min_depth ? min_depth : 1u;
    return FNX_OK;
}

int fnx_set_blend_math(int mode) {
    if (mode != 0 && mode != 1) return fail(FNX_ERR_INVALID_ARG, "blend math mode must be 0 (exact) or 1 (fast)");
    g_blend_math = mode;
    return FNX_OK;
}
int fnx_get_blend_math(void) { return g_blend_math; }
int fnx_request_zero3(float *rows3) {
    t_zero_request = rows3;
    return FNX_OK;
}
int fnx_request_gradient_limit(int grad_splat_limit) {
    t_grad_limit_request = grad_splat_limit < 0 ? 0xFFFFFFFFu : (uint32_t)grad_splat_limit;
    return FNX_OK;
}
size_t fnx_sort_state_bytes(int P) { return fnx::sort_state_layout(P).total; }
size_t fnx_segment_scratch_bytes(int width, int height) {
    return fnx::seg_layout(fnx::tiles_x(width) * fnx::tiles_y(height)).total;
}
int fnx_segment_scratch_read(const char *scratch, int width, int height, int view, fnx_stream_t stream, uint32_t out[16]) {
    if (!scratch || !out || view < 0) return fail(FNX_ERR_INVALID_ARG, "bad argument");
    const fnx::SegLayout L = fnx::seg_layout(fnx::tiles_x(width) * fnx::tiles_y(height));
    const char *src = aligned(const_cast<char *>(scratch)) + L.total * (size_t)view + L.ctl;
    hipError_t e = hipMemcpyAsync(out, src, 16 * sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return fail(FNX_ERR_HIP, "segment_scratch_read: %s", hipGetErrorString(e));
    return FNX_OK;
}
int fnx_sort_state_read(const char *sort_state, int P, int view, fnx_stream_t stream, uint32_t out[3]) {
    if (!sort_state || !out || P < 0 || view < 0) return fail(FNX_ERR_INVALID_ARG, "bad argument");
    const fnx::SortStateLayout SL = fnx::sort_state_layout(P);
    uint32_t h[fnx::COH_HDR_WORDS];
    const char *src = aligned(sort_state) + SL.total * (size_t)view + SL.hdr;
    hipError_t e = hipMemcpyAsync(h, src, sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return fail(FNX_ERR_HIP, "sort_state_read: %s", hipGetErrorString(e));
    out[0] = h[fnx::COH_REPAIRS];
    out[1] = h[fnx::COH_FALLBACKS];
    out[2] = h[fnx::COH_WHY];
    return FNX_OK;
}
int fnx_sort_state_outliers(const char *sort_state, int P, int view, fnx_stream_t stream, uint32_t *out) {
    if (!sort_state || !out || P < 0 || view < 0) return fail(FNX_ERR_INVALID_ARG, "bad argument");
    const fnx::SortStateLayout SL = fnx::sort_state_layout(P);
    const char *src = aligned(sort_state) + SL.total * (size_t)view + SL.hdr + 4 * fnx::COH_OUTLIERS;
    hipError_t e = hipMemcpyAsync(out, src, 4, hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return fail(FNX_ERR_HIP, "sort_state_outliers: %s", hipGetErrorString(e));
    return FNX_OK;
}
int fnx_set_sort_narrow(int on) {
    g_sort_narrow = on ? 1 : 0;
    return FNX_OK;
}
int fnx_set_lean_geometry(int on) {
    g_lean_geometry = on ? 1 : 0;
    return FNX_OK;
}
int fnx_set_deep_kernel(int mode) {
    if (mode < 0 || mode > 5)
        return fail(FNX_ERR_INVALID_ARG, "deep kernel mode must be 0 (per-tile kernel only), 1 / 2 (lab super-batch kernel: auto / always), "
                                         "3 / 4 (staging waves: deep tiles / all tiles) or 5 (staging waves by the launch's view count, the default)");
    g_deep_kernel = mode;
    return FNX_OK;
}

int fnx_set_backward_form(int form) {
    if (form != 0 && form != 1) return fail(FNX_ERR_INVALID_ARG, "backward form must be 0 (pixels as lanes) or 1 (entries as lanes)");
    fnx::set_backward_form(form);
    return FNX_OK;
}
int fnx_get_backward_form(void) { return fnx::get_backward_form(); }

int fnx_profile_enable(int on) {
    g_prof_on = on != 0;
    for (auto &c : g_prof) c.used = 0;
    return FNX_OK;
}

int fnx_profile_read(int which, double *total_ms, int *launches) {
    if (which < 0 || which >= kProfClasses || !total_ms || !launches) return fail(FNX_ERR_INVALID_ARG, "bad argument");
    ProfClass &c = g_prof[which];
    double tot = 0.0;
    for (size_t i = 0; i < c.used; i++) {
        float ms = 0.f;
        hipError_t e = hipEventSynchronize(c.pool[i].second);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, c.pool[i].first, c.pool[i].second);
        if (e != hipSuccess) return fail(FNX_ERR_HIP, "profile_read: %s", hipGetErrorString(e));
        tot += ms;
    }
    *total_ms = tot;
    *launches = (int)c.used;
    return FNX_OK;
}

int fnx_mark_visible(int P, const float *means3D, const float *viewmatrix, const float *projmatrix, uint8_t *present,
                     fnx_stream_t stream) {
    (void)projmatrix;
    if (P == 0) return FNX_OK;
    if (P < 0 || !means3D || !viewmatrix || !present) return fail(FNX_ERR_INVALID_ARG, "bad argument");
    fnx::launch_mark_visible((hipStream_t)stream, P, means3D, viewmatrix, present);
    return hip_check("mark_visible");
}

}  // extern "C"
