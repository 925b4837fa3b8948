// Host+device view of the three opaque scratch blobs (layout is this library's own; the
// reference's GeometryState/ImageState/BinningState live in ch3 rasterizer_impl.h:27-64).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <hip/hip_runtime.h>

#include "../../include/fnx_raster.h"

#include <mutex>

namespace fnx {

// Host-side launch parameters that depend on the device (compute units, resident workgroups of a kernel variant):
// looked up once per (device, slot) under a mutex -- a process may drive several devices, from several threads.
constexpr int kMaxDevices = 64;
template <typename F>
inline int per_device_cached(int (&cache)[kMaxDevices], std::mutex &mu, F &&query) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= kMaxDevices) return query(dev);
    std::lock_guard<std::mutex> lock(mu);
    if (cache[dev] == 0) cache[dev] = query(dev);
    return cache[dev];
}
inline int device_cu_count() {
    static int cache[kMaxDevices];
    static std::mutex mu;
    return per_device_cached(cache, mu, [](int dev) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    });
}

constexpr size_t kAlign = 256;
inline size_t align_up(size_t x) { return (x + kAlign - 1) / kAlign * kAlign; }

inline int tiles_x(int W) { return (W + 15) / 16; }
inline int tiles_y(int H) { return (H + 15) / 16; }

// Instances are counted and emitted per block of kSplatBlock consecutive depth ranks; the depth
// sort works on chunks of kSortChunk keys per workgroup.
constexpr int kSplatBlock = 1024;
constexpr int kSortChunk = 1024;
constexpr int kMaxTiles = 16384;  // LDS tile histogram: 64 KiB; also the tile field of a backward work item (kItemTileBits)
// The blend kernels walk a tile's list in batches of kBlendBatch entries.  The forward leaves, for every batch b >= 1 it
// blends, the per-pixel state in front of the batch (transmittance + accumulated colour, 16 bytes per pixel) in slot
// (first list position of the tile) / kBlendBatch + b - 1 -- unique, because a tile owns ceil(len / 256) - 1 <=
// floor(len / 256) such slots -- and one work item (tile | batch << 14) per batch that holds a contributor, so the
// backward can run the batches of a tile as independent workgroups.
constexpr int kBlendBatch = 256;
// Backward work item = tile | batch << kItemTileBits: kMaxTiles tiles, 2^(32 - kItemTileBits) batches of kBlendBatch
// entries per tile, i.e. a tile list of at most kMaxTileList entries.  A view's lists cannot be longer than its instance
// capacity, so the forward rejects capacity + static instances above kMaxTileList (FNX_ERR_UNSUPPORTED).
constexpr int kItemTileBits = 14;
constexpr uint32_t kItemTileMask = (1u << kItemTileBits) - 1u;
constexpr uint64_t kMaxTileList = (uint64_t)kBlendBatch << (32 - kItemTileBits);  // 2^26 entries
static_assert(kMaxTiles <= (1 << kItemTileBits), "a backward work item holds the tile in kItemTileBits bits");
constexpr size_t kBlendStateBytes = 256 * 16;
inline size_t blend_state_slots(size_t list_capacity) { return list_capacity / kBlendBatch + 2; }
// Depth sort digits: 9 bits.  Keys are sorted relative to the smallest visible key, so three passes order any
// view whose depths span less than 2^27 ulps (a far/near ratio of ~2^16); the fourth pass runs only beyond that.
constexpr int kSortBits = 9;
constexpr int kSortRadix = 1 << kSortBits;
constexpr int kKeyBlock = 256;  // splats per preprocess workgroup = per entry of the block-minimum array
// u32 words inside the geometry blob's sort_hist region
struct SortScratch {
    size_t hist, hist_rel, totals, ctl, kmin_blk, kmax_blk, blk_total, emit_ctl, emit_items, words;
};
// smallest visible key; 1 if the fourth pass is needed (and will run); bit length of the view's key span; 1 if the span
// needs a pass that was not launched (fnx_set_sort_narrow)
enum { SORT_CTL_KMIN = 0, SORT_CTL_WIDE = 1, SORT_CTL_SPAN = 2, SORT_CTL_OVERFLOW = 3 };
// Emission work items: a rank block with many instances (the nearest, largest splats) is split into up to
// kEmitBands items, each a band of tile rows (a rectangle clipped to a band of rows is still a rectangle).
#ifndef FNX_EMIT_BANDS
#define FNX_EMIT_BANDS 8
#endif
#ifndef FNX_EMIT_BAND_TARGET
#define FNX_EMIT_BAND_TARGET 8192
#endif
constexpr int kEmitBands = FNX_EMIT_BANDS;
constexpr uint32_t kEmitBandTarget = FNX_EMIT_BAND_TARGET;  // instances per item aimed at
enum { EMIT_CTL_ITEMS = 0, EMIT_CTL_TICKET = 1 };  // items of this view; next ticket (view 0's word, all views)
__host__ __device__ inline uint32_t emit_item_pack(uint32_t blk, uint32_t band, uint32_t nbands) {
    return (blk << 8) | (band << 4) | nbands;  // nbands <= 8, band < 8, blk < 2^24
}
inline SortScratch sort_scratch(int P) {
    const size_t p = (size_t)(P > 0 ? P : 0), nsb = (p + kSortChunk - 1) / kSortChunk;
    SortScratch o;
    o.hist = 0;
    o.hist_rel = o.hist + nsb * kSortRadix;
    o.totals = o.hist_rel + nsb * kSortRadix;
    o.ctl = o.totals + kSortRadix;
    o.kmin_blk = o.ctl + 16;
    o.kmax_blk = o.kmin_blk + (p + kKeyBlock - 1) / kKeyBlock;
    o.blk_total = o.kmax_blk + nsb;  // instances per rank block (rank_hist -> emit)
    o.emit_ctl = o.blk_total + (p + kSplatBlock - 1) / kSplatBlock;
    o.emit_items = o.emit_ctl + 4;
    o.words = o.emit_items + (p + kSplatBlock - 1) / kSplatBlock * kEmitBands;
    return o;
}
inline int splat_blocks(int P) { return (P + kSplatBlock - 1) / kSplatBlock; }

// Temporal-coherence depth sort (fnx_raster_opts_t.sort_mode = FNX_SORT_COHERENT, raster_binning.hip): the caller's
// persistent per-view state.  Header words, the (key, id) bounds every repair workgroup publishes for its chunk of ranks
// (first | last, u64 each), inv[id] = depth rank of splat id in the previous call's order, and what the OUTLIERS need
// (splats that travel further than a repair window reaches, a handful per call: fringe particles whose interpolated
// velocity is noise): samples[j] = the previous order's depth bits at rank kCohSampleStep * j, by which the preprocess
// tells that a splat has left its neighbourhood; such a splat's record goes to olist[] instead of the slot of its
// previous rank (which gets a hole record), and holes[b] counts the holes in ranks [1024 b, 1024 (b + 1)).
enum { COH_MAGIC = 0, COH_EPOCH = 1, COH_ARRIVED = 2, COH_FAIL = 3, COH_FALLBACKS = 4, COH_REPAIRS = 5,
       COH_WHY = 6,  // sticky: why calls fell back (1 record not of this call, 2 bucket overflow, 4 chunk not increasing, 8 chunk boundary, 16 unseeded)
       COH_NOUT = 7,        // outliers appended by this call's preprocess (reset by the repair kernel)
       COH_SAMPLES_OK = 8,  // 1: samples[] describe the order inv[] refers to (a repair call wrote both)
       COH_OUTLIERS = 9,    // running total of outliers taken (statistics)
       COH_HDR_WORDS = 64 };
constexpr int kCohOutlierCap = 256;          // per view and call; further candidates stay in the slots of their previous ranks
constexpr int kCohSampleStep = 128;          // ranks between two samples
constexpr int kCohSampleReach = 3;           // a splat stays where it is while its key lies within [sample(jb - 3), sample(jb + 4)], jb = rank / 128: at most 511 ranks
constexpr uint32_t kCohHoleId = 0xFFFFFFFEu;  // id of a hole record
struct SortStateLayout {
    size_t hdr, bounds, inv, samples, holes, olist, total;
};
inline SortStateLayout sort_state_layout(int P) {
    const size_t p = (size_t)(P > 0 ? P : 0), nc = (p + kSplatBlock - 1) / kSplatBlock;
    SortStateLayout o;
    size_t off = 0;
    o.hdr = off;     off = align_up(off + COH_HDR_WORDS * 4);
    o.bounds = off;  off = align_up(off + nc * 16);
    o.inv = off;     off = align_up(off + p * 4);
    o.samples = off; off = align_up(off + (p / kCohSampleStep + 2) * 4);
    o.holes = off;   off = align_up(off + (p / 1024 + 2) * 4);
    o.olist = off;   off = align_up(off + (size_t)kCohOutlierCap * 16);
    o.total = off + kAlign;
    return o;
}
// What the preprocess needs to leave its records in the previous depth order (state == nullptr: not in this mode)
struct CohRef {
    uint4 *krec;    // view 0's record array inside the geometry blob (stride: ViewBatch::geom)
    char *state;    // aligned start of view 0's sort state
    size_t stride;  // bytes between consecutive views' states
    size_t hdr, inv, samples, holes, olist;
};
// The blend forward refreshes inv[id] = rank from the sorted pairs on its way (pairs == nullptr: nothing to do): one
// million scattered 4-byte stores disappear under a throughput-bound kernel instead of extending the sort's launch.
struct InvUpdate {
    const uint2 *pairs;  // view 0's (depth bits, id) pairs in rank order (stride: ViewBatch::geom)
    char *state;         // aligned start of view 0's sort state
    size_t stride, inv;
    int P;
};
// the temporal-coherence records pack a tile rectangle into four bytes
inline bool coherent_sort_supported(int W, int H) { return tiles_x(W) <= 255 && tiles_y(H) <= 255; }
// Stamp of the records one call's preprocess writes for one view's sort state: the state's call counter mixed with a
// per-state nonce (its address), so that a stale record left in the recycled geometry blob by ANOTHER state's call -- whose
// counter may hold the same value -- cannot pass for one of this call's.
__host__ __device__ inline uint32_t coh_stamp(const uint32_t *hdr) {
    return hdr[COH_EPOCH] * 2654435761u + (uint32_t)((uintptr_t)hdr >> 8) * 0x9E3779B1u;
}
__host__ __device__ inline uint32_t coh_magic(int P) { return (0xC0DE0000u ^ ((uint32_t)P * 2654435761u)) | 1u; }
inline int sort_blocks(int P) { return (P + kSortChunk - 1) / kSortChunk; }

inline void geom_layout(int P, int W, int H, fnx_geom_layout_t *o) {
    size_t off = 0;
    size_t p = (size_t)(P > 0 ? P : 0);
    size_t t = (size_t)tiles_x(W) * tiles_y(H);
    size_t nb = (size_t)splat_blocks(P > 0 ? P : 0);
    o->depths = off;        off = align_up(off + p * 4);
    o->clamped = off;       off = align_up(off + p * 3);
    o->radii = off;         off = align_up(off + p * 4);
    o->means2D = off;       off = align_up(off + p * 8);
    o->cov3D = off;         off = align_up(off + p * 24);
    o->conic_opacity = off; off = align_up(off + p * 16);
    o->rgb = off;           off = align_up(off + p * 12);
    o->tiles_touched = off; off = align_up(off + p * 4);
    // sort_key0|sort_key1 and sort_val0|sort_val1 are used as two P-entry (key, id) pair buffers: keep each
    // couple adjacent (the raw depth keys of the preprocess occupy the first P words of the first couple)
    o->sort_key0 = off;     off = align_up(off + p * 4);
    o->sort_key1 = off;     off = align_up(off + p * 4);
    o->sort_val0 = off;     off = align_up(off + p * 4);
    o->sort_val1 = off;     off = align_up(off + p * 4);
    o->rect = off;          off = align_up(off + p * 8);
    o->rect_sorted = off;   off = align_up(off + p * 8);
    o->krec = off;          off = align_up(off + p * 16);
    o->sort_hist = off;     off = align_up(off + sort_scratch(P).words * 4);
    o->blk_hist = off;      off = align_up(off + nb * t * 2);
    o->blk_rel = off;       off = align_up(off + nb * t * 4);
    o->blend_rec = off;     off = align_up(off + p * 64);
    o->total = off + kAlign;
}

inline void image_layout(int W, int H, fnx_image_layout_t *o) {
    size_t n = (size_t)W * H, t = (size_t)tiles_x(W) * tiles_y(H);
    size_t off = 0;
    o->header = off;      off = align_up(off + 32);
    o->final_T = off;     off = align_up(off + n * 4);
    o->n_contrib = off;   off = align_up(off + n * 4 * 2);  // [n_contrib | the limited backward's walking limit per pixel]
    o->ranges = off;      off = align_up(off + t * 8);
    o->tile_count = off;  off = align_up(off + t * 4);
    o->dyn_start = off;   off = align_up(off + t * 4);
    o->acc_final = off;   off = align_up(off + n * 4 * 3);
    o->tile_order = off;  off = align_up(off + t * 4);
    o->tile_deep = off;   off = align_up(off + t);
    o->total = off + kAlign;
}

// R = capacity of the per-call (dynamic) instances; R_static > 0: static-split layout (the merged point_list holds
// R + R_static ids, the dynamic instances are emitted as (depth bits, id) pairs).
inline void binning_layout(int64_t R, int64_t R_static, bool split, fnx_binning_layout_t *o) {
    size_t r = (size_t)(R > 0 ? R : 0), rs = (size_t)(split && R_static > 0 ? R_static : 0);
    size_t off = 0;
    o->point_list = off; off = align_up(off + (r + rs) * 4);
    o->pairs = off;      off = align_up(off + (split ? r * 8 : 0));
    // forward -> backward hand-over per batch of kBlendBatch list entries (see blend_state_slots)
    o->bstate = off;     off = align_up(off + blend_state_slots(r + rs) * kBlendStateBytes);
    o->bwd_items = off;  off = align_up(off + (blend_state_slots(r + rs) + kMaxTiles) * 4);
    o->block_masks = off; off = align_up(off + (r + rs) * 2);
    o->total = off + kAlign;
}

// Dual mode (fnx_raster_dual_t): the second image's per-batch hand-over (transmittance + accumulated value, 8 bytes per
// pixel) sits behind everything else, so the plain layouts above do not move.
inline size_t binning_dual_offset(int64_t R, int64_t R_static, bool split) {
    fnx_binning_layout_t L;
    binning_layout(R, R_static, split, &L);
    return L.total - kAlign;
}
inline size_t binning_dual_total(int64_t R, int64_t R_static, bool split) {
    size_t r = (size_t)(R > 0 ? R : 0), rs = (size_t)(split && R_static > 0 ? R_static : 0);
    return align_up(binning_dual_offset(R, R_static, split) + blend_state_slots(r + rs) * (kBlendStateBytes / 2)) + kAlign;
}

// Static splat set binned once (fnx_static_finalize_views): everything the per-iteration kernels need from it.
inline void static_layout(int P_static, int W, int H, int64_t R_static, fnx_static_layout_t *o) {
    size_t p = (size_t)(P_static > 0 ? P_static : 0), r = (size_t)(R_static > 0 ? R_static : 0);
    size_t t = (size_t)tiles_x(W) * tiles_y(H);
    size_t off = 0;
    o->header = off;    off = align_up(off + 32);
    o->starts = off;    off = align_up(off + (t + 1) * 4);
    o->radii = off;     off = align_up(off + p * 4);
    o->blend_rec = off; off = align_up(off + p * 64);
    o->pairs = off;     off = align_up(off + r * 8);
    o->total = off + kAlign;
}

// View-batched launches: V cameras over the same Gaussians, the view is grid dimension y of every
// kernel.  Per-view scratch (the three blobs) and per-view user arrays are V equal slices; the blob
// strides in bytes live here, the user-array strides follow from P, W, H and the channel count.
constexpr int kMaxViews = FNX_MAX_VIEWS;
struct ViewBatch {
    size_t geom, img, bin;  // bytes between consecutive views' blobs (0 for a single view)
    size_t bin_pairs;       // byte offset of the (key, id) pair array inside a binning blob (static-split mode)
    size_t bin_bstate, bin_items;  // byte offsets of the per-batch blend state and the backward work items
    size_t bin_masks;              // byte offset of the per-entry block masks (forward -> backward)
    size_t radii_stride;    // elements between consecutive views' radii (= total splat count)
    float tan_fovx[kMaxViews], tan_fovy[kMaxViews], focal_x[kMaxViews], focal_y[kMaxViews];
};

// Reference to the static blobs of a view batch (base == nullptr: no static set).  Static splats carry the ids
// [id0, id0 + P) of the caller's arrays; the per-call (dynamic) splats the ids [0, id0).
struct StaticRef {
    const char *base;  // aligned start of view 0's blob
    size_t stride;     // bytes between consecutive views' blobs
    uint32_t id0;
    int P;
    size_t starts, radii, rec, pairs;  // byte offsets inside a blob
};
enum { SHDR_NUM_RENDERED = 0, SHDR_P = 1, SHDR_ID0 = 2 };

// Dual mode of the blend kernels (fnx_raster_dual_t, round 5): a SECOND, single-channel image over the per-call splats
// only (ids below the gradient limit), blended -- and differentiated -- in the same pass over the same lists as the first.
// img1 == nullptr: off.  The second image's per-pixel arrays live in image blobs of their own (same layout and stride as
// the first image's), its value per splat is channel 0 of the splat's colour.
struct DualRef {
    char *img1;                            // aligned start of view 0's second image blob (stride: ViewBatch::img)
    size_t final_T, n_contrib, acc_final;  // byte offsets inside an image blob
    const float *bg1;                      // [1] background of the second image
    float *out_color1, *out_depth1;        // forward outputs [V,1,H,W]
    const float *dL_dpix1;                 // backward input [V,1,H,W]
    size_t bin_bstate1;                    // byte offset of the second image's per-batch state inside a binning blob
};

// Segmented blend forward (fnx_raster_opts_t.segment_scratch, round 5): the list of a DEEP tile is cut into segments of
// kSegBatches batches (the first: kSegBatches0) that independent workgroups blend at the same time, each from
// transmittance 1 and colour 0; the workgroup of a tile that finishes last puts the segments together per pixel
// (colour += T_in colour_s, T_in *= T_s), checks that no decision of a segment's walk depends on the T_in it did not know
// (the stop rule T < 1e-4, the median-depth entry) and blends the rest of the list again from the first segment where
// one does (raster_forward.hip).  Per view: a control block, the work list of the blend launch (segments of the
// segmented tiles first, then the other tiles in tile order), per tile the first record slot and an arrival counter,
// per segment a record per pixel.
#ifndef FNX_SEG_BATCHES
#define FNX_SEG_BATCHES 4
#endif
#ifndef FNX_SEG_BATCHES0
#define FNX_SEG_BATCHES0 8
#endif
constexpr uint32_t kSegBatches = FNX_SEG_BATCHES, kSegBatches0 = FNX_SEG_BATCHES0;
constexpr uint32_t kSegMax = 2048;      // segments per view (record slots); tiles beyond the budget stay whole
constexpr uint32_t kSegPerTileMax = 255;
// work item of the blend forward: tile | segment << 14 | segments of the tile << 24 (0: the whole tile)
enum { SEG_CTL_WORK = 0, SEG_CTL_SEGMENTS = 1, SEG_CTL_TILES = 2, SEG_CTL_REPAIRED = 3, SEG_CTL_REPAIR_BATCHES = 4,
       SEG_CTL_PARITY = 5, SEG_CTL_WORDS = 16 };
// What a segment's walk cannot know is the transmittance T_in in front of it.  It starts from a HINT: the T_in the same
// pixel had at the same segment boundary in the previous forward of this view batch (kept in the scratch, two generations:
// a call reads the previous one's while it writes its own); without one (first call, a tile not cut last time) from 1.
// The hint only has to be good enough that a pixel hinted as stopped is stopped and that the walk sees T cross 1/2 in the
// segment where it does; everything else is scaled to the true T_in afterwards.
struct SegLayout {
    size_t ctl, items, tile_slot, arrive, meta, rec_a, rec_b, hint, total;  // byte offsets inside a view's scratch
};
__host__ __device__ inline SegLayout seg_layout(int T) {
    SegLayout o;
    size_t off = 0, t = (size_t)T;
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    o.ctl = off;       off = up(off + SEG_CTL_WORDS * 4);
    o.items = off;     off = up(off + (t + kSegMax) * 4);
    o.tile_slot = off; off = up(off + 2 * t * 4);  // [generation][tile]: first record slot | segments << 16 (0xFFFFFFFF: whole)
    o.arrive = off;    off = up(off + t * 4);
    o.meta = off;      off = up(off + (size_t)kSegMax * 16);
    o.rec_a = off;     off = up(off + (size_t)kSegMax * 256 * 16);
    o.rec_b = off;     off = up(off + (size_t)kSegMax * 256 * 16);
    o.hint = off;      off = up(off + 2 * (size_t)kSegMax * 256 * 4);  // [generation][slot][pixel]: working T in front of the segment
    o.total = off + 256;
    return o;
}
struct SegRef {
    char *base;     // aligned start of view 0's scratch (nullptr: off)
    size_t stride;  // bytes between consecutive views' scratch
    SegLayout L;
};
// first batch of segment s of a tile, and the segments a list of `len` entries is cut into (0: not worth cutting)
__host__ __device__ inline uint32_t seg_first_batch(uint32_t s) { return s == 0 ? 0u : kSegBatches0 + (s - 1u) * kSegBatches; }
__host__ __device__ inline uint32_t seg_count_for(uint32_t len) {
    const uint32_t nb = (len + 255u) >> 8;
    if (nb < kSegBatches0 + kSegBatches) return 0u;
    const uint32_t n = 1u + (nb - kSegBatches0 + kSegBatches - 1u) / kSegBatches;
    return n > kSegPerTileMax ? 0u : n;
}

// header words inside the image blob
// HDR_BIN_CAPACITY: the binning capacity stage 2 ran with (the binning blob's layout depends on it): the backward pass
// refuses a view whose stored value differs from its own argument (status FNX_ERR_CAPACITY)
enum { HDR_NUM_RENDERED = 0, HDR_STATUS = 1, HDR_CAPACITY = 2, HDR_NUM_STATIC = 3, HDR_BWD_ITEMS = 4, HDR_DEEP_COUNT = 5,
       HDR_BIN_CAPACITY = 6, HDR_BWD_TICKET = 7, HDR_BWD_DONE = 8,
       // the forward's gradient limit: ids >= it were treated as gradient-free when the per-pixel walking limits of the
       // backward (second half of n_contrib) and its work items were laid down; a backward with a larger limit is refused
       HDR_DYN_LIMIT = 9,
       // list entries the blend forward staged (batches it blended, clipped to the lists) and the blend backward walked
       // (its work items' batches): what the two kernels actually read, for the roofline record (one atomic per workgroup)
       HDR_FWD_ENTRIES = 10, HDR_BWD_ENTRIES = 11 };  // words 12 .. 63 of the header block are scratch

}  // namespace fnx
