// Host+device view of the three opaque scratch blobs (layout is this library's own; the
// reference's GeometryState/ImageState/BinningState live in ch3 rasterizer_impl.h:27-64).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "../../include/fnx_raster.h"

namespace fnx {

constexpr size_t kAlign = 256;
inline size_t align_up(size_t x) { return (x + kAlign - 1) / kAlign * kAlign; }

inline int tiles_x(int W) { return (W + 15) / 16; }
inline int tiles_y(int H) { return (H + 15) / 16; }

inline void geom_layout(int P, fnx_geom_layout_t *o) {
    size_t off = 0;
    size_t p = (size_t)(P > 0 ? P : 0);
    o->depths = off;        off = align_up(off + p * 4);
    o->clamped = off;       off = align_up(off + p * 3);
    o->radii = off;         off = align_up(off + p * 4);
    o->means2D = off;       off = align_up(off + p * 8);
    o->cov3D = off;         off = align_up(off + p * 24);
    o->conic_opacity = off; off = align_up(off + p * 16);
    o->rgb = off;           off = align_up(off + p * 12);
    o->tiles_touched = off; off = align_up(off + p * 4);
    o->total = off + kAlign;
}

inline void image_layout(int W, int H, fnx_image_layout_t *o) {
    size_t n = (size_t)W * H, t = (size_t)tiles_x(W) * tiles_y(H);
    size_t off = 0;
    o->header = off;      off = align_up(off + 32);
    o->final_T = off;     off = align_up(off + n * 4);
    o->n_contrib = off;   off = align_up(off + n * 4);
    o->ranges = off;      off = align_up(off + t * 8);
    o->tile_count = off;  off = align_up(off + t * 4);
    o->tile_cursor = off; off = align_up(off + t * 4);
    o->total = off + kAlign;
}

inline void binning_layout(int64_t R, fnx_binning_layout_t *o) {
    size_t r = (size_t)(R > 0 ? R : 0);
    size_t off = 0;
    o->point_list = off; off = align_up(off + r * 4);
    o->pairs = off;      off = align_up(off + r * 8);
    o->total = off + kAlign;
}

// header words inside the image blob
enum { HDR_NUM_RENDERED = 0, HDR_STATUS = 1, HDR_CAPACITY = 2 };

}  // namespace fnx
