// Binning of splat instances into per-tile, depth-ordered lists -- the MI355X-native replacement
// for the reference's key duplication + device-wide 64-bit radix sort + range detection
// (ch3/cuda_rasterizer/rasterizer_impl.cu:67-128,259-296).
//
// The reference sorts R = sum(tiles touched) 64-bit (tile | depth) keys.  Here only the P splats
// are sorted, once, by depth; the R instances are never sorted globally:
//
//   depth sort      stable LSD radix sort of (depth bits -> id), 9-bit digits, wave64 ballot ranking:
//                   sorted_ids[rank] = splat id in (depth bits, id) order.  Keys are taken relative to
//                   the smallest visible key, so 3 passes suffice unless the depths span >= 2^27 ulps
//                   (then a 4th runs; its kernels return at once otherwise)
//   rank_hist       splats in RANK order, blocks of 1024 consecutive ranks: how many splats of the
//                   block touch each tile -> blk_hist[block][tile]               (LDS atomics only)
//   tile_colscan    column prefix of that matrix -> blk_rel[block][tile], tile totals
//   tile_scan       (raster_forward.hip) tile totals -> [start,end) ranges, num_rendered
//   emit            each rank block owns the slice [start + blk_rel, +blk_hist) of every tile's
//                   segment.  Slices of consecutive blocks are consecutive in depth, so a tile's list
//                   is ordered as soon as every slice is.  Work items (tile_scan): a rank block, or a band
//                   of tile rows of a heavy one; persistent workgroups take them by ticket.  Per sub-batch
//                   of <= 256 ranks every tile gets an LDS bitmask of the ranks touching it (atomicOr);
//                   an instance's position in its slice is the number of set bits below its own, and the
//                   splat id goes straight to its final position.  No global atomics, no per-tile sort.
//
// (depth bits, id) is a total order and the reference's radix sort is stable over keys emitted in
// id order, so the resulting lists are bit-identical to the reference's point_list.
#include "fnx_device.h"
#include "fnx_state.h"

namespace fnx {

// ---------------------------------------------------------------------------------------------
// Column prefix over blocks.  Workgroup = 64 columns x 16 waves; wave w owns a contiguous
// band of blocks, lane = column.  Pass 1 sums the band, LDS combines bands, pass 2 writes prefixes.
template <typename CountT>
__global__ void __launch_bounds__(1024)
colscan_kernel(int T, int NB, const CountT *__restrict__ blk_hist, uint32_t *__restrict__ blk_rel,
               uint32_t *__restrict__ tile_count, size_t hist_stride, size_t count_stride,
               const uint32_t *__restrict__ kmax_blk, uint32_t *__restrict__ ctl, int only_if_wide, int narrow) {
    __shared__ uint32_t s_band[16][64];
    if (ctl) {  // depth-sort passes
        ctl = view_at(ctl, hist_stride, blockIdx.y);
        if (only_if_wide && ctl[SORT_CTL_WIDE] == 0u) return;
        if (kmax_blk && blockIdx.x == 0 && threadIdx.x < 64) {
            // first pass: largest relative key of the view -> is the fourth pass needed?
            kmax_blk = view_at(kmax_blk, hist_stride, blockIdx.y);
            uint32_t m = 0;
            for (int b = threadIdx.x; b < NB; b += 64) m = max(m, kmax_blk[b]);
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off));
            if (threadIdx.x == 0) {
                const uint32_t wide = (m >> (3 * kSortBits)) ? 1u : 0u;
                ctl[SORT_CTL_WIDE] = narrow ? 0u : wide;  // narrow: the fourth pass is not launched
                ctl[SORT_CTL_OVERFLOW] = narrow ? wide : 0u;
                ctl[SORT_CTL_SPAN] = m ? 32u - (uint32_t)__clz((int)m) : 0u;
            }
        }
    }
    blk_hist = view_at(blk_hist, hist_stride, blockIdx.y);
    blk_rel = view_at(blk_rel, hist_stride, blockIdx.y);
    tile_count = view_at(tile_count, count_stride, blockIdx.y);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + lane;
    const int per = (NB + 15) / 16;
    const int b0 = w * per, b1 = min(NB, b0 + per);
    // the band is read once, eight rows at a time (independent loads in flight; the kernel is pure latency),
    // and kept in registers for the second pass when it fits
    constexpr int kKeep = 24;
    uint32_t keep[kKeep];
    uint32_t sum = 0;
    if (t < T) {
#pragma unroll
        for (int k = 0; k < kKeep; k++) keep[k] = (b0 + k < b1) ? (uint32_t)blk_hist[(size_t)(b0 + k) * T + t] : 0u;
#pragma unroll
        for (int k = 0; k < kKeep; k++) sum += keep[k];
        for (int b = b0 + kKeep; b < b1; b++) sum += blk_hist[(size_t)b * T + t];
    }
    s_band[w][lane] = sum;
    __syncthreads();
    uint32_t run = 0;
    for (int k = 0; k < w; k++) run += s_band[k][lane];
    if (t < T) {
#pragma unroll
        for (int k = 0; k < kKeep; k++) {
            if (b0 + k < b1) blk_rel[(size_t)(b0 + k) * T + t] = run;
            run += keep[k];
        }
        for (int b = b0 + kKeep; b < b1; b++) {
            blk_rel[(size_t)b * T + t] = run;
            run += blk_hist[(size_t)b * T + t];
        }
        if (w == 15) {
            // band 15 may be empty; the grand total is the sum of all bands
            uint32_t tot = 0;
            for (int k = 0; k < 16; k++) tot += s_band[k][lane];
            tile_count[t] = tot;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Depth sort: LSD radix, kSortBits-bit digits, stable.  Workgroup = 256 threads = 4 waves over a chunk of
// kSortChunk = 1024 keys; wave w owns keys [256 w, 256 w + 256) of the chunk in 4 steps of 64.
// Pass 0 reads the raw keys and sorts key' = key - kmin (culled splats: 0; their rank is irrelevant, they emit
// nothing), where kmin is the smallest visible key of the view; the later passes read key' as scattered.
enum { SORT_FIRST = 0, SORT_MIDDLE = 1, SORT_THIRD = 2, SORT_FOURTH = 3 };

__device__ __forceinline__ uint32_t relative_key(uint32_t key, uint32_t kmin) {
    return key == 0xFFFFFFFFu ? 0u : key - kmin;
}

__global__ void __launch_bounds__(256)
sort_hist_kernel(int P, const uint32_t *__restrict__ raw_keys, const uint2 *__restrict__ pairs, int pass,
                 uint32_t *__restrict__ hist, const uint32_t *__restrict__ kmin_blk, uint32_t *__restrict__ kmax_blk,
                 uint32_t *__restrict__ ctl, size_t geom_stride) {
    __shared__ uint32_t s_h[kSortRadix];
    __shared__ uint32_t s_red[4];
    raw_keys = view_at(raw_keys, geom_stride, blockIdx.y);
    pairs = view_at(pairs, geom_stride, blockIdx.y);
    hist = view_at(hist, geom_stride, blockIdx.y);
    ctl = view_at(ctl, geom_stride, blockIdx.y);
    if (pass == SORT_FOURTH && ctl[SORT_CTL_WIDE] == 0u) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int shift = pass * kSortBits;
    uint32_t kmin = 0;
    if (pass == SORT_FIRST) {
        // every workgroup reduces the preprocess workgroups' minima (a few KiB from L2); workgroup 0 publishes it
        kmin_blk = view_at(kmin_blk, geom_stride, blockIdx.y);
        const int nkb = (P + kKeyBlock - 1) / kKeyBlock;
        uint32_t m = 0xFFFFFFFFu;
        for (int b = tid; b < nkb; b += 256) m = min(m, kmin_blk[b]);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) m = min(m, (uint32_t)__shfl_xor((int)m, off));
        if (lane == 0) s_red[w] = m;
        __syncthreads();
        kmin = min(min(s_red[0], s_red[1]), min(s_red[2], s_red[3]));
        if (blockIdx.x == 0 && tid == 0) ctl[SORT_CTL_KMIN] = kmin;
        __syncthreads();
    }
    for (int d = tid; d < kSortRadix; d += 256) s_h[d] = 0;
    __syncthreads();
    const int base = blockIdx.x * kSortChunk;
    uint32_t kmax = 0;
#pragma unroll
    for (int k = 0; k < kSortChunk / 256; k++) {
        const int i = base + k * 256 + tid;
        if (i < P) {
            uint32_t key;
            if (pass == SORT_FIRST) {
                key = relative_key(raw_keys[i], kmin);
                kmax = max(kmax, key);
            } else {
                key = pairs[i].x;
            }
            atomicAdd(&s_h[(key >> shift) & (kSortRadix - 1)], 1u);
        }
    }
    if (pass == SORT_FIRST) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, off));
        if (lane == 0) s_red[w] = kmax;
    }
    __syncthreads();
    for (int d = tid; d < kSortRadix; d += 256)
        hist[(size_t)blockIdx.x * kSortRadix + d] = s_h[d];  // block-major rows of kSortRadix digits
    if (pass == SORT_FIRST && tid == 0) {
        kmax_blk = view_at(kmax_blk, geom_stride, blockIdx.y);
        kmax_blk[blockIdx.x] = max(max(s_red[0], s_red[1]), max(s_red[2], s_red[3]));
    }
}

// Stable scatter of one pass.  Rank of a key inside its 64-key step: lanes holding the same digit
// are found with kSortBits ballots; the rank is the number of lower lanes in that set.
// The tile rectangles are brought into rank order by the last pass that runs (the third, or the fourth).
__global__ void __launch_bounds__(256)
sort_scatter_kernel(int P, const uint32_t *__restrict__ raw_keys, const uint2 *__restrict__ pairs_in,
                    uint2 *__restrict__ pairs_out, int pass,
                    const uint32_t *__restrict__ hist_rel, const uint32_t *__restrict__ digit_total,
                    const uint32_t *__restrict__ ctl, const uint2 *__restrict__ rect, uint2 *__restrict__ rect_sorted,
                    size_t geom_stride) {
    __shared__ uint32_t s_cnt[4][kSortRadix];   // per-wave digit counts, then per-wave running offsets
    __shared__ uint32_t s_wtot[4];
    bool with_rect = false;
    uint32_t kmin = 0;
    {
        const int vw = blockIdx.y;
        ctl = view_at(ctl, geom_stride, vw);
        const bool wide = ctl[SORT_CTL_WIDE] != 0u;
        if (pass == SORT_FOURTH && !wide) return;
        if (pass == SORT_FIRST) kmin = ctl[SORT_CTL_KMIN];
        with_rect = (pass == SORT_THIRD && !wide) || pass == SORT_FOURTH;
        raw_keys = view_at(raw_keys, geom_stride, vw);
        pairs_in = view_at(pairs_in, geom_stride, vw);
        pairs_out = view_at(pairs_out, geom_stride, vw);
        hist_rel = view_at(hist_rel, geom_stride, vw);
        digit_total = view_at(digit_total, geom_stride, vw);
        rect = view_at(rect, geom_stride, vw);
        rect_sorted = view_at(rect_sorted, geom_stride, vw);
    }
    const int shift = pass * kSortBits;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int base = blockIdx.x * kSortChunk + w * 256;
    // the digit totals and this chunk's column prefixes are requested up front, with the keys (one round trip, not two)
    static_assert(kSortRadix == 512, "two digits per thread");
    const uint32_t t0 = digit_total[2 * tid], t1 = digit_total[2 * tid + 1];
    const uint2 rel = *reinterpret_cast<const uint2 *>(hist_rel + (size_t)blockIdx.x * kSortRadix + 2 * tid);
#pragma unroll
    for (int k = 0; k < 4; k++)
        for (int d = tid; d < kSortRadix; d += 256) s_cnt[k][d] = 0;
    __syncthreads();
    uint2 kv[4];  // (relative key, splat id)
    bool valid[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int i = base + k * 64 + lane;
        valid[k] = i < P;
        kv[k] = make_uint2(0u, 0u);
        if (valid[k]) {
            kv[k] = pass == SORT_FIRST ? make_uint2(relative_key(raw_keys[i], kmin), (uint32_t)i) : pairs_in[i];
            atomicAdd(&s_cnt[w][(kv[k].x >> shift) & (kSortRadix - 1)], 1u);
        }
    }
    __syncthreads();
    // thread = digit pair (2 tid, 2 tid + 1): global base of the digit (exclusive scan of the digit totals) +
    // keys of this digit in earlier chunks + earlier waves of this chunk
    {
        const uint32_t tot = t0 + t1;
        uint32_t inc = tot;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = (uint32_t)__shfl_up((int)inc, off);
            if (lane >= off) inc += v;
        }
        if (lane == 63) s_wtot[w] = inc;
        __syncthreads();
        uint32_t dbase = inc - tot;
        for (int k = 0; k < w; k++) dbase += s_wtot[k];
        uint32_t run0 = dbase + rel.x, run1 = dbase + t0 + rel.y;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint2 c = *reinterpret_cast<const uint2 *>(&s_cnt[k][2 * tid]);
            *reinterpret_cast<uint2 *>(&s_cnt[k][2 * tid]) = make_uint2(run0, run1);
            run0 += c.x;
            run1 += c.y;
        }
    }
    __syncthreads();
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    volatile uint32_t *run_off = s_cnt[w];  // updated by one lane, read by the others: keep it out of registers
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t d = (kv[k].x >> shift) & (kSortRadix - 1);
        unsigned long long same = __ballot(valid[k]);
#pragma unroll
        for (int b = 0; b < kSortBits; b++) {
            const unsigned long long m = __ballot((d >> b) & 1u);
            same &= ((d >> b) & 1u) ? m : ~m;
        }
        if (valid[k]) {
            const uint32_t r = (uint32_t)__popcll(same & lt_mask);
            const uint32_t pos = run_off[d] + r;
            pairs_out[pos] = kv[k];
            if (with_rect) rect_sorted[pos] = rect[kv[k].y];
        }
        // the wave's LDS reads above are issued before this write (in-order per wave)
        if (valid[k] && (same & lt_mask) == 0ull) run_off[d] = run_off[d] + (uint32_t)__popcll(same);
    }
}

// ---------------------------------------------------------------------------------------------
// Per-(rank block, tile) instance counts.  A 256-thread workgroup owns kSplatBlock = 1024
// consecutive depth ranks (4 per thread).  The kernel is bound by LDS atomic throughput, so a splat
// does not add 1 to every tile of its rectangle: per tile row it adds +1 at the first column and -1
// just past the last one (2 atomics per row instead of one per tile), and the counts are the running
// sums along each tile row, formed once per block.
__global__ void __launch_bounds__(256)
rank_hist_kernel(int P, int T, const uint2 *__restrict__ rect_sorted, int gx, int gy, uint16_t *__restrict__ blk_hist,
                 uint32_t *__restrict__ blk_total, const ViewBatch vb) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_hist[];  // T row-delta counters (mod 2^32)
    __shared__ uint32_t s_total;
    {
        const int vw = blockIdx.y;
        rect_sorted = view_at(rect_sorted, vb.geom, vw);
        blk_hist = view_at(blk_hist, vb.geom, vw);
        blk_total = view_at(blk_total, vb.geom, vw);
    }
    if (threadIdx.x == 0) s_total = 0;
    for (int i = threadIdx.x; i < T; i += 256) s_hist[i] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kSplatBlock / 256; k++) {
        const int rank = blockIdx.x * kSplatBlock + k * 256 + threadIdx.x;
        if (rank >= P) break;
        const uint2 r = rect_sorted[rank];
        const int x0 = r.x & 0xFFFFu, x1 = r.x >> 16, y0 = r.y & 0xFFFFu, y1 = r.y >> 16;
        if (x1 > x0)
            for (int y = y0; y < y1; y++) {
                atomicAdd(&s_hist[y * gx + x0], 1u);
                if (x1 < gx) atomicAdd(&s_hist[y * gx + x1], 0xFFFFFFFFu);  // -1
            }
    }
    __syncthreads();
    {  // running sum along every tile row: a half-wave (rows of <= 32 tiles: two rows per wave at a time) or a wave per
       // row, 32 / 64 tiles per step with a shuffle scan (one thread per row walked the row through LDS serially)
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        const int seg = gx <= 32 ? 32 : 64, rows_at_once = 64 / seg;
        const int sl = lane & (seg - 1), sub = lane / seg;
        for (int y0 = w * rows_at_once; y0 < gy; y0 += 4 * rows_at_once) {
            const int y = y0 + sub;
            uint32_t carry = 0;
            for (int x0 = 0; x0 < gx; x0 += seg) {
                const int x = x0 + sl;
                const bool in = y < gy && x < gx;
                uint32_t v = in ? s_hist[y * gx + x] : 0u;
                for (int off = 1; off < seg; off <<= 1) {
                    const uint32_t t = (uint32_t)__shfl_up((int)v, off, seg);
                    if (sl >= off) v += t;
                }
                if (in) s_hist[y * gx + x] = v + carry;
                carry += (uint32_t)__shfl((int)v, seg - 1, seg);
            }
        }
    }
    __syncthreads();
    uint16_t *row = blk_hist + (size_t)blockIdx.x * T;
    uint32_t sum = 0;
    for (int i = threadIdx.x; i < T; i += 256) {
        row[i] = (uint16_t)s_hist[i];
        sum += s_hist[i];
    }
    // instances of the whole block: emit splits heavy blocks over several workgroups
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += (uint32_t)__shfl_xor((int)sum, off);
    if ((threadIdx.x & 63) == 0) atomicAdd(&s_total, sum);
    __syncthreads();
    if (threadIdx.x == 0) blk_total[blockIdx.x] = s_total;
}

// ---------------------------------------------------------------------------------------------
// Instance emission, final order.  Same rank blocks as rank_hist.  A block's splats are handled in
// sub-batches of at most kEmitSpan consecutive ranks (and at most kEmitStage instances); for every
// tile an LDS bitmask over the sub-batch's ranks records which of them touch the tile (atomicOr),
// and an instance's place inside the block's slice of its tile is the number of set bits below its
// own: depth order without sorting and without ordered atomics.  Tiles are handled in windows of at
// most kEmitTileWindow (the LDS arrays are per window).
#ifndef FNX_EMIT_THREADS
#define FNX_EMIT_THREADS 512
#endif
#ifndef FNX_EXP_EMIT
#define FNX_EXP_EMIT 0  // timing experiments (tools/build_variant.py): 10 loads only, 11 no bitmask / output, 12 no store, 13 lane-contiguous store
#endif
#ifdef FNX_EXP_CLOCK  // developer timing: per-workgroup [start, end, sub-batches, instances] of the last emit launch
__device__ unsigned long long g_emit_clock[4 * 16384];
extern "C" int fnx_debug_emit_clock(unsigned long long *host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_emit_clock), (size_t)n * 8);
}
#endif
constexpr int kEmitThreads = FNX_EMIT_THREADS;          // many waves per workgroup: the passes are latency-bound
constexpr int kEmitPer = kSplatBlock / kEmitThreads;   // splats loaded per thread
#ifndef FNX_EMIT_CHUNK
#define FNX_EMIT_CHUNK 8
#endif
constexpr int kEmitChunk = FNX_EMIT_CHUNK;             // instances per thread and sub-batch (kept in registers)
constexpr int kEmitStage = kEmitChunk * kEmitThreads;  // instances per sub-batch
#ifndef FNX_EMIT_MASK_WORDS
#define FNX_EMIT_MASK_WORDS 8  // 256 ranks per sub-batch; measured on config 3 (5 views): 2 -> 120, 4 -> 86, 8 -> 72, 16 -> 107, 32 -> 115 us
#endif
constexpr int kEmitMaskWords = FNX_EMIT_MASK_WORDS;    // per-tile bitmask: 32 ranks per word
constexpr int kEmitSpan = 32 * kEmitMaskWords;         // ranks per sub-batch
constexpr int kEmitTileWindow = kEmitMaskWords <= 8 ? 2048 : (kEmitMaskWords <= 16 ? 1792 : 960);  // <= kEmitStage: one splat never overflows a sub-batch; window x (words + 1) x 4 B of LDS
static_assert(kSplatBlock == 1024, "emit packs the rank-in-block into 10 bits");
static_assert(kEmitTileWindow <= kEmitStage && kEmitTileWindow <= (1 << 22), "entry packing");

// Cursor over the instances of a rank block in (splat, tile row, tile column) order, restricted to
// the tile window [tw0, tw1).  s_pre = inclusive prefix of the per-splat instance counts.
struct InstanceWalk {
    int l, x, y, x0, x1, y1;
    __device__ __forceinline__ void load(const uint2 *s_rect, int ll) {
        const uint2 r = s_rect[ll];
        l = ll;
        x0 = r.x & 0xFFFFu;
        x1 = r.x >> 16;
        y = r.y & 0xFFFFu;
        y1 = r.y >> 16;
        x = x0;
    }
    __device__ __forceinline__ int tile(int gx) const { return y * gx + x; }
    // position on instance number `target` (0-based over the whole block, window-restricted counts)
    __device__ __forceinline__ void seek(const uint32_t *s_pre, const uint2 *s_rect, int l_lo, int l_hi, uint32_t target,
                                         int gx, int tw0, int tw1, bool whole) {
        int lo = l_lo, hi = l_hi;  // first splat whose inclusive prefix exceeds target (it lies in [l_lo, l_hi])
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (s_pre[mid] > target) hi = mid; else lo = mid + 1;
        }
        load(s_rect, lo);
        uint32_t skip = target - (lo > 0 ? s_pre[lo - 1] : 0u);
        if (whole) {
            const uint32_t wdt = (uint32_t)max(x1 - x0, 1);
            y += (int)(skip / wdt);
            x += (int)(skip % wdt);
        } else {
            if (!in_window(gx, tw0, tw1)) advance_to_window(s_rect, gx, tw0, tw1);
            for (; skip > 0; skip--) next(s_rect, gx, tw0, tw1, false);
        }
    }
    __device__ __forceinline__ bool in_window(int gx, int tw0, int tw1) const {
        const int t = y * gx + x;
        return t >= tw0 && t < tw1 && x < x1 && y < y1;
    }
    __device__ __forceinline__ void step(const uint2 *s_rect) {
        if (++x >= x1) {
            x = x0;
            if (++y >= y1) {
                int ll = l + 1;
                while (ll < kSplatBlock - 1 && s_rect[ll].x == 0u && s_rect[ll].y == 0u) ll++;
                load(s_rect, min(ll, kSplatBlock - 1));
            }
        }
    }
    __device__ __forceinline__ void advance_to_window(const uint2 *s_rect, int gx, int tw0, int tw1) {
        for (int guard = 0; guard < (1 << 24) && !in_window(gx, tw0, tw1); guard++) {
            if (l >= kSplatBlock - 1 && (y >= y1 || x1 <= x0)) break;
            step(s_rect);
        }
    }
    __device__ __forceinline__ void next(const uint2 *s_rect, int gx, int tw0, int tw1, bool whole) {
        step(s_rect);
        if (!whole) advance_to_window(s_rect, gx, tw0, tw1);
    }
};

// Persistent workgroups: each takes tickets from one counter; ticket t is item t / V of view t % V, so the
// items come in block order across all views -- the front blocks (nearest, largest splats, most instances) first,
// and the long items do not end up in the tail.  (Launching one workgroup per possible item and returning early
// from the unused ones costs ~45 ns of dispatch per workgroup.)
// PAIRS (static-split mode): an instance is written as the pair (depth bits, id) instead of the id alone -- the blend
// kernel merges the tile's list with the static splats' by depth.
template <bool PAIRS>
__global__ void __launch_bounds__(kEmitThreads)
emit_kernel(int P, int T, const uint2 *__restrict__ sorted3_all, const uint2 *__restrict__ sorted4_all,
            const uint32_t *__restrict__ sort_ctl, const uint2 *__restrict__ rect_sorted_all, int gx, int gy,
            const uint32_t *__restrict__ starts_all, const uint32_t *__restrict__ blk_rel_all,
            uint32_t *__restrict__ emit_ctl, const uint32_t *__restrict__ emit_items,
            uint32_t *__restrict__ point_list_all, uint32_t *__restrict__ header_all, uint32_t capacity, int TW, int V,
            const ViewBatch vb) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];  // s_mask[TW][kEmitMaskWords] | s_cur[TW]
    __shared__ uint32_t s_id[kSplatBlock];
    __shared__ uint32_t s_key[PAIRS ? kSplatBlock : 1];  // depth bits of the compacted splats
    __shared__ uint2 s_rect[kSplatBlock];    // (x0 | x1 << 16, y0 | y1 << 16), tile coordinates
    __shared__ uint32_t s_pre[kSplatBlock];  // inclusive prefix of the per-splat instance counts (current window)
    __shared__ uint32_t s_wsum[kEmitThreads / 64];
    __shared__ uint32_t s_items[kMaxViews];
    __shared__ uint32_t s_ticket;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid < V) s_items[tid] = view_at(emit_ctl, vb.geom, tid)[EMIT_CTL_ITEMS];
    lds_barrier();
    uint32_t most = 0;
    for (int v = 0; v < V; v++) most = max(most, s_items[v]);
  for (;;) {
    lds_barrier();  // the previous item is done with the LDS arrays (and s_ticket has been read)
    if (tid == 0) s_ticket = atomicAdd(&emit_ctl[EMIT_CTL_TICKET], 1u);
    lds_barrier();
    const uint32_t ticket = s_ticket;
    const int vw = (int)(ticket % (uint32_t)V);
    const uint32_t item_no = ticket / (uint32_t)V;
    if (item_no >= most) break;
    if (item_no >= s_items[vw]) continue;
    const uint32_t item = view_at(emit_items, vb.geom, vw)[item_no];
    const int blk = (int)(item >> 8), band = (int)((item >> 4) & 15u), nbands = (int)(item & 15u);
    const int ry0 = (int)((long long)band * gy / nbands), ry1 = (int)((long long)(band + 1) * gy / nbands);
    const uint2 *__restrict__ rect_sorted = view_at(rect_sorted_all, vb.geom, vw);
    const uint32_t *__restrict__ blk_rel = view_at(blk_rel_all, vb.geom, vw);
    const uint32_t *__restrict__ starts = view_at(starts_all, vb.img, vw);  // first list position of every tile
    uint32_t *__restrict__ header = view_at(header_all, vb.img, vw);
    uint32_t *__restrict__ point_list = view_at(point_list_all, vb.bin, vw);
    uint2 *__restrict__ pair_list = reinterpret_cast<uint2 *>(reinterpret_cast<char *>(point_list) + vb.bin_pairs);
    const uint32_t kmin = PAIRS ? view_at(sort_ctl, vb.geom, vw)[SORT_CTL_KMIN] : 0u;  // sorted keys are relative to it
    // ids in depth order: where the third sort pass left them, or the fourth if it had to run
    const uint2 *__restrict__ sorted_ids =  // (relative key, id) pairs
        view_at(view_at(sort_ctl, vb.geom, vw)[SORT_CTL_WIDE] ? sorted4_all : sorted3_all, vb.geom, vw);
    if (header[HDR_NUM_RENDERED] > capacity) {
        if (item_no == 0 && tid == 0) {
            header[HDR_STATUS] = FNX_ERR_CAPACITY;
            header[HDR_CAPACITY] = capacity;
        }
        continue;
    }
    uint32_t *s_mask = s_dyn, *s_cur = s_dyn + (size_t)TW * kEmitMaskWords;
    const uint32_t *rel = blk_rel + (size_t)blk * T;
#ifdef FNX_EXP_CLOCK
    const unsigned long long clk0 = wall_clock64();
    unsigned long long n_sub = 0, n_inst = 0;
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = clock64();
#define FNX_PH(i) { const unsigned long long tn = clock64(); ph[i] += tn - tlast; tlast = tn; }
#else
#define FNX_PH(i)
#endif
    // the block's splats that touch the band, compacted in rank order: local index l (consecutive entries per
    // thread, so the workgroup-wide prefix over l is a per-thread running sum on top of a prefix over threads).
    // Positions in the compacted list order the instances exactly like the ranks do.
    {
        uint32_t id[kEmitPer], key[kEmitPer];
        uint2 rc[kEmitPer];
        uint32_t have = 0;
#pragma unroll
        for (int k = 0; k < kEmitPer; k++) {
            const int rank = blk * kSplatBlock + kEmitPer * tid + k;
            id[k] = key[k] = 0;
            rc[k] = make_uint2(0u, 0u);
            if (rank < P) {
                const uint2 kv = sorted_ids[rank];
                id[k] = kv.y;
                key[k] = kv.x + kmin;
                const uint2 rect = rect_sorted[rank];
                // clip the tile rows to the band
                const uint32_t y0 = max(rect.y & 0xFFFFu, (uint32_t)ry0), y1 = min(rect.y >> 16, (uint32_t)ry1);
                if (y1 > y0 && (rect.x >> 16) > (rect.x & 0xFFFFu)) {
                    rc[k] = make_uint2(rect.x, y0 | (y1 << 16));
                    have++;
                }
            }
        }
        uint32_t inc = have;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = (uint32_t)__shfl_up((int)inc, off);
            if (lane >= off) inc += v;
        }
        if (lane == 63) s_wsum[w] = inc;
        lds_barrier();
        uint32_t pos = inc - have, n_c = 0;
        for (int k = 0; k < kEmitThreads / 64; k++) {
            if (k < w) pos += s_wsum[k];
            n_c += s_wsum[k];
        }
#pragma unroll
        for (int k = 0; k < kEmitPer; k++)
            if (rc[k].x | rc[k].y) {
                s_id[pos] = id[k];
                if (PAIRS) s_key[PAIRS ? pos : 0] = key[k];
                s_rect[pos] = rc[k];
                pos++;
            }
        for (int l = tid; l < kSplatBlock; l += kEmitThreads)
            if ((uint32_t)l >= n_c) s_rect[l] = make_uint2(0u, 0u);
    }
    if (FNX_EXP_EMIT == 10) continue;
    FNX_PH(0)
    const int band_t0 = ry0 * gx, band_t1 = ry1 * gx;
    for (int tw0 = band_t0; tw0 < band_t1; tw0 += TW) {
        const int tw1 = min(band_t1, tw0 + TW);
        const bool whole = (tw0 == band_t0 && tw1 == band_t1);
        lds_barrier();
        for (int i = tid; i < tw1 - tw0; i += kEmitThreads) {
            s_cur[i] = starts[tw0 + i] + rel[tw0 + i];
#pragma unroll
            for (int q = 0; q < kEmitMaskWords; q++) s_mask[q * TW + i] = 0u;  // word-major: lanes = tiles, no bank conflicts
        }
        // per-splat instance counts inside this tile window, and their prefix over the block
        uint32_t c[kEmitPer], run = 0;
#pragma unroll
        for (int k = 0; k < kEmitPer; k++) {
            const uint2 rect = s_rect[kEmitPer * tid + k];
            const int x0 = rect.x & 0xFFFFu, x1 = rect.x >> 16, y0 = rect.y & 0xFFFFu, y1 = rect.y >> 16;
            uint32_t n = 0;
            if (whole) {
                n = (uint32_t)((x1 - x0) * (y1 - y0));
            } else {
                for (int y = y0; y < y1; y++) {  // tiles of row y inside [tw0, tw1)
                    const int a = max(y * gx + x0, tw0), b = min(y * gx + x1, tw1);
                    n += (uint32_t)max(b - a, 0);
                }
            }
            c[k] = n;
            run += n;
        }
        uint32_t inc = run;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = (uint32_t)__shfl_up((int)inc, off);
            if (lane >= off) inc += v;
        }
        if (lane == 63) s_wsum[w] = inc;
        lds_barrier();
        uint32_t pre = inc - run;
        for (int k = 0; k < w; k++) pre += s_wsum[k];
#pragma unroll
        for (int k = 0; k < kEmitPer; k++) {
            pre += c[k];
            s_pre[kEmitPer * tid + k] = pre;
        }
        lds_barrier();
        const uint32_t total = s_pre[kSplatBlock - 1];
        FNX_PH(1)
        uint32_t done = 0;  // instances of splats < l0
        for (int l0 = 0; done < total;) {
            // sub-batch [l0, l1): at most kEmitSpan ranks and kEmitStage instances
            int lo = l0, hi = min(kSplatBlock, l0 + kEmitSpan);  // l1 = first l with s_pre[l] - done > kEmitStage
            if (s_pre[hi - 1] - done <= (uint32_t)kEmitStage) {
                lo = hi;  // the usual case: the rank span ends the sub-batch, not the instance count (one LDS read, no search)
            } else {
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (s_pre[mid] - done > (uint32_t)kEmitStage) hi = mid; else lo = mid + 1;
                }
            }
            const int l1 = lo;
            const uint32_t batch = s_pre[l1 - 1] - done;
            FNX_PH(2)
            // The batch's instances, enumerated in (splat, row, column) order, are dealt out in equal
            // contiguous chunks of at most kEmitChunk: a thread finds the splat of its first instance by
            // binary search in the prefix, then walks rectangles, so the work per thread is the same
            // whatever the splat sizes.  The entries (window tile << 10 | rank in block) stay in registers.
            const uint32_t chunk = (batch + (uint32_t)kEmitThreads - 1u) / (uint32_t)kEmitThreads;
            const uint32_t i0 = min(batch, (uint32_t)tid * chunk), i1 = min(batch, i0 + chunk);
            uint32_t ent[kEmitChunk];
            if (batch > 0) {
                InstanceWalk wk;
                wk.load(s_rect, l0);
                if (i0 < i1) wk.seek(s_pre, s_rect, l0, l1 - 1, done + i0, gx, tw0, tw1, whole);
#pragma unroll
                for (int k = 0; k < kEmitChunk; k++) {
                    ent[k] = 0xFFFFFFFFu;
                    if (i0 + k < i1) {
                        ent[k] = ((uint32_t)(wk.tile(gx) - tw0) << 10) | (uint32_t)wk.l;
                        if (i0 + k + 1 < i1) wk.next(s_rect, gx, tw0, tw1, whole);
                    }
                }
                FNX_PH(3)
                const int nw = (l1 - l0 + 31) >> 5;  // bitmask words in use in this sub-batch
#pragma unroll
                for (int k = 0; k < kEmitChunk; k++)
                    if (ent[k] != 0xFFFFFFFFu && FNX_EXP_EMIT != 11) {
                        const uint32_t b = (ent[k] & 1023u) - (uint32_t)l0;
                        atomicOr(&s_mask[(b >> 5) * TW + (ent[k] >> 10)], 1u << (b & 31u));
                    }
                lds_barrier();
                FNX_PH(4)
#pragma unroll
                for (int k = 0; k < kEmitChunk; k++)
                    if (ent[k] != 0xFFFFFFFFu && FNX_EXP_EMIT != 11) {
                        const uint32_t t = ent[k] >> 10, l = ent[k] & 1023u, b = l - (uint32_t)l0;
                        const uint32_t bw = b >> 5;
                        uint32_t r = 0;
#pragma unroll
                        for (uint32_t q = 0; q < (uint32_t)kEmitMaskWords; q++) {  // independent reads (words past nw are 0)
                            const uint32_t m = s_mask[q * TW + t];
                            r += q < bw ? __popc(m) : (q == bw ? __popc(m & ((1u << (b & 31u)) - 1u)) : 0u);
                        }
                        if (FNX_EXP_EMIT == 12) {  // no scattered store
                            if (r == 0x7FFFFFFFu) point_list[0] = s_id[l];
                        } else if (FNX_EXP_EMIT == 13) {  // store, but lane-contiguous
                            point_list[(size_t)blk * 8192 + (size_t)k * kEmitThreads + tid + (r >> 30)] = s_id[l];
                        } else if (PAIRS) {
                            pair_list[s_cur[t] + r] = make_uint2(s_key[PAIRS ? l : 0], s_id[l]);
                        } else {
                            point_list[s_cur[t] + r] = s_id[l];
                        }
                    }
                lds_barrier();
                FNX_PH(5)
                for (int i = tid; i < tw1 - tw0; i += kEmitThreads) {
                    uint32_t n = 0;
#pragma unroll
                    for (int q = 0; q < kEmitMaskWords; q++) n += __popc(s_mask[q * TW + i]);  // words past nw are 0
                    if (n) {
                        s_cur[i] += n;
#pragma unroll
                        for (int q = 0; q < kEmitMaskWords; q++) s_mask[q * TW + i] = 0u;
                    }
                }
                (void)nw;
                lds_barrier();
                FNX_PH(6)
            }
            done += batch;
            l0 = l1;
#ifdef FNX_EXP_CLOCK
            n_sub++;
            n_inst += batch;
#endif
        }
    }
#ifdef FNX_EXP_CLOCK
    if (tid == 0) {
        const int wg = (int)ticket;
        if (wg < 16000) {
            g_emit_clock[4 * wg] = clk0;
            g_emit_clock[4 * wg + 1] = wall_clock64();
            g_emit_clock[4 * wg + 2] = n_sub;
            g_emit_clock[4 * wg + 3] = n_inst;
        }
        if (wg == 0 || wg == 200)
            for (int i = 0; i < 8; i++) g_emit_clock[4 * 16000 + (wg ? 8 : 0) + i] = ph[i];
    }
#endif
  }  // tickets
}

// ---------------------------------------------------------------------------------------------
void launch_tile_colscan(hipStream_t s, int T, int P, const uint16_t *blk_hist, uint32_t *blk_rel,
                         uint32_t *tile_count, int V, const ViewBatch &vb) {
    hipLaunchKernelGGL((colscan_kernel<uint16_t>), dim3((T + 63) / 64, V), dim3(1024), 0, s, T, splat_blocks(P),
                       blk_hist, blk_rel, tile_count, vb.geom, vb.img, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                       0, 0);
}

// `raw_keys` holds the depth keys (preprocess), `scratch` the geometry blob's sort_hist region; pairs_a / pairs_b
// are two P-entry (relative key, id) buffers (pairs_a may alias raw_keys: it is first written by the second pass).
// After the call the pairs in (depth bits, id) order are in pairs_b (three passes) or pairs_a (four: scratch ctl
// word SORT_CTL_WIDE is 1).
void launch_depth_sort(hipStream_t s, int P, const uint32_t *raw_keys, uint2 *pairs_a, uint2 *pairs_b,
                       uint32_t *scratch, const uint2 *rect, uint2 *rect_sorted, int V, const ViewBatch &vb, int narrow) {
    const int NSB = sort_blocks(P);
    const SortScratch L = sort_scratch(P);
    uint32_t *hist = scratch + L.hist, *hist_rel = scratch + L.hist_rel, *totals = scratch + L.totals,
             *ctl = scratch + L.ctl, *kmin_blk = scratch + L.kmin_blk, *kmax_blk = scratch + L.kmax_blk;
    uint2 *pin = pairs_a, *pout = pairs_b;
    for (int pass = 0; pass < (narrow ? 3 : 4); pass++) {
        hipLaunchKernelGGL(sort_hist_kernel, dim3(NSB, V), dim3(256), 0, s, P, raw_keys, pin, pass, hist, kmin_blk,
                           kmax_blk, ctl, vb.geom);
        hipLaunchKernelGGL((colscan_kernel<uint32_t>), dim3(kSortRadix / 64, V), dim3(1024), 0, s, kSortRadix, NSB, hist,
                           hist_rel, totals, vb.geom, vb.geom, pass == SORT_FIRST ? kmax_blk : (const uint32_t *)nullptr,
                           ctl, pass == SORT_FOURTH ? 1 : 0, narrow);
        hipLaunchKernelGGL(sort_scatter_kernel, dim3(NSB, V), dim3(256), 0, s, P, raw_keys, pin, pout, pass, hist_rel,
                           totals, ctl, rect, rect_sorted, vb.geom);
        uint2 *t = pin; pin = pout; pout = t;
    }
}

void launch_rank_hist(hipStream_t s, int P, int W, int H, const uint2 *rect_sorted, uint16_t *blk_hist,
                      uint32_t *blk_total, int V, const ViewBatch &vb) {
    const int gx = tiles_x(W), gy = tiles_y(H), T = gx * gy;
    hipLaunchKernelGGL(rank_hist_kernel, dim3(splat_blocks(P), V), dim3(256), (size_t)T * 4, s, P, T, rect_sorted, gx,
                       gy, blk_hist, blk_total, vb);
}

void launch_emit(hipStream_t s, int P, int W, int H, const uint2 *sorted3, const uint2 *sorted4,
                 const uint32_t *sort_ctl, const uint2 *rect_sorted, const uint32_t *starts, const uint32_t *blk_rel,
                 uint32_t *emit_ctl, const uint32_t *emit_items, uint32_t *point_list, uint32_t *header,
                 uint32_t capacity, int pairs, int V, const ViewBatch &vb) {
    const int gx = tiles_x(W), gy = tiles_y(H), T = gx * gy;
    const int TW = T < kEmitTileWindow ? T : kEmitTileWindow;
    const size_t lds = (size_t)TW * 4 * (kEmitMaskWords + 1);
    const void *kernel = pairs ? (const void *)emit_kernel<true> : (const void *)emit_kernel<false>;
    // resident workgroups for this LDS size (host-side queries, cached per kernel variant and device)
    const bool large = lds > 40 * 1024;  // static + dynamic LDS exceeds the default 64 KiB limit
    static int cache[2][2][kMaxDevices];
    static std::mutex mu;
    const int n_cu = device_cu_count();
    const int wgs = per_device_cached(cache[pairs ? 1 : 0][large ? 1 : 0], mu, [&](int) {
        if (large)
            (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      kEmitTileWindow * 4 * (kEmitMaskWords + 1));
        int per_cu = 0;
        const size_t lds_max = (size_t)kEmitTileWindow * 4 * (kEmitMaskWords + 1);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kEmitThreads,
                                                         large ? lds_max : (size_t)40 * 1024) !=
                hipSuccess || per_cu <= 0)
            per_cu = 1;
        (void)hipGetLastError();
        return n_cu * per_cu;
    });
    const int items_bound = splat_blocks(P) * V * kEmitBands;  // never more workgroups than items
    const dim3 grid(wgs < items_bound ? wgs : items_bound), block(kEmitThreads);
    if (pairs)
        hipLaunchKernelGGL(emit_kernel<true>, grid, block, lds, s, P, T, sorted3, sorted4, sort_ctl, rect_sorted, gx, gy,
                           starts, blk_rel, emit_ctl, emit_items, point_list, header, capacity, TW, V, vb);
    else
        hipLaunchKernelGGL(emit_kernel<false>, grid, block, lds, s, P, T, sorted3, sorted4, sort_ctl, rect_sorted, gx, gy,
                           starts, blk_rel, emit_ctl, emit_items, point_list, header, capacity, TW, V, vb);
}

// ---------------------------------------------------------------------------------------------
// Static-split mode, once per frame: after the static subset went through preprocess / sort / emit as a splat set
// of its own (local ids 0 .. P_static-1), pack what the per-iteration kernels need into the view's static blob:
// (depth bits, global id) pairs in list order, the exclusive tile prefix, radii and blend records.
__global__ void __launch_bounds__(256)
static_pack_kernel(int P, int T, const uint32_t *__restrict__ point_list, const uint32_t *__restrict__ dyn_start,
                   const uint32_t *__restrict__ header, const int *__restrict__ radii, const float4 *__restrict__ blend_rec,
                   char *__restrict__ blob, size_t blob_stride, size_t off_header, size_t off_starts, size_t off_radii,
                   size_t off_rec, size_t off_pairs, uint32_t id0, uint32_t r_capacity, const ViewBatch vb) {
    const int vw = blockIdx.y;
    point_list = view_at(point_list, vb.bin, vw);
    dyn_start = view_at(dyn_start, vb.img, vw);
    header = view_at(header, vb.img, vw);
    radii += (size_t)vw * P;  // the caller's [V, P_static] array of the static stage 1
    blend_rec = view_at(blend_rec, vb.geom, vw);
    blob += blob_stride * vw;
    uint32_t *o_header = reinterpret_cast<uint32_t *>(blob + off_header);
    uint32_t *o_starts = reinterpret_cast<uint32_t *>(blob + off_starts);
    int *o_radii = reinterpret_cast<int *>(blob + off_radii);
    float4 *o_rec = reinterpret_cast<float4 *>(blob + off_rec);
    uint2 *o_pairs = reinterpret_cast<uint2 *>(blob + off_pairs);
    const uint32_t R = header[HDR_NUM_RENDERED];
    const bool fits = R <= r_capacity && header[HDR_STATUS] == 0u;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i == 0) {
        o_header[SHDR_NUM_RENDERED] = fits ? R : 0u;
        o_header[SHDR_P] = (uint32_t)P;
        o_header[SHDR_ID0] = id0;
    }
    if (i <= (size_t)T) o_starts[i] = !fits ? 0u : (i < (size_t)T ? dyn_start[i] : R);
    if (i < (size_t)P) o_radii[i] = radii[i];
    if (i < 4 * (size_t)P) o_rec[i] = blend_rec[i];
    if (fits && i < (size_t)R) {
        const uint32_t id = point_list[i];
        o_pairs[i] = make_uint2(__float_as_uint(blend_rec[4 * (size_t)id + 1].w), id + id0);
    }
}

void launch_static_pack(hipStream_t s, int P, int W, int H, const uint32_t *point_list, const uint32_t *dyn_start,
                        const uint32_t *header, const int *radii, const float4 *blend_rec, char *blob,
                        size_t blob_stride, const fnx_static_layout_t &L, uint32_t id0, uint32_t r_capacity, int V,
                        const ViewBatch &vb) {
    const int T = tiles_x(W) * tiles_y(H);
    size_t n = (size_t)r_capacity;
    if (4 * (size_t)P > n) n = 4 * (size_t)P;
    if ((size_t)T + 1 > n) n = (size_t)T + 1;
    hipLaunchKernelGGL(static_pack_kernel, dim3((unsigned)((n + 255) / 256), V), dim3(256), 0, s, P, T, point_list,
                       dyn_start, header, radii, blend_rec, blob, blob_stride, L.header, L.starts, L.radii, L.blend_rec,
                       L.pairs, id0, r_capacity, vb);
}

}  // namespace fnx
