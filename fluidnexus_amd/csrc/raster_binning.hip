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
#include "lab/fnx_lab.h"  // experiment switches (all off in the production build)

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
// Pass 0 reads the raw keys and sorts key' = key - kmin (culled splats: their own depth, clamped, see relative_key; their
// rank is irrelevant, they emit nothing), where kmin is the smallest visible key of the view; the later passes read
// key' as scattered.
enum { SORT_FIRST = 0, SORT_MIDDLE = 1, SORT_THIRD = 2, SORT_FOURTH = 3 };

// Culled splats carry bit 31 (visible depths are positive floats above 0.2: bit 31 clear) over their own depth bits: they
// emit nothing, so any deterministic place in the order is correct, and keeping them AT THEIR DEPTH (clamped into the
// range three passes order) means a splat that enters or leaves a view does not jump through the whole order -- the
// temporal-coherence sort below relies on small displacements between consecutive calls.
__device__ __forceinline__ bool key_culled(uint32_t key) { return (key & 0x80000000u) != 0u; }
__device__ __forceinline__ uint32_t relative_key(uint32_t key, uint32_t kmin) {
    if (key_culled(key)) {
        const uint32_t d = key & 0x7FFFFFFFu;
        return min(d > kmin ? d - kmin : 0u, (1u << (3 * kSortBits)) - 1u);
    }
    return key - kmin;
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
                const uint32_t raw = raw_keys[i];
                key = relative_key(raw, kmin);
                if (!key_culled(raw)) kmax = max(kmax, key);  // the span that decides about the fourth pass: visible splats only
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
                    size_t geom_stride, char *__restrict__ coh_state, size_t coh_stride, SortStateLayout SL) {
    __shared__ uint32_t s_cnt[4][kSortRadix];   // per-wave digit counts, then per-wave running offsets
    __shared__ uint32_t s_wtot[4];
    bool with_rect = false;
    uint32_t kmin = 0, kmin_all = 0;  // kmin_all: for the samples of the final order (the pairs' keys are relative to it)
    {
        const int vw = blockIdx.y;
        ctl = view_at(ctl, geom_stride, vw);
        const bool wide = ctl[SORT_CTL_WIDE] != 0u;
        if (pass == SORT_FOURTH && !wide) return;
        if (pass == SORT_FIRST) kmin = ctl[SORT_CTL_KMIN];
        with_rect = (pass == SORT_THIRD && !wide) || pass == SORT_FOURTH;
        if (with_rect && coh_state) kmin_all = ctl[SORT_CTL_KMIN];
        raw_keys = view_at(raw_keys, geom_stride, vw);
        pairs_in = view_at(pairs_in, geom_stride, vw);
        pairs_out = view_at(pairs_out, geom_stride, vw);
        hist_rel = view_at(hist_rel, geom_stride, vw);
        digit_total = view_at(digit_total, geom_stride, vw);
        rect = view_at(rect, geom_stride, vw);
        rect_sorted = view_at(rect_sorted, geom_stride, vw);
        if (coh_state) coh_state += coh_stride * (size_t)vw;
    }
    // the pass that leaves the final order also seeds the caller's temporal-coherence state (rank of every splat)
    uint32_t *__restrict__ coh_inv = (coh_state && with_rect) ? reinterpret_cast<uint32_t *>(coh_state + SL.inv) : nullptr;
    if (coh_inv && blockIdx.x == 0 && threadIdx.x == 0) {
        uint32_t *hdr = reinterpret_cast<uint32_t *>(coh_state + SL.hdr);
        hdr[COH_MAGIC] = coh_magic(P);
        hdr[COH_ARRIVED] = 0u;
        hdr[COH_FAIL] = 0u;
        hdr[COH_SAMPLES_OK] = 1u;  // written below, rank by rank
        hdr[COH_NOUT] = 0u;
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
            if (coh_inv) {
                coh_inv[kv[k].y] = pos;
                // the samples the next (first repair) call's preprocess tells its outliers by: the depth bits at every
                // kCohSampleStep-th rank.  (With culled splats in the view this order -- culled last -- is not the repair
                // calls' order and the samples of the tail mean little: the first repair call then takes its full sort,
                // as it would without them.)
                if (pos % (uint32_t)kCohSampleStep == 0u)
                    reinterpret_cast<uint32_t *>(coh_state + SL.samples)[pos / (uint32_t)kCohSampleStep] = (kv[k].x + kmin_all) & 0x7FFFFFFFu;
            }
        }
        // the wave's LDS reads above are issued before this write (in-order per wave)
        if (valid[k] && (same & lt_mask) == 0ull) run_off[d] = run_off[d] + (uint32_t)__popcll(same);
    }
}


// ---------------------------------------------------------------------------------------------
// Temporal-coherence depth sort (sort_mode = FNX_SORT_COHERENT): ONE launch instead of nine, and no gather.
//
// Between two optimiser steps the positions move by ~1e-4 m, i.e. a splat's depth rank by a few hundred places in a
// 200 k-splat plume (1e6 ranks per metre of depth).  The caller's persistent state holds inv[id] = the splat's rank in
// the PREVIOUS call's depth order.  The preprocess kernel, which has the new depth and tile rectangle of splat id in
// registers, writes them as one 16-byte record (depth bits, id, rectangle, call stamp) to slot inv[id] of a per-call
// array: a scattered write that costs the bandwidth-bound kernel nothing measurable, and leaves the records in the
// previous order.  Workgroup c of this kernel then reads the window of kCohWin consecutive records around its chunk of
// kCohOut ranks with coalesced loads (measured before: gathering keys and rectangles by id cost 35 of 89 us), sorts
// the window in LDS, and writes the middle kCohOut positions: that is the chunk of the globally sorted order whenever
// no element has to travel further than kCohMargin ranks.
//
// The window sort is a sample sort: every 16th window element is a splitter (256 of them, ranked by counting), an
// element's bucket is the number of splitters below it (binary search), a bucket holds ~16 elements in any order, and
// the position inside the bucket is counted directly.  Its cost does not depend on how far the elements moved.  All
// comparisons are on (depth bits, id) as one u64: ties in depth order by id, no limit on the key span.  Every phase is a
// handful of dependent LDS round trips, so it is the number of waves in flight that sets the time (256 threads x 16
// elements measured 42 us for the sort phases alone): kCohThreads threads per workgroup, two workgroups per unit.
//
// The result is VERIFIED, not assumed.  A record counts only if it carries this call's stamp and an id in range, i.e.
// if the preprocess of THIS call wrote it (with the splat's current depth); every workgroup checks that its chunk is
// strictly increasing in (depth bits, id) and publishes its first and last element; the workgroup that arrives last
// checks the chunk boundaries.  N strictly increasing records of this call with ids in [0, P) are the sorted
// permutation -- whatever the state held.  If a check fails (a new frame, a large move, an unseeded state) that same
// workgroup sorts the view by itself from the preprocess' plain arrays (stable 8-bit LSD passes: ~2 ms for 200 k
// splats, rare and counted), so the call is always exact; (depth bits, id) is a total order, hence point_list is
// bit-identical to the radix path's (tests/test_coherent_sort_gpu.py).  Culled splats (bit 31 of the preprocess key)
// are ordered by their own depth bits in this mode: they emit nothing, and a splat that enters or leaves a view
// does not move in the order.
constexpr int kCohOut = 2 * kSplatBlock;
constexpr int kCohMargin = 1024;
constexpr int kCohWin = kCohOut + 2 * kCohMargin;
#ifndef FNX_COH_THREADS
#define FNX_COH_THREADS 512
#endif
constexpr int kCohThreads = FNX_COH_THREADS;
constexpr int kCohPer = kCohWin / kCohThreads;  // consecutive window positions per thread
constexpr int kCohMaxBucket = 256;              // a bucket larger than this fails the call (cannot happen with distinct elements in practice)
constexpr int kCohSampleThreads = 16 / kCohPer;  // threads per splitter: one splitter per 16 window positions
constexpr int kCohParts = kCohThreads / 256;     // the 256 samples are ranked by kCohParts threads each
static_assert(kCohPer * kCohSampleThreads == 16 && kCohParts * 256 == kCohThreads && kCohOut % kCohThreads == 0, "thread counts");
typedef unsigned long long u64;
// LDS layout of sort_repair_kernel: the output chunk | the bucketed window | splitters | bucket counts | starts | per
// bucketed position: its bucket, then its final window position (the window itself lives in registers)
constexpr int kCohE = kCohPer + 1;                   // elements per thread: its window positions + one outlier
constexpr int kCohCap = kCohWin + kCohOutlierCap;    // elements a workgroup sorts at most
static_assert(kCohOutlierCap <= kCohThreads && kCohE * kCohThreads >= kCohCap, "one outlier per thread");
constexpr size_t kCohLdsA = (size_t)kCohOut * 8;
constexpr size_t kCohLdsT = (size_t)kCohCap * 8;
constexpr size_t kCohLdsBytes = kCohLdsA + kCohLdsT + 256 * 8 + 2 * 272 * 4 + (size_t)kCohCap * 2;
static_assert(kCohLdsBytes + 256 <= 64 * 1024, "static LDS");
constexpr int kCohPosBias = 1024;  // final window positions are stored biased (holes in front of the window shift them below 0)

__device__ __forceinline__ uint32_t sort_key_bits(uint32_t raw) { return raw & 0x7FFFFFFFu; }  // depth bits, culled or not
__device__ __forceinline__ uint2 unpack_rect8(uint32_t r) {
    return make_uint2((r & 0xFFu) | (((r >> 8) & 0xFFu) << 16), ((r >> 16) & 0xFFu) | ((r >> 24) << 16));
}

// Whole-view stable LSD sort by the first 256 threads of ONE workgroup (the fallback above), on the absolute depth bits:
// raw -> pb -> pa -> pb -> pa, then pa is copied to pb (pa aliases the raw keys, which only the first pass reads).
// `lds`: >= 1288 words.
__device__ void coh_fallback_sort(int P, const uint32_t *raw, uint2 *pa, uint2 *pb, uint32_t *inv, const uint2 *rect,
                                  uint2 *rect_sorted, uint32_t *lds) {
    uint32_t *s_base = lds;              // [256] next free position of every digit
    uint32_t *s_cnt = lds + 256;         // [4][256] per-wave digit counts -> per-wave running offsets
    uint32_t *s_wt = lds + 256 + 1024;   // [4]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nchunks = (P + 1023) / 1024;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    for (int pass = 0; pass < 4; pass++) {
        const int shift = 8 * pass;
        const uint2 *src = (pass & 1) ? pb : pa;
        uint2 *dst = (pass & 1) ? pa : pb;
        s_base[tid] = 0u;
        __syncthreads();
        for (int i = tid; i < P; i += 256) {
            const uint32_t key = pass == 0 ? sort_key_bits(raw[i]) : src[i].x;
            atomicAdd(&s_base[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        {  // exclusive prefix over the 256 digits (thread = digit)
            const uint32_t cnt = s_base[tid];
            uint32_t inc = cnt;
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t v = (uint32_t)__shfl_up((int)inc, off);
                if (lane >= off) inc += v;
            }
            if (lane == 63) s_wt[w] = inc;
            __syncthreads();
            uint32_t ex = inc - cnt;
            for (int k = 0; k < w; k++) ex += s_wt[k];
            __syncthreads();
            s_base[tid] = ex;
        }
        for (int ch = 0; ch < nchunks; ch++) {
            const int base = ch * 1024 + w * 256;
#pragma unroll
            for (int k = 0; k < 4; k++) s_cnt[k * 256 + tid] = 0u;
            __syncthreads();
            uint2 kv[4];
            bool valid[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int i = base + k * 64 + lane;
                valid[k] = i < P;
                kv[k] = make_uint2(0u, 0u);
                if (valid[k]) {
                    kv[k] = pass == 0 ? make_uint2(sort_key_bits(raw[i]), (uint32_t)i) : src[i];
                    atomicAdd(&s_cnt[w * 256 + ((kv[k].x >> shift) & 255u)], 1u);
                }
            }
            __syncthreads();
            {
                uint32_t run = s_base[tid];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t c = s_cnt[k * 256 + tid];
                    s_cnt[k * 256 + tid] = run;
                    run += c;
                }
                s_base[tid] = run;
            }
            __syncthreads();
            volatile uint32_t *run_off = s_cnt + w * 256;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t d = (kv[k].x >> shift) & 255u;
                unsigned long long same = __ballot(valid[k]);
#pragma unroll
                for (int b = 0; b < 8; b++) {
                    const unsigned long long m = __ballot((d >> b) & 1u);
                    same &= ((d >> b) & 1u) ? m : ~m;
                }
                if (valid[k]) dst[run_off[d] + (uint32_t)__popcll(same & lt_mask)] = kv[k];
                // the wave's LDS reads above are issued before this write (in-order per wave)
                if (valid[k] && (same & lt_mask) == 0ull) run_off[d] = run_off[d] + (uint32_t)__popcll(same);
            }
            __syncthreads();
        }
        // this workgroup reads back what it wrote: stores out to L2, stale L1 lines dropped
        __threadfence();
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    for (int i = tid; i < P; i += 256) {
        const uint2 kv = pa[i];
        pb[i] = kv;
        inv[kv.y] = (uint32_t)i;
        rect_sorted[i] = rect[kv.y];
    }
}

// (defined with rank_hist_kernel below) per-(rank block, tile) instance counts of one block of kSplatBlock ranks
template <class RectAt>
__device__ __forceinline__ void rank_hist_block(int blk, int tid, int P, int T, int gx, int gy, RectAt rect_at,
                                                uint32_t *s_hist, uint32_t *s_total, uint16_t *__restrict__ blk_hist,
                                                uint32_t *__restrict__ blk_total, bool write);
// What the coherent sort needs to do rank_hist_kernel's work on its way (T == 0: not fused, the kernel is launched)
struct FusedHist {
    int T, gx, gy;
    uint16_t *blk_hist;   // view 0's arrays (stride: the geometry blob's)
    uint32_t *blk_total;
};
static_assert(kCohOut / kSplatBlock == kCohThreads / 256, "one 256-thread group per rank block of the chunk");
constexpr int kFusedHistMaxTiles = (int)(kCohLdsA / 4) / (kCohOut / kSplatBlock);  // the groups' histograms share the output chunk's LDS

__global__ void __launch_bounds__(kCohThreads)  // two workgroups per unit (60 KiB of LDS each)
sort_repair_kernel(int P, const uint32_t *__restrict__ raw_keys, const uint4 *__restrict__ krec,
                   uint2 *__restrict__ pairs_tmp, uint2 *__restrict__ pairs_out, uint32_t *__restrict__ scratch,
                   SortScratch L, const uint2 *__restrict__ rect, uint2 *__restrict__ rect_sorted,
                   char *__restrict__ state, size_t state_stride, SortStateLayout SL, size_t geom_stride, FusedHist fh) {
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[kCohLdsBytes];
    __shared__ uint32_t s_htot[kCohThreads / 256];
    u64 *s_out = reinterpret_cast<u64 *>(s_raw);                           // the output chunk is staged here
    u64 *s_t = reinterpret_cast<u64 *>(s_raw + kCohLdsA);                  // bucketed copy of the window
    u64 *s_split = reinterpret_cast<u64 *>(s_raw + kCohLdsA + kCohLdsT);   // [256] sorted sample of the window
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(s_raw + kCohLdsA + kCohLdsT + 256 * 8);  // [257] bucket sizes
    uint32_t *s_start = s_cnt + 272;                                        // [257] bucket starts (before that: sample ranks)
    uint16_t *s_bid = reinterpret_cast<uint16_t *>(s_start + 272);          // [kCohWin] bucket, then final position
    __shared__ uint32_t s_wsum[4];
    __shared__ uint32_t s_flag, s_last;
    __shared__ uint32_t s_holes_before, s_out_below;  // holes in front of the window; outliers below everything the window sorts
    const int vw = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int c = blockIdx.x, nc = gridDim.x;
    raw_keys = view_at(raw_keys, geom_stride, vw);
    krec = view_at(krec, geom_stride, vw);
    pairs_tmp = view_at(pairs_tmp, geom_stride, vw);
    pairs_out = view_at(pairs_out, geom_stride, vw);
    scratch = view_at(scratch, geom_stride, vw);
    rect = view_at(rect, geom_stride, vw);
    rect_sorted = view_at(rect_sorted, geom_stride, vw);
    state += state_stride * (size_t)vw;
    fh.blk_hist = view_at(fh.blk_hist, geom_stride, vw);
    fh.blk_total = view_at(fh.blk_total, geom_stride, vw);
    uint32_t *ctl = scratch + L.ctl;
    uint32_t *hdr = reinterpret_cast<uint32_t *>(state + SL.hdr);
    u64 *bounds = reinterpret_cast<u64 *>(state + SL.bounds);
    uint32_t *__restrict__ inv = reinterpret_cast<uint32_t *>(state + SL.inv);
    uint32_t *__restrict__ samples = reinterpret_cast<uint32_t *>(state + SL.samples);
    uint32_t *__restrict__ holes = reinterpret_cast<uint32_t *>(state + SL.holes);
    const uint4 *__restrict__ olist = reinterpret_cast<const uint4 *>(state + SL.olist);
    if (tid == 0) {
        s_flag = 0u;
        s_holes_before = 0u;
        s_out_below = 0u;
    }
    if (tid < 272) {
        s_cnt[tid] = 0u;
        s_start[tid] = 0u;
    }
    // a position nobody writes (impossible while every record is valid) must not pass for a splat: invalid id
#pragma unroll
    for (int k = 0; k < kCohOut / kCohThreads; k++) s_out[k * kCohThreads + tid] = ~0ull;
    const uint32_t magic = coh_magic(P);
    const bool seeded = hdr[COH_MAGIC] == magic;  // written only by the workgroup that arrives last, after everyone read it
    const uint32_t epoch = coh_stamp(hdr);        // the stamp this call's preprocess put on its records
    uint32_t bad = 0;  // COH_WHY bits
    const int n_out = min(kCohOut, P - c * kCohOut);
    if (seeded) {
        // ---- the window: previous ranks [g0, g0 + kCohWin), thread t owns four consecutive ones.  Ranks outside the
        // array take no part in the sort: the n_left window positions in front of rank 0 (chunk 0 only) shift every real
        // element's position, the ones behind rank P - 1 lie behind every output.
        const long long gt = (long long)c * kCohOut - kCohMargin + (long long)kCohPer * tid;
        const int n_left = (int)max(0ll, (long long)kCohMargin - (long long)c * kCohOut);
        uint4 rec[kCohE];  // [kCohPer]: the thread's outlier, if any
        bool real[kCohE];
#pragma unroll
        for (int k = 0; k < kCohPer; k++) {
            const long long g = gt + k;
            real[k] = g >= 0 && g < (long long)P;
            // UNCONDITIONAL loads (an index outside the array reads the nearest record, dropped below): under `if (real[k])`
            // every load sat in a branch of its own with s_waitcnt vmcnt(0) behind it -- eight memory round trips in a row at
            // the head of a latency-bound kernel (round 5)
            rec[k] = krec[min(max(g, 0ll), (long long)P - 1)];
        }

        // the call's outliers (records the preprocess kept out of the slots of their previous ranks): every workgroup
        // takes all of them, one per thread; those its window has no place for only count (below) or drop out (above)
        const uint32_t n_outl = min(hdr[COH_NOUT], (uint32_t)kCohOutlierCap);  // (candidates beyond the list's size stayed in their slots)
        real[kCohPer] = (uint32_t)tid < n_outl;
        rec[kCohPer] = olist[min((uint32_t)tid, (uint32_t)kCohOutlierCap - 1u)];  // (unconditional, as above)
#pragma unroll
        for (int k = 0; k < kCohE; k++)
            if (!real[k]) rec[k] = make_uint4(0u, 0u, 0u, 0u);
        uint32_t holes_part = 0;  // holes in front of the window's first rank (a multiple of 1024): that many fewer elements there
        for (int b = tid, nb = (int)(max(0ll, (long long)c * kCohOut - kCohMargin) >> 10); b < nb; b += kCohThreads)
            holes_part += holes[b];
        u64 v[kCohE];
#pragma unroll
        for (int k = 0; k < kCohE; k++) {
            if (real[k] && rec[k].w == epoch && rec[k].y == kCohHoleId && k < kCohPer) real[k] = false;  // its splat is in the outlier list
            if (real[k] && (rec[k].w != epoch || rec[k].y >= (uint32_t)P)) {  // not written by this call's preprocess
                bad |= 1u;
                real[k] = false;
            }
            v[k] = real[k] ? (((u64)rec[k].x << 32) | rec[k].y) : 0ull;
        }
        uint32_t at[kCohE];  // where the element lies in the bucketed copy
        if (FNX_EXP_COH & 4) {
#pragma unroll
            for (int k = 0; k < kCohPer; k++) {
                at[k] = (uint32_t)(kCohPer * tid + k);
                s_bid[at[k]] = (uint16_t)at[k];
                const int o = kCohPer * tid + k - kCohMargin;
                if (o >= 0 && o < kCohOut) s_out[o] = v[k];
            }
            __syncthreads();
        } else {
        // splitters: a REGULAR SAMPLE of the window (every 16th position), sorted by counting.  Consecutive sorted
        // samples are ~16 window elements apart whatever the disorder (the longest of 256 gaps stays near 16 ln 256),
        // so no bucket grows with the displacement.  A sample position without a real element contributes the largest
        // value (empty buckets).
        if (tid % kCohSampleThreads == 0) s_t[tid / kCohSampleThreads] = real[0] ? v[0] : ~0ull;  // s_t is free until the bucketed copy is written
        __syncthreads();
        if (holes_part) atomicAdd(&s_holes_before, holes_part);  // (zeroed before the barrier above, read after later ones)
        {
            constexpr int kSpan = 256 / kCohParts;
            const int j = tid & 255, first = (tid >> 8) * kSpan;  // sample j against samples [first, first + kSpan)
            const u64 mine = s_t[j];
            uint32_t r = 0;
#pragma unroll 2
            for (int i0 = 0; i0 < kSpan; i0 += 8) {
                u64 q[8];
#pragma unroll
                for (int i = 0; i < 8; i++) q[i] = s_t[first + i0 + i];
#pragma unroll
                for (int i = 0; i < 8; i++) r += (q[i] < mine || (q[i] == mine && first + i0 + i < j)) ? 1u : 0u;
            }
            atomicAdd(&s_start[j], r);
        }
        __syncthreads();
        if (tid < 256) s_split[s_start[tid]] = s_t[tid];
        __syncthreads();
        // bucket = number of splitters below the element: 0 .. 256
        // an outlier takes part in this window's sort if it lies between the window's smallest and largest sample (the
        // first / last window of the view: from the bottom / to the top); below that it counts into the positions
        // (every element of buckets >= 1 lies above it; bucket 0 sits in the lower margin), above that it is somebody else's
        if (real[kCohPer]) {
            u64 top = 0ull;  // largest sample that is a real element
            {
                int lo = 0, hi = 256;  // first of the "no element" samples (they sort last)
#pragma unroll
                for (int step = 0; step < 9; step++) {
                    const int mid = (lo + hi) >> 1;
                    if (lo < hi) {
                        if (s_split[mid] != ~0ull) lo = mid + 1;
                        else hi = mid;
                    }
                }
                top = lo > 0 ? s_split[lo - 1] : 0ull;
            }
            // (the last TWO windows reach the end of the order: an outlier that belongs there must not drop out of the one
            // whose chunk it may still touch)
            const bool first = c == 0, last = c >= nc - 2;
            if (!first && v[kCohPer] < s_split[0]) {
                atomicAdd(&s_out_below, 1u);
                real[kCohPer] = false;
            } else if (!last && v[kCohPer] > top) {
                real[kCohPer] = false;
            }
        }
        uint32_t bk[kCohE];
#pragma unroll
        for (int k = 0; k < kCohE; k++) {
            int lo = 0, hi = 256;
            if (k < kCohPer || real[k]) {  // (most waves hold no outlier)
#pragma unroll
                for (int step = 0; step < 9; step++) {
                    const int mid = (lo + hi) >> 1;
                    if (lo < hi) {
                        if (s_split[mid] < v[k]) lo = mid + 1;
                        else hi = mid;
                    }
                }
            }
            bk[k] = (uint32_t)lo;
        }
#pragma unroll
        for (int k = 0; k < kCohE; k++) at[k] = real[k] ? atomicAdd(&s_cnt[bk[k]], 1u) : 0u;  // slot inside the bucket
        __syncthreads();
        uint32_t n_mine = 0, inc = 0;
        if (tid < 256) {  // bucket starts: exclusive prefix of the sizes (thread = bucket)
            n_mine = s_cnt[tid];
            inc = n_mine;
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t tu = (uint32_t)__shfl_up((int)inc, off);
                if (lane >= off) inc += tu;
            }
            if (lane == 63) s_wsum[w] = inc;
        }
        __syncthreads();
        if (tid < 256) {
            uint32_t ex = inc - n_mine;
            for (int k = 0; k < w; k++) ex += s_wsum[k];
            s_start[tid] = ex;
            if (tid == 255) s_start[256] = ex + n_mine;  // elements above every splitter
            if (n_mine > (uint32_t)kCohMaxBucket || s_cnt[256] > (uint32_t)kCohMaxBucket) bad |= 2u;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kCohE; k++)
            if (real[k]) {
                at[k] += s_start[bk[k]];
                s_t[at[k]] = v[k];
                s_bid[at[k]] = (uint16_t)bk[k];
            }
        __syncthreads();
        // Position of every bucketed element = n_left + bucket start + elements of its bucket below it; the middle
        // kCohOut positions are the chunk.  Threads take the BUCKETED positions (lanes = consecutive positions: a wave
        // spans four or five buckets, so its lanes loop about equally long and read the same LDS words: broadcasts).
        const uint32_t n_real = s_start[256] + s_cnt[256];
        // rank of the window's first sorted element in the whole order = ranks in front of the window - their holes +
        // the outliers below the window
        const int shift = (int)s_out_below - (int)s_holes_before;
#pragma unroll
        for (int k = 0; k < kCohE; k++) {
            if ((uint32_t)(k * kCohThreads) >= n_real) break;  // (uniform: the last round exists for the outliers)
            const uint32_t p = (uint32_t)(k * kCohThreads + tid);
            const bool on = p < n_real;
            const uint32_t b = on ? s_bid[p] : 0u;
            const u64 x = on ? s_t[p] : 0ull;
            const uint32_t base = s_start[b], n = on ? min(s_cnt[b], (uint32_t)kCohMaxBucket) : 0u;
            uint32_t below = 0;
            for (uint32_t j = 0; j < n; j += 4) {
                const u64 q0 = s_t[base + j], q1 = s_t[base + j + 1], q2 = s_t[base + j + 2], q3 = s_t[base + j + 3];
                below += (q0 < x) ? 1u : 0u;
                below += (j + 1 < n && q1 < x) ? 1u : 0u;
                below += (j + 2 < n && q2 < x) ? 1u : 0u;
                below += (j + 3 < n && q3 < x) ? 1u : 0u;
            }
            const int fw = n_left + (int)(base + below) + shift;  // final window position
            const int o = fw - kCohMargin;
            if (on) {
                s_bid[p] = (uint16_t)(fw + kCohPosBias);  // for the element's owner (only this thread read the bucket stored here)
                if (o >= 0 && o < kCohOut) s_out[o] = x;
            }
        }
        __syncthreads();
        }  // FNX_EXP_COH & 4
        // the chunk, coalesced: (depth bits, id) pairs; strictly increasing?
#pragma unroll
        for (int k = 0; k < kCohOut / kCohThreads; k++) {
            const int o = k * kCohThreads + tid;
            if (o < n_out) {
                const u64 x = s_out[o];
                if ((uint32_t)x < (uint32_t)P) {
                    pairs_out[(size_t)c * kCohOut + o] = make_uint2((uint32_t)(x >> 32), (uint32_t)x);
                    if (o % kCohSampleStep == 0) samples[(c * kCohOut + o) / kCohSampleStep] = (uint32_t)(x >> 32);
                } else {
                    bad |= 4u;
                }
                if (o + 1 < n_out && !(x < s_out[o + 1])) bad |= 4u;
            }
        }
        // every element's owner still has its rectangle in registers: it goes to the element's final position in LDS (the
        // bucketed copy is dead by now), then out in rank order with coalesced stores.  (The splats' new ranks, inv[],
        // are written by the blend forward on its way: InvUpdate.)
        if (!(FNX_EXP_COH & 2)) {
            uint32_t *s_orect = reinterpret_cast<uint32_t *>(s_t);
#pragma unroll
            for (int k = 0; k < kCohE; k++) {
                if (!real[k]) continue;
                const int o = (int)s_bid[at[k]] - kCohPosBias - kCohMargin;
                if (o >= 0 && o < n_out) s_orect[o] = rec[k].z;
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < kCohOut / kCohThreads; k++) {
                const int o = k * kCohThreads + tid;
                if (o < n_out) rect_sorted[(size_t)c * kCohOut + o] = unpack_rect8(s_orect[o]);
            }
        }
        if (tid == 0) {
            bounds[2 * c] = s_out[0];
            bounds[2 * c + 1] = s_out[n_out - 1];
        }
        if (fh.T && !(FNX_EXP_COH & 2)) {
            // rank_hist_kernel's work for the chunk's rank blocks, from the rectangles in LDS: one 256-thread group per
            // block, their histograms where the output chunk was staged (it is out by now)
            __syncthreads();
            const int grp = tid >> 8, blk = c * (kCohOut / kSplatBlock) + grp;
            const uint32_t *s_orect = reinterpret_cast<const uint32_t *>(s_t);
            rank_hist_block(blk, tid & 255, P, fh.T, fh.gx, fh.gy,
                            [&](int rank) { return unpack_rect8(s_orect[rank - c * kCohOut]); },
                            reinterpret_cast<uint32_t *>(s_raw) + (size_t)grp * fh.T, &s_htot[grp], fh.blk_hist, fh.blk_total,
                            blk * kSplatBlock < P);
        }
    }
    if (FNX_EXP_COH) bad = 0;
    if (bad) atomicOr(&s_flag, bad);
    __syncthreads();
    if (tid == 0) {
        if (s_flag) atomicOr(&hdr[COH_FAIL], s_flag);
        __threadfence();
        s_last = (atomicAdd(&hdr[COH_ARRIVED], 1u) == (uint32_t)nc - 1u) ? 1u : 0u;
    }
    __syncthreads();
    // The tail below is the work of four waves.  The other four END here, and the barriers of the tail are then between the
    // remaining waves only: that is the documented behaviour of s_barrier on this ISA (terminated waves no longer count
    // towards a workgroup's barrier), which this library -- gfx950 only -- relies on here and nowhere else; HIP's portable
    // model would call a barrier behind a partial exit undefined.  Keeping eight waves alive through
    // coh_fallback_sort / rank_hist_block would mean predicating every phase of those 256-thread routines.
    if (!s_last || tid >= 256) return;
    // ---- the workgroup that arrives last: verify the chunk boundaries, repair by a full sort if need be, publish
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    uint32_t why = seeded ? 0u : 16u;
    if (seeded) {
        why |= __hip_atomic_load(&hdr[COH_FAIL], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int b = tid; b + 1 < nc; b += 256)
            if (!(bounds[2 * b + 1] < bounds[2 * b + 2])) why |= 8u;  // last of chunk b < first of chunk b + 1
        if (FNX_EXP_COH) why = 0u;
    }
    if (tid == 0) s_flag = 0u;
    __syncthreads();
    if (why) atomicOr(&s_flag, why);
    __syncthreads();
    const bool fail = s_flag != 0u;
    if (fail) {
        coh_fallback_sort(P, raw_keys, pairs_tmp, pairs_out, inv, rect, rect_sorted, reinterpret_cast<uint32_t *>(s_raw));
        __syncthreads();
        for (int j = tid; j * kCohSampleStep < P; j += 256) samples[j] = pairs_out[(size_t)j * kCohSampleStep].x;
        __threadfence();
        if (fh.T) {  // ... and the counts of every rank block again, from the order just written
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            for (int blk = 0; blk < (P + kSplatBlock - 1) / kSplatBlock; blk++)
                rank_hist_block(blk, tid, P, fh.T, fh.gx, fh.gy, [&](int rank) { return rect_sorted[rank]; },
                                reinterpret_cast<uint32_t *>(s_raw), &s_htot[0], fh.blk_hist, fh.blk_total, true);
        }
    }
    for (int b = tid; b < P / 1024 + 2; b += 256) holes[b] = 0u;  // every workgroup has read them: clean for the next call
    if (tid == 0) {
        hdr[COH_OUTLIERS] = hdr[COH_OUTLIERS] + min(hdr[COH_NOUT], (uint32_t)kCohOutlierCap);
        hdr[COH_NOUT] = 0u;
        hdr[COH_SAMPLES_OK] = 1u;     // samples[] and inv[] (the blend forward's job, from these pairs) describe this call's order
        ctl[SORT_CTL_KMIN] = 0u;      // this mode's pairs hold the depth bits themselves
        ctl[SORT_CTL_WIDE] = 0u;      // emit reads the order from the three-pass buffer
        ctl[SORT_CTL_OVERFLOW] = 0u;  // keys are compared whole: no span limit in this mode
        ctl[SORT_CTL_SPAN] = 0u;      // not measured here (the radix call that seeded the state reported it)
        hdr[COH_MAGIC] = magic;
        hdr[COH_EPOCH] = hdr[COH_EPOCH] + 1u;
        hdr[COH_ARRIVED] = 0u;
        hdr[COH_FAIL] = 0u;
        hdr[COH_REPAIRS] = hdr[COH_REPAIRS] + 1u;
        if (fail) {
            hdr[COH_FALLBACKS] = hdr[COH_FALLBACKS] + 1u;
            hdr[COH_WHY] = hdr[COH_WHY] | s_flag;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Per-(rank block, tile) instance counts.  A 256-thread workgroup owns kSplatBlock = 1024
// consecutive depth ranks (4 per thread).  The kernel is bound by LDS atomic throughput, so a splat
// does not add 1 to every tile of its rectangle: per tile row it adds +1 at the first column and -1
// just past the last one (2 atomics per row instead of one per tile), and the counts are the running
// sums along each tile row, formed once per block.
// The block's work as a device function: `tid` in [0, 256) of a 256-thread group whose threads all reach the same
// __syncthreads() (a whole workgroup, or one half of the coherent sort's 512-thread workgroup beside the other half);
// `rect_at(rank)` = tile rectangle of a rank of the block, s_hist: T words of LDS of the group's own, s_total: one word.
template <class RectAt>
__device__ __forceinline__ void rank_hist_block(int blk, int tid, int P, int T, int gx, int gy, RectAt rect_at,
                                                uint32_t *s_hist, uint32_t *s_total, uint16_t *__restrict__ blk_hist,
                                                uint32_t *__restrict__ blk_total, bool write) {
    if (tid == 0) *s_total = 0;
    for (int i = tid; i < T; i += 256) s_hist[i] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kSplatBlock / 256; k++) {
        const int rank = blk * kSplatBlock + k * 256 + tid;
        if (rank >= P) break;
        const uint2 r = rect_at(rank);
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
        const int lane = tid & 63, w = tid >> 6;
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
    uint16_t *row = blk_hist + (size_t)blk * T;
    uint32_t sum = 0;
    for (int i = tid; i < T; i += 256) {
        if (write) row[i] = (uint16_t)s_hist[i];
        sum += s_hist[i];
    }
    // instances of the whole block: emit splits heavy blocks over several workgroups
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += (uint32_t)__shfl_xor((int)sum, off);
    if ((tid & 63) == 0) atomicAdd(s_total, sum);
    __syncthreads();
    if (tid == 0 && write) blk_total[blk] = *s_total;
}

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
    rank_hist_block((int)blockIdx.x, (int)threadIdx.x, P, T, gx, gy, [&](int rank) { return rect_sorted[rank]; }, s_hist,
                    &s_total, blk_hist, blk_total, true);
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
// Round 5: the COUNTING form of an item (default; FNX_EMIT_COUNTING=0 keeps only the bitmask form, which also serves
// images whose bands do not fit one tile window).  Instead of 256-rank sub-batches with a bitmask per tile, a chunk of
// up to 6 x the tile window instances (a whole rank block of config 3) is placed by a counting sort over the tiles held in
// LDS: one LDS atomic per instance hands out a slot inside its tile's segment (arrival order), an exclusive scan of the
// per-tile counts places the segments, and every staged entry then counts the entries of its own segment with a lower rank
// -- its place in the tile's list -- and goes straight out.  Six barriers per ~4 600 instances instead of four per ~1 100,
// no per-sub-batch pass over all tiles' mask words.  Deterministic (the count fixes the order), bit-identical lists.
#ifndef FNX_EMIT_COUNTING
#define FNX_EMIT_COUNTING 1
#endif
// the counting form lays counts [TW] | starts [TW] | staged [6 TW] into the dynamic LDS and expects s_cur at 8 TW
// (= TW * kEmitMaskWords): fewer mask words would put the staging area over s_cur and past the allocation
static_assert(!FNX_EMIT_COUNTING || FNX_EMIT_MASK_WORDS >= 8, "the counting emit needs FNX_EMIT_MASK_WORDS >= 8");
constexpr int kCountPer = 12;          // instances per thread and chunk at most (registers)
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
        // unconditional loads at clamped ranks, all of the thread's in flight together (cf. sort_repair_kernel); the empty asm
        // keeps the compiler from sinking them back into the branches that use them, one memory round trip each
        uint2 kv_[kEmitPer], rect_[kEmitPer];
#pragma unroll
        for (int k = 0; k < kEmitPer; k++) {
            const int rank = min(blk * kSplatBlock + kEmitPer * tid + k, P - 1);
            kv_[k] = sorted_ids[rank];
            rect_[k] = rect_sorted[rank];
        }
#pragma unroll
        for (int k = 0; k < kEmitPer; k++) asm volatile("" : "+v"(kv_[k].x), "+v"(kv_[k].y), "+v"(rect_[k].x), "+v"(rect_[k].y));
#pragma unroll
        for (int k = 0; k < kEmitPer; k++) {
            const int rank = blk * kSplatBlock + kEmitPer * tid + k;
            id[k] = key[k] = 0;
            rc[k] = make_uint2(0u, 0u);
            const uint2 kv = kv_[k], rect = rect_[k];
            if (rank < P) {
                id[k] = kv.y;
                key[k] = kv.x + kmin;
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
        for (int i = tid; i < tw1 - tw0; i += kEmitThreads) s_cur[i] = starts[tw0 + i] + rel[tw0 + i];
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
        if (FNX_EMIT_COUNTING && whole && FNX_EXP_EMIT == 0) {
            // ---- counting form (see FNX_EMIT_COUNTING): s_dyn = counts [TW] | segment starts [TW] | staged entries [6 TW] | s_cur [TW]
            const int ntw = tw1 - tw0;
            uint32_t *s_cnt = s_dyn, *s_off = s_dyn + TW, *s_stage = s_dyn + 2 * (size_t)TW;
            const uint32_t cap = min((uint32_t)(6 * TW), (uint32_t)(kCountPer * kEmitThreads));
            for (int l0 = 0; done < total;) {
                // chunk [l0, l1): as many ranks as fit `cap` instances (at least one splat: a single splat's instances never
                // exceed the window, and TW <= cap)
                int lo = l0, hi = kSplatBlock;
                if (s_pre[hi - 1] - done <= cap) {
                    lo = hi;
                } else {
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        if (s_pre[mid] - done > cap) hi = mid; else lo = mid + 1;
                    }
                }
                const int l1 = lo;
                const uint32_t batch = s_pre[l1 - 1] - done;
                for (int i = tid; i < ntw; i += kEmitThreads) s_cnt[i] = 0u;
                lds_barrier();
                // the chunk's instances in (splat, row, column) order, dealt out in equal contiguous runs; one atomic per
                // instance: its slot inside the tile's segment (arrival order)
                const uint32_t run = (batch + (uint32_t)kEmitThreads - 1u) / (uint32_t)kEmitThreads;
                const uint32_t i0 = min(batch, (uint32_t)tid * run), i1 = min(batch, i0 + run);
                uint32_t ent[kCountPer], slot[kCountPer];
                {
                    InstanceWalk wk;
                    wk.load(s_rect, l0);
                    if (i0 < i1) wk.seek(s_pre, s_rect, l0, l1 - 1, done + i0, gx, tw0, tw1, true);
#pragma unroll
                    for (int k = 0; k < kCountPer; k++) {
                        ent[k] = 0xFFFFFFFFu;
                        slot[k] = 0u;
                        if (i0 + k < i1) {
                            const uint32_t t = (uint32_t)(wk.tile(gx) - tw0);
                            slot[k] = atomicAdd(&s_cnt[t], 1u);
                            ent[k] = (t << 10) | (uint32_t)wk.l;
                            if (i0 + k + 1 < i1) wk.next(s_rect, gx, tw0, tw1, true);
                        }
                    }
                }
                lds_barrier();
                // exclusive scan of the counts over the window's tiles -> segment starts (consecutive tiles per thread)
                {
                    const int per = (ntw + kEmitThreads - 1) / kEmitThreads;
                    const int b0 = tid * per, b1 = min(ntw, b0 + per);
                    uint32_t sum = 0;
                    for (int i = b0; i < b1; i++) sum += s_cnt[i];
                    uint32_t inc = sum;
                    for (int off = 1; off < 64; off <<= 1) {
                        const uint32_t v = (uint32_t)__shfl_up((int)inc, off);
                        if (lane >= off) inc += v;
                    }
                    if (lane == 63) s_wsum[w] = inc;
                    lds_barrier();
                    uint32_t at = inc - sum;
                    for (int k = 0; k < w; k++) at += s_wsum[k];
                    for (int i = b0; i < b1; i++) {
                        s_off[i] = at;
                        at += s_cnt[i];
                    }
                }
                lds_barrier();
#pragma unroll
                for (int k = 0; k < kCountPer; k++)
                    if (ent[k] != 0xFFFFFFFFu) s_stage[s_off[ent[k] >> 10] + slot[k]] = ent[k];
                lds_barrier();
                // an entry's place inside its segment = the number of the segment's entries with a lower rank (they share the
                // tile bits: whole words compare): counted by the entry's own thread, so a long segment -- a tile that takes
                // dozens of consecutive ranks -- costs every one of its entries a pass over it instead of one thread a
                // quadratic sort.  Consecutive threads hold consecutive staged entries: a segment goes out to one stretch of
                // its tile's list.
                for (uint32_t i = tid; i < batch; i += kEmitThreads) {
                    const uint32_t e = s_stage[i], t = e >> 10, l = e & 1023u;
                    const uint32_t b0 = s_off[t], n = s_cnt[t];
                    uint32_t r = 0;
                    for (uint32_t k = 0; k < n; k++) r += s_stage[b0 + k] < e ? 1u : 0u;
                    const uint32_t pos = s_cur[t] + r;
                    if (PAIRS) pair_list[pos] = make_uint2(s_key[PAIRS ? l : 0], s_id[l]);
                    else point_list[pos] = s_id[l];
                }
                lds_barrier();
                for (int i = tid; i < ntw; i += kEmitThreads) s_cur[i] += s_cnt[i];
                done += batch;
                l0 = l1;
#ifdef FNX_EXP_CLOCK
                n_sub++;
                n_inst += batch;
#endif
            }
            continue;  // the window (= the item's band) is done
        }
        // ---- bitmask form
        for (int i = tid; i < tw1 - tw0; i += kEmitThreads) {
#pragma unroll
            for (int q = 0; q < kEmitMaskWords; q++) s_mask[q * TW + i] = 0u;  // word-major: lanes = tiles, no bank conflicts
        }
        lds_barrier();
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
// `coh_state` (may be NULL): the caller's persistent temporal-coherence state, V x sort_state_layout(P).total bytes.
// coherent != 0: one launch of sort_repair_kernel over the records `krec` the preprocess left in the previous order;
// otherwise the radix passes, which leave the state seeded when it is given.  (coh_state: already aligned.)
void launch_depth_sort(hipStream_t s, int P, const uint32_t *raw_keys, uint2 *pairs_a, uint2 *pairs_b,
                       uint32_t *scratch, const uint2 *rect, uint2 *rect_sorted, int V, const ViewBatch &vb, int narrow,
                       char *coh_state, int coherent, const uint4 *krec, int W, int H, uint16_t *blk_hist,
                       uint32_t *blk_total, int *hist_done) {
    *hist_done = 0;
    const int NSB = sort_blocks(P);
    const SortScratch L = sort_scratch(P);
    const SortStateLayout SL = sort_state_layout(P);
    if (coherent && coh_state) {
        FusedHist fh{0, tiles_x(W), tiles_y(H), blk_hist, blk_total};
        if (fh.gx * fh.gy <= kFusedHistMaxTiles && blk_hist && blk_total) {
            fh.T = fh.gx * fh.gy;
            *hist_done = 1;  // the caller does not launch rank_hist_kernel
        }
        hipLaunchKernelGGL(sort_repair_kernel, dim3((P + kCohOut - 1) / kCohOut, V), dim3(kCohThreads), 0, s, P, raw_keys,
                           krec, pairs_a, pairs_b, scratch, L, rect, rect_sorted, coh_state, SL.total, SL, vb.geom, fh);
        return;
    }
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
                           totals, ctl, rect, rect_sorted, vb.geom, coh_state, SL.total, SL);
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
