// Binning of splat instances into per-tile, depth-ordered lists -- the MI355X-native replacement
// for the reference's key duplication + device-wide 64-bit radix sort + range detection
// (ch3/cuda_rasterizer/rasterizer_impl.cu:67-128,259-296).
//
// The reference sorts R = sum(tiles touched) 64-bit (tile | depth) keys.  Here only the P splats
// are sorted, once, by depth; the R instances are never sorted:
//
//   preprocess      (raster_forward.hip) counts, per block of 1024 splats, how many touch each tile
//                   -> blk_hist[block][tile]                                   (LDS atomics only)
//   tile_colscan    column prefix of that matrix -> blk_rel[block][tile], tile totals
//   tile_scan       (raster_forward.hip) tile totals -> [start,end) ranges, num_rendered
//   depth sort      stable LSD radix sort of (depth bits -> id), 4 x 8-bit passes, wave64 ballot
//                   ranking; gives every splat its rank in the (depth bits, id) order
//   emit            each splat block writes the ranks of its instances into its reserved slice of
//                   every tile's segment (slot = range start + blk_rel + LDS cursor; no global atomic)
//   tile_order      one workgroup per tile sets one bit per instance in an LDS bitmap over the
//                   ranks and walks the bitmap: the instances come out in depth order, O(n + P/32)
//                   per tile, no comparison sort.  160 KiB of LDS holds the bitmap for 1.2 M splats;
//                   beyond that the rank range is processed in windows.
//
// (depth bits, id) is a total order and the reference's radix sort is stable over keys emitted in
// id order, so the resulting lists are bit-identical to the reference's point_list.
#include "fnx_device.h"
#include "fnx_state.h"

namespace fnx {

// ---------------------------------------------------------------------------------------------
// Column prefix over splat blocks.  Workgroup = 64 tiles x 16 waves; wave w owns a contiguous
// band of blocks, lane = tile.  Pass 1 sums the band, LDS combines bands, pass 2 writes prefixes.
template <typename CountT>
__global__ void __launch_bounds__(1024)
colscan_kernel(int T, int NB, const CountT *__restrict__ blk_hist, uint32_t *__restrict__ blk_rel,
               uint32_t *__restrict__ tile_count, size_t hist_stride, size_t count_stride) {
    __shared__ uint32_t s_band[16][64];
    blk_hist = view_at(blk_hist, hist_stride, blockIdx.y);
    blk_rel = view_at(blk_rel, hist_stride, blockIdx.y);
    tile_count = view_at(tile_count, count_stride, blockIdx.y);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + lane;
    const int per = (NB + 15) / 16;
    const int b0 = w * per, b1 = min(NB, b0 + per);
    uint32_t sum = 0;
    if (t < T)
        for (int b = b0; b < b1; b++) sum += blk_hist[(size_t)b * T + t];
    s_band[w][lane] = sum;
    __syncthreads();
    uint32_t run = 0;
    for (int k = 0; k < w; k++) run += s_band[k][lane];
    if (t < T) {
        for (int b = b0; b < b1; b++) {
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
// Depth sort: LSD radix, 8-bit digits, stable.  Workgroup = 256 threads = 4 waves over a chunk of
// kSortChunk = 1024 keys; wave w owns keys [256 w, 256 w + 256) of the chunk in 4 steps of 64.
__global__ void __launch_bounds__(256)
sort_hist_kernel(int P, const uint32_t *__restrict__ keys, int shift, int NSB, uint32_t *__restrict__ hist,
                 size_t geom_stride) {
    __shared__ uint32_t s_h[256];
    keys = view_at(keys, geom_stride, blockIdx.y);
    hist = view_at(hist, geom_stride, blockIdx.y);
    s_h[threadIdx.x] = 0;
    __syncthreads();
    const int base = blockIdx.x * kSortChunk;
#pragma unroll
    for (int k = 0; k < kSortChunk / 256; k++) {
        const int i = base + k * 256 + threadIdx.x;
        if (i < P) atomicAdd(&s_h[(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(size_t)blockIdx.x * 256 + threadIdx.x] = s_h[threadIdx.x];  // block-major rows of 256 digits
}

// Stable scatter of one pass.  Rank of a key inside its 64-key step: lanes holding the same digit
// are found with 8 ballots; the rank is the number of lower lanes in that set.
__global__ void __launch_bounds__(256)
sort_scatter_kernel(int P, const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
                    uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out, int shift,
                    const uint32_t *__restrict__ hist_rel, const uint32_t *__restrict__ digit_total,
                    uint32_t *__restrict__ rank_of, int first_pass, int last_pass, size_t geom_stride) {
    __shared__ uint32_t s_cnt[4][256];   // per-wave digit counts, then per-wave running offsets
    {
        const int vw = blockIdx.y;
        keys_in = view_at(keys_in, geom_stride, vw);
        vals_in = view_at(vals_in, geom_stride, vw);
        keys_out = view_at(keys_out, geom_stride, vw);
        vals_out = view_at(vals_out, geom_stride, vw);
        hist_rel = view_at(hist_rel, geom_stride, vw);
        digit_total = view_at(digit_total, geom_stride, vw);
        rank_of = view_at(rank_of, geom_stride, vw);
    }
    __shared__ uint32_t s_wtot[4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int base = blockIdx.x * kSortChunk + w * 256;
#pragma unroll
    for (int k = 0; k < 4; k++) s_cnt[k][tid] = 0;
    __syncthreads();
    uint32_t key[4];
    bool valid[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int i = base + k * 64 + lane;
        valid[k] = i < P;
        key[k] = valid[k] ? keys_in[i] : 0xFFFFFFFFu;
        if (valid[k]) atomicAdd(&s_cnt[w][(key[k] >> shift) & 255u], 1u);
    }
    __syncthreads();
    // thread = digit: global base of the digit (exclusive scan of the 256 digit totals) + keys of
    // this digit in earlier chunks + earlier waves of this chunk
    {
        const uint32_t tot = digit_total[tid];
        uint32_t inc = tot;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = (uint32_t)__shfl_up((int)inc, off);
            if (lane >= off) inc += v;
        }
        if (lane == 63) s_wtot[w] = inc;
        __syncthreads();
        uint32_t dbase = inc - tot;
        for (int k = 0; k < w; k++) dbase += s_wtot[k];
        uint32_t run = dbase + hist_rel[(size_t)blockIdx.x * 256 + tid];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t c = s_cnt[k][tid];
            s_cnt[k][tid] = run;
            run += c;
        }
    }
    __syncthreads();
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    volatile uint32_t *run_off = s_cnt[w];  // updated by one lane, read by the others: keep it out of registers
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t d = (key[k] >> shift) & 255u;
        unsigned long long same = __ballot(valid[k]);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const unsigned long long m = __ballot((d >> b) & 1u);
            same &= ((d >> b) & 1u) ? m : ~m;
        }
        if (valid[k]) {
            const uint32_t r = (uint32_t)__popcll(same & lt_mask);
            const uint32_t pos = run_off[d] + r;
            const int i = base + k * 64 + lane;
            const uint32_t v = first_pass ? (uint32_t)i : vals_in[i];
            keys_out[pos] = key[k];
            vals_out[pos] = v;
            if (last_pass) rank_of[v] = pos;
        }
        // the wave's LDS reads above are issued before this write (in-order per wave)
        if (valid[k] && (same & lt_mask) == 0ull) run_off[d] = run_off[d] + (uint32_t)__popcll(same);
    }
}

// ---------------------------------------------------------------------------------------------
// Instance emission.  Same splat blocks as preprocess; LDS cursor per tile starts at the block's
// reserved offset inside the tile segment.
__global__ void __launch_bounds__(256)
emit_kernel(int P, int T, const float2 *__restrict__ means2D, const int *__restrict__ radii, int gx, int gy,
            const uint32_t *__restrict__ ranges, const uint32_t *__restrict__ blk_rel,
            const uint32_t *__restrict__ rank_of, uint32_t *__restrict__ bins, uint32_t *__restrict__ header,
            uint32_t capacity, const ViewBatch vb) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_cur[];
    {
        const int vw = blockIdx.y;
        means2D = view_at(means2D, vb.geom, vw);
        blk_rel = view_at(blk_rel, vb.geom, vw);
        rank_of = view_at(rank_of, vb.geom, vw);
        radii += (size_t)vw * P;
        ranges = view_at(ranges, vb.img, vw);
        header = view_at(header, vb.img, vw);
        bins = view_at(bins, vb.bin, vw);
    }
    if (header[HDR_NUM_RENDERED] > capacity) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            header[HDR_STATUS] = FNX_ERR_CAPACITY;
            header[HDR_CAPACITY] = capacity;
        }
        return;
    }
    const uint32_t *rel = blk_rel + (size_t)blockIdx.x * T;
    for (int i = threadIdx.x; i < T; i += 256) s_cur[i] = ranges[2 * i] + rel[i];
    __syncthreads();
    for (int k = 0; k < kSplatBlock / 256; k++) {
        const int idx = blockIdx.x * kSplatBlock + k * 256 + threadIdx.x;
        if (idx >= P) break;
        const int rad = radii[idx];
        if (rad > 0) {
            const float2 p = means2D[idx];
            int x0, y0, x1, y1;
            tile_rect(p.x, p.y, rad, gx, gy, x0, y0, x1, y1);
            const uint32_t rk = rank_of[idx];
            for (int y = y0; y < y1; y++)
                for (int x = x0; x < x1; x++) bins[atomicAdd(&s_cur[y * gx + x], 1u)] = rk;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Per-tile ordering through an LDS bitmap over depth ranks.
__global__ void __launch_bounds__(256)
tile_order_kernel(int P, const uint32_t *__restrict__ ranges, const uint32_t *__restrict__ bins,
                  const uint32_t *__restrict__ sorted_ids, uint32_t *__restrict__ point_list,
                  const uint32_t *__restrict__ header, uint32_t capacity, int win_words, const ViewBatch vb) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_bits[];  // win_words words + 4 wave totals
    {
        const int vw = blockIdx.y;
        ranges = view_at(ranges, vb.img, vw);
        header = view_at(header, vb.img, vw);
        bins = view_at(bins, vb.bin, vw);
        point_list = view_at(point_list, vb.bin, vw);
        sorted_ids = view_at(sorted_ids, vb.geom, vw);
    }
    __shared__ uint32_t s_wave[4];
    if (header[HDR_NUM_RENDERED] > capacity) return;
    const uint32_t start = ranges[2 * blockIdx.x], end = ranges[2 * blockIdx.x + 1];
    const uint32_t n = end - start;
    if (n == 0) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int words_total = (P + 31) >> 5;
    // words per thread, odd => consecutive threads hit distinct LDS banks
    const int wpt = ((win_words + 255) / 256) | 1;
    uint32_t out = start;
    for (int wbase = 0; wbase < words_total; wbase += win_words) {
        const int nw = min(win_words, words_total - wbase);
        for (int i = tid; i < nw; i += 256) s_bits[i] = 0u;
        __syncthreads();
        const uint32_t lo = (uint32_t)wbase << 5, hi = lo + ((uint32_t)nw << 5);
        for (uint32_t i = tid; i < n; i += 256) {
            const uint32_t r = bins[start + i];
            if (r >= lo && r < hi) atomicOr(&s_bits[(r - lo) >> 5], 1u << (r & 31u));
        }
        __syncthreads();
        const int w0 = tid * wpt, w1 = min(nw, w0 + wpt);
        uint32_t cnt = 0;
        for (int i = w0; i < w1; i++) cnt += __popc(s_bits[i]);
        // workgroup exclusive scan of cnt: wave scan with shuffles, then 4 wave totals
        uint32_t inc = cnt;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = (uint32_t)__shfl_up((int)inc, off);
            if (lane >= off) inc += v;
        }
        if (lane == 63) s_wave[w] = inc;
        __syncthreads();
        uint32_t pre = inc - cnt;
        for (int k = 0; k < w; k++) pre += s_wave[k];
        const uint32_t total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        uint32_t o = out + pre;
        for (int i = w0; i < w1; i++) {
            uint32_t m = s_bits[i];
            const uint32_t rbase = lo + ((uint32_t)i << 5);
            while (m) {
                const int b = __ffs((int)m) - 1;
                point_list[o++] = sorted_ids[rbase + b];
                m &= m - 1u;
            }
        }
        out += total;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
void launch_tile_colscan(hipStream_t s, int T, int P, const uint16_t *blk_hist, uint32_t *blk_rel,
                         uint32_t *tile_count, int V, const ViewBatch &vb) {
    hipLaunchKernelGGL((colscan_kernel<uint16_t>), dim3((T + 63) / 64, V), dim3(1024), 0, s, T, splat_blocks(P),
                       blk_hist, blk_rel, tile_count, vb.geom, vb.img);
}

// keys0 holds the depth keys; after the call vals0 = ids in (depth bits, id) order, rank_of = inverse.
// hist: u32[NSB*256] chunk histograms, hist_rel: u32[NSB*256] their prefix over chunks, totals: u32[256].
void launch_depth_sort(hipStream_t s, int P, uint32_t *keys0, uint32_t *keys1, uint32_t *vals0, uint32_t *vals1,
                       uint32_t *hist, uint32_t *hist_rel, uint32_t *totals, uint32_t *rank_of, int V,
                       const ViewBatch &vb) {
    const int NSB = sort_blocks(P);
    uint32_t *kin = keys0, *kout = keys1, *vin = vals0, *vout = vals1;
    for (int pass = 0; pass < 4; pass++) {
        const int shift = pass * 8;
        hipLaunchKernelGGL(sort_hist_kernel, dim3(NSB, V), dim3(256), 0, s, P, kin, shift, NSB, hist, vb.geom);
        hipLaunchKernelGGL((colscan_kernel<uint32_t>), dim3(4, V), dim3(1024), 0, s, 256, NSB, hist, hist_rel, totals,
                           vb.geom, vb.geom);
        hipLaunchKernelGGL(sort_scatter_kernel, dim3(NSB, V), dim3(256), 0, s, P, kin, vin, kout, vout, shift, hist_rel,
                           totals, rank_of, pass == 0 ? 1 : 0, pass == 3 ? 1 : 0, vb.geom);
        uint32_t *t = kin; kin = kout; kout = t;
        t = vin; vin = vout; vout = t;
    }
    // 4 passes: the result is back in (keys0, vals0)
}

void launch_emit(hipStream_t s, int P, int W, int H, const float2 *means2D, const int *radii, const uint32_t *ranges,
                 const uint32_t *blk_rel, const uint32_t *rank_of, uint32_t *bins, uint32_t *header,
                 uint32_t capacity, int V, const ViewBatch &vb) {
    const int gx = tiles_x(W), gy = tiles_y(H), T = gx * gy;
    hipLaunchKernelGGL(emit_kernel, dim3(splat_blocks(P), V), dim3(256), (size_t)T * 4, s, P, T, means2D, radii, gx, gy,
                       ranges, blk_rel, rank_of, bins, header, capacity, vb);
}

void launch_tile_order(hipStream_t s, int P, int T, const uint32_t *ranges, const uint32_t *bins,
                       const uint32_t *sorted_ids, uint32_t *point_list, const uint32_t *header, uint32_t capacity,
                       int V, const ViewBatch &vb) {
    const int words_total = (P + 31) >> 5;
    const int kMaxWinWords = 36 * 1024;  // 144 KiB bitmap window (1.18 M ranks)
    const int win_words = words_total < kMaxWinWords ? words_total : kMaxWinWords;
    static bool attr_set = false;
    if (!attr_set) {  // allow > 64 KiB of dynamic LDS for the bitmap window
        (void)hipFuncSetAttribute((const void *)tile_order_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  kMaxWinWords * 4);
        attr_set = true;
    }
    hipLaunchKernelGGL(tile_order_kernel, dim3(T, V), dim3(256), (size_t)win_words * 4, s, P, ranges, bins, sorted_ids,
                       point_list, header, capacity, win_words, vb);
}

}  // namespace fnx
