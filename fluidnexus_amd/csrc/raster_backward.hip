// Backward pass of the gfx950 Gaussian rasteriser.
//
//   blend_backward : per batch of 256 entries of a tile's list, front to back, pixel gradients -> per-splat
//                    screen-space gradients (ch3 backward.cu:384-536)
//   geom_backward  : per splat, screen-space gradients -> means / cov3D / scale / rotation / SH
//                    (ch3 backward.cu:137-263 computeCov2DCUDA fused with :332-381 preprocessCUDA;
//                    same arithmetic, same order of the three mean-gradient terms)
//
// The reference issues 6+C global fp32 atomics per contributing (pixel, splat) pair.  Here each
// pair's contribution is first summed across the 16 lanes of its 4x4 block with DPP row operations,
// then across the tile's 16 blocks in LDS, and one set of atomics per (tile, batch, splat) reaches HBM:
// ~256x fewer device-scope atomics, same sums up to fp32 association order.
#include "fnx_device.h"
#include "fnx_fold_asm.h"
#include "fnx_state.h"

#ifndef FNX_LDS_BARRIER
#define FNX_LDS_BARRIER 1  // 0: plain __syncthreads() inside the item loop (timing experiments)
#endif
#if FNX_LDS_BARRIER
#define FNX_LOOP_BARRIER() fnx::lds_barrier()
#else
#define FNX_LOOP_BARRIER() __syncthreads()
#endif
#ifdef FNX_EXP_WG_ATOMICS  // timing experiment: flush with workgroup-scope (XCD-local L2) atomics -- WRONG sums across XCDs
#define FNX_FLUSH_ADD(p, v) __hip_atomic_fetch_add((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#else
#define FNX_FLUSH_ADD(p, v) unsafeAtomicAdd((p), (v))
#endif
#include "lab/fnx_lab.h"  // experiment switches (all off in the production build)
#include <cstdlib>
#ifndef FNX_LANES_DUAL
#define FNX_LANES_DUAL 1  // the dual mode through the entries-as-lanes kernel too (0: always the row form)
#endif
#ifndef FNX_BWD_FORM_DEFAULT
#define FNX_BWD_FORM_DEFAULT 1  // 1: entries as lanes (raster_backward_lanes.h), 0: pixels as lanes (this file); the dual mode always takes 0
#endif

#ifdef FNX_EXP_BCLK  // developer timing: per-phase cycles of wave 0 / lane 0 of every workgroup, summed over the launch
__device__ unsigned long long g_bwd_clock[32];
extern "C" int fnx_debug_bwd_clock(unsigned long long *host, int reset) {
    if (reset) {
        unsigned long long z[32] = {0};
        return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_bwd_clock), z, sizeof(z));
    }
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_bwd_clock), sizeof(g_bwd_clock));
}
#define FNX_BCLK(i) { const unsigned long long tn = clock64(); if (tid == 0) atomicAdd(&g_bwd_clock[i], tn - t_last); t_last = tn; }
#else
#define FNX_BCLK(i)
#endif

namespace fnx {

// DPP row operations (gfx9 encodings): quad permutes 0xb1 / 0x4e, row_shr:4 / :8 = 0x114 / 0x118, row_ror:8 = 0x128,
// row_half_mirror = 0x141.
#define FNX_DPP(v, ctrl, rmask) \
    __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), (rmask), 0xf, false))
// Cross-lane sums of the blend backward: every 16-lane row of the wave holds the NV values of a DIFFERENT list entry
// (its 4x4 block's); each row's sums go to acc[v][slot] of that row's entry (`slot` is per lane, uniform within a row).  Rows with no
// active lane add nothing.  Four values are folded together: 16 -> 8 lanes pairs (a | b) and (c | d) across the row
// halves (row_ror:8), 8 -> 4 pairs the pairs across the half-row's quads (row_half_mirror), two quad permutes finish:
// the quads of a row then hold the totals of a, c, b, d -- 11 VALU operations and ONE LDS atomic (4 lanes per row)
// for four values instead of 4 x (5 + 1).
template <int NV, int STRIDE>
__device__ __forceinline__ void row_fold_accumulate(const float (&val)[NV], float (*acc)[STRIDE], uint32_t slot, int lane,
                                                    bool row_active) {
    const bool hi8 = (lane & 8) != 0, hi4 = (lane & 4) != 0;
    const int vq = ((lane >> 3) & 1) | (((lane >> 2) & 1) << 1);  // which value of a group this lane's quad ends up with
    const bool writer = row_active && (lane & 3) == 0;
#pragma unroll
    for (int g = 0; g + 4 <= NV; g += 4) {
        const float a = val[g], b = val[g + 1], c = val[g + 2], d = val[g + 3];
        const float s01 = (hi8 ? b : a) + FNX_DPP((hi8 ? a : b), 0x128, 0xf);
        const float s23 = (hi8 ? d : c) + FNX_DPP((hi8 ? c : d), 0x128, 0xf);
        float t = (hi4 ? s23 : s01) + FNX_DPP((hi4 ? s01 : s23), 0x141, 0xf);
        t += FNX_DPP(t, 0xb1, 0xf);
        t += FNX_DPP(t, 0x4e, 0xf);
        if (writer) atomicAdd(&acc[g + vq][slot], t);
    }
#pragma unroll
    for (int v = NV & ~3; v < NV; v++) {  // the one to three values left over: a plain row sum each
        float t = val[v];
        t += FNX_DPP(t, 0x128, 0xf);
        t += FNX_DPP(t, 0x141, 0xf);
        t += FNX_DPP(t, 0xb1, 0xf);
        t += FNX_DPP(t, 0x4e, 0xf);
        if (writer && (lane & 15) == 4 * (v & 3)) atomicAdd(&acc[v][slot], t);
    }
}

// Gradient pass over the tile lists.  The reference walks a tile's list back to front, one workgroup per tile,
// rebuilding T by division and the colour behind an entry by a recurrence (backward.cu:384-536): its running time is
// the LENGTH of the deepest list, and a scene whose lists do not saturate early (a semi-transparent plume: thousands
// of contributing entries per pixel) leaves most of the chip idle behind a few hundred sequential walks.  Here the
// unit of work is one BATCH of 256 list entries (fnx_state.h, kBlendBatch): the forward stored the per-pixel
// (T, accumulated colour) in front of every batch and appended one work item per batch that holds a contributor, so
// the batches of a tile are independent workgroups.  Inside a batch the entries are walked FRONT to back with the
// forward's own arithmetic (T and the colour prefix come out bit-identical to the forward's), and
//     dL/dalpha_i = sum_ch dL_ch (c_i T_i - S_i / (1 - alpha_i)) - T_final / (1 - alpha_i) (bg . dL),
//     S_i = colour accumulated BEHIND entry i = final accumulated colour - prefix including i,
// which is the reference's (c - accum_rec) T with accum_rec = S_i / T_{i+1} written without the recurrence.
// Same structure as the forward blend: a wave is an 8x8 quadrant, each of its 16-lane rows a 4x4 block with its own
// compacted list (block_mask_exact), the four rows walk four lists at once, and the per-entry sums are row
// reductions; a block only receives entries in front of its own deepest contributor.
// MODE 0: all screen-space gradients (mean2D.xy | conic.xyw | opacity | colour[C]);
// MODE 1: geometry only (mean2D.xy | conic.xyw) -- the caller does not need opacity / colour gradients;
// MODE 2: fixed positions (conic.xyw | opacity | colour[C]) -- the caller does not need the gradient of the 2D means
//         (the visual-particle stage: positions are not optimised);
// MODE 3: positions only -- the sums of MODE 1, but the flush carries them through the geometry backward of its
//         (splat, view) (geom_backward_view: linear in them) and adds the result to dL/dmean3D, summed over the views:
//         3 global atomics per (tile, batch, splat) instead of 5, no per-view arrays, no geometry kernel behind it.
// Splats with id >= grad_limit still take part in the blend recurrences but produce no gradient.
#ifndef FNX_BWD_WAVES
#define FNX_BWD_WAVES 4  // waves per SIMD the register allocation of the blend backward aims at
#endif
template <bool WANT_COV = true>
__device__ inline void geom_backward_view(const float3 mean, const float *cov3D, const float *view, const float *proj,
                                          float h_x, float h_y, float tan_fovx, float tan_fovy, float gc0, float gc1,
                                          float gc2, float g0, float g1, float *gm, float *dcv);

// FAST: the arithmetic of the forward's fast mode (blend_forward_kernel: pre-scaled coefficients, log2(o) in the
// exponent, one v_exp_f32), so that both passes see the same alphas; the per-entry sums of four consecutive entries are
// folded together, one moment at a time (five 4-value folds per four entries instead of four 4 + 1 folds).
// DUAL (fnx_raster_dual_t; C = 3, MODE 3): the forward blended a second, single-channel image over the dynamic entries in
// the same pass (blend_forward_kernel); its loss gradient dL/dpixel1 enters every dynamic entry's dL/dalpha through the
// second image's own transmittance / "what lies behind" recurrences, evaluated in the same walk on the same alphas:
//     dL/dalpha_i = [first image's term] + [T1_i (c0_i dL1) - rest1_i / (1 - alpha_i)]   (dynamic entries in front of the
//     second image's last contributor), and the moments / flush carry the sum.
template <int C, int MODE, bool FAST, bool DUAL = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(FNX_BWD_WAVES, FNX_BWD_WAVES)))
blend_backward_kernel(int T, int gx, const uint32_t *__restrict__ ranges, const uint32_t *__restrict__ point_list, int W,
                      int H, const float *__restrict__ bg, const float4 *__restrict__ blend_rec,
                      const float *__restrict__ final_Ts, const uint32_t *__restrict__ n_contrib,
                      const float *__restrict__ acc_final, const float *__restrict__ dL_dpixels,
                      float *__restrict__ dL_dmean2D, float *__restrict__ dL_dconic, float *__restrict__ dL_dopacity,
                      float *__restrict__ dL_dcolors, const uint32_t *__restrict__ header, uint32_t capacity,
                      uint32_t grad_limit, int P, int n_views, const StaticRef st, const ViewBatch vb,
                      const float *__restrict__ means3D, const float *__restrict__ cov3Ds, size_t cov3D_stride,
                      const float *__restrict__ viewmatrix, const float *__restrict__ projmatrix,
                      float *__restrict__ dL_dmean3D, uint32_t *__restrict__ status_out, const DualRef du) {
    static_assert(!DUAL || (C == 3 && MODE == 3), "dual mode: 3 channels, positions-only backward");
    constexpr uint32_t kListStatic = 0x8000u, kListOffMask = 0x1FFFu;  // as in the forward's lists
    constexpr bool kMeans = MODE != 2, kAppearance = MODE == 0 || MODE == 2, kFusedGeom = MODE == 3;
    constexpr int kConic = kMeans ? 2 : 0, kOpac = kConic + 3, kCol = kOpac + 1;  // slots of the per-entry sums
    constexpr int NV = kAppearance ? kCol + C : kOpac;
#ifndef FNX_BWD_GROUP
#define FNX_BWD_GROUP 4
#endif
    constexpr int kGroup = FNX_BWD_GROUP;  // list entries per step of the gradient loop
    constexpr int kListStride = (256 + kGroup + 7) & ~7;
    // staged batch: three 16-byte records per slot; slot 256 is a NULL record (opacity 0: alpha = 0, no gradient) that
    // the tail of every block list points to, so the loop needs no end-of-list test per entry
    __shared__ uint32_t s_id[256];
    __shared__ float4 s_ra[257];  // x, y, conic a, conic b
    __shared__ float4 s_rb[257];  // conic c, opacity, -, 1.0 if the splat wants gradients (id < grad_limit) else 0.0
    __shared__ float4 s_rc[257];  // colour (C channels)
    __shared__ float4 s_rd[FAST ? 256 : 1];  // FAST: the entry's own conic and opacity for the flush (a, b, c, o)
    // per-entry sums; FAST pads the rows so that the four values a quad adds at once fall into different banks
    constexpr int kAccStride = FAST ? 272 : 256;
    __shared__ float s_acc[NV][kAccStride];
    // per-block lists of LDS byte offsets (slot * 16), depth order; lists 4 w .. 4 w + 3 are built, padded and read by
    // wave w alone
    __shared__ __attribute__((aligned(16))) uint16_t s_list[16][kListStride];
    __shared__ uint16_t s_mask[256];
    __shared__ __attribute__((aligned(16))) uint32_t s_max[16];
    __shared__ uint32_t s_first[kMaxViews + 1];  // ticket of every view's first work item; [n_views] = all items
    __shared__ float s_pfv[kFusedGeom ? 4 : 1][192];   // positions-only flush: the gradients change lanes here
    __shared__ uint32_t s_pfi[kFusedGeom ? 4 : 1][64];
    __shared__ unsigned long long s_dynmask[DUAL ? 4 : 1];  // DUAL: per staging wave, which slots hold dynamic entries
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;  // ch3 backward.cu:444-445
    if (tid == 0) {
        s_ra[256] = make_float4(0.f, 0.f, 0.f, 0.f);
        s_rb[256] = FAST ? make_float4(0.f, -200.0f, 0.f, 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
        s_rc[256] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // The work items of ALL views form one queue (view 0's first); workgroup b takes items b, b + grid, b + 2 grid, ...,
    // so every workgroup samples the whole queue (all views, shallow and deep tiles alike) and the launch is sized to
    // the workgroups that are resident at a time (launch_blend_backward).  Handing the items out in queue order through
    // an atomic ticket measured 7 % slower (neighbouring tiles' batches then run at the same time and meet on the same
    // splats' accumulators).  A view whose forward overflowed its binning capacity contributes no items.
    // (thread v reads view v's header words, all five in flight together, thread 0 adds the views up behind a barrier: read
    //  by one thread, view after view, with the short-circuit conditions below, they were twenty-five memory round trips in a
    //  row at the head of every workgroup -- round 5)
    __shared__ uint32_t s_view_items[kMaxViews];
    if (tid < n_views) {
        const int v = tid;
        const uint32_t *h = view_at(header, vb.img, v);
        const uint32_t h_limit = h[HDR_DYN_LIMIT], h_cap = h[HDR_BIN_CAPACITY], h_nr = h[HDR_NUM_RENDERED],
                       h_status = h[HDR_STATUS], h_items = h[HDR_BWD_ITEMS];
        // a view whose forward overflowed, or ran with another binning capacity than this call's (the blob's layout
        // depends on it), contributes nothing; the mismatch is left in the view's status word (fnx_read_status)
        // ... or whose forward laid the work items and walking limits down for a SMALLER gradient limit than this call's
        // (fnx_request_gradient_limit; HDR_DYN_LIMIT): splats this call differentiates would be cut off
        // (dual mode: the second image was blended over ids < HDR_DYN_LIMIT; this call must differentiate exactly those)
        const bool cut = grad_limit > h_limit || (DUAL && grad_limit != h_limit);
        const bool mismatch = h_cap != capacity || cut;
        if (mismatch && blockIdx.x == 0) {
            const_cast<uint32_t *>(h)[HDR_STATUS] = cut ? FNX_ERR_INVALID_ARG : FNX_ERR_CAPACITY;
            // ... and in the caller's status row of the view, where a deferred check looks (a refused backward
            // returns zero gradients: it must not pass unnoticed)
            if (status_out) status_out[8 * v + HDR_STATUS] = cut ? FNX_ERR_INVALID_ARG : FNX_ERR_CAPACITY;
        }
        s_view_items[v] = (mismatch || h_nr > capacity || h_status != 0u) ? 0u : h_items;
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        for (int v = 0; v < n_views; v++) {
            s_first[v] = run;
            run += s_view_items[v];
        }
        s_first[n_views] = run;
    }
    // An item starts with a chain of dependent reads (item -> tile range -> ids -> records), ~0.7 us each, against
    // ~10 us of blending: the chain of the NEXT item runs under the current one (its descriptor and range are read at
    // the top, its ids and masks before the walk, its records behind the walk), and an item finds its inputs in
    // registers (config 3: 527 -> 511 us).
    struct Fetched {
        uint32_t item;  // descriptor, kNoItem behind the end of the queue
        int vw;
        uint32_t r0, r1;  // the tile's list
        uint32_t id, qm;  // entry q0 + tid of the batch and its block mask (valid: r0 + q0 + tid < r1)
        float4 ra;
        float rbx, rby, rcz, rcw, rdx;
    };
    constexpr uint32_t kNoItem = 0xFFFFFFFFu;
    auto fetch_item = [&](uint32_t t, Fetched &f) {
        f.item = kNoItem;
        f.vw = 0;
        if (t < s_first[n_views]) {
            while (t >= s_first[f.vw + 1]) f.vw++;
            const uint32_t *items = reinterpret_cast<const uint32_t *>(
                reinterpret_cast<const char *>(view_at(point_list, vb.bin, f.vw)) + vb.bin_items);
            f.item = items[t - s_first[f.vw]];
        }
    };
    auto fetch_range = [&](Fetched &f) {
        f.r0 = f.r1 = 0;
        if (f.item != kNoItem) {
            const uint2 rg = reinterpret_cast<const uint2 *>(view_at(ranges, vb.img, f.vw))[f.item & kItemTileMask];
            f.r0 = rg.x;
            f.r1 = rg.y;
        }
    };
    auto fetch_ids = [&](Fetched &f) {
        f.id = 0;
        f.qm = 0;
        const uint32_t pos = f.r0 + ((f.item >> kItemTileBits) << 8) + (uint32_t)tid;
        if (f.item != kNoItem && pos < f.r1) {
            const uint32_t *pl = view_at(point_list, vb.bin, f.vw);
            f.id = pl[pos];
            f.qm = reinterpret_cast<const uint16_t *>(reinterpret_cast<const char *>(pl) + vb.bin_masks)[pos];
        }
    };
    auto fetch_records = [&](Fetched &f) {
        const uint32_t pos = f.r0 + ((f.item >> kItemTileBits) << 8) + (uint32_t)tid;
        if (f.item != kNoItem && pos < f.r1) {
            // static-split mode: records of splats with id >= st.id0 live in the view's static blob
            const float4 *rec = (st.base && f.id >= st.id0)
                ? reinterpret_cast<const float4 *>(st.base + st.stride * f.vw + st.rec) + 4 * (size_t)(f.id - st.id0)
                : view_at(blend_rec, vb.geom, f.vw) + 4 * (size_t)f.id;
            const float4 rb = rec[1], rc = rec[2];
            f.ra = rec[0];
            f.rbx = rb.x;
            f.rby = rb.y;
            f.rcz = rc.z;
            f.rcw = rc.w;
            f.rdx = C > 2 ? rec[3].x : 0.f;
        }
    };
    __syncthreads();  // s_first is written
    // The forward appends a tile's batches as consecutive items.  A workgroup takes the queue in CHUNKS of kChunk
    // consecutive items (chunk c of workgroup b: items (c grid + b) kChunk ...), so most of its items continue the tile of
    // the one before: the pixels' state (final T, last contributor, dL/dpixel, what lies behind the whole list) then
    // stays in registers, and the record in front of the next batch is prefetched under the walk.
#ifndef FNX_EARLY_GATHER
#define FNX_EARLY_GATHER 1
#endif
#ifndef FNX_BWD_DYNAMIC
#define FNX_BWD_DYNAMIC 1
#endif
#ifndef FNX_BWD_CHUNK
#define FNX_BWD_CHUNK 1  // measured on config 3 (mode 3): 1 -> 398 us, 2 -> 403, 4 -> 436, 8 -> 465, one contiguous range -> 483
#endif
    constexpr uint32_t kChunk = FNX_BWD_CHUNK;
    (void)kChunk;
#ifdef FNX_BWD_BLOCKED  // experiment: every workgroup takes ONE contiguous range of the queue
    const uint32_t per_wg = (s_first[n_views] + gridDim.x - 1) / gridDim.x;
    auto ticket_of = [&](uint32_t sidx) -> uint32_t { return sidx < per_wg ? blockIdx.x * per_wg + sidx : 0xFFFFFFF0u; };
#elif FNX_BWD_DYNAMIC
    // Items are handed out by a device counter (the first two per workgroup are fixed: b, b + grid), so a workgroup
    // that drew light items takes more of them; ticket t is item (t * prime) mod n -- a bijection (the prime does not
    // divide n) that keeps workgroups running at the same time on items far apart in the queue: neighbouring tiles'
    // batches processed together meet on the same splats' accumulators (an unscrambled ticket measured 7 % slower than
    // the fixed stride, this one 6 % faster: config 3, 399 -> 375 us).
    __shared__ uint32_t s_tk;
    const uint32_t n_all = max(s_first[n_views], 1u);
    const unsigned long long mult = (n_all % 7919u) ? 7919ull : 7927ull;
    auto scramble = [&](uint32_t t) -> uint32_t { return t < n_all ? (uint32_t)(((unsigned long long)t * mult) % n_all) : 0xFFFFFFF0u; };
    uint32_t dyn_next = 0xFFFFFFF0u;  // the ticket drawn during the previous item
    auto ticket_of = [&](uint32_t sidx) -> uint32_t {
        return sidx < 2 ? scramble(blockIdx.x + sidx * gridDim.x) : scramble(dyn_next);
    };
#else
    auto ticket_of = [&](uint32_t sidx) -> uint32_t { return ((sidx / kChunk) * gridDim.x + blockIdx.x) * kChunk + sidx % kChunk; };
#endif
    uint32_t walked[1] = {0u};  // list entries of this workgroup's items (thread 0)
    Fetched cur, nxt;
    fetch_item(ticket_of(0), cur);
    fetch_item(ticket_of(1), nxt);
    fetch_range(cur);
    fetch_ids(cur);
    fetch_records(cur);
    // The pixels' inputs of the NEXT item (final T, last contributor, dL/dpixel, the tile's accumulated colour, the
    // hand-over record in front of its batch) are requested behind the walk of the current one, when the walk's registers
    // are free, and arrive under the flush: an item then starts without a round trip to memory (~2-3 us of ~20 per item).
    struct PixelsAhead {
        bool valid;
        float T_final;
        uint32_t last_contributor;
        float dL[C], total[C];
        float4 stt;
        float T_final1, dL1, total1;  // DUAL: the second image's pixel
        uint32_t last1;
        float2 stt1;
    } ahead;
    ahead.valid = false;
    auto load_ahead = [&](const Fetched &f) {  // the pixel inputs of item f (valid)
        const int nv = f.vw, ntile = (int)(f.item & kItemTileMask);
        const uint32_t nb_ = f.item >> kItemTileBits;
        const int npx = (ntile % gx) * FNX_TILE_X + blend_pixel_x(w, lane), npy = (ntile / gx) * FNX_TILE_Y + blend_pixel_y(w, lane);
        const bool nin = npx < W && npy < H;
        const uint32_t npix = (uint32_t)W * npy + npx;
        ahead.T_final = nin ? view_at(final_Ts, vb.img, nv)[npix] : 0.f;
        ahead.last_contributor = nin ? (view_at(n_contrib, vb.img, nv) + (size_t)W * H)[npix] : 0u;
        const float *nacc = view_at(acc_final, vb.img, nv), *ndl = dL_dpixels + (size_t)nv * C * H * W;
#pragma unroll
        for (int ch = 0; ch < C; ch++) {
            ahead.dL[ch] = nin ? ndl[(size_t)ch * H * W + npix] : 0.f;
            ahead.total[ch] = nin ? nacc[(size_t)ch * H * W + npix] : 0.f;
        }
        ahead.stt = make_float4(1.f, 0.f, 0.f, 0.f);
        if (nb_) {
            const float4 *nbs = reinterpret_cast<const float4 *>(
                reinterpret_cast<const char *>(view_at(point_list, vb.bin, nv)) + vb.bin_bstate);
            ahead.stt = nbs[((size_t)(f.r0 >> 8) + nb_ - 1) * 256 + tid];
        }
        if (DUAL) {
            const char *i1 = du.img1 + vb.img * (size_t)nv;
            ahead.T_final1 = nin ? reinterpret_cast<const float *>(i1 + du.final_T)[npix] : 0.f;
            ahead.last1 = nin ? reinterpret_cast<const uint32_t *>(i1 + du.n_contrib)[npix] : 0u;
            ahead.dL1 = nin ? du.dL_dpix1[(size_t)nv * H * W + npix] : 0.f;
            ahead.total1 = nin ? reinterpret_cast<const float *>(i1 + du.acc_final)[npix] : 0.f;
            ahead.stt1 = make_float2(1.f, 0.f);
            if (nb_)
                ahead.stt1 = reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(view_at(point_list, vb.bin, nv)) +
                                                              du.bin_bstate1)[((size_t)(f.r0 >> 8) + nb_ - 1) * 256 + tid];
        }
    };
    // The FIRST item's pixels are requested here, like every later item's behind the walk of the one before: a load of them
    // left inside the loop "for the first item only" gave every item an s_waitcnt vmcnt(0) at its head -- behind the
    // previous item's flush atomics (the phase clocks' "item head": 17 % of a workgroup's time, round 5).
    ahead.valid = cur.item != kNoItem;
    if (ahead.valid) load_ahead(cur);
    // ... and everything the first item was requested is waited for HERE: the compiler's wait-count pass merges the loop's two
    // entries, and a register that is still in flight on the way in costs every iteration an s_waitcnt vmcnt(0) at its first
    // use -- on the way round that is a wait for the flush atomics.
    asm volatile("" ::"v"(cur.id), "v"(cur.qm), "v"(cur.ra.x), "v"(cur.ra.y), "v"(cur.ra.z), "v"(cur.ra.w), "v"(cur.rbx),
                 "v"(cur.rby), "v"(cur.rcz), "v"(cur.rcw), "v"(cur.rdx), "v"(ahead.T_final), "v"(ahead.last_contributor),
                 "v"(ahead.dL[0]), "v"(ahead.dL[C > 1 ? 1 : 0]), "v"(ahead.dL[C > 2 ? 2 : 0]), "v"(ahead.total[0]),
                 "v"(ahead.total[C > 1 ? 1 : 0]), "v"(ahead.total[C > 2 ? 2 : 0]), "v"(ahead.stt.x), "v"(ahead.stt.y),
                 "v"(ahead.stt.z), "v"(ahead.stt.w), "v"(nxt.item));
    if (DUAL)
        asm volatile("" ::"v"(ahead.T_final1), "v"(ahead.last1), "v"(ahead.dL1), "v"(ahead.total1), "v"(ahead.stt1.x),
                     "v"(ahead.stt1.y));
#ifdef FNX_EXP_BCLK
    unsigned long long t_last = clock64();
#endif
    for (uint32_t sidx = 0;; sidx++) {
        FNX_BCLK(0)  // loop back-edge (flush of the previous item ends here)
        // no barrier here: before the next one (behind the block maxima) an item writes s_max only, and the previous
        // item's last readers of s_max passed two barriers ago
        if (cur.item == kNoItem) break;  // tickets only grow along a workgroup's sequence: nothing behind the queue's end
        const int vw = cur.vw;
        Fetched nx2;
#if FNX_BWD_DYNAMIC
        // The ticket for the item after next.  Written as an instruction, not as atomicAdd(): the compiler turns a
        // uniform atomic add into a wave-aggregated one whose result it reads back at once (v_readfirstlane), i.e. with
        // s_waitcnt vmcnt(0) HERE -- at the head of every item, behind the previous item's flush atomics (24 % of an
        // item).  The returned value is only needed behind the walk; the wait for it is spelled out there.
        uint32_t drawn = 0;
        if (tid == 0) {
            uint32_t *tk = const_cast<uint32_t *>(header) + HDR_BWD_TICKET;
            const uint32_t one = 1u;
            asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(drawn) : "v"(tk), "v"(one) : "memory");
        }
#else
        fetch_item(ticket_of(sidx + 2), nx2);
#endif
        fetch_range(nxt);
        // per-view scratch, pixel gradients and screen-space accumulators of the item's view
        float *dL_dmean2D_v = dL_dmean2D + (size_t)vw * P * 3;
        float *dL_dconic_v = dL_dconic + (size_t)vw * P * 4;
        float *dL_dopacity_v = kAppearance ? dL_dopacity + (size_t)vw * P : nullptr;
        float *dL_dcolors_v = kAppearance ? dL_dcolors + (size_t)vw * P * C : nullptr;
        const uint32_t item = cur.item;
        const int tile = (int)(item & kItemTileMask);
        const uint32_t b = item >> kItemTileBits;
        const int tx = tile % gx, ty = tile / gx;
        const int row = lane >> 4;
        const int px = tx * FNX_TILE_X + blend_pixel_x(w, lane), py = ty * FNX_TILE_Y + blend_pixel_y(w, lane);
        const float pxf = (float)px, pyf = (float)py;
        const uint32_t r0 = cur.r0;
        const uint32_t q0 = b << 8;  // list position of the batch's first entry
        walked[0] += min(256u, cur.r1 - r0 - q0);

        float T_final, dL_dpixel[C], total[C];
        uint32_t last_contributor;
        float4 stt = make_float4(1.f, 0.f, 0.f, 0.f);
        {  // (always prefetched: before the loop for the first item, behind the previous item's walk otherwise)
            T_final = ahead.T_final;
            last_contributor = ahead.last_contributor;
#pragma unroll
            for (int ch = 0; ch < C; ch++) {
                dL_dpixel[ch] = ahead.dL[ch];
                total[ch] = ahead.total[ch];
            }
            stt = ahead.stt;
        }
        // DUAL: the second image's pixel (final T, last contributor, dL/dpixel, accumulated value, state in front of the batch)
        float T_final1 = 0.f, dL1 = 0.f, total1 = 0.f;
        uint32_t last1 = 0;
        float2 stt1 = make_float2(1.f, 0.f);
        if (DUAL) {
            T_final1 = ahead.T_final1;
            dL1 = ahead.dL1;
            total1 = ahead.total1;
            last1 = ahead.last1;
            stt1 = ahead.stt1;
        }
        float Tr1 = b ? stt1.x : 1.0f;
        // what lies behind the entry, second image: (total1 - prefix1) dL1 + its background's share
        float rest1 = DUAL ? (total1 * dL1 - (b ? stt1.y : 0.f) * dL1) + T_final1 * du.bg1[0] * dL1 : 0.f;
        float bg_dot_dpixel = 0.f, total_dot = 0.f;
#pragma unroll
        for (int ch = 0; ch < C; ch++) {
            bg_dot_dpixel += bg[ch] * dL_dpixel[ch];
            total_dot += total[ch] * dL_dpixel[ch];
        }
        const float tail_dot = T_final * bg_dot_dpixel;  // the background's share of what lies behind an entry
        float Tr = b ? stt.x : 1.0f, pre[C];
        pre[0] = b ? stt.y : 0.f;
        if (C > 1) pre[C > 1 ? 1 : 0] = b ? stt.z : 0.f;
        if (C > 2) pre[C > 2 ? 2 : 0] = b ? stt.w : 0.f;
        // Everything the gradient needs from the colour channels is their dot product with dL/dpixel:
        //   sum_ch dL_ch (c_ch T - S_ch / (1 - alpha)) = T (c . dL) - (S . dL) / (1 - alpha),
        //   S . dL = (total . dL) - (prefix . dL), and the prefix's dot product is itself a running sum of
        //   alpha_i T_i (c_i . dL): one scalar recurrence instead of one per channel.
        float pre_dot = 0.f;
#pragma unroll
        for (int ch = 0; ch < C; ch++) pre_dot += pre[ch] * dL_dpixel[ch];
        // what lies behind the entry being processed, as one running value: (total - prefix) . dL + the background's
        // share; an applied entry takes its own alpha T (c . dL) out of it (fused multiply-adds: the gradients are
        // compared within fp32 summation tolerance, only power / alpha repeat the forward's arithmetic exactly)
        float rest = (total_dot - pre_dot) + tail_dot;

        // entry q (0-based from the front) is used by a pixel iff q < its n_contrib_v (backward.cu:467-469):
        // a block needs nothing behind its own max, the batch nothing behind the max of the tile's blocks
        uint32_t m = DUAL ? max(last_contributor, last1) : last_contributor;
        for (int off = 8; off >= 1; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off));
        if ((lane & 15) == 0) s_max[4 * w + row] = m;
        FNX_BCLK(1)  // item head: pixel inputs, block maxima
        // LDS-only barriers in the item loop: the waves exchange nothing through global memory, and a plain
        // __syncthreads() would also wait for the previous item's flush atomics and for the next item's prefetches
        FNX_LOOP_BARRIER();
        FNX_BCLK(2)  // wait at barrier A
        // the 16 block maxima in registers (four 16-byte LDS reads instead of 2 x 16 scalar ones)
        uint32_t bmax[16];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint4 m4 = reinterpret_cast<const uint4 *>(s_max)[k];
            bmax[4 * k] = m4.x;
            bmax[4 * k + 1] = m4.y;
            bmax[4 * k + 2] = m4.z;
            bmax[4 * k + 3] = m4.w;
        }
        uint32_t qmax = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) qmax = max(qmax, bmax[k]);
        const uint32_t cnt = min(256u, qmax - min(qmax, q0));

        // stage entries q = q0 + t (slot t), zero the slot accumulators
        if (DUAL) {  // which staged entries are dynamic (= take gradients: the forward's limit equals this call's)
            const unsigned long long dm = __ballot((uint32_t)tid < cnt && cur.id < grad_limit);
            if (lane == 0) s_dynmask[DUAL ? w : 0] = dm;
        }
        uint32_t qm = 0;
        if ((uint32_t)tid < cnt) {
            const uint32_t q = q0 + tid;  // cnt <= r1 - r0 - q0: the entry was fetched
            const uint32_t id = cur.id;
            qm = cur.qm;
#pragma unroll
            for (int k = 0; k < 16; k++)
                if (q >= bmax[k]) qm &= ~(1u << k);
            s_id[tid] = id;
            if (FAST) {
                constexpr float kL2e = 1.44269504088896341f;
                s_ra[tid] = make_float4(cur.ra.x, cur.ra.y, (-0.5f * kL2e) * cur.ra.z, (-kL2e) * cur.ra.w);
                // the walk is bound by LDS cycles: two 16-byte reads per entry (+ 8 bytes with three channels), no
                // 12-byte reads (ds_read_b96 costs twice a ds_read_b128)
                const float wants_f = id < grad_limit ? 1.0f : 0.0f;
                s_rb[tid] = make_float4((-0.5f * kL2e) * cur.rbx, __builtin_amdgcn_logf(fmaxf(cur.rby, 0.0f)), cur.rcz,
                                        C == 3 ? cur.rcw : wants_f);
                if (C == 3) s_rc[tid] = make_float4(cur.rdx, wants_f, 0.f, 0.f);
                s_rd[FAST ? tid : 0] = make_float4(cur.ra.z, cur.ra.w, cur.rbx, cur.rby);
            } else {
                s_ra[tid] = cur.ra;
                s_rb[tid] = make_float4(cur.rbx, cur.rby, 0.f, id < grad_limit ? 1.0f : 0.0f);
                s_rc[tid] = make_float4(cur.rcz, C > 1 ? cur.rcw : 0.f, C > 2 ? cur.rdx : 0.f, 0.f);
            }
        }
#pragma unroll
        for (int v = 0; v < NV; v++) s_acc[v][tid] = 0.f;
        s_mask[tid] = (uint16_t)qm;
        {  // this wave's four lists start out as NULL pointers (slot 256) from end to end
            const uint4 nul = make_uint4(0x10001000u, 0x10001000u, 0x10001000u, 0x10001000u);
            uint4 *mine = reinterpret_cast<uint4 *>(&s_list[4 * w][0]);
            for (int i = lane; i < 4 * kListStride / 8; i += 64) mine[i] = nul;
        }
        FNX_BCLK(3)  // staging writes
        FNX_LOOP_BARRIER();
        FNX_BCLK(4)  // wait at barrier B
        uint32_t len[4] = {0u, 0u, 0u, 0u};  // wave-uniform lengths of the wave's four lists
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t mk = (uint32_t)s_mask[64 * k + lane] >> (4 * w);
            const uint32_t flag = (DUAL && !((s_dynmask[DUAL ? k : 0] >> lane) & 1ull)) ? kListStatic : 0u;
#pragma unroll
            for (int bb = 0; bb < 4; bb++) {
                const bool bit = (mk >> bb) & 1u;
                const unsigned long long bm = __ballot(bit);
                if (bit) s_list[4 * w + bb][len[bb] + (uint32_t)__popcll(bm & lt_mask)] = (uint16_t)(((64 * k + lane) * 16) | flag);
                len[bb] += (uint32_t)__popcll(bm);
            }
        }
        const uint32_t n_w = max(max(len[0], len[1]), max(len[2], len[3]));  // steps of the wave = its longest list
        const uint16_t *mylist = s_list[4 * w + row];
        fetch_ids(nxt);  // in flight during the walk
        FNX_BCLK(5)  // list build
        // entry q0 + slot lies in front of the pixel's last contributor <=> its LDS offset (16 slot) is below this bound
        const uint32_t lim_off = last_contributor > q0 ? min(last_contributor - q0, 4096u) << 4 : 0u;
        const uint32_t lim_off1 = (DUAL && last1 > q0) ? min(last1 - q0, 4096u) << 4 : 0u;
        // Straight-line steps of kGroup entries (all LDS reads of a step issued together, no lane predicates): an entry
        // the pixel does not take has alpha = 0, which leaves T and the colour prefix exactly as they are (x * 1, + 0)
        // and zeroes every gradient term; the NULL record behind a list's end is such an entry for every pixel.
        if (FAST) {
            static_assert(!FAST || kGroup == 4, "the fast walk folds the sums of four entries together");
            const int vq = ((lane >> 3) & 1) | (((lane >> 2) & 1) << 1);  // entry of a step whose sums this lane's quad ends up with
            const bool hi8 = (lane & 8) != 0, hi4 = (lane & 4) != 0;
            (void)hi8;
            (void)hi4;
            for (uint32_t i0 = 0; i0 < (FNX_ABLATE == 3 ? 0u : n_w); i0 += 4) {
                uint32_t jw[2];
                jw[0] = reinterpret_cast<const uint32_t *>(mylist + i0)[0];
                jw[1] = reinterpret_cast<const uint32_t *>(mylist + i0)[1];
                // per entry of the step only (w, dx, dy[, alpha T]) stay in registers; the NV values of an entry are formed
                // from them block by block right in front of their fold (all NV x 4 at once spill with colour sums)
                // (kLazy; with five values they are formed inside the loop: moving them behind it cost 5 % of the walk)
                constexpr bool kLazy = NV > 5;
                constexpr int kBlockA = NV <= 5 ? NV : 4, kBlockB = NV - kBlockA;
                float w_[4], dx_[4], dy_[4], dc_[kAppearance ? 4 : 1];
                float valA[kBlockA][4];
                bool any_emit = false;
                float4 ra[4], rb[4];
                float2 rc[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t off = (jw[k >> 1] >> (16 * (k & 1))) & (DUAL ? kListOffMask : 0xFFFFu);
                    ra[k] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s_ra) + off);
                    rb[k] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s_rb) + off);
                    rc[k] = C == 3 ? *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(s_rc) + off)
                                   : make_float2(0.f, 0.f);
                }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t raw = (jw[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
                    const uint32_t off = raw & (DUAL ? kListOffMask : 0xFFFFu);
                    const float dx = ra[k].x - pxf, dy = ra[k].y - pyf;
                    const float u = __builtin_fmaf(ra[k].z, dx, ra[k].w * dy);
                    const float q = __builtin_fmaf(u, dx, (rb[k].x * dy) * dy);  // log2(e) * power
                    const float e = __builtin_amdgcn_exp2f(q + rb[k].y);         // o G
                    const float alpha = fminf(0.99f, e);
                    const bool hit = !(q > 0.0f) && !(alpha < 1.0f / 255.0f);
                    const bool active = hit && (off < lim_off);
                    const bool emits = active && (C == 3 ? rc[k].y : rb[k].w) != 0.0f;
                    const float a = active ? alpha : 0.0f;
                    const float one_m = 1 - a;
                    const float inv_1ma = __builtin_amdgcn_rcpf(one_m);
                    const float Tb = Tr;
                    float c_dot = rb[k].z * dL_dpixel[0];
                    if (C > 1) c_dot = __builtin_fmaf(rb[k].w, dL_dpixel[C > 1 ? 1 : 0], c_dot);
                    if (C > 2) c_dot = __builtin_fmaf(rc[k].x, dL_dpixel[C > 2 ? 2 : 0], c_dot);
                    const float aT = a * Tb;
                    rest = __builtin_fmaf(-aT, c_dot, rest);
                    Tr = Tb * one_m;
                    float dL_dalpha = __builtin_fmaf(Tb, c_dot, -(rest * inv_1ma));
                    if (DUAL) dL_dalpha = emits ? dL_dalpha : 0.0f;  // (the second image may emit where the first does not)
                    bool emits1 = false;
                    if (DUAL) {  // the second image's share: dynamic entries in front of ITS last contributor
                        const bool active1 = hit && !(raw & kListStatic) && (off < lim_off1);
                        const float a1 = active1 ? alpha : 0.0f;
                        // (1 - alpha) is the same number in both images wherever both take the entry
                        const float inv1 = active1 ? (active ? inv_1ma : __builtin_amdgcn_rcpf(1 - a1)) : 1.0f;
                        const float Tb1 = Tr1, c_dot1 = rb[k].z * dL1;
                        rest1 = __builtin_fmaf(-(a1 * Tb1), c_dot1, rest1);
                        Tr1 = Tb1 * (1 - a1);
                        const float dL_dalpha1 = __builtin_fmaf(Tb1, c_dot1, -(rest1 * inv1));
                        emits1 = active1;  // a dynamic entry takes gradients (the forward's limit equals this call's)
                        dL_dalpha += emits1 ? dL_dalpha1 : 0.0f;
                    }
                    // G dL/dG = G o dL/dalpha = (o G) dL/dalpha: the clamp at 0.99 has no mask in the reference (A.10)
                    const float wgt = ((emits || emits1) ? e : 0.0f) * dL_dalpha;
                    if constexpr (kLazy) {
                        w_[k] = wgt;
                        dx_[k] = dx;
                        dy_[k] = dy;
                        if (kAppearance) dc_[kAppearance ? k : 0] = emits ? aT : 0.0f;
                    } else {
                        const float wx = wgt * dx, wy = wgt * dy;
                        if (kMeans) {
                            valA[0][k] = wx;
                            valA[kMeans ? 1 : 0][k] = wy;
                        }
                        valA[kConic][k] = wx * dx;
                        valA[kConic + 1][k] = wx * dy;
                        valA[kConic + 2][k] = wy * dy;
                        if (kAppearance) {
                            valA[kAppearance ? kOpac : 0][k] = wgt;
                            const float dchannel_dcolor = emits ? aT : 0.0f;
#pragma unroll
                            for (int ch = 0; ch < C; ch++) valA[kAppearance && kCol + ch < kBlockA ? kCol + ch : 0][k] = dchannel_dcolor * dL_dpixel[ch];
                        }
                    }
                    any_emit |= emits || emits1;
                }
                if (FNX_ABLATE != 2 && __ballot(any_emit) != 0ull) {
                    // this quad's target: the slot of entry vq of the step (row-uniform), the NULL slot's sums are dropped
                    const uint32_t o01 = jw[0], o23 = jw[1];
                    const uint32_t osel = (vq & 2) ? o23 : o01;
                    const uint32_t slot = ((osel >> (16 * (vq & 1))) & (DUAL ? kListOffMask : 0xFFFFu)) >> 4;
                    const int kq = lane & 3;
                    // value v of entry k: [w dx, w dy,] w dx dx, w dx dy, w dy dy [, w (the flush divides by o), alpha T dL_ch]
                    auto value = [&](int v, int k) -> float {
                        if (kMeans && v == 0) return w_[k] * dx_[k];
                        if (kMeans && v == 1) return w_[k] * dy_[k];
                        if (v == kConic) return (w_[k] * dx_[k]) * dx_[k];
                        if (v == kConic + 1) return (w_[k] * dx_[k]) * dy_[k];
                        if (v == kConic + 2) return (w_[k] * dy_[k]) * dy_[k];
                        if (kAppearance && v == kOpac) return w_[k];
                        return kAppearance ? dc_[kAppearance ? k : 0] * dL_dpixel[v - kCol < C && v >= kCol ? v - kCol : 0] : 0.0f;
                    };
                    // hand-scheduled DPP butterflies (fnx_fold_asm.h) over blocks of values; afterwards the four lanes of a
                    // quad hold the same totals and lane k of the quad adds value 4 j + k: ceil(n / 4) LDS atomics per block
                    // of n values instead of n (the walk is bound by LDS cycles, not by instructions)
                    {
                        float (&val)[kBlockA][4] = valA;
                        if constexpr (kLazy) {
#pragma unroll
                            for (int v = 0; v < kBlockA; v++)
#pragma unroll
                                for (int k = 0; k < 4; k++) val[v][k] = value(v, k);
                        }
                        fold_rows_asm<kBlockA>(val);
#pragma unroll
                        for (int j = 0; j < (kBlockA + 3) / 4; j++) {
                            float t = val[4 * j][0];
                            if (4 * j + 1 < kBlockA) t = kq == 1 ? val[4 * j + 1 < kBlockA ? 4 * j + 1 : 0][0] : t;
                            if (4 * j + 2 < kBlockA) t = kq == 2 ? val[4 * j + 2 < kBlockA ? 4 * j + 2 : 0][0] : t;
                            if (4 * j + 3 < kBlockA) t = kq == 3 ? val[4 * j + 3 < kBlockA ? 4 * j + 3 : 0][0] : t;
                            const int v = 4 * j + kq;
                            if (slot < 256u && v < kBlockA) atomicAdd(&s_acc[0][0] + v * kAccStride + (slot & 255u), t);
                        }
                    }
                    if constexpr (kBlockB > 0) {
                        float val[kBlockB][4];
#pragma unroll
                        for (int v = 0; v < kBlockB; v++)
#pragma unroll
                            for (int k = 0; k < 4; k++) val[v][k] = value(kBlockA + v, k);
                        fold_rows_asm<kBlockB>(val);
#pragma unroll
                        for (int j = 0; j < (kBlockB + 3) / 4; j++) {
                            float t = val[4 * j][0];
                            if (4 * j + 1 < kBlockB) t = kq == 1 ? val[4 * j + 1 < kBlockB ? 4 * j + 1 : 0][0] : t;
                            if (4 * j + 2 < kBlockB) t = kq == 2 ? val[4 * j + 2 < kBlockB ? 4 * j + 2 : 0][0] : t;
                            if (4 * j + 3 < kBlockB) t = kq == 3 ? val[4 * j + 3 < kBlockB ? 4 * j + 3 : 0][0] : t;
                            const int v = 4 * j + kq;
                            if (slot < 256u && v < kBlockB)
                                atomicAdd(&s_acc[0][0] + (kBlockA + v) * kAccStride + (slot & 255u), t);
                        }
                    }
                }
            }
        } else
        for (uint32_t i0 = 0; i0 < (FNX_ABLATE == 3 ? 0u : n_w); i0 += kGroup) {
            uint32_t jw[(kGroup + 1) / 2];
#pragma unroll
            for (int k = 0; k < (kGroup + 1) / 2; k++) jw[k] = reinterpret_cast<const uint32_t *>(mylist + i0)[k];
            float4 ra[kGroup], rb[kGroup], rc[kGroup];
#pragma unroll
            for (int k = 0; k < kGroup; k++) {
                const uint32_t off = (jw[k >> 1] >> (16 * (k & 1))) & (DUAL ? kListOffMask : 0xFFFFu);
                ra[k] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s_ra) + off);
                rb[k] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s_rb) + off);
                rc[k] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s_rc) + off);
            }
#pragma unroll
            for (int k = 0; k < kGroup; k++) {
                const uint32_t raw = (jw[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
                const uint32_t off = raw & (DUAL ? kListOffMask : 0xFFFFu), slot = off >> 4;
                const float dx = ra[k].x - pxf, dy = ra[k].y - pyf;
                const float power = -0.5f * (ra[k].z * dx * dx + rb[k].x * dy * dy) - ra[k].w * dx * dy;
                const float G = exp_fixed_in_range(fmaxf(power, -87.0f));
                const float alpha = fminf(0.99f, rb[k].y * G);
                // the forward's decision (blend_forward_kernel), for the entries in front of the pixel's last contributor
                const bool hit = !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
                const bool active = hit && (off < lim_off);
                const bool wants = rb[k].w != 0.0f;  // uniform within a row
                const float a = active ? alpha : 0.0f;
                // one hardware reciprocal (<= 1 ulp) serves both divisions by (1 - alpha) of backward.cu:482,510;
                // the backward is compared within fp32 summation tolerance, not bit for bit
                const float one_m = 1 - a;
                const float inv_1ma = __builtin_amdgcn_rcpf(one_m);
                const float Tb = Tr;  // transmittance in front of the entry
                const float col[3] = {rc[k].x, rc[k].y, rc[k].z};
                float c_dot = col[0] * dL_dpixel[0];
#pragma unroll
                for (int ch = 1; ch < C; ch++) c_dot = __builtin_fmaf(col[ch], dL_dpixel[ch], c_dot);
                rest = __builtin_fmaf(-(a * Tb), c_dot, rest);
                Tr = Tb * one_m;  // the forward's test_T
                const float dL_dalpha = __builtin_fmaf(Tb, c_dot, -(rest * inv_1ma));
                bool emits = active && wants;
                float dL_da = emits ? dL_dalpha : 0.0f;
                bool active1 = false;
                if (DUAL) {  // the second image's share (see the fast walk)
                    active1 = hit && !(raw & kListStatic) && (off < lim_off1);
                    const float a1 = active1 ? alpha : 0.0f;
                    const float inv1 = active1 ? (active ? inv_1ma : __builtin_amdgcn_rcpf(1 - a1)) : 1.0f;
                    const float Tb1 = Tr1, c_dot1 = col[0] * dL1;
                    rest1 = __builtin_fmaf(-(a1 * Tb1), c_dot1, rest1);
                    Tr1 = Tb1 * (1 - a1);
                    dL_da += active1 ? __builtin_fmaf(Tb1, c_dot1, -(rest1 * inv1)) : 0.0f;
                    emits = emits || active1;
                }
                const float dL_dG = rb[k].y * dL_da;
                const float Gm = (active || active1) ? G : 0.0f;  // power > 0 can push the range-limited exp out of range: never into a sum
                // Per (pixel, entry) only the weighted moments of the offset are formed: w = G dL/dG,
                // (w dx, w dy, w dx^2, w dx dy, w dy^2).  The entry's own constants -- its conic in dG/d(mean), the
                // -1/2 of the conic gradients, the pixel scale -- multiply the SUMS once per entry when they are flushed.
                const float wgt = Gm * dL_dG;
                const float wx = wgt * dx, wy = wgt * dy;
                float val[NV];
                if (kMeans) {
                    val[0] = wx;
                    val[kMeans ? 1 : 0] = wy;
                }
                val[kConic] = wx * dx;
                val[kConic + 1] = wx * dy;
                val[kConic + 2] = wy * dy;
                if (kAppearance) {
                    val[kAppearance ? kOpac : 0] = Gm * dL_da;
                    const float dchannel_dcolor = emits ? a * Tb : 0.0f;
#pragma unroll
                    for (int ch = 0; ch < C; ch++) val[kAppearance ? kCol + ch : 0] = dchannel_dcolor * dL_dpixel[ch];
                }
#if FNX_ABLATE == 2
                { float sink = 0.f; _Pragma("unroll") for (int v = 0; v < NV; v++) sink += val[v]; asm volatile("" ::"v"(sink)); }
#else
                if (__ballot(emits) != 0ull) row_fold_accumulate<NV, kAccStride>(val, s_acc, slot & 255u, lane, wants);
#endif
            }
        }
        FNX_BCLK(6)  // walk
#if FNX_BWD_DYNAMIC
        // the ticket drawn at the top has long arrived; published BEFORE the prefetches below are issued: s_waitcnt counts
        // in order, so waiting for this atomic's return behind them would drain them all in front of the barrier
        if (tid == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the compiler does not track the asm's return value
            s_tk = 2u * gridDim.x + drawn;
        }
#endif
        fetch_records(nxt);  // in flight while the accumulators are flushed
        ahead.valid = nxt.item != kNoItem;
        if (ahead.valid) load_ahead(nxt);  // ... and so are the next item's pixels
        // positions-only mode: the flush needs the splat's mean and world covariance; requested here, in front of the
        // barrier (the waves wait for the slowest walk anyway), instead of as a round trip inside the flush
        float gmean[3] = {0.f, 0.f, 0.f}, gcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (kFusedGeom && FNX_EARLY_GATHER && (uint32_t)tid < cnt) {
            const uint32_t gid = s_id[tid];
            if (gid < grad_limit) {
                const float *cv = view_at(cov3Ds, cov3D_stride, vw) + 6 * (size_t)gid;
#pragma unroll
                for (int k = 0; k < 3; k++) gmean[k] = means3D[3 * (size_t)gid + k];
#pragma unroll
                for (int k = 0; k < 6; k++) gcov[k] = cv[k];
            }
        }
        FNX_LOOP_BARRIER();
        FNX_BCLK(7)  // prefetches + wait at barrier C
#if FNX_BWD_DYNAMIC
        dyn_next = s_tk;
        fetch_item(ticket_of(sidx + 2), nx2);
#endif
        // Flush: the values first, the global atomics last.  vmcnt counts loads, stores and atomics in order, and the
        // atomics sit in a lane-divergent branch: whatever load is waited for behind them is waited for with vmcnt(0),
        // i.e. until the atomics have retired (measured: 24 % of an item at the head of the NEXT one).  So every
        // prefetched register is touched -- waited for -- in front of the atomics, and nothing needs a wait behind them
        // until the next item's own loads return.
        constexpr int kFl = kFusedGeom ? 3 : (kMeans ? 2 : 0) + 3 + (kAppearance ? 1 + C : 0);
        float fl[kFl];
#pragma unroll
        for (int k = 0; k < kFl; k++) fl[k] = 0.f;
        bool do_flush = false;
        uint32_t fid = 0;
        if ((uint32_t)tid < cnt) {
            const uint32_t id = s_id[tid];
            float a[NV];
            bool any = false;
#pragma unroll
            for (int v = 0; v < NV; v++) {
                a[v] = s_acc[v][tid];
                any |= (a[v] != 0.f);
            }
            if (any && FNX_ABLATE != 1) {
                do_flush = true;
                fid = id;
                // moments -> gradients (backward.cu:512-533): dG/d(delta) = -G (a dx + b dy, c dy + b dx), conic terms
                // -1/2 G (dx^2, dx dy, dy^2), each times dL/dG
                float4 ra = s_ra[tid];
                float cc = s_rb[tid].x;
                if (FAST) {  // the entry's own conic (the staged coefficients are pre-scaled) and opacity
                    const float4 rd = s_rd[FAST ? tid : 0];
                    ra.z = rd.x;
                    ra.w = rd.y;
                    cc = rd.z;
                    if (kAppearance) a[kAppearance ? kOpac : 0] = a[kAppearance ? kOpac : 0] / rd.w;
                }
                const float g0 = kMeans ? -(ra.z * a[0] + ra.w * a[kMeans ? 1 : 0]) * ddelx_dx : 0.f;
                const float g1 = kMeans ? -(cc * a[kMeans ? 1 : 0] + ra.w * a[0]) * ddely_dy : 0.f;
                if (kFusedGeom) {
                    if (!FNX_EARLY_GATHER) {
                        const float *cv = view_at(cov3Ds, cov3D_stride, vw) + 6 * (size_t)id;
#pragma unroll
                        for (int k = 0; k < 3; k++) gmean[k] = means3D[3 * (size_t)id + k];
#pragma unroll
                        for (int k = 0; k < 6; k++) gcov[k] = cv[k];
                    }
                    const float3 mean = make_float3(gmean[0], gmean[1], gmean[2]);
                    float gv[3];
                    geom_backward_view<false>(mean, gcov, viewmatrix + 16 * vw,
                                       projmatrix + 16 * vw, vb.focal_x[vw], vb.focal_y[vw], vb.tan_fovx[vw], vb.tan_fovy[vw],
                                       -0.5f * a[kConic], -0.5f * a[kConic + 1], -0.5f * a[kConic + 2], g0, g1, gv, nullptr);
#pragma unroll
                    for (int k = 0; k < 3; k++) fl[k < kFl ? k : 0] = gv[k];
                } else {
                    int o = 0;
                    if (kMeans) {
                        fl[o++ < kFl ? o - 1 : 0] = g0;
                        fl[o++ < kFl ? o - 1 : 0] = g1;
                    }
                    fl[o++ < kFl ? o - 1 : 0] = -0.5f * a[kConic];
                    fl[o++ < kFl ? o - 1 : 0] = -0.5f * a[kConic + 1];
                    fl[o++ < kFl ? o - 1 : 0] = -0.5f * a[kConic + 2];
                    if (kAppearance) {
                        fl[o++ < kFl ? o - 1 : 0] = a[kAppearance ? kOpac : 0];
#pragma unroll
                        for (int ch = 0; ch < C; ch++) fl[o++ < kFl ? o - 1 : 0] = a[kAppearance ? kCol + ch : 0];
                    }
                }
            }
        }
        asm volatile("" ::"v"(nxt.id), "v"(nxt.qm), "v"(nxt.ra.x), "v"(nxt.ra.y), "v"(nxt.ra.z), "v"(nxt.ra.w), "v"(nxt.rbx),
                     "v"(nxt.rby), "v"(nxt.rcz), "v"(nxt.rcw), "v"(nxt.rdx), "v"(ahead.T_final), "v"(ahead.last_contributor),
                     "v"(ahead.dL[0]), "v"(ahead.dL[C > 1 ? 1 : 0]), "v"(ahead.dL[C > 2 ? 2 : 0]), "v"(ahead.total[0]),
                     "v"(ahead.total[C > 1 ? 1 : 0]), "v"(ahead.total[C > 2 ? 2 : 0]), "v"(ahead.stt.x), "v"(ahead.stt.y),
                     "v"(ahead.stt.z), "v"(ahead.stt.w), "v"(nx2.item));
        if (DUAL)
            asm volatile("" ::"v"(ahead.T_final1), "v"(ahead.last1), "v"(ahead.dL1), "v"(ahead.total1), "v"(ahead.stt1.x),
                         "v"(ahead.stt1.y));
        if (kFusedGeom) {
            // one atomic instruction for the three components of a splat's gradient (lane l: component l % 3 of entry l / 3,
            // through a wave-private strip of LDS): a global fp32 atomic is priced per memory request, and an instruction then
            // touches 22 lines instead of 64 (raster_backward_lanes.h, tools/micro/atomic_rate.hip)
            if (__ballot(do_flush) != 0ull) {
                s_pfi[w][lane] = do_flush ? fid : 0xFFFFFFFFu;
#pragma unroll
                for (int k = 0; k < 3; k++) s_pfv[w][3 * lane + k] = fl[k < kFl ? k : 0];
#pragma unroll
                for (int rnd = 0; rnd < 3; rnd++) {
                    const int v = 64 * rnd + lane;
                    const uint32_t id_ = s_pfi[w][v / 3];
                    const float val = s_pfv[w][v];
                    if (id_ != 0xFFFFFFFFu) FNX_FLUSH_ADD(&dL_dmean3D[3 * (size_t)id_ + (v % 3)], val);
                }
            }
        } else if (do_flush) {
            {
                int o = 0;
                if (kMeans) {
                    FNX_FLUSH_ADD(&dL_dmean2D_v[3 * (size_t)fid + 0], fl[o++ < kFl ? o - 1 : 0]);
                    FNX_FLUSH_ADD(&dL_dmean2D_v[3 * (size_t)fid + 1], fl[o++ < kFl ? o - 1 : 0]);
                }
                FNX_FLUSH_ADD(&dL_dconic_v[4 * (size_t)fid + 0], fl[o++ < kFl ? o - 1 : 0]);
                FNX_FLUSH_ADD(&dL_dconic_v[4 * (size_t)fid + 1], fl[o++ < kFl ? o - 1 : 0]);
                FNX_FLUSH_ADD(&dL_dconic_v[4 * (size_t)fid + 3], fl[o++ < kFl ? o - 1 : 0]);
                if (kAppearance) {
                    FNX_FLUSH_ADD(&dL_dopacity_v[fid], fl[o++ < kFl ? o - 1 : 0]);
#pragma unroll
                    for (int ch = 0; ch < C; ch++) FNX_FLUSH_ADD(&dL_dcolors_v[(size_t)fid * C + ch], fl[o++ < kFl ? o - 1 : 0]);
                }
            }
        }
        cur = nxt;
        nxt.item = nx2.item;
        nxt.vw = nx2.vw;
    }
    // (all views' entries are counted in view 0's word: the items of a workgroup come from every view)
    if (tid == 0 && walked[0]) atomicAdd(const_cast<uint32_t *>(header) + HDR_BWD_ENTRIES, walked[0]);
#if FNX_BWD_DYNAMIC
    // the last workgroup to run out of items re-arms the counters: another backward over the same forward (a second
    // autograd pass, a test that calls it again) starts from ticket 0 like the first
    if (tid == 0) {
        uint32_t *h0 = const_cast<uint32_t *>(header);
        if (atomicAdd(h0 + HDR_BWD_DONE, 1u) == gridDim.x - 1u) {
            __hip_atomic_store(h0 + HDR_BWD_TICKET, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(h0 + HDR_BWD_DONE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// SH colour backward (ch3 backward.cu:20-132)
// dL_dsh is accumulated (+=) into the caller's zero-filled array, so the views of a batch add up; the
// mean-gradient term is returned in gmean[3].
__device__ inline void sh_backward(int idx, int deg, int M, const float *means, const float *campos, const float *shs,
                                   const uint8_t *clamped, const float *dL_dcolor, float *gmean, float *dL_dshs) {
    const float ox = means[3 * idx] - campos[0], oy = means[3 * idx + 1] - campos[1], oz = means[3 * idx + 2] - campos[2];
    const float len = sqrtf(ox * ox + oy * oy + oz * oz);
    const float x = ox / len, y = oy / len, z = oz / len;
    const float *sh = shs + (size_t)idx * M * 3;
    float *dL_dsh = dL_dshs + (size_t)idx * M * 3;
    float g[3];
#pragma unroll
    for (int c = 0; c < 3; c++) g[c] = dL_dcolor[3 * idx + c] * (clamped[3 * idx + c] ? 0.0f : 1.0f);
    float ddx[3] = {0.f, 0.f, 0.f}, ddy[3] = {0.f, 0.f, 0.f}, ddz[3] = {0.f, 0.f, 0.f};
#define SH(k, c) sh[(k) * 3 + (c)]
#define DSH(k, c) dL_dsh[(k) * 3 + (c)]
#pragma unroll
    for (int c = 0; c < 3; c++) DSH(0, c) += kSH0 * g[c];
    if (deg > 0) {
        const float d1 = -kSH1 * y, d2 = kSH1 * z, d3 = -kSH1 * x;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            DSH(1, c) += d1 * g[c];
            DSH(2, c) += d2 * g[c];
            DSH(3, c) += d3 * g[c];
            ddx[c] = -kSH1 * SH(3, c);
            ddy[c] = -kSH1 * SH(1, c);
            ddz[c] = kSH1 * SH(2, c);
        }
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z;
            const float xy = x * y, yz = y * z, xz = x * z;
            const float d4 = kSH2[0] * xy, d5 = kSH2[1] * yz, d6 = kSH2[2] * (2.f * zz - xx - yy), d7 = kSH2[3] * xz,
                        d8 = kSH2[4] * (xx - yy);
#pragma unroll
            for (int c = 0; c < 3; c++) {
                DSH(4, c) += d4 * g[c];
                DSH(5, c) += d5 * g[c];
                DSH(6, c) += d6 * g[c];
                DSH(7, c) += d7 * g[c];
                DSH(8, c) += d8 * g[c];
                ddx[c] += kSH2[0] * y * SH(4, c) + kSH2[2] * 2.f * -x * SH(6, c) + kSH2[3] * z * SH(7, c) +
                          kSH2[4] * 2.f * x * SH(8, c);
                ddy[c] += kSH2[0] * x * SH(4, c) + kSH2[1] * z * SH(5, c) + kSH2[2] * 2.f * -y * SH(6, c) +
                          kSH2[4] * 2.f * -y * SH(8, c);
                ddz[c] += kSH2[1] * y * SH(5, c) + kSH2[2] * 2.f * 2.f * z * SH(6, c) + kSH2[3] * x * SH(7, c);
            }
            if (deg > 2) {
                const float d9 = kSH3[0] * y * (3.f * xx - yy), d10 = kSH3[1] * xy * z,
                            d11 = kSH3[2] * y * (4.f * zz - xx - yy),
                            d12 = kSH3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy),
                            d13 = kSH3[4] * x * (4.f * zz - xx - yy), d14 = kSH3[5] * z * (xx - yy),
                            d15 = kSH3[6] * x * (xx - 3.f * yy);
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    DSH(9, c) += d9 * g[c];
                    DSH(10, c) += d10 * g[c];
                    DSH(11, c) += d11 * g[c];
                    DSH(12, c) += d12 * g[c];
                    DSH(13, c) += d13 * g[c];
                    DSH(14, c) += d14 * g[c];
                    DSH(15, c) += d15 * g[c];
                    ddx[c] += (kSH3[0] * SH(9, c) * 3.f * 2.f * xy + kSH3[1] * SH(10, c) * yz +
                               kSH3[2] * SH(11, c) * -2.f * xy + kSH3[3] * SH(12, c) * -3.f * 2.f * xz +
                               kSH3[4] * SH(13, c) * (-3.f * xx + 4.f * zz - yy) + kSH3[5] * SH(14, c) * 2.f * xz +
                               kSH3[6] * SH(15, c) * 3.f * (xx - yy));
                    ddy[c] += (kSH3[0] * SH(9, c) * 3.f * (xx - yy) + kSH3[1] * SH(10, c) * xz +
                               kSH3[2] * SH(11, c) * (-3.f * yy + 4.f * zz - xx) +
                               kSH3[3] * SH(12, c) * -3.f * 2.f * yz + kSH3[4] * SH(13, c) * -2.f * xy +
                               kSH3[5] * SH(14, c) * -2.f * yz + kSH3[6] * SH(15, c) * -3.f * 2.f * xy);
                    ddz[c] += (kSH3[1] * SH(10, c) * xy + kSH3[2] * SH(11, c) * 4.f * 2.f * yz +
                               kSH3[3] * SH(12, c) * 3.f * (2.f * zz - xx - yy) +
                               kSH3[4] * SH(13, c) * 4.f * 2.f * xz + kSH3[5] * SH(14, c) * (xx - yy));
                }
            }
        }
    }
#undef SH
#undef DSH
    const float vx = ddx[0] * g[0] + ddx[1] * g[1] + ddx[2] * g[2];
    const float vy = ddy[0] * g[0] + ddy[1] * g[1] + ddy[2] * g[2];
    const float vz = ddz[0] * g[0] + ddz[1] * g[1] + ddz[2] * g[2];
    // gradient through the direction normalisation (ch3 auxiliary.h:95-105)
    const float sum2 = ox * ox + oy * oy + oz * oz;
    const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    const float o0 = ((+sum2 - ox * ox) * vx - oy * ox * vy - oz * ox * vz) * invsum32;
    const float o1 = (-ox * oy * vx + (sum2 - oy * oy) * vy - oz * oy * vz) * invsum32;
    const float o2 = (-ox * oz * vx - oy * oz * vy + (sum2 - oz * oz) * vz) * invsum32;
    gmean[0] = o0;
    gmean[1] = o1;
    gmean[2] = o2;
}

__device__ __forceinline__ float dot3(const float *a, const float *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// scale/rotation backward (ch3 backward.cu:267-327; raw-quaternion gradient, :326)
__device__ inline void cov3d_backward(int idx, const float *scale, float mod, const float *rot,
                                      const float *dL_dcov3Ds, float *dL_dscales, float *dL_drots) {
    const float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    const M3 R = m3_cols(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                         2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                         2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
    M3 S = m3_cols(1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f);
    const float s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
    S.m[0][0] = s[0];
    S.m[1][1] = s[1];
    S.m[2][2] = s[2];
    const M3 Mm = m3_mul(S, R);
    const float *d = dL_dcov3Ds + 6 * (size_t)idx;
    const M3 dSigma = m3_cols(d[0], 0.5f * d[1], 0.5f * d[2], 0.5f * d[1], d[3], 0.5f * d[4], 0.5f * d[2],
                              0.5f * d[4], d[5]);
    M3 M2;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int rr = 0; rr < 3; rr++) M2.m[c][rr] = 2.0f * Mm.m[c][rr];
    const M3 dM = m3_mul(M2, dSigma);
    const M3 Rt = m3_t(R);
    M3 dMt = m3_t(dM);
    dL_dscales[3 * (size_t)idx + 0] = dot3(Rt.m[0], dMt.m[0]);
    dL_dscales[3 * (size_t)idx + 1] = dot3(Rt.m[1], dMt.m[1]);
    dL_dscales[3 * (size_t)idx + 2] = dot3(Rt.m[2], dMt.m[2]);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        dMt.m[0][k] *= s[0];
        dMt.m[1][k] *= s[1];
        dMt.m[2][k] *= s[2];
    }
#define D(i, j) dMt.m[i][j]
    const float qx = 2 * z * (D(0, 1) - D(1, 0)) + 2 * y * (D(2, 0) - D(0, 2)) + 2 * x * (D(1, 2) - D(2, 1));
    const float qy = 2 * y * (D(1, 0) + D(0, 1)) + 2 * z * (D(2, 0) + D(0, 2)) + 2 * r * (D(1, 2) - D(2, 1)) -
                     4 * x * (D(2, 2) + D(1, 1));
    const float qz = 2 * x * (D(1, 0) + D(0, 1)) + 2 * r * (D(2, 0) - D(0, 2)) + 2 * z * (D(1, 2) + D(2, 1)) -
                     4 * y * (D(2, 2) + D(0, 0));
    const float qw = 2 * r * (D(0, 1) - D(1, 0)) + 2 * x * (D(2, 0) + D(0, 2)) + 2 * y * (D(1, 2) + D(2, 1)) -
                     4 * z * (D(1, 1) + D(0, 0));
#undef D
    dL_drots[4 * (size_t)idx + 0] = qx;
    dL_drots[4 * (size_t)idx + 1] = qy;
    dL_drots[4 * (size_t)idx + 2] = qz;
    dL_drots[4 * (size_t)idx + 3] = qw;
}

// Screen-space gradients of one splat in one view -> contribution to the mean (gm) and the world
// covariance (dcv): EWA Jacobian with clamp masks, then the projection Jacobian of the 2D mean.
template <bool WANT_COV>
__device__ inline void geom_backward_view(const float3 mean, const float *cov3D, const float *view, const float *proj,
                                          float h_x, float h_y, float tan_fovx, float tan_fovy, float gc0, float gc1,
                                          float gc2, float g0, float g1, float *gm, float *dcv) {
    float3 t = xform4x3(mean, view);
    const float limx = 1.3f * tan_fovx;
    const float limy = 1.3f * tan_fovy;
    const float txtz = t.x / t.z;
    const float tytz = t.y / t.z;
    t.x = fminf(limx, fmaxf(-limx, txtz)) * t.z;
    t.y = fminf(limy, fmaxf(-limy, tytz)) * t.z;
    const float x_grad_mul = txtz < -limx || txtz > limx ? 0 : 1;
    const float y_grad_mul = tytz < -limy || tytz > limy ? 0 : 1;
    const M3 J = m3_cols(h_x / t.z, 0.0f, -(h_x * t.x) / (t.z * t.z), 0.0f, h_y / t.z, -(h_y * t.y) / (t.z * t.z), 0.f,
                         0.f, 0.f);
    const M3 Wm = m3_cols(view[0], view[4], view[8], view[1], view[5], view[9], view[2], view[6], view[10]);
    const M3 Vrk = m3_cols(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
    const M3 Tm_ = m3_mul(Wm, J);
    M3 cov2D = m3_mul(m3_mul(m3_t(Tm_), m3_t(Vrk)), Tm_);
    const float a = cov2D.m[0][0] += 0.3f;
    const float b = cov2D.m[0][1];
    const float c = cov2D.m[1][1] += 0.3f;
    const float denom = a * c - b * b;
    float dL_da = 0, dL_db = 0, dL_dc = 0;
    const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
#define Tm(i, j) Tm_.m[i][j]
#define Vm(i, j) Vrk.m[i][j]
    if (denom2inv != 0) {
        dL_da = denom2inv * (-c * c * gc0 + 2 * b * c * gc1 + (denom - a * c) * gc2);
        dL_dc = denom2inv * (-a * a * gc2 + 2 * a * b * gc1 + (denom - a * c) * gc0);
        dL_db = denom2inv * 2 * (b * c * gc0 - (denom + 2 * b * b) * gc1 + a * b * gc2);
        if (WANT_COV) {  // (the positions-only backward has no use for the covariance's gradient)
            dcv[0] = (Tm(0, 0) * Tm(0, 0) * dL_da + Tm(0, 0) * Tm(1, 0) * dL_db + Tm(1, 0) * Tm(1, 0) * dL_dc);
            dcv[3] = (Tm(0, 1) * Tm(0, 1) * dL_da + Tm(0, 1) * Tm(1, 1) * dL_db + Tm(1, 1) * Tm(1, 1) * dL_dc);
            dcv[5] = (Tm(0, 2) * Tm(0, 2) * dL_da + Tm(0, 2) * Tm(1, 2) * dL_db + Tm(1, 2) * Tm(1, 2) * dL_dc);
            dcv[1] = 2 * Tm(0, 0) * Tm(0, 1) * dL_da + (Tm(0, 0) * Tm(1, 1) + Tm(0, 1) * Tm(1, 0)) * dL_db +
                     2 * Tm(1, 0) * Tm(1, 1) * dL_dc;
            dcv[2] = 2 * Tm(0, 0) * Tm(0, 2) * dL_da + (Tm(0, 0) * Tm(1, 2) + Tm(0, 2) * Tm(1, 0)) * dL_db +
                     2 * Tm(1, 0) * Tm(1, 2) * dL_dc;
            dcv[4] = 2 * Tm(0, 2) * Tm(0, 1) * dL_da + (Tm(0, 1) * Tm(1, 2) + Tm(0, 2) * Tm(1, 1)) * dL_db +
                     2 * Tm(1, 1) * Tm(1, 2) * dL_dc;
        }
    } else if (WANT_COV) {
#pragma unroll
        for (int i = 0; i < 6; i++) dcv[i] = 0;
    }
    const float dL_dT00 = 2 * (Tm(0, 0) * Vm(0, 0) + Tm(0, 1) * Vm(0, 1) + Tm(0, 2) * Vm(0, 2)) * dL_da +
                          (Tm(1, 0) * Vm(0, 0) + Tm(1, 1) * Vm(0, 1) + Tm(1, 2) * Vm(0, 2)) * dL_db;
    const float dL_dT01 = 2 * (Tm(0, 0) * Vm(1, 0) + Tm(0, 1) * Vm(1, 1) + Tm(0, 2) * Vm(1, 2)) * dL_da +
                          (Tm(1, 0) * Vm(1, 0) + Tm(1, 1) * Vm(1, 1) + Tm(1, 2) * Vm(1, 2)) * dL_db;
    const float dL_dT02 = 2 * (Tm(0, 0) * Vm(2, 0) + Tm(0, 1) * Vm(2, 1) + Tm(0, 2) * Vm(2, 2)) * dL_da +
                          (Tm(1, 0) * Vm(2, 0) + Tm(1, 1) * Vm(2, 1) + Tm(1, 2) * Vm(2, 2)) * dL_db;
    const float dL_dT10 = 2 * (Tm(1, 0) * Vm(0, 0) + Tm(1, 1) * Vm(0, 1) + Tm(1, 2) * Vm(0, 2)) * dL_dc +
                          (Tm(0, 0) * Vm(0, 0) + Tm(0, 1) * Vm(0, 1) + Tm(0, 2) * Vm(0, 2)) * dL_db;
    const float dL_dT11 = 2 * (Tm(1, 0) * Vm(1, 0) + Tm(1, 1) * Vm(1, 1) + Tm(1, 2) * Vm(1, 2)) * dL_dc +
                          (Tm(0, 0) * Vm(1, 0) + Tm(0, 1) * Vm(1, 1) + Tm(0, 2) * Vm(1, 2)) * dL_db;
    const float dL_dT12 = 2 * (Tm(1, 0) * Vm(2, 0) + Tm(1, 1) * Vm(2, 1) + Tm(1, 2) * Vm(2, 2)) * dL_dc +
                          (Tm(0, 0) * Vm(2, 0) + Tm(0, 1) * Vm(2, 1) + Tm(0, 2) * Vm(2, 2)) * dL_db;
#undef Tm
#undef Vm
    const float dL_dJ00 = Wm.m[0][0] * dL_dT00 + Wm.m[0][1] * dL_dT01 + Wm.m[0][2] * dL_dT02;
    const float dL_dJ02 = Wm.m[2][0] * dL_dT00 + Wm.m[2][1] * dL_dT01 + Wm.m[2][2] * dL_dT02;
    const float dL_dJ11 = Wm.m[1][0] * dL_dT10 + Wm.m[1][1] * dL_dT11 + Wm.m[1][2] * dL_dT12;
    const float dL_dJ12 = Wm.m[2][0] * dL_dT10 + Wm.m[2][1] * dL_dT11 + Wm.m[2][2] * dL_dT12;
    const float tz = 1.f / t.z;
    const float tz2 = tz * tz;
    const float tz3 = tz2 * tz;
    const float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
    const float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
    const float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * t.x) * tz3 * dL_dJ02 +
                         (2 * h_y * t.y) * tz3 * dL_dJ12;
    // term 1 of the mean gradient: through the covariance (assigned, ch3 backward.cu:262)
    float gm0 = view[0] * dL_dtx + view[1] * dL_dty + view[2] * dL_dtz;
    float gm1 = view[4] * dL_dtx + view[5] * dL_dty + view[6] * dL_dtz;
    float gm2 = view[8] * dL_dtx + view[9] * dL_dty + view[10] * dL_dtz;

    // term 2: through the projected 2D mean (ch3 backward.cu:358-372)
    const float4 m_hom = xform4x4(mean, proj);
    const float m_w = 1.0f / (m_hom.w + 0.0000001f);
    const float mul1 = (proj[0] * mean.x + proj[4] * mean.y + proj[8] * mean.z + proj[12]) * m_w * m_w;
    const float mul2 = (proj[1] * mean.x + proj[5] * mean.y + proj[9] * mean.z + proj[13]) * m_w * m_w;
    const float dmx = (proj[0] * m_w - proj[3] * mul1) * g0 + (proj[1] * m_w - proj[3] * mul2) * g1;
    const float dmy = (proj[4] * m_w - proj[7] * mul1) * g0 + (proj[5] * m_w - proj[7] * mul2) * g1;
    const float dmz = (proj[8] * m_w - proj[11] * mul1) * g0 + (proj[9] * m_w - proj[11] * mul2) * g1;
    gm[0] = gm0 + dmx;
    gm[1] = gm1 + dmy;
    gm[2] = gm2 + dmz;
}

}  // namespace fnx
#include "raster_backward_lanes.h"
namespace fnx {

// One thread per splat, all views of the batch in turn (a splat's result is a sum over the views
// that see it, formed in view order in registers and written once): conic gradient -> cov2D ->
// cov3D and mean, view-dependent colour (SH), then scale / rotation from the summed cov3D gradient
// (linear in it).  With one view this is the reference's per-call arithmetic, term for term.
template <int C>
__global__ void __launch_bounds__(256)
geom_backward_kernel(int P, int D, int M, const float *__restrict__ means3D, const int *__restrict__ radii,
                     const float *__restrict__ shs, const uint8_t *__restrict__ clamped,
                     const float *__restrict__ scales, const float *__restrict__ rotations, float scale_modifier,
                     const float *__restrict__ cov3Ds, size_t cov3D_stride, const float *__restrict__ view,
                     const float *__restrict__ proj, const float *__restrict__ campos,
                     const float *__restrict__ dL_dmean2D, const float *__restrict__ dL_dconics,
                     const float *__restrict__ dL_dopacity_views, const float *__restrict__ dL_dcolor_views,
                     float *__restrict__ dL_dopacity, float *__restrict__ dL_dcolor, float *__restrict__ dL_dmeans,
                     float *__restrict__ dL_dcov, float *__restrict__ dL_dsh, float *__restrict__ dL_dscale,
                     float *__restrict__ dL_drot, int grad_limit, int V, int sum_appearance, const ViewBatch vb) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P || idx >= grad_limit) return;
    const float3 mean = make_float3(means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]);
    float gm[3] = {0.f, 0.f, 0.f}, dcv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    bool seen = false;
    for (int v = 0; v < V; v++) {
        if (!(radii[(size_t)v * P + idx] > 0)) continue;
        const size_t o = (size_t)v * P + idx;
        const float *cov3D = view_at(cov3Ds, cov3D_stride, v) + 6 * (size_t)idx;
        float gv[3], dv[6];
        geom_backward_view(mean, cov3D, view + 16 * v, proj + 16 * v, vb.focal_x[v], vb.focal_y[v], vb.tan_fovx[v],
                           vb.tan_fovy[v], dL_dconics[4 * o], dL_dconics[4 * o + 1], dL_dconics[4 * o + 3],
                           dL_dmean2D[3 * o], dL_dmean2D[3 * o + 1], gv, dv);
        if (shs) {  // term 3: view-dependent colour (ch3 backward.cu:374-376)
            float gs[3];
            sh_backward(idx, D, M, means3D, campos + 3 * v, shs, view_at(clamped, vb.geom, v),
                        dL_dcolor_views + (size_t)v * P * 3, gs, dL_dsh);
            gv[0] += gs[0];
            gv[1] += gs[1];
            gv[2] += gs[2];
        }
#pragma unroll
        for (int k = 0; k < 3; k++) gm[k] = seen ? gm[k] + gv[k] : gv[k];
#pragma unroll
        for (int k = 0; k < 6; k++) dcv[k] = seen ? dcv[k] + dv[k] : dv[k];
        seen = true;
    }
    if (!seen) return;
#pragma unroll
    for (int k = 0; k < 3; k++) dL_dmeans[3 * (size_t)idx + k] = gm[k];
#pragma unroll
    for (int k = 0; k < 6; k++) dL_dcov[6 * (size_t)idx + k] = dcv[k];
    if (scales) cov3d_backward(idx, scales + 3 * (size_t)idx, scale_modifier, rotations + 4 * (size_t)idx, dL_dcov, dL_dscale, dL_drot);
    if (sum_appearance) {  // V > 1: fold the per-view opacity / colour accumulators
        float so = 0.f;
        for (int v = 0; v < V; v++) so += dL_dopacity_views[(size_t)v * P + idx];
        dL_dopacity[idx] = so;
        if (dL_dcolor) {
#pragma unroll
            for (int ch = 0; ch < C; ch++) {
                float sc = 0.f;
                for (int v = 0; v < V; v++) sc += dL_dcolor_views[((size_t)v * P + idx) * C + ch];
                dL_dcolor[(size_t)idx * C + ch] = sc;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The grid is exactly the workgroups that are resident at a time (the occupancy query: 4 per compute unit at the current register count): a larger grid
// would run its surplus as a second, half-empty round.
template <int C, int MODE, bool FAST, bool DUAL = false, typename... A>
static void launch_blend_backward_tf(int n_cu, hipStream_t s, A... args) {
    static int cache[kMaxDevices];  // resident workgroups per compute unit of THIS kernel variant, per device
    static std::mutex mu;
    const int per_cu = per_device_cached(cache, mu, [](int) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, blend_backward_kernel<C, MODE, FAST, DUAL>, 256, 0) != hipSuccess || n <= 0) n = 4;
        return n;
    });
    hipLaunchKernelGGL((blend_backward_kernel<C, MODE, FAST, DUAL>), dim3(n_cu * per_cu), dim3(256), 0, s, args...);
}
// entries-as-lanes form (raster_backward_lanes.h); takes the row form's arguments without the dual reference
template <int C, int MODE, bool FAST, bool DUAL = false, typename... A>
static void launch_blend_backward_lanes_tf(int n_cu, hipStream_t s, A... args) {
    static int cache[kMaxDevices];
    static std::mutex mu;
    const int per_cu = per_device_cached(cache, mu, [](int) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, blend_backward_lanes_kernel<C, MODE, FAST, DUAL>, 256, 0) != hipSuccess || n <= 0) n = 4;
        return n;
    });
    hipLaunchKernelGGL((blend_backward_lanes_kernel<C, MODE, FAST, DUAL>), dim3(n_cu * per_cu), dim3(256), 0, s, args...);
}
int g_backward_form = -1;  // fnx_set_backward_form: 0 = one pixel per lane (rows), 1 = one entry per lane; -1: FNX_BWD_FORM or the default
static int backward_form() {
    if (g_backward_form < 0) {
        const char *e = getenv("FNX_BWD_FORM");
        g_backward_form = e ? (atoi(e) != 0) : FNX_BWD_FORM_DEFAULT;
    }
    return g_backward_form;
}
void set_backward_form(int form) { g_backward_form = form ? 1 : 0; }
int get_backward_form() { return backward_form(); }
template <int C, int MODE, typename... A>
static void launch_blend_backward_t(int fast, int n_cu, hipStream_t s, const DualRef &du, A... args) {
    if (backward_form() == 1) {
        if (fast) launch_blend_backward_lanes_tf<C, MODE, true>(n_cu, s, args..., du);
        else launch_blend_backward_lanes_tf<C, MODE, false>(n_cu, s, args..., du);
        return;
    }
    if (fast) launch_blend_backward_tf<C, MODE, true>(n_cu, s, args..., du);
    else launch_blend_backward_tf<C, MODE, false>(n_cu, s, args..., du);
}

void launch_blend_backward(int C, int mode, hipStream_t s, int P, int W, int H, const uint32_t *ranges,
                           const uint32_t *point_list, const float *bg, const float4 *blend_rec, const float *final_Ts,
                           const uint32_t *n_contrib, const float *acc_final, const float *dL_dpixels,
                           float *dL_dmean2D, float *dL_dconic,
                           float *dL_dopacity, float *dL_dcolors, const uint32_t *header, uint32_t capacity,
                           uint32_t grad_limit, int V, const ViewBatch &vb, const StaticRef &st, const float *means3D,
                           const float *cov3Ds, size_t cov3D_stride, const float *viewmatrix, const float *projmatrix,
                           float *dL_dmean3D, int fast, uint32_t *status_out, const DualRef &du) {
    const int gx = tiles_x(W), T = gx * tiles_y(H);
    // persistent workgroups striding over the view's work items (their number is only known on the device)
    const int n_cu = device_cu_count();
    if (du.img1 && backward_form() == 1 && FNX_LANES_DUAL) {  // dual mode, entries-as-lanes form
        if (fast) launch_blend_backward_lanes_tf<3, 3, true, true>(n_cu, s, T, gx, ranges, point_list, W, H, bg, blend_rec, final_Ts, n_contrib, acc_final, dL_dpixels, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolors, header, capacity, grad_limit, P, V, st, vb, means3D, cov3Ds, cov3D_stride, viewmatrix, projmatrix, dL_dmean3D, status_out, du);
        else launch_blend_backward_lanes_tf<3, 3, false, true>(n_cu, s, T, gx, ranges, point_list, W, H, bg, blend_rec, final_Ts, n_contrib, acc_final, dL_dpixels, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolors, header, capacity, grad_limit, P, V, st, vb, means3D, cov3Ds, cov3D_stride, viewmatrix, projmatrix, dL_dmean3D, status_out, du);
        return;
    }
    if (du.img1) {  // dual mode (the caller has checked: 3 channels, positions only)
        if (fast) launch_blend_backward_tf<3, 3, true, true>(n_cu, s, T, gx, ranges, point_list, W, H, bg, blend_rec, final_Ts, n_contrib, acc_final, dL_dpixels, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolors, header, capacity, grad_limit, P, V, st, vb, means3D, cov3Ds, cov3D_stride, viewmatrix, projmatrix, dL_dmean3D, status_out, du);
        else launch_blend_backward_tf<3, 3, false, true>(n_cu, s, T, gx, ranges, point_list, W, H, bg, blend_rec, final_Ts, n_contrib, acc_final, dL_dpixels, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolors, header, capacity, grad_limit, P, V, st, vb, means3D, cov3Ds, cov3D_stride, viewmatrix, projmatrix, dL_dmean3D, status_out, du);
        return;
    }
    if (C == 3 && mode == 3) launch_blend_backward_t<3, 3>(fast, n_cu, s, du, T, gx, ranges, point_list, W, H, bg, blend_rec, final_Ts, n_contrib, acc_final, dL_dpixels, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolors, header, capacity, grad_limit, P, V, st, vb, means3D, cov3Ds, cov3D_stride, viewmatrix, projmatrix, dL_dmean3D, status_out);
    else if (mode == 3) launch_blend_backward_t<1, 3>(fast, n_cu, s, du, T, gx, ranges, point_list, W, H, bg, blend_rec, final_Ts, n_contrib, acc_final, dL_dpixels, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolors, header, capacity, grad_limit, P, V, st, vb, means3D, cov3Ds, cov3D_stride, viewmatrix, projmatrix, dL_dmean3D, status_out);
    else if (C == 3 && mode == 2) launch_blend_backward_t<3, 2>(fast, n_cu, s, du, T, gx, ranges, point_list, W, H, bg, blend_rec, final_Ts, n_contrib, acc_final, dL_dpixels, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolors, header, capacity, grad_limit, P, V, st, vb, means3D, cov3Ds, cov3D_stride, viewmatrix, projmatrix, dL_dmean3D, status_out);
    else if (mode == 2) launch_blend_backward_t<1, 2>(fast, n_cu, s, du, T, gx, ranges, point_list, W, H, bg, blend_rec, final_Ts, n_contrib, acc_final, dL_dpixels, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolors, header, capacity, grad_limit, P, V, st, vb, means3D, cov3Ds, cov3D_stride, viewmatrix, projmatrix, dL_dmean3D, status_out);
    else if (C == 3 && mode == 0) launch_blend_backward_t<3, 0>(fast, n_cu, s, du, T, gx, ranges, point_list, W, H, bg, blend_rec, final_Ts, n_contrib, acc_final, dL_dpixels, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolors, header, capacity, grad_limit, P, V, st, vb, means3D, cov3Ds, cov3D_stride, viewmatrix, projmatrix, dL_dmean3D, status_out);
    else if (C == 3) launch_blend_backward_t<3, 1>(fast, n_cu, s, du, T, gx, ranges, point_list, W, H, bg, blend_rec, final_Ts, n_contrib, acc_final, dL_dpixels, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolors, header, capacity, grad_limit, P, V, st, vb, means3D, cov3Ds, cov3D_stride, viewmatrix, projmatrix, dL_dmean3D, status_out);
    else if (mode == 0) launch_blend_backward_t<1, 0>(fast, n_cu, s, du, T, gx, ranges, point_list, W, H, bg, blend_rec, final_Ts, n_contrib, acc_final, dL_dpixels, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolors, header, capacity, grad_limit, P, V, st, vb, means3D, cov3Ds, cov3D_stride, viewmatrix, projmatrix, dL_dmean3D, status_out);
    else launch_blend_backward_t<1, 1>(fast, n_cu, s, du, T, gx, ranges, point_list, W, H, bg, blend_rec, final_Ts, n_contrib, acc_final, dL_dpixels, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolors, header, capacity, grad_limit, P, V, st, vb, means3D, cov3Ds, cov3D_stride, viewmatrix, projmatrix, dL_dmean3D, status_out);
}

void launch_geom_backward(int C, hipStream_t s, int P, int D, int M, const float *means3D, const int *radii,
                          const float *shs, const uint8_t *clamped, const float *scales, const float *rotations,
                          float scale_modifier, const float *cov3Ds, size_t cov3D_stride, const float *view,
                          const float *proj, const float *campos, const float *dL_dmean2D, const float *dL_dconic,
                          const float *dL_dopacity_views, const float *dL_dcolor_views, float *dL_dopacity,
                          float *dL_dcolor, float *dL_dmean3D, float *dL_dcov3D, float *dL_dsh, float *dL_dscale,
                          float *dL_drot, int grad_limit, int V, int sum_appearance, const ViewBatch &vb) {
    const int n = grad_limit < P ? grad_limit : P;
    if (n <= 0) return;
#define FNX_LAUNCH_GB(CC)                                                                                              \
    hipLaunchKernelGGL((geom_backward_kernel<CC>), dim3((n + 255) / 256), dim3(256), 0, s, P, D, M, means3D, radii, shs, \
                       clamped, scales, rotations, scale_modifier, cov3Ds, cov3D_stride, view, proj, campos,           \
                       dL_dmean2D, dL_dconic, dL_dopacity_views, dL_dcolor_views, dL_dopacity, dL_dcolor, dL_dmean3D,  \
                       dL_dcov3D, dL_dsh, dL_dscale, dL_drot, grad_limit, V, sum_appearance, vb)
    if (C == 3) FNX_LAUNCH_GB(3);
    else FNX_LAUNCH_GB(1);
#undef FNX_LAUNCH_GB
}

}  // namespace fnx
