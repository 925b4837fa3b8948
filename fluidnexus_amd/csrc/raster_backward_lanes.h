// Blend backward, ENTRIES-AS-LANES form (round 6; included by raster_backward.hip behind blend_backward_kernel, whose
// work items, staging formats, flush and ticket protocol it shares).
//
// blend_backward_kernel walks four block lists per wave, one 4x4 block per 16-lane row, a LANE = a PIXEL: every lane is
// a serial chain over its block's entries (T <- T (1 - alpha), "what lies behind" <- ...), every (pixel, entry) reads the
// entry's record from LDS, every per-entry sum is a cross-lane fold + an LDS atomic, a wave lasts as long as the longest of
// its four lists and a workgroup as long as its slowest quadrant (VALU 31 % of the issue rate, LDS pipe 51 %
// busy at 12 cycles per instruction, 41 % of the wave cycles waiting).
//
// Here a wave takes ONE block at a time with all 64 lanes: lane (r, e) = row r of the block's pixels x entry e of a
// CHUNK of 16 consecutive entries of the block's list.  The lane keeps its entry's record in registers and steps over the
// four pixels j of its row:
//   * alpha(entry e, pixel (j, r)) -- the forward's arithmetic, operation for operation (decisions are bit-equal);
//   * the transmittance in front of the entry is the pixel's carry times the EXCLUSIVE PRODUCT of (1 - alpha) over the
//     chunk's earlier lanes: a 16-lane multiplicative scan, four DPP row shifts (the reference's recurrence
//     ch3 backward.cu:446-535 re-associated; the forward's recurrence forward.cu:297-361);
//   * "what lies behind the entry" = carry - the INCLUSIVE SUM of alpha T (c . dL): an additive scan of the same shape;
//   * the per-entry sums stay in the lane: over its four pixels dy is one number, so three sums (w, w dx, w dx^2) carry
//     all five moments; the four rows' sums of an entry are folded with v_permlane{32,16}_swap and reach the
//     workgroup's accumulators with one LDS atomic per four values and chunk.
// The four pixels of a step are independent of each other (only the carries chain, from chunk to chunk), the inner loop
// reads no LDS, and the blocks of a tile are dealt to the waves across the quadrants, one block of each quadrant per wave.
// Sums are re-associated against both the reference and blend_backward_kernel: gradients agree within the backward's
// stated fp32 bound; integer decisions (which entries a pixel takes) are the forward's.
#pragma once
#include "raster_backward_lanes_prims.h"

namespace fnx {

// S floats per entry (the bits of USED say which of them) added into arr[S id + component] with lane l taking component
// l % S of entry l / S: an instruction touches 64 / S rows instead of 64.  strip: S x 64 floats of wave-private LDS,
// ids: the wave's 64 entry ids (0xFFFFFFFF: nothing to add) -- LDS is in order per wave, no barrier.
template <int S, unsigned USED>
__device__ __forceinline__ void packed_flush_add(float *strip, const uint32_t *ids, float *arr, int lane, const float (&v)[S]) {
#pragma unroll
    for (int k = 0; k < S; k++) strip[S * lane + k] = v[k];
#pragma unroll
    for (int rnd = 0; rnd < S; rnd++) {
        const int idx = 64 * rnd + lane, c = idx % S;
        const uint32_t id_ = ids[idx / S];
        const float val = strip[idx];
        if (((USED >> c) & 1u) && id_ != 0xFFFFFFFFu) FNX_FLUSH_ADD(&arr[(size_t)S * id_ + c], val);
    }
}

#ifdef FNX_EXP_BCLK  // developer timing (tools/bwd_phases.py --lanes): lane 0 of EVERY wave, g_bwd_clock of raster_backward.hip
// (summed in registers, one set of atomics per wave at the end of the kernel: an atomic per phase and item made the
//  instrumented kernel 16 x slower and charged its own round trips to whichever phase waited for memory next)
#define FNX_LCLK(i) { const unsigned long long tn = clock64(); lclk[i] += tn - t_last; t_last = tn; }
#define FNX_LCNT(i, n) { lclk[i] += (unsigned long long)(n); }
#define FNX_LSUB0() { t_sub = clock64(); }
#define FNX_LSUB(i) { const unsigned long long tn = clock64(); lclk[i] += tn - t_sub; t_sub = tn; }
#else
#define FNX_LCLK(i)
#define FNX_LCNT(i, n)
#define FNX_LSUB0()
#define FNX_LSUB(i)
#endif
// timing experiments (results WRONG): bit 1 no global flush atomics, 2 no mean / covariance gathers, 4 no pixel loads,
// 8 no geometry in the flush, 16 no walk, 32 no record loads, 64 no LDS atomics in the walk, 128 plain LDS stores instead
#ifndef FNX_LABLATE
#define FNX_LABLATE 0
#endif
// 1: every wave adds into an accumulator array of its OWN (plain LDS read-add-write: ds_add_f32 is what the LDS pipe of the
// shared-accumulator build is busy with, 27 us of 222) and the flush adds the four copies; 16 KB more LDS = three
// workgroups per compute unit, so this build also takes three waves per SIMD and requests the flush's gathers in front of
// the walk (the registers are there).  The staged conic / opacity copy (s_rd) goes: the flush rescales the staged values.
#ifndef FNX_LANES_PRIVATE_ACC
#define FNX_LANES_PRIVATE_ACC 1  // measured: 222 -> 203.5 us (the same build with the gathers behind the walk: 208)
#endif
#ifndef FNX_LANES_NO_FOLD
#define FNX_LANES_NO_FOLD 0
#endif
#ifndef FNX_LANES_DYNAMIC_BLOCKS
#define FNX_LANES_DYNAMIC_BLOCKS 0
#endif
#ifndef FNX_LANES_CHUNK_PREFETCH
#define FNX_LANES_CHUNK_PREFETCH 0  // 1: the next chunk's list word and records requested a chunk ahead (10 registers): 236 against 222 us
#endif
#ifndef FNX_LANES_EARLY_RECORDS
#define FNX_LANES_EARLY_RECORDS 0
#endif
#ifndef FNX_LANES_EARLY_GATHER
#define FNX_LANES_EARLY_GATHER 1  // (only in the builds with private accumulators: kEarlyGather)
#endif
#ifndef FNX_BWDL_WAVES
#define FNX_BWDL_WAVES (FNX_LANES_PRIVATE_ACC ? 3 : 4)  // waves per SIMD the register allocation aims at
#endif

// DUAL (fnx_raster_dual_t; C = 3, MODE 3): the second, single-channel image of the per-call splats (blend_forward_kernel) adds
// its share of every dynamic entry's dL/dalpha through its own transmittance / what-lies-behind scans over the same alphas
// (see blend_backward_kernel); three waves per SIMD (the second image's pixel state and scans need the registers).
template <int C, int MODE, bool FAST, bool DUAL = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DUAL ? 3 : FNX_BWDL_WAVES, DUAL ? 3 : FNX_BWDL_WAVES)))
blend_backward_lanes_kernel(int T, int gx, const uint32_t *__restrict__ ranges, const uint32_t *__restrict__ point_list, int W,
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
    constexpr bool kMeans = MODE != 2, kAppearance = MODE == 0 || MODE == 2, kFusedGeom = MODE == 3;
    constexpr int kConic = kMeans ? 2 : 0, kOpac = kConic + 3, kCol = kOpac + 1;  // slots of the per-entry sums (as the row form)
    constexpr int NV = kAppearance ? kCol + C : kOpac;
    constexpr int kListStride = 272;  // 256 entries + a chunk of NULL pointers
    constexpr uint32_t kNullOff = 256u * 16u;
    // staged batch (same records as blend_backward_kernel): slot 256 is a NULL record (alpha = 0)
    __shared__ uint32_t s_id[256];
    __shared__ float4 s_ra[257];  // x, y, conic a, conic b   (FAST: the conic pre-scaled by -log2(e) / 2, -log2(e))
    __shared__ float4 s_rb[257];  // conic c, opacity (FAST: log2 opacity), FAST: colour 0, colour 1 | wants; exact: -, wants
    __shared__ float4 s_rc[257];  // FAST C = 3: colour 2, wants; exact: colour
    // (not for the dual mode and the modes with colour sums: their LDS is at three workgroups' worth already)
#ifndef FNX_LANES_PRIVATE_MAXNV
#define FNX_LANES_PRIVATE_MAXNV 9  // sums per entry up to which a wave gets its own accumulators (7, 9: two workgroups per compute unit, still mode 2: 319-326 -> 307 us, mode 0: 0.259 -> 0.241 ms per view forward + backward; the dual mode: 536 -> 655, off)
#endif
#ifndef FNX_LANES_PRIVATE_DUAL
#define FNX_LANES_PRIVATE_DUAL 0
#endif
    constexpr bool kPrivAcc = FNX_LANES_PRIVATE_ACC != 0 && (!DUAL || FNX_LANES_PRIVATE_DUAL) && NV <= FNX_LANES_PRIVATE_MAXNV;
    constexpr bool kEarlyGather = FNX_LANES_EARLY_GATHER != 0 && kPrivAcc;  // (the private-accumulator builds run three waves per SIMD)
    __shared__ float4 s_rd[(FAST && !kPrivAcc) ? 256 : 1];  // FAST: the entry's own conic and opacity for the flush
    constexpr int kAccStride = 272;
    constexpr int kAccCopies = kPrivAcc ? 4 : 1;  // kPrivAcc: one accumulator array per wave
    __shared__ float s_acc[kAccCopies][NV][kAccStride];
    __shared__ __attribute__((aligned(16))) uint16_t s_list[4][kListStride];  // the list of the block a wave is walking
    __shared__ uint16_t s_mask[256];
    __shared__ uint32_t s_bmax[16];
    // the pixels' state in front of the batch, index 16 block + 4 row + column (= the staging thread's index)
    __shared__ __attribute__((aligned(16))) float s_pT[256], s_pR[256], s_pD[C][256];
    __shared__ __attribute__((aligned(16))) uint32_t s_pL[256];
    // DUAL: the second image's pixel (T in front of the batch, what lies behind, walking limit, dL/dpixel)
    __shared__ __attribute__((aligned(16))) float s_qT[DUAL ? 256 : 4], s_qR[DUAL ? 256 : 4], s_qD[DUAL ? 256 : 4];
    __shared__ __attribute__((aligned(16))) uint32_t s_qL[DUAL ? 256 : 4];
    __shared__ uint32_t s_first[kMaxViews + 1];
    __shared__ uint32_t s_view_items[kMaxViews];
    __shared__ uint32_t s_tk;
#if FNX_LANES_DYNAMIC_BLOCKS
    __shared__ uint32_t s_next_block;
#endif
    __shared__ float s_mz[kFusedGeom ? 256 : 1];  // positions-only flush: the splats' world z (x, y ride in spare record words)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;  // ch3 backward.cu:444-445
    if (tid == 0) {
        s_ra[256] = make_float4(0.f, 0.f, 0.f, 0.f);
        s_rb[256] = FAST ? make_float4(0.f, -200.0f, 0.f, 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
        s_rc[256] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (tid < n_views) {  // (as blend_backward_kernel: which views contribute items)
        const int v = tid;
        const uint32_t *h = view_at(header, vb.img, v);
        const uint32_t h_limit = h[HDR_DYN_LIMIT], h_cap = h[HDR_BIN_CAPACITY], h_nr = h[HDR_NUM_RENDERED],
                       h_status = h[HDR_STATUS], h_items = h[HDR_BWD_ITEMS];
        const bool cut = grad_limit > h_limit || (DUAL && grad_limit != h_limit);
        const bool mismatch = h_cap != capacity || cut;
        if (mismatch && blockIdx.x == 0) {
            const_cast<uint32_t *>(h)[HDR_STATUS] = cut ? FNX_ERR_INVALID_ARG : FNX_ERR_CAPACITY;
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
    struct Fetched {
        uint32_t item;
        int vw;
        uint32_t r0, r1;
        uint32_t id, qm;
        float4 ra;
        float rbx, rby, rcz, rcw, rdx;
        float mx, my, mz;  // the splat's world position (record words 13 .. 15, written by the preprocess)
    };
    constexpr uint32_t kNoItem = 0xFFFFFFFFu;
    __syncthreads();  // s_first is written
    const uint32_t n_total = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_first[n_views]);
    // The descriptor and the tile range of an item are the same for every thread: the ticket is made wave-uniform
    // (v_readfirstlane), so these are SCALAR loads -- they count against lgkmcnt, not against vmcnt, i.e. they are not
    // queued behind the flush's global atomics (vmcnt retires in order: a vector load issued behind the atomics is only
    // seen when they have all returned, and the ablations priced that at a third of the kernel).
    auto fetch_item = [&](uint32_t t_, Fetched &f) {
        const uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t_);
        f.item = kNoItem;
        f.vw = 0;
        if (t < n_total) {
            uint32_t first = 0;
            for (int v = 1; v < n_views; v++) {
                const uint32_t fv = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_first[v]);
                if (t >= fv) f.vw = v, first = fv;
            }
            const uint32_t *items = reinterpret_cast<const uint32_t *>(
                reinterpret_cast<const char *>(view_at(point_list, vb.bin, f.vw)) + vb.bin_items);
            f.item = items[t - first];
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
        if (FNX_LABLATE & 32) {
            f.ra = make_float4(100.f, 100.f, 1.f, 0.f);
            f.rbx = 1.f, f.rby = 0.5f, f.rcz = f.rcw = f.rdx = 0.3f;
            f.mx = f.my = f.mz = 0.1f;
        } else if (f.item != kNoItem && pos < f.r1) {
            const float4 *rec = (st.base && f.id >= st.id0)
                ? reinterpret_cast<const float4 *>(st.base + st.stride * f.vw + st.rec) + 4 * (size_t)(f.id - st.id0)
                : view_at(blend_rec, vb.geom, f.vw) + 4 * (size_t)f.id;
            const float4 rb = rec[1], rc = rec[2];
            f.ra = rec[0];
            f.rbx = rb.x;
            f.rby = rb.y;
            f.rcz = rc.z;
            f.rcw = rc.w;
            if (kFusedGeom) {
                const float4 rd = rec[3];
                f.rdx = rd.x;
                f.mx = rd.y;
                f.my = rd.z;
                f.mz = rd.w;
            } else {
                f.rdx = C > 2 ? rec[3].x : 0.f;
            }
        }
    };
    // tickets: as blend_backward_kernel (the first two items of a workgroup are fixed, the rest drawn from a device
    // counter, scrambled so that workgroups running at the same time work far apart in the queue)
    const uint32_t n_all = max(n_total, 1u);
    const unsigned long long mult = (n_all % 7919u) ? 7919ull : 7927ull;
    auto scramble = [&](uint32_t t) -> uint32_t { return t < n_all ? (uint32_t)(((unsigned long long)t * mult) % n_all) : 0xFFFFFFF0u; };
    uint32_t dyn_next = 0xFFFFFFF0u;
    auto ticket_of = [&](uint32_t sidx) -> uint32_t {
        return sidx < 2 ? scramble(blockIdx.x + sidx * gridDim.x) : scramble(dyn_next);
    };
    uint32_t walked = 0u;
    Fetched cur, nxt;
    fetch_item(ticket_of(0), cur);
    fetch_item(ticket_of(1), nxt);
    fetch_range(cur);
    fetch_range(nxt);
    fetch_ids(cur);
    fetch_records(cur);
    // the staging thread's pixel (tid = 16 block + 4 row + column, the forward's lane order: fnx_device.h blend_pixel_*)
    struct PixelsAhead {
        float T_final;
        uint32_t last_contributor;
        float dL[C], total[C];
        float4 stt;
        float T_final1, dL1, total1;  // DUAL: the second image's pixel
        uint32_t last1;
        float2 stt1;
    } ahead;
    auto load_ahead = [&](const Fetched &f) {
        if (FNX_LABLATE & 4) {
            ahead.T_final = 0.5f;
            ahead.last_contributor = 0xFFFFu;
            for (int ch = 0; ch < C; ch++) ahead.dL[ch] = 0.1f, ahead.total[ch] = 0.2f;
            ahead.stt = make_float4(1.f, 0.f, 0.f, 0.f);
            return;
        }
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
    if (cur.item != kNoItem) load_ahead(cur);
    // everything the first item was requested is waited for here (see blend_backward_kernel: a register in flight on the
    // way into the loop costs every iteration a vmcnt(0) at its first use)
    asm volatile("" ::"v"(cur.id), "v"(cur.qm), "v"(cur.ra.x), "v"(cur.ra.y), "v"(cur.ra.z), "v"(cur.ra.w), "v"(cur.rbx),
                 "v"(cur.rby), "v"(cur.rcz), "v"(cur.rcw), "v"(cur.rdx), "v"(cur.mx), "v"(cur.my), "v"(cur.mz), "v"(ahead.T_final), "v"(ahead.last_contributor),
                 "v"(ahead.dL[0]), "v"(ahead.dL[C > 1 ? 1 : 0]), "v"(ahead.dL[C > 2 ? 2 : 0]), "v"(ahead.total[0]),
                 "v"(ahead.total[C > 1 ? 1 : 0]), "v"(ahead.total[C > 2 ? 2 : 0]), "v"(ahead.stt.x), "v"(ahead.stt.y),
                 "v"(ahead.stt.z), "v"(ahead.stt.w), "v"(nxt.item));
    if (DUAL)
        asm volatile("" ::"v"(ahead.T_final1), "v"(ahead.last1), "v"(ahead.dL1), "v"(ahead.total1), "v"(ahead.stt1.x), "v"(ahead.stt1.y));
    const int r = lane >> 4, e = lane & 15;
#ifdef FNX_EXP_BCLK
    unsigned long long lclk[32];
    for (int i = 0; i < 32; i++) lclk[i] = 0;
    unsigned long long t_last = clock64(), t_sub = t_last;
#endif
    // The gradients of an item are ADDED at the top of the NEXT iteration, behind the request for the ids of the item after
    // it: from there the atomics have a whole staging + walk to retire before anything is waited for (the first wait is for
    // those ids, behind the walk), and every other wait of an iteration is for a load that was issued in front of them.
    // Issued at the end of their own iteration they sat in front of the next item's first loads: 100 of 304 us (ablation).
    constexpr int kFl = kFusedGeom ? 3 : (kMeans ? 2 : 0) + 3 + (kAppearance ? 1 + C : 0);
    float pf_fl[kFl];
    bool pf_do = false;
    uint32_t pf_id = 0;
    int pf_vw = 0;
    // Positions-only mode: the three components of a splat's gradient go out in ONE instruction, not three.  A global fp32
    // atomic is priced per memory REQUEST (a 128-byte line per instruction), not per lane -- tools/micro/atomic_rate.hip:
    // 4.7 M atomics at random splats 233 us (20 G/s), at consecutive addresses 46 us -- and a tile's entries hit 64 different
    // splats per wave: with lane l adding component l % 3 of entry l / 3 an instruction touches 22 lines instead of 64, a
    // third of the requests for the same sums.  The values change lanes through a wave-private strip of LDS (in-order per
    // wave: no barrier).
    // The same for the per-view screen-space arrays of the other modes: the components of one array go out together
    // (mean2D: 2 of a 3-float row, conic: 3 of a 4-float row, colours: C), lane l adding component l % S of entry l / S.
    __shared__ float s_pfv[4][256];
    __shared__ uint32_t s_pfi[4][64];
    auto issue_pending = [&]() {
        if (__ballot(pf_do) == 0ull) return;
        s_pfi[w][lane] = pf_do ? pf_id : 0xFFFFFFFFu;
        if (kFusedGeom) {
            const float v[3] = {pf_fl[0], pf_fl[1 < kFl ? 1 : 0], pf_fl[2 < kFl ? 2 : 0]};
            packed_flush_add<3, 7u>(s_pfv[w], s_pfi[w], dL_dmean3D, lane, v);
        } else {
            // (a wave's items come from one view: pf_vw is wave-uniform)
            const int vw_ = __builtin_amdgcn_readfirstlane(pf_vw);
            int o = 0;
            if (kMeans) {
                const float v[3] = {pf_fl[0], pf_fl[1 < kFl ? 1 : 0], 0.f};
                packed_flush_add<3, 3u>(s_pfv[w], s_pfi[w], dL_dmean2D + (size_t)vw_ * P * 3, lane, v);
                o = 2;
            }
            {
                const float v[4] = {pf_fl[o < kFl ? o : 0], pf_fl[o + 1 < kFl ? o + 1 : 0], 0.f, pf_fl[o + 2 < kFl ? o + 2 : 0]};
                packed_flush_add<4, 11u>(s_pfv[w], s_pfi[w], dL_dconic + (size_t)vw_ * P * 4, lane, v);
                o += 3;
            }
            if (kAppearance) {
                if (pf_do) FNX_FLUSH_ADD(&(dL_dopacity + (size_t)vw_ * P)[pf_id], pf_fl[o < kFl ? o : 0]);
                o += 1;
                if (C == 3) {
                    const float v[3] = {pf_fl[o < kFl ? o : 0], pf_fl[o + 1 < kFl ? o + 1 : 0], pf_fl[o + 2 < kFl ? o + 2 : 0]};
                    packed_flush_add<3, 7u>(s_pfv[w], s_pfi[w], dL_dcolors + (size_t)vw_ * P * 3, lane, v);
                } else if (pf_do) {
                    FNX_FLUSH_ADD(&(dL_dcolors + (size_t)vw_ * P)[pf_id], pf_fl[o < kFl ? o : 0]);
                }
            }
        }
        pf_do = false;
    };
    for (uint32_t sidx = 0;; sidx++) {
        FNX_LCLK(0)  // loop back-edge
        if (cur.item == kNoItem) break;
        const int vw = cur.vw;
        Fetched nx2;
        uint32_t drawn = 0;
        if (tid == 0) {  // the ticket for the item after next (see blend_backward_kernel on why this is an instruction)
            uint32_t *tk = const_cast<uint32_t *>(header) + HDR_BWD_TICKET;
            const uint32_t one = 1u;
            asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(drawn) : "v"(tk), "v"(one) : "memory");
        }
        fetch_ids(nxt);   // (its range was requested at the end of the previous iteration)
        issue_pending();  // the previous item's gradients
        FNX_LCLK(7)       // ids request + flush atomics
        const uint32_t item = cur.item;
        const int tile = (int)(item & kItemTileMask);
        const uint32_t b = item >> kItemTileBits;
        const int tx = tile % gx, ty = tile / gx;
        const uint32_t q0 = b << 8;
        const uint32_t cnt = min(256u, cur.r1 - cur.r0 - q0);  // entries of the batch
        walked += cnt;

        // ---- staging: this thread's pixel, this thread's entry -------------------------------------------------------
        {
            const float T_final = ahead.T_final;
            const uint32_t last_contributor = ahead.last_contributor;
            float bg_dot = 0.f, total_dot = 0.f, pre_dot = 0.f;
#pragma unroll
            for (int ch = 0; ch < C; ch++) {
                bg_dot += bg[ch] * ahead.dL[ch];
                total_dot += ahead.total[ch] * ahead.dL[ch];
            }
            const float pre[3] = {b ? ahead.stt.y : 0.f, b ? ahead.stt.z : 0.f, b ? ahead.stt.w : 0.f};
#pragma unroll
            for (int ch = 0; ch < C; ch++) pre_dot += pre[ch] * ahead.dL[ch];
            // what lies behind the batch's first entry: (total - prefix) . dL + the background's share
            s_pT[tid] = b ? ahead.stt.x : 1.0f;
            s_pR[tid] = (total_dot - pre_dot) + T_final * bg_dot;
            // entry q0 + slot lies in front of the pixel's last contributor <=> its LDS offset (16 slot) is below this
            s_pL[tid] = last_contributor > q0 ? min(last_contributor - q0, 256u) << 4 : 0u;
#pragma unroll
            for (int ch = 0; ch < C; ch++) s_pD[ch][tid] = ahead.dL[ch];
            if (DUAL) {
                s_qT[DUAL ? tid : 0] = b ? ahead.stt1.x : 1.0f;
                s_qR[DUAL ? tid : 0] = (ahead.total1 * ahead.dL1 - (b ? ahead.stt1.y : 0.f) * ahead.dL1) + ahead.T_final1 * du.bg1[0] * ahead.dL1;
                s_qL[DUAL ? tid : 0] = ahead.last1 > q0 ? min(ahead.last1 - q0, 256u) << 4 : 0u;
                s_qD[DUAL ? tid : 0] = ahead.dL1;
            }
            uint32_t m = DUAL ? max(last_contributor, ahead.last1) : last_contributor;  // a block needs nothing behind its own deepest contributor
            for (int off = 8; off >= 1; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off));
            if (e == 0) s_bmax[tid >> 4] = m;
        }
        if ((uint32_t)tid < cnt) {
            const uint32_t id = cur.id;
            s_id[tid] = id;
            const float wants_f = id < grad_limit ? 1.0f : 0.0f;
            if (FAST) {
                constexpr float kL2e = 1.44269504088896341f;
                s_ra[tid] = make_float4(cur.ra.x, cur.ra.y, (-0.5f * kL2e) * cur.ra.z, (-kL2e) * cur.ra.w);
                s_rb[tid] = make_float4((-0.5f * kL2e) * cur.rbx, __builtin_amdgcn_logf(fmaxf(cur.rby, 0.0f)), cur.rcz,
                                        C == 3 ? cur.rcw : wants_f);
                if (C == 3 || kFusedGeom) s_rc[tid] = make_float4(cur.rdx, wants_f, kFusedGeom ? cur.mx : 0.f, kFusedGeom ? cur.my : 0.f);
                if (!kPrivAcc) s_rd[(FAST && !kPrivAcc) ? tid : 0] = make_float4(cur.ra.z, cur.ra.w, cur.rbx, cur.rby);
            } else {
                s_ra[tid] = cur.ra;
                s_rb[tid] = make_float4(cur.rbx, cur.rby, kFusedGeom ? cur.mx : 0.f, wants_f);
                s_rc[tid] = make_float4(cur.rcz, C > 1 ? cur.rcw : 0.f, C > 2 ? cur.rdx : 0.f, kFusedGeom ? cur.my : 0.f);
            }
        }
        if (kFusedGeom) s_mz[kFusedGeom ? tid : 0] = cur.mz;
#pragma unroll
        for (int v = 0; v < NV; v++)
#pragma unroll
            for (int cp = 0; cp < kAccCopies; cp++) s_acc[cp][v][tid] = 0.f;  // (own slot in every copy: only this thread reads it)
        s_mask[tid] = (uint16_t)((uint32_t)tid < cnt ? cur.qm : 0u);
#if FNX_LANES_DYNAMIC_BLOCKS
        if (tid == 0) s_next_block = 4u;  // blocks 0 .. 3 (in draw order) are the waves' first
#endif
        FNX_LCLK(1)  // staging
        fnx::lds_barrier();  // B: the batch is staged
        FNX_LCLK(2)  // wait at barrier B
        if (w == 0) { FNX_LCNT(8, 1) FNX_LCNT(9, cnt) }

#if FNX_LANES_EARLY_RECORDS
        fetch_records(nxt);  // lab: the next item's records in flight during the walk (14 registers)
#endif
        // this item's covariance (the mean came with the record) for the flush: requested HERE, in front of the walk, by the
        // builds that have the registers for it (kEarlyGather: three waves per SIMD) -- requested behind the walk the round trip
        // sits in front of every flush (ablation: 18 of 239 us); at four waves per SIMD it spills (249 against 240 us)
        float gmean[3] = {0.f, 0.f, 0.f}, gcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        auto request_cov = [&](uint32_t gid) {
            if (FNX_LABLATE & 2) {
                gmean[0] = 0.1f, gmean[1] = 0.2f, gmean[2] = -0.3f;
                gcov[0] = gcov[3] = gcov[5] = 1e-4f;
            } else if (gid < grad_limit) {
                const float *cv = view_at(cov3Ds, cov3D_stride, vw) + 6 * (size_t)gid;
#pragma unroll
                for (int kx = 0; kx < 6; kx++) gcov[kx] = cv[kx];
            }
        };
        if (kEarlyGather && kFusedGeom && (uint32_t)tid < cnt) request_cov(cur.id);
        // ---- walk: one block of every quadrant per wave ----------------------------------------------------------------
#if FNX_LANES_DYNAMIC_BLOCKS
        // the sixteen blocks are handed out by an LDS counter (a wave that drew short lists takes more of them); the first
        // block of every wave is fixed (no round trip in front of it)
#pragma unroll 1
        for (int kb = w; kb < ((FNX_LABLATE & 16) ? 0 : 16);) {
            const int k = ((kb & 3) << 2) | (kb >> 2);  // consecutive draws go to different quadrants
            const int qi = k >> 2, bsub = k & 3;
            {
                int nk = 0;
                if (lane == 0) nk = (int)atomicAdd(&s_next_block, 1u);
                kb = __builtin_amdgcn_readfirstlane(nk);  // the NEXT block of this wave, requested under this one's walk
            }
#else
#pragma unroll 1
        for (int qi = 0; qi < ((FNX_LABLATE & 16) ? 0 : 4); qi++) {
            const int bsub = (w + qi) & 3;
            const int k = 4 * qi + bsub;  // bit of the block in the entries' masks (fnx_device.h block_mask_exact)
#endif
            const uint32_t bm_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_bmax[k]);
            const uint32_t nlim = bm_ > q0 ? min(bm_ - q0, 256u) : 0u;
            if (nlim == 0u) continue;
            uint16_t *mylist = s_list[w];
            uint32_t len = 0;
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                const uint32_t slot = 64 * kk + lane;
                const bool bit = (((uint32_t)s_mask[slot] >> k) & 1u) && slot < nlim;
                const unsigned long long bmk = __ballot(bit);
                if (bit) mylist[len + (uint32_t)__popcll(bmk & lt_mask)] = (uint16_t)(slot * 16);
                len += (uint32_t)__popcll(bmk);
            }
            if (len == 0u) continue;
            if (lane < 16) mylist[len + lane] = (uint16_t)kNullOff;  // the last chunk's tail points to the NULL record
            const uint32_t nchunks = (len + 15u) >> 4;
            FNX_LCNT(10, len) FNX_LCNT(11, nchunks) FNX_LCNT(12, 1)
            // the row's four pixels
            const float4 T4 = reinterpret_cast<const float4 *>(s_pT)[4 * k + r];
            const float4 R4 = reinterpret_cast<const float4 *>(s_pR)[4 * k + r];
            const uint4 L4 = reinterpret_cast<const uint4 *>(s_pL)[4 * k + r];
            float Tc[4] = {T4.x, T4.y, T4.z, T4.w}, Rc[4] = {R4.x, R4.y, R4.z, R4.w};
            const uint32_t Lj[4] = {L4.x, L4.y, L4.z, L4.w};
            float dLj[C][4];
#pragma unroll
            for (int ch = 0; ch < C; ch++) {
                const float4 d4 = reinterpret_cast<const float4 *>(s_pD[ch])[4 * k + r];
                dLj[ch][0] = d4.x;
                dLj[ch][1] = d4.y;
                dLj[ch][2] = d4.z;
                dLj[ch][3] = d4.w;
            }
            float Tc1[4] = {1.f, 1.f, 1.f, 1.f}, Rc1[4] = {0.f, 0.f, 0.f, 0.f}, dL1j[4] = {0.f, 0.f, 0.f, 0.f};
            uint32_t L1j[4] = {0u, 0u, 0u, 0u};
            if (DUAL) {
                const float4 t4 = reinterpret_cast<const float4 *>(s_qT)[DUAL ? 4 * k + r : 0];
                const float4 r4 = reinterpret_cast<const float4 *>(s_qR)[DUAL ? 4 * k + r : 0];
                const float4 d4 = reinterpret_cast<const float4 *>(s_qD)[DUAL ? 4 * k + r : 0];
                const uint4 l4 = reinterpret_cast<const uint4 *>(s_qL)[DUAL ? 4 * k + r : 0];
                Tc1[0] = t4.x, Tc1[1] = t4.y, Tc1[2] = t4.z, Tc1[3] = t4.w;
                Rc1[0] = r4.x, Rc1[1] = r4.y, Rc1[2] = r4.z, Rc1[3] = r4.w;
                dL1j[0] = d4.x, dL1j[1] = d4.y, dL1j[2] = d4.z, dL1j[3] = d4.w;
                L1j[0] = l4.x, L1j[1] = l4.y, L1j[2] = l4.z, L1j[3] = l4.w;
            }
            const float pxf0 = (float)(tx * FNX_TILE_X + 8 * (qi & 1) + 4 * (bsub & 1));
            const float pyf = (float)(ty * FNX_TILE_Y + 8 * (qi >> 1) + 4 * (bsub >> 1) + r);
            // the chunk's records are requested one chunk ahead
#if FNX_LANES_CHUNK_PREFETCH
            uint32_t off_n = mylist[e];
            float4 ra_n = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s_ra) + off_n);
            float4 rb_n = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s_rb) + off_n);
            float4 rc_n = (!FAST || C == 3) ? *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s_rc) + off_n)
                                            : make_float4(0.f, 0.f, 0.f, 0.f);
#endif
#pragma unroll 1
            for (uint32_t c = 0; c < nchunks; c++) {
#if FNX_LANES_CHUNK_PREFETCH
                const uint32_t off = off_n;
                const float4 ra = ra_n, rb = rb_n, rc = rc_n;
                if (c + 1 < nchunks) {
                    off_n = mylist[16 * (c + 1) + e];
                    ra_n = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s_ra) + off_n);
                    rb_n = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s_rb) + off_n);
                    if (!FAST || C == 3) rc_n = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s_rc) + off_n);
                }
#else
                const uint32_t off = mylist[16 * c + e];
                const float4 ra = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s_ra) + off);
                const float4 rb = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s_rb) + off);
                const float4 rc = (!FAST || C == 3) ? *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s_rc) + off)
                                                    : make_float4(0.f, 0.f, 0.f, 0.f);
#endif
                const bool wants = (FAST ? (C == 3 ? rc.y : rb.w) : rb.w) != 0.0f;
                float col[3];
                if (FAST) {
                    col[0] = rb.z;
                    col[1] = C > 1 ? rb.w : 0.f;
                    col[2] = C > 2 ? rc.x : 0.f;
                } else {
                    col[0] = rc.x;
                    col[1] = rc.y;
                    col[2] = rc.z;
                }
                const float dy = ra.y - pyf;
                // per chunk: the terms of the exponent that do not depend on the column
                const float k1 = FAST ? ra.w * dy : 0.f;                    // FAST: (-log2e b) dy
                const float k0 = (rb.x * dy) * dy;                          // (c dy) dy, FAST: pre-scaled
                float dx[4], ev[4], a[4], om[4], inv[4], cd[4];
                float a1[4], om1[4], inv1[4];  // DUAL: the second image's alpha (dynamic entries in front of ITS last contributor)
                bool emits[4], act1[4];
                bool any_emit = false;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    dx[j] = ra.x - (pxf0 + (float)j);
                    float alpha;
                    bool hit;
                    if (FAST) {
                        const float u = __builtin_fmaf(ra.z, dx[j], k1);
                        const float q = __builtin_fmaf(u, dx[j], k0);  // log2(e) * power
                        ev[j] = __builtin_amdgcn_exp2f(q + rb.y);      // o G
                        alpha = fminf(0.99f, ev[j]);
                        hit = !(q > 0.0f) && !(alpha < 1.0f / 255.0f);
                    } else {
                        const float power = -0.5f * (ra.z * dx[j] * dx[j] + k0) - ra.w * dx[j] * dy;
                        const float G = exp_fixed_in_range(fmaxf(power, -87.0f));
                        alpha = fminf(0.99f, rb.y * G);
                        hit = !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
                        ev[j] = G;
                    }
                    const bool active = hit && (off < Lj[j]);
                    emits[j] = active && wants;
                    any_emit |= emits[j];
                    a[j] = active ? alpha : 0.0f;
                    act1[j] = DUAL && hit && wants && (off < L1j[j]);  // a dynamic entry takes gradients (limit = the forward's)
                    if (!FAST) ev[j] = (active || act1[j]) ? ev[j] : 0.0f;  // power > 0 can push the range-limited exp out of range
                    om[j] = 1 - a[j];
                    if (DUAL) {  // (1 - alpha) is the same number in both images wherever both take the entry: one reciprocal
                        const float ih = __builtin_amdgcn_rcpf(1 - (hit ? alpha : 0.0f));
                        inv[j] = active ? ih : 1.0f;
                        inv1[j] = act1[j] ? ih : 1.0f;
                        a1[j] = act1[j] ? alpha : 0.0f;
                        om1[j] = 1 - a1[j];
                        any_emit |= act1[j];
                    } else {
                        inv[j] = __builtin_amdgcn_rcpf(om[j]);
                    }
                    float cdot = col[0] * dLj[0][j];
                    if (C > 1) cdot = __builtin_fmaf(col[1], dLj[C > 1 ? 1 : 0][j], cdot);
                    if (C > 2) cdot = __builtin_fmaf(col[2], dLj[C > 2 ? 2 : 0][j], cdot);
                    cd[j] = cdot;
                }
                // transmittance behind every entry of the chunk: the carry times the inclusive product of (1 - alpha)
                float Tn[4] = {om[0], om[1], om[2], om[3]};
                row_scan_mul4(Tn);
                float Tb[4], aT[4], term[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    Tn[j] = Tc[j] * Tn[j];
                    Tb[j] = row_prev(Tn[j], Tc[j]);  // transmittance in FRONT of the entry
                    aT[j] = a[j] * Tb[j];
                    term[j] = aT[j] * cd[j];
                }
                // what lies behind the entry: the carry minus alpha T (c . dL) of the entries up to and including it
                row_scan_add4(term);
                float Tn1[4] = {om1[0], om1[1], om1[2], om1[3]}, Tb1[4], term1[4], cd1[4];
                if (DUAL) {  // the second image: its value per splat is channel 0 of the colour
                    row_scan_mul4(Tn1);
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        Tn1[j] = Tc1[j] * Tn1[j];
                        Tb1[j] = row_prev(Tn1[j], Tc1[j]);
                        cd1[j] = col[0] * dL1j[j];
                        term1[j] = (a1[j] * Tb1[j]) * cd1[j];
                    }
                    row_scan_add4(term1);
                }
                float S0 = 0.f, S1 = 0.f, S2 = 0.f, SO = 0.f, SC[C];  // SO: exact arithmetic's opacity sum (G dL/dalpha)
#pragma unroll
                for (int ch = 0; ch < C; ch++) SC[ch] = 0.f;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float rest = Rc[j] - term[j];
                    float dL_dalpha = __builtin_fmaf(Tb[j], cd[j], -(rest * inv[j]));
                    bool takes = emits[j];
                    if (DUAL) {
                        const float rest1 = Rc1[j] - term1[j];
                        const float dL_dalpha1 = __builtin_fmaf(Tb1[j], cd1[j], -(rest1 * inv1[j]));
                        dL_dalpha = (emits[j] ? dL_dalpha : 0.0f) + (act1[j] ? dL_dalpha1 : 0.0f);
                        takes = emits[j] || act1[j];
                        Tc1[j] = row_last(Tn1[j]);
                        Rc1[j] = row_last(rest1);
                    }
                    // FAST: G dL/dG = (o G) dL/dalpha; exact: G (o dL/dalpha) (ch3 backward.cu:505-512)
                    const float Ge = takes ? ev[j] : 0.0f;
                    const float wgt = FAST ? Ge * dL_dalpha : Ge * (rb.y * dL_dalpha);
                    if (!FAST && kAppearance) SO = __builtin_fmaf(Ge, dL_dalpha, SO);
                    S0 += wgt;
                    S1 = __builtin_fmaf(wgt, dx[j], S1);
                    S2 = __builtin_fmaf(wgt * dx[j], dx[j], S2);
                    if (kAppearance) {
                        const float dch = emits[j] ? aT[j] : 0.0f;
#pragma unroll
                        for (int ch = 0; ch < C; ch++) SC[ch] = __builtin_fmaf(dch, dLj[ch][j], SC[ch]);
                    }
                    Tc[j] = row_last(Tn[j]);
                    Rc[j] = row_last(rest);
                }
                if (__ballot(any_emit) != 0ull) {
                    FNX_LCNT(13, 1)
                    // the lane's sums over its four pixels -> the entry's moments (dy is the row's)
                    float m[NV];
                    const float S0y = S0 * dy;
                    if (kMeans) {
                        m[0] = S1;
                        m[kMeans ? 1 : 0] = S0y;
                    }
                    m[kConic] = S2;
                    m[kConic + 1] = S1 * dy;
                    m[kConic + 2] = S0y * dy;
                    if (kAppearance) {
                        // the row form sums w = (o G) dL/dalpha and divides by o at the flush (FAST); exact: G dL/dalpha
                        m[kAppearance ? kOpac : 0] = FAST ? S0 : SO;
#pragma unroll
                        for (int ch = 0; ch < C; ch++) m[kAppearance ? kCol + ch : 0] = SC[ch];
                    }
                    const uint32_t slot = off >> 4;
#if FNX_LANES_NO_FOLD  // every row adds its own sums: NV LDS atomics with four lanes per address instead of the swaps
                    if (slot < 256u) {
#pragma unroll
                        for (int v = 0; v < NV; v++) atomicAdd(&s_acc[0][0][0] + v * kAccStride + slot, m[v]);
                    }
#else
                    const int vq = ((r & 1) << 1) | (r >> 1);  // which value of a group of four this row ends up with
#pragma unroll
                    for (int g = 0; g < NV; g += 4) {
                        auto mv = [&](int i) -> float { return i < NV ? m[i < NV ? i : 0] : 0.0f; };
                        const float t = rows_fold4(mv(g), mv(g + 1), mv(g + 2), mv(g + 3));
                        if (FNX_LABLATE & 64) {
                            asm volatile("" ::"v"(t));
                        } else if (FNX_LABLATE & 128) {  // plain store instead of the atomic: WRONG sums, what ds_add_f32 costs
                            if (slot < 256u && g + vq < NV) (&s_acc[0][0][0])[(g + vq) * kAccStride + slot] = t;
                        } else if (kPrivAcc) {  // this wave's own array: no other wave touches it before barrier C
                            if (slot < 256u && g + vq < NV) {
                                float *pacc = &s_acc[kPrivAcc ? w : 0][0][0] + (g + vq) * kAccStride + slot;
                                *pacc = *pacc + t;
                            }
                        } else if (slot < 256u && g + vq < NV) atomicAdd(&s_acc[0][0][0] + (g + vq) * kAccStride + slot, t);
                    }
#endif
                }
            }
        }
        FNX_LCLK(3)  // walk (list builds + chunks)
        // the ticket drawn at the top has long arrived (published before the prefetches below are issued)
        FNX_LSUB0()
        if (tid == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            s_tk = 2u * gridDim.x + drawn;
        }
        FNX_LSUB(16)
#ifdef FNX_EXP_BCLK
        asm volatile("" ::"v"(nxt.id), "v"(nxt.qm));
        FNX_LSUB(17)  // wait for the next item's ids
#endif
        if (!kEarlyGather && kFusedGeom && (uint32_t)tid < cnt) request_cov(s_id[tid]);
#if !FNX_LANES_EARLY_RECORDS
        fetch_records(nxt);  // in flight while the accumulators are flushed
#endif
        FNX_LSUB(18)
        if (nxt.item != kNoItem) load_ahead(nxt);
        FNX_LSUB(19)
        FNX_LSUB(20)  // gather requests
        FNX_LCLK(4)  // prefetch requests
        fnx::lds_barrier();  // C: every block of the batch is walked
        FNX_LCLK(5)  // wait at barrier C
        FNX_LSUB0()
        dyn_next = s_tk;
        fetch_item(ticket_of(sidx + 2), nx2);
        FNX_LSUB(21)
        // ---- flush: thread t turns the sums of entry t into gradients (as blend_backward_kernel); added at the next top ----
#pragma unroll
        for (int kx = 0; kx < kFl; kx++) pf_fl[kx] = 0.f;
        pf_vw = vw;  // (every lane: issue_pending reads it as a wave-uniform value)
        if ((uint32_t)tid < cnt) {
            const uint32_t id = s_id[tid];
            float a[NV];
            bool any = false;
#pragma unroll
            for (int v = 0; v < NV; v++) {
                a[v] = s_acc[0][v][tid];
#pragma unroll
                for (int cp = 1; cp < kAccCopies; cp++) a[v] += s_acc[cp < kAccCopies ? cp : 0][v][tid];
                any |= (a[v] != 0.f);
            }
            if (any) {
                pf_do = !(FNX_LABLATE & 1);
                pf_id = id;
                float4 ra = s_ra[tid];
                float cc = s_rb[tid].x;
                if (FAST && kPrivAcc) {  // the staged coefficients are the conic's, pre-scaled; log2 of the opacity
                    constexpr float kL2e = 1.44269504088896341f;
                    const float lo = s_rb[tid].y;
                    ra.z = ra.z * (-2.0f / kL2e);
                    ra.w = ra.w * (-1.0f / kL2e);
                    cc = cc * (-2.0f / kL2e);
                    if (kAppearance) a[kAppearance ? kOpac : 0] = a[kAppearance ? kOpac : 0] / __builtin_amdgcn_exp2f(lo);
                } else if (FAST) {
                    const float4 rd = s_rd[(FAST && !kPrivAcc) ? tid : 0];
                    ra.z = rd.x;
                    ra.w = rd.y;
                    cc = rd.z;
                    if (kAppearance) a[kAppearance ? kOpac : 0] = a[kAppearance ? kOpac : 0] / rd.w;
                }
                const float g0 = kMeans ? -(ra.z * a[0] + ra.w * a[kMeans ? 1 : 0]) * ddelx_dx : 0.f;
                const float g1 = kMeans ? -(cc * a[kMeans ? 1 : 0] + ra.w * a[0]) * ddely_dy : 0.f;
                if (kFusedGeom && (FNX_LABLATE & 8)) {
                    pf_fl[0] = g0 + gmean[0] + gcov[0], pf_fl[1 < kFl ? 1 : 0] = g1 + gmean[1] + gcov[3], pf_fl[2 < kFl ? 2 : 0] = a[kConic] + gmean[2] + gcov[5];
                } else if (kFusedGeom) {
                    // the splat's world position came with its record (preprocess: record words 13 .. 15)
                    const float3 mean = FAST ? make_float3(s_rc[tid].z, s_rc[tid].w, s_mz[kFusedGeom ? tid : 0])
                                             : make_float3(s_rb[tid].z, s_rc[tid].w, s_mz[kFusedGeom ? tid : 0]);
                    (void)gmean;
                    float gv[3];
                    geom_backward_view<false>(mean, gcov, viewmatrix + 16 * vw, projmatrix + 16 * vw, vb.focal_x[vw],
                                              vb.focal_y[vw], vb.tan_fovx[vw], vb.tan_fovy[vw], -0.5f * a[kConic],
                                              -0.5f * a[kConic + 1], -0.5f * a[kConic + 2], g0, g1, gv, nullptr);
#pragma unroll
                    for (int kx = 0; kx < 3; kx++) pf_fl[kx < kFl ? kx : 0] = gv[kx];
                } else {
                    int o = 0;
                    if (kMeans) {
                        pf_fl[o++ < kFl ? o - 1 : 0] = g0;
                        pf_fl[o++ < kFl ? o - 1 : 0] = g1;
                    }
                    pf_fl[o++ < kFl ? o - 1 : 0] = -0.5f * a[kConic];
                    pf_fl[o++ < kFl ? o - 1 : 0] = -0.5f * a[kConic + 1];
                    pf_fl[o++ < kFl ? o - 1 : 0] = -0.5f * a[kConic + 2];
                    if (kAppearance) {
                        pf_fl[o++ < kFl ? o - 1 : 0] = a[kAppearance ? kOpac : 0];
#pragma unroll
                        for (int ch = 0; ch < C; ch++) pf_fl[o++ < kFl ? o - 1 : 0] = a[kAppearance ? kCol + ch : 0];
                    }
                }
            }
        }
        if (FNX_LABLATE & 1) {
            float sink = 0.f;
            for (int kx = 0; kx < kFl; kx++) sink += pf_fl[kx];
            asm volatile("" ::"v"(sink));
        }
        FNX_LSUB(22)  // sums -> gradients (incl. the wait for the gathered mean / covariance)
        fetch_range(nx2);  // scalar; the descriptor was requested behind barrier C.  In front of the wait below: they overlap
        // the next item's records and pixels (requested in front of barrier C) are waited for here, with no atomic between
        // their request and this wait
        asm volatile("" ::"v"(nxt.id), "v"(nxt.qm), "v"(nxt.ra.x), "v"(nxt.ra.y), "v"(nxt.ra.z), "v"(nxt.ra.w), "v"(nxt.rbx),
                     "v"(nxt.rby), "v"(nxt.rcz), "v"(nxt.rcw), "v"(nxt.rdx), "v"(nxt.mx), "v"(nxt.my), "v"(nxt.mz), "v"(ahead.T_final), "v"(ahead.last_contributor),
                     "v"(ahead.dL[0]), "v"(ahead.dL[C > 1 ? 1 : 0]), "v"(ahead.dL[C > 2 ? 2 : 0]), "v"(ahead.total[0]),
                     "v"(ahead.total[C > 1 ? 1 : 0]), "v"(ahead.total[C > 2 ? 2 : 0]), "v"(ahead.stt.x), "v"(ahead.stt.y),
                     "v"(ahead.stt.z), "v"(ahead.stt.w));
        if (DUAL)
            asm volatile("" ::"v"(ahead.T_final1), "v"(ahead.last1), "v"(ahead.dL1), "v"(ahead.total1), "v"(ahead.stt1.x), "v"(ahead.stt1.y));
        FNX_LSUB(23)  // wait for the next item's records / pixels
        FNX_LCLK(6)   // flush: sums -> gradients (+ the wait for the prefetched registers)
        FNX_LCNT(14, __popcll(__ballot(pf_do)))
        cur = nxt;
        nxt.item = nx2.item;
        nxt.vw = nx2.vw;
        nxt.r0 = nx2.r0;
        nxt.r1 = nx2.r1;
    }
    issue_pending();  // the last item's gradients
#ifdef FNX_EXP_BCLK
    if (lane == 0)
        for (int i = 0; i < 32; i++) atomicAdd(&g_bwd_clock[i], lclk[i]);
#endif
    if (tid == 0 && walked) atomicAdd(const_cast<uint32_t *>(header) + HDR_BWD_ENTRIES, walked);
    if (tid == 0) {  // the last workgroup re-arms the counters (as blend_backward_kernel)
        uint32_t *h0 = const_cast<uint32_t *>(header);
        if (atomicAdd(h0 + HDR_BWD_DONE, 1u) == gridDim.x - 1u) {
            __hip_atomic_store(h0 + HDR_BWD_TICKET, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(h0 + HDR_BWD_DONE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace fnx
