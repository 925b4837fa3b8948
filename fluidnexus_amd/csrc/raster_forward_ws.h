// Blend forward with STAGING WAVES (round 6; fnx_raster_opts_t.deep_kernel = 3 / 4).  Textually included by
// raster_forward.hip inside namespace fnx, behind the per-tile kernel (blend_forward_kernel) whose staging code, lists
// and inner loops it shares: per pixel the arithmetic and its order are that kernel's, bit for bit, in both arithmetics.
//
// Why: the launch of the per-tile kernel ends when its deepest tile ends -- a one-view forward of config 3 takes 232 us,
// the five-view one 304 -- and a deep tile is a chain of ~30 batches in which the four waves of the workgroup first STAGE a
// batch (merge of the two streams, records, block masks, per-block lists: 36 % of the tile's cycles, barriers 12 %) and
// then WALK it (49 %), one after the other (per-phase clocks: profiles/r06_lab_staging_waves.md).  A lone wave on a SIMD
// issues an instruction every 8-9 cycles in the walk (dependent chains), the SIMD could issue three times that.
//
// Here a workgroup is 512 threads: waves 0-3 ("walkers") own the 256 pixels and only walk, waves 4-7 ("stagers") own the
// 256 slots of a batch and only stage, one batch ahead, into the other half of double-buffered LDS arrays.  The two
// groups never meet in a hardware barrier inside the batch loop (gfx950 has one barrier per workgroup): the stagers
// synchronise among themselves through an LDS counter, and batches are handed over through LDS words
//     s_staged      batches staged so far (written by stager wave 0 once all four have finished one)
//     s_walked[w]   batches walker w has finished (a buffer is refilled once every walker has left it)
//     s_live[w]     which of walker w's four 4x4 blocks still blend (bits 0-3), bit 4: none of its pixels does
//     s_nbatches    how many batches will be staged in all (written when the stagers stop: every pixel stopped, or
//                   the list ended)
// polled with s_sleep between reads.  Everything ordered through them is LDS traffic: s_waitcnt lgkmcnt(0) in front of a
// write of a flag, the flag read in front of the data (LDS operations of a wave complete in order).
//
// deep_only != 0: the workgroups take the tiles tile_scan_kernel put in front of the tile order because they went deep in
// the previous forward (depth hints), beside the per-tile kernel on a helper stream; that kernel skips them.  deep_only
// == 0: every tile, instead of the per-tile kernel (then this kernel also refreshes the coherent sort's inverse ranks
// and copies the status words out, as the per-tile kernel does).
#ifdef FNX_EXP_CLOCK  // developer timing: per-phase cycles of lane 0 of every wave of workgroup (rank 0, view 0)
__device__ unsigned long long g_ws_clock[128];
extern "C" int fnx_debug_ws_clock(unsigned long long *host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_ws_clock), sizeof(g_ws_clock));
}
#define FNX_WCLK(i) { const unsigned long long tn = clock64(); if (wg_rank == 0 && wg_view == 0 && lane == 0) g_ws_clock[16 * w8 + (i)] += tn - t_last; t_last = tn; }
#else
#define FNX_WCLK(i)
#endif
template <int C, bool SPLIT, bool FAST, bool DUAL = false>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(FNX_FWD_WAVES, FNX_FWD_WAVES)))
blend_forward_ws_kernel(int T, int gx, const uint32_t *__restrict__ ranges, uint32_t *__restrict__ point_list, int W,
                        int H, const float4 *__restrict__ blend_rec, const float *__restrict__ bg,
                        float *__restrict__ final_T, uint32_t *__restrict__ n_contrib, float *__restrict__ out_color,
                        float *__restrict__ out_depth, uint32_t *__restrict__ header, uint32_t capacity,
                        uint32_t *__restrict__ status_out, const uint32_t *__restrict__ tile_count,
                        const uint32_t *__restrict__ dyn_start, float *__restrict__ acc_final,
                        const uint32_t *__restrict__ tile_order, const uint8_t *__restrict__ tile_deep,
                        uint32_t *__restrict__ depth_hint, const StaticRef st, int materialize_all, const ViewBatch vb,
                        int deep_only, uint32_t dyn_limit, const InvUpdate iu, const DualRef du) {
    static_assert(!DUAL || (C == 3 && SPLIT), "dual mode: three channels, static-split lists");
    const char *static_blob = nullptr;
    // workgroup -> (view, rank in the view's tile order): blend_forward_kernel's mapping
    const int wg_linear = blockIdx.y * gridDim.x + blockIdx.x, n_views = gridDim.y;
    const int wg_view = (wg_linear >> 3) % n_views, wg_rank = ((wg_linear >> 3) / n_views) * 8 + (wg_linear & 7);
    if (wg_rank >= T) return;
    if (!deep_only && iu.pairs) {
        const uint2 *pr = view_at(iu.pairs, vb.geom, wg_view);
        uint32_t *inv = reinterpret_cast<uint32_t *>(iu.state + iu.stride * (size_t)wg_view + iu.inv);
        const int per = (iu.P + T - 1) / T, r1_ = min(iu.P, (wg_rank + 1) * per);
        for (int r = wg_rank * per + (int)threadIdx.x; r < r1_; r += 512) {
            const uint32_t id = pr[r].y;
            if (id < (uint32_t)iu.P) inv[id] = (uint32_t)r;
        }
    }
    {
        const int vw = wg_view;
        ranges = view_at(ranges, vb.img, vw);
        final_T = view_at(final_T, vb.img, vw);
        n_contrib = view_at(n_contrib, vb.img, vw);
        header = view_at(header, vb.img, vw);
        point_list = view_at(point_list, vb.bin, vw);
        blend_rec = view_at(blend_rec, vb.geom, vw);
        out_color += (size_t)vw * C * H * W;
        out_depth += (size_t)vw * H * W;
        acc_final = view_at(acc_final, vb.img, vw);
        tile_order = view_at(tile_order, vb.img, vw);
        tile_deep = view_at(tile_deep, vb.img, vw);
        if (depth_hint) depth_hint += (size_t)vw * T;
        if (SPLIT) {
            tile_count = view_at(tile_count, vb.img, vw);
            dyn_start = view_at(dyn_start, vb.img, vw);
            static_blob = st.base + st.stride * vw;
        }
    }
    char *img1 = DUAL ? du.img1 + vb.img * (size_t)wg_view : nullptr;  // DUAL: the second image's per-pixel arrays
    constexpr int kGroup = FNX_FWD_GROUP;
    constexpr int kListStride = (256 + kGroup + 7) & ~7;
#ifndef FNX_WS_BUFFERS
#define FNX_WS_BUFFERS 2
#endif
    constexpr int kBuf = FNX_WS_BUFFERS;  // staged batches in LDS at a time
    __shared__ float4 s_ra[kBuf][257];
    __shared__ float4 s_rb[kBuf][257];
    __shared__ float4 s_rc[kBuf][257];
    __shared__ __attribute__((aligned(16))) uint16_t s_list[kBuf][16][kListStride];
    __shared__ __attribute__((aligned(16))) uint32_t s_len[kBuf][16];  // lengths of the lists of a staged batch
    __shared__ unsigned long long s_dynmask[kBuf][4];
    __shared__ uint16_t s_mask[256];
    __shared__ uint32_t s_wk[SPLIT ? 2 : 1][SPLIT ? 256 : 1];
    __shared__ uint32_t s_wi[SPLIT ? 2 : 1][SPLIT ? 256 : 1];
    __shared__ uint32_t s_adv;
    __shared__ uint32_t s_sbar;      // arrivals at the stagers' own barrier
    __shared__ uint32_t s_staged;    // see above
    __shared__ uint32_t s_nbatches;
    __shared__ uint32_t s_stop;      // the stagers' common decision, taken by stager wave 0 between two of their barriers
    __shared__ __attribute__((aligned(16))) uint32_t s_walked[4];
    __shared__ __attribute__((aligned(16))) uint32_t s_live[4];
    __shared__ uint32_t s_qmax[4], s_qdyn[4], s_qcnt[4];
    if (!deep_only) {
        if (wg_rank == 0 && threadIdx.x == 0) {
            header[HDR_BIN_CAPACITY] = capacity;
            header[HDR_DYN_LIMIT] = dyn_limit;
        }
        if (status_out && wg_rank == 0 && threadIdx.x < 8)
            status_out[8 * wg_view + threadIdx.x] = threadIdx.x == HDR_BIN_CAPACITY ? capacity : header[threadIdx.x];
    }
    if (header[HDR_NUM_RENDERED] > capacity || header[HDR_STATUS] == (uint32_t)FNX_ERR_SORT_SPAN) return;
    if (deep_only && (uint32_t)wg_rank >= (header[HDR_DEEP_COUNT] & 0xFFFFFFu)) return;
    const int tile = (int)tile_order[wg_rank];
    if (tile_deep[tile] >= 2) __builtin_amdgcn_s_setprio(FNX_DEEP_PRIO);
    else if (deep_only && !tile_deep[tile]) return;
    const int tx = tile % gx, ty = tile / gx;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool stager = w8 >= 4;
    const int w = w8 & 3;       // the walker's quadrant / the stager's quarter of the slots (and the quadrant it builds lists for)
    const int sid = tid & 255;  // pixel of a walker thread, slot of a stager thread
    if (tid < kBuf) {
        s_ra[tid][256] = make_float4(0.f, 0.f, 0.f, 0.f);
        s_rb[tid][256] = FAST ? make_float4(0.f, -200.0f, 0.f, 0.f) : make_float4(0.f, 0.f, -87.0f, 0.f);
        s_rc[tid][256] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (tid == 0) {
        s_sbar = 0u;
        s_staged = 0u;
        s_nbatches = 0xFFFFFFFFu;
        s_stop = 0u;
    }
    if (tid < 4) {
        s_walked[tid] = 0u;
        s_live[tid] = 0xFu;
    }
    const float tile_x0 = (float)(tx * FNX_TILE_X), tile_y0 = (float)(ty * FNX_TILE_Y);
    const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
    const uint32_t b_lo = r0, b_hi = r1;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    auto lds_load = [](const uint32_t *p) -> uint32_t {
        return __hip_atomic_load(const_cast<uint32_t *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto lds_store = [](uint32_t *p, uint32_t v) {
        __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    // walker-side results (zero in the stagers)
    uint32_t last_contributor = 0, last_dyn = 0, n_blended = 0;
    __syncthreads();
#ifdef FNX_EXP_CLOCK
    unsigned long long t_last = clock64();
    if (wg_rank == 0 && wg_view == 0 && lane == 0) { for (int i = 0; i < 16; i++) g_ws_clock[16 * w8 + i] = 0; g_ws_clock[16 * w8 + 15] = r1 - r0; }
#endif
    if (stager) {
        // ---------------------------------------------------------------- stagers ----
        uint32_t sbar_n = 0;
        auto sbar = [&]() {  // barrier of the four staging waves
            sbar_n += 4u;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(&s_sbar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            while (lds_load(&s_sbar) < sbar_n) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
        };
        uint32_t id_ahead = 0;
        const uint2 *sp = nullptr, *fp = nullptr;
        const float4 *rec_s = nullptr;
        uint32_t ns = 0, nf = 0, si = 0, fj = 0;
        uint2 ws = make_uint2(0u, 0u), wf = ws;
        auto record_of = [&](uint32_t id) -> const float4 * {
            return (SPLIT && id >= st.id0) ? rec_s + 4 * (size_t)(id - st.id0) : blend_rec + 4 * (size_t)id;
        };
        auto load_windows = [&]() {
            ws = (si + (uint32_t)sid < ns) ? sp[si + sid] : make_uint2(0xFFFFFFFFu, 0u);
            wf = (fj + (uint32_t)sid < nf) ? fp[fj + sid] : make_uint2(0xFFFFFFFFu, 0u);
        };
        auto store_windows = [&]() {
            s_wk[0][SPLIT ? sid : 0] = ws.x;
            s_wi[0][SPLIT ? sid : 0] = ws.y;
            s_wk[SPLIT ? 1 : 0][SPLIT ? sid : 0] = wf.x;
            s_wi[SPLIT ? 1 : 0][SPLIT ? sid : 0] = wf.y;
        };
        auto merge_batch = [&](uint32_t cnt_next) -> uint32_t {  // blend_forward_kernel's merge path
            uint32_t id = 0;
            if ((uint32_t)sid < cnt_next) {
                const uint32_t nsw = min(256u, ns - si), nfw = min(256u, nf - fj);
                const uint32_t *ks = s_wk[0], *kf = s_wk[SPLIT ? 1 : 0];
                const uint32_t t = (uint32_t)sid;
#if FNX_MERGE_FAST_PATH
                if (nfw >= cnt_next && (nsw == 0u || kf[cnt_next - 1u] <= ks[0])) {
                    if (t == 0u) s_adv = 0u;
                    return s_wi[SPLIT ? 1 : 0][t];
                }
                if (nsw >= cnt_next && (nfw == 0u || ks[cnt_next - 1u] < kf[0])) {
                    if (t == 0u) s_adv = cnt_next;
                    return s_wi[0][t];
                }
#endif
                uint32_t lo = t > nfw ? t - nfw : 0u, hi = min(t, nsw);
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (ks[mid] < kf[t - mid - 1]) lo = mid + 1; else hi = mid;
                }
                const uint32_t i = lo, j = t - lo;
                const bool from_static = !(j < nfw && (i >= nsw || kf[j] <= ks[i]));
                id = from_static ? s_wi[0][i] : s_wi[SPLIT ? 1 : 0][j];
                if (t == cnt_next - 1) s_adv = i + (from_static ? 1u : 0u);
            }
            return id;
        };
        // Two record sets in flight: set A holds the even batches, set B the odd ones; the records of batch b + 2 are requested
        // into the set batch b has just been staged from, and the merge runs TWO batches ahead -- a set's loads have a whole
        // iteration to arrive (with one set the stagers' chain held a global round trip per batch: requested at the end of a
        // batch's staging, consumed at the top of the next; the per-tile kernel hides that under its walk, the stagers have
        // nothing else to do).
        struct RecSet {
            float4 pa, pb, pc;
            float pd;
            uint32_t id;
        };
        RecSet A, B;
        A.pa = A.pb = A.pc = B.pa = B.pb = B.pc = make_float4(0.f, 0.f, 0.f, 0.f);
        A.pd = B.pd = 0.f;
        A.id = B.id = 0u;
        auto request = [&](RecSet &R, const float4 *rec) {
            R.pa = rec[0];
            R.pb = rec[1];
            R.pc = rec[2];
            if (C > 2) R.pd = rec[3].x;
        };
        if (SPLIT) {
            const uint32_t *starts = reinterpret_cast<const uint32_t *>(static_blob + st.starts);
            const uint32_t s0 = starts[tile];
            ns = starts[tile + 1] - s0;
            sp = reinterpret_cast<const uint2 *>(static_blob + st.pairs) + s0;
            rec_s = reinterpret_cast<const float4 *>(static_blob + st.rec);
            nf = tile_count[tile];
            fp = reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(point_list) + vb.bin_pairs) + dyn_start[tile];
            load_windows();
            store_windows();
            sbar();
            const uint32_t cnt0 = min(256u, b_hi - b_lo);
            A.id = merge_batch(cnt0);
            sbar();
            if (cnt0) {
                const uint32_t a = s_adv;
                si += a;
                fj += cnt0 - a;
            }
            if ((uint32_t)sid < cnt0) request(A, record_of(A.id));
            load_windows();
            const uint32_t cnt1 = b_lo + 256u < b_hi ? min(256u, b_hi - b_lo - 256u) : 0u;
            store_windows();
            sbar();
            B.id = merge_batch(cnt1);
            sbar();
            if (cnt1) {
                const uint32_t a = s_adv;
                si += a;
                fj += cnt1 - a;
            }
            load_windows();
            request(B, record_of(B.id));  // unconditional (see blend_forward_kernel); a slot beyond the batch reads splat 0's
        } else {
            if (b_lo + (uint32_t)sid < b_hi) {
                A.id = point_list[b_lo + sid];
                request(A, blend_rec + 4 * (size_t)A.id);
            }
            if (b_lo + 256u + (uint32_t)sid < b_hi) {
                B.id = point_list[b_lo + 256u + sid];
                request(B, blend_rec + 4 * (size_t)B.id);
            }
            if (b_lo + 512u + (uint32_t)sid < b_hi) id_ahead = point_list[b_lo + 512u + sid];
        }
        bool blending = true;
        uint32_t published = 0;
        // one batch: staged from R (its records arrived an iteration ago), then R is refilled with batch b + 2's
        auto stage = [&](RecSet &R, const uint32_t b, const uint32_t base) -> bool {
            const int p = (int)(b % (uint32_t)kBuf);
            FNX_WCLK(0)
            if (blending) {
                // the stagers' common decision of the previous batch: every pixel has stopped -> nothing more to stage
                if (lds_load(&s_stop)) {
                    if (w == 0 && lane == 0) {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        lds_store(&s_nbatches, published);
                    }
                    if (!SPLIT || !materialize_all) return false;
                    blending = false;
                }
            }
            if (blending && b >= (uint32_t)kBuf) {  // buffer p is free once every walker has finished batch b - kBuf
                for (;;) {
                    const uint32_t m = min(min(lds_load(&s_walked[0]), lds_load(&s_walked[1])),
                                           min(lds_load(&s_walked[2]), lds_load(&s_walked[3])));
                    if (m >= b + 1u - (uint32_t)kBuf) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                asm volatile("" ::: "memory");
            }
            FNX_WCLK(1)
            const uint32_t cnt = min(256u, b_hi - base);
            const uint32_t my_id = R.id;
            uint32_t qm = 0;
            if (blending) {
                const unsigned long long dm = __ballot((uint32_t)sid < cnt && my_id < dyn_limit);
                if (lane == 0) s_dynmask[p][w] = dm;
                if ((uint32_t)sid < cnt) {
                    const float4 pa = R.pa, pb = R.pb, pc = R.pc;
                    const float pd = R.pd;
                    qm = block_mask_exact(pa.x, pa.y, pa.z, pa.w, pb.x, pb.z, pc.x, pc.y, tile_x0, tile_y0);
                    if (FAST) {
                        constexpr float kL2e = 1.44269504088896341f;
                        s_ra[p][sid] = make_float4(pa.x, pa.y, (-0.5f * kL2e) * pa.z, (-kL2e) * pa.w);
                        const float lo = __builtin_amdgcn_logf(fmaxf(pb.y, 0.0f));
                        if (C == 3) {
                            s_rb[p][sid] = make_float4((-0.5f * kL2e) * pb.x, lo, pc.z, pc.w);
                            s_rc[p][sid] = make_float4(pd, pb.w, 0.f, 0.f);
                        } else {
                            s_rb[p][sid] = make_float4((-0.5f * kL2e) * pb.x, lo, pc.z, pb.w);
                        }
                    } else {
                        s_ra[p][sid] = pa;
                        s_rb[p][sid] = pb;
                        s_rc[p][sid] = make_float4(pc.z, C > 1 ? pc.w : 0.f, C > 2 ? pd : 0.f, pb.w);
                    }
                }
            }
            s_mask[sid] = (uint16_t)qm;
            if (blending) {
                const uint4 nul = make_uint4(0x10001000u, 0x10001000u, 0x10001000u, 0x10001000u);
                uint4 *mine = reinterpret_cast<uint4 *>(&s_list[p][4 * w][0]);
                for (int i = lane; i < 4 * kListStride / 8; i += 64) mine[i] = nul;
            }
            if (SPLIT) store_windows();  // (requested at the end of the previous batch: behind the masks, not in front of them)
            FNX_WCLK(2)
            sbar();
            FNX_WCLK(3)
            if (blending) {
                uint32_t len[4] = {0u, 0u, 0u, 0u};
                const uint32_t live = lds_load(&s_live[w]);  // walker w's blocks that still blend (as of its last batch)
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t mk = (uint32_t)s_mask[64 * k + lane] >> (4 * w);
                    // DUAL: a static splat's entry is marked in its list word (staging wave k left the batch's dynamic flags)
                    const uint32_t flag = (DUAL && !((s_dynmask[p][k] >> lane) & 1ull)) ? kListStatic : 0u;
#pragma unroll
                    for (int bb = 0; bb < 4; bb++) {
                        const bool bit = ((mk >> bb) & 1u) && ((live >> bb) & 1u);
                        const unsigned long long m = __ballot(bit);
                        if (bit) s_list[p][4 * w + bb][len[bb] + (uint32_t)__popcll(m & lt_mask)] = (uint16_t)(((64 * k + lane) * 16) | flag);
                        len[bb] += (uint32_t)__popcll(m);
                    }
                }
                if (lane == 0) *reinterpret_cast<uint4 *>(&s_len[p][4 * w]) = make_uint4(len[0], len[1], len[2], len[3]);
                // the decision about the NEXT batch, the same for all four staging waves (read behind the barrier below)
                if (w == 0 && lane == 0) {
                    const uint32_t all = lds_load(&s_live[0]) & lds_load(&s_live[1]) & lds_load(&s_live[2]) & lds_load(&s_live[3]);
                    if (all & 16u) lds_store(&s_stop, 1u);
                }
            }
            FNX_WCLK(4)
            uint32_t n2_cnt = 0, n2_id = 0;  // the batch after next: merged here, its records requested below
            if (SPLIT) {
                n2_cnt = base + 512u < b_hi ? min(256u, b_hi - base - 512u) : 0u;
                n2_id = merge_batch(n2_cnt);
            }
            FNX_WCLK(5)
            if (SPLIT && (uint32_t)sid < cnt) point_list[base + sid] = my_id;
            if ((uint32_t)sid < cnt && blending)
                reinterpret_cast<uint16_t *>(reinterpret_cast<char *>(point_list) + vb.bin_masks)[base + sid] = (uint16_t)qm;
            sbar();
            FNX_WCLK(6)
            if (blending) {
                published = b + 1u;
                if (w == 0 && lane == 0) lds_store(&s_staged, published);  // (every stager's LDS writes were complete at the barrier)
            }
            if (SPLIT) {
                if (n2_cnt) {
                    const uint32_t a = s_adv;
                    si += a;
                    fj += n2_cnt - a;
                }
                // the windows FIRST: vmcnt retires in order, and the next batch's staging waits for the windows, not for these
                // records (they are consumed two batches on)
                load_windows();
                R.id = n2_id;
                request(R, record_of(n2_id));  // unconditional (see blend_forward_kernel)
            } else {
                if (base + 512u + (uint32_t)sid < b_hi) {
                    R.id = id_ahead;
                    request(R, blend_rec + 4 * (size_t)id_ahead);
                }
                if (base + 768u + (uint32_t)sid < b_hi) id_ahead = point_list[base + 768u + sid];
            }
            FNX_WCLK(7)
            return true;
        };
        {
            uint32_t b = 0, base = b_lo;
            for (;;) {
                if (base >= b_hi || !stage(A, b, base)) break;
                b++;
                base += 256u;
                if (base >= b_hi || !stage(B, b, base)) break;
                b++;
                base += 256u;
            }
        }
        if (blending && w == 0 && lane == 0) {  // the list ended (or every pixel stopped at its last batch)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            lds_store(&s_nbatches, published);
        }
    } else {
        // ---------------------------------------------------------------- walkers ----
        const int row = lane >> 4;
        const int px = tx * FNX_TILE_X + blend_pixel_x(w, lane), py = ty * FNX_TILE_Y + blend_pixel_y(w, lane);
        const bool inside = px < W && py < H;
        const uint32_t pix_id = (uint32_t)W * py + px;
        const float pxf = (float)px, pyf = (float)py;
        float alive = inside ? 1.0f : 0.0f;
        float Tr = 1.0f;
        uint32_t dyn_before = 0;
        float acc[C];
#pragma unroll
        for (int ch = 0; ch < C; ch++) acc[ch] = 0.f;
        float Dm = 15.0f;
        DualPixel d1;  // DUAL: the second image's pixel
        d1.acc = 0.f;
        d1.Tr = 1.0f;
        d1.alive = inside ? 1.0f : 0.0f;
        d1.Dm = 15.0f;
        d1.hit_off = 0xFFFFFFFFu;
        uint32_t last_contributor1 = 0;
        float4 *bstate = reinterpret_cast<float4 *>(reinterpret_cast<char *>(point_list) + vb.bin_bstate) +
                         (size_t)(r0 >> 8) * 256 + sid;
        float2 *bstate1 = DUAL ? reinterpret_cast<float2 *>(reinterpret_cast<char *>(point_list) + du.bin_bstate1) +
                                     (size_t)(r0 >> 8) * 256 + sid : nullptr;
        uint32_t b = 0;
        for (uint32_t base = b_lo; base < b_hi; base += 256, b++) {
            const int p = (int)(b % (uint32_t)kBuf);
            const bool wave_done = __all(alive == 0.0f && (!DUAL || d1.alive == 0.0f));
            FNX_WCLK(0)
            bool ended = false;
            for (;;) {  // the batch, or the word that there will be none
                if (lds_load(&s_staged) > b) break;
                if (lds_load(&s_nbatches) <= b) {
                    ended = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            asm volatile("" ::: "memory");
            if (ended) break;
            FNX_WCLK(1)
            if (base != r0)  // hand-over record of the backward pass: the pixel's state in front of this batch
                bstate[(size_t)(b - 1u) * 256] = make_float4(Tr, acc[0], acc[C > 1 ? 1 : 0], acc[C > 2 ? 2 : 0]);
            if (DUAL && base != r0) bstate1[(size_t)(b - 1u) * 256] = make_float2(d1.Tr, d1.acc);
            if (!wave_done) {
                n_blended = b + 1u;
                const uint4 ln = *reinterpret_cast<const uint4 *>(&s_len[p][4 * w]);
                const uint32_t n_w = max(max(ln.x, ln.y), max(ln.z, ln.w));
                const uint16_t *mylist = s_list[p][4 * w + row];
                const uint32_t pos0 = base - r0 + 1;
                uint32_t hit_off = 0xFFFFFFFFu;
                d1.hit_off = 0xFFFFFFFFu;
                if (FAST) {
                    fast_walk<C, DUAL>(mylist, n_w, s_ra[p], s_rb[p], s_rc[p], pxf, pyf, acc, Tr, alive, Dm, hit_off, &d1);
                } else
                for (uint32_t i0 = 0; i0 < n_w; i0 += kGroup) {
                    if (__all(alive == 0.0f && (!DUAL || d1.alive == 0.0f))) break;
                    uint32_t jw[kGroup / 2];
#pragma unroll
                    for (int k = 0; k < kGroup / 2; k++) jw[k] = reinterpret_cast<const uint32_t *>(mylist + i0)[k];
                    float a_h[kGroup];
                    float4 rc[kGroup];
#pragma unroll
                    for (int k = 0; k < kGroup; k++) {
                        const uint32_t off = (jw[k >> 1] >> (16 * (k & 1))) & (DUAL ? kListOffMask : 0xFFFFu);
                        const float4 ra = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s_ra[p]) + off);
                        const float4 rb = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s_rb[p]) + off);
                        rc[k] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s_rc[p]) + off);
                        const float dx = ra.x - pxf, dy = ra.y - pyf;
                        const float power = -0.5f * (ra.z * dx * dx + rb.x * dy * dy) - ra.w * dx * dy;
                        const float alpha = fminf(0.99f, rb.y * exp_fixed_in_range(fmaxf(power, -87.0f)));
                        a_h[k] = (!(power > 0.0f) && !(alpha < 1.0f / 255.0f)) ? alpha : 0.0f;
                    }
                    if (DUAL) {  // the second image: the same entries, static ones with alpha 0, its own T and stop rule
#pragma unroll
                        for (int k = 0; k < kGroup; k++) {
                            const uint32_t raw = (jw[k >> 1] >> (16 * (k & 1))) & 0xFFFFu, off = raw & kListOffMask;
                            const float ae = ((raw & kListStatic) ? 0.0f : a_h[k]) * d1.alive;
                            const float test_T = d1.Tr * (1 - ae);
                            const bool stop = test_T < 0.0001f;
                            const float a_eff = stop ? 0.0f : ae;
                            d1.acc = d1.acc + rc[k].x * a_eff * d1.Tr;
                            d1.Dm = (d1.Tr > 0.5f && test_T < 0.5f) ? rc[k].w : d1.Dm;
                            d1.Tr = stop ? d1.Tr : test_T;
                            d1.hit_off = (a_eff > 0.0f) ? off : d1.hit_off;
                            d1.alive = stop ? 0.0f : d1.alive;
                        }
                    }
#pragma unroll
                    for (int k = 0; k < kGroup; k++) {
                        const uint32_t off = (jw[k >> 1] >> (16 * (k & 1))) & (DUAL ? kListOffMask : 0xFFFFu);
                        const float ae = a_h[k] * alive;
                        const float test_T = Tr * (1 - ae);
                        const bool stop = test_T < 0.0001f;
                        const float a_eff = stop ? 0.0f : ae;
                        acc[0] = acc[0] + rc[k].x * a_eff * Tr;
                        if (C > 1) acc[C > 1 ? 1 : 0] = acc[C > 1 ? 1 : 0] + rc[k].y * a_eff * Tr;
                        if (C > 2) acc[C > 2 ? 2 : 0] = acc[C > 2 ? 2 : 0] + rc[k].z * a_eff * Tr;
                        Dm = (Tr > 0.5f && test_T < 0.5f) ? rc[k].w : Dm;
                        Tr = stop ? Tr : test_T;
                        hit_off = (a_eff > 0.0f) ? off : hit_off;
                        alive = stop ? 0.0f : alive;
                    }
                }
                {
                    auto dyn_at_or_below = [&](uint32_t upto) -> uint32_t {
                        int ws_ = (int)(upto >> 6);
                        unsigned long long m = s_dynmask[p][ws_] & ((2ull << (upto & 63u)) - 1ull);
                        while (m == 0ull && ws_ > 0) m = s_dynmask[p][--ws_];
                        return m ? (uint32_t)(64 * ws_ + 63 - __clzll((long long)m)) : 0xFFFFFFFFu;
                    };
                    if (hit_off != 0xFFFFFFFFu) {
                        last_contributor = pos0 + (hit_off >> 4);
                        const uint32_t d = dyn_at_or_below(hit_off >> 4);
                        last_dyn = d != 0xFFFFFFFFu ? pos0 + d : dyn_before;
                    }
                    const uint32_t d_all = dyn_at_or_below(255u);
                    if (d_all != 0xFFFFFFFFu) dyn_before = pos0 + d_all;
                    if (DUAL && d1.hit_off != 0xFFFFFFFFu) last_contributor1 = pos0 + (d1.hit_off >> 4);
                }
            }
            FNX_WCLK(2)
#ifdef FNX_EXP_CLOCK
            if (wg_rank == 0 && wg_view == 0 && lane == 0) g_ws_clock[16 * w8 + 9] += 1;
#endif
            // this wave has left buffer p; which of its blocks go on
            const unsigned long long lv = __ballot(alive != 0.0f || (DUAL && d1.alive != 0.0f));
            const uint32_t live4 = ((lv & 0xFFFFull) ? 1u : 0u) | ((lv & 0xFFFF0000ull) ? 2u : 0u) |
                                   ((lv & 0xFFFF00000000ull) ? 4u : 0u) | ((lv & 0xFFFF000000000000ull) ? 8u : 0u);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) {
                lds_store(&s_live[w], live4 | (lv ? 0u : 16u));
                lds_store(&s_walked[w], b + 1u);
            }
        }
        if (lane == 0) {  // (a walker that has left never holds a buffer)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            lds_store(&s_live[w], 16u);
            lds_store(&s_walked[w], 0xFFFFFFFFu);
        }
        if (inside) {
            final_T[pix_id] = Tr;
            n_contrib[pix_id] = last_contributor;
            n_contrib[(size_t)W * H + pix_id] = last_dyn;
#pragma unroll
            for (int ch = 0; ch < C; ch++) {
                out_color[(size_t)ch * H * W + pix_id] = acc[ch] + Tr * bg[ch];
                acc_final[(size_t)ch * H * W + pix_id] = acc[ch];
            }
            out_depth[pix_id] = Dm;
            if (DUAL) {  // the second image: every contributor of its is a dynamic entry, so both halves of n_contrib agree
                reinterpret_cast<float *>(img1 + du.final_T)[pix_id] = d1.Tr;
                uint32_t *nc1 = reinterpret_cast<uint32_t *>(img1 + du.n_contrib);
                nc1[pix_id] = last_contributor1;
                nc1[(size_t)W * H + pix_id] = last_contributor1;
                reinterpret_cast<float *>(img1 + du.acc_final)[pix_id] = d1.acc;
                du.out_color1[(size_t)wg_view * H * W + pix_id] = d1.acc + d1.Tr * du.bg1[0];
                du.out_depth1[(size_t)wg_view * H * W + pix_id] = d1.Dm;
            }
        }
        uint32_t m = DUAL ? max(last_contributor, last_contributor1) : last_contributor;
        uint32_t md = DUAL ? max(last_dyn, last_contributor1) : last_dyn;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            m = max(m, (uint32_t)__shfl_xor((int)m, off));
            md = max(md, (uint32_t)__shfl_xor((int)md, off));
        }
        if (lane == 0) {
            s_qmax[w] = m;
            s_qdyn[w] = md;
            s_qcnt[w] = n_blended;
        }
    }
    __syncthreads();
    const uint32_t qmax = max(max(s_qmax[0], s_qmax[1]), max(s_qmax[2], s_qmax[3]));
    const uint32_t nb = (max(max(s_qdyn[0], s_qdyn[1]), max(s_qdyn[2], s_qdyn[3])) + 255u) >> 8;
    if (nb) {
        if (tid == 0) s_adv = atomicAdd(&header[HDR_BWD_ITEMS], nb);
        __syncthreads();
        uint32_t *items = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(point_list) + vb.bin_items) + s_adv;
        for (uint32_t k = tid; k < nb; k += 512) items[k] = (uint32_t)tile | (k << kItemTileBits);
    }
    if (depth_hint && tid == 0) depth_hint[tile] = qmax;
    if (tid == 0) {
        // list entries of the batches the tile blended (the per-tile kernel's count: batches some pixel was alive in front of)
        const uint32_t nbl = max(max(s_qcnt[0], s_qcnt[1]), max(s_qcnt[2], s_qcnt[3]));
        const uint32_t staged = min(256u * nbl, b_hi - b_lo);
        if (staged) atomicAdd(&header[HDR_FWD_ENTRIES], staged);
    }
}
