// LAB (opt-in, off in every measured configuration: fnx_raster_opts_t.deep_kernel): the super-batch blend forward for deep
// tiles.  Textually included by raster_forward.hip inside namespace fnx, behind the per-tile kernel it shares its inner
// loops with.  DESIGN.md 4.6 has the measurements (a one-view forward 269 -> 199 us; the iteration does not follow).
// K5b: the blend forward of DEEP tiles (FAST arithmetic only).  A tile whose list does not saturate -- a semi-transparent
// plume in front of a distant wall: thousands of contributing entries per pixel -- is a serial walk in K5, and the
// launch ends when the deepest one ends (~10 us per 256-entry batch, 30 batches).  Here a workgroup of G x 256 threads
// takes a SUPER-BATCH of G x 256 entries at a time: group g = threads [256 g, 256 g + 256) owns sub-batch g, every
// group covers all 256 pixels of the tile (same pixel <-> lane map as K5).  Per super-batch:
//   stage     all G x 256 entries at once (merge of the two streams, records, block masks, lists) -- the latency-bound
//             part of a batch is paid once per super-batch instead of once per batch;
//   pass A    groups 0 .. G-2 walk their sub-batch for the transmittance alone (fast_walk_transmittance: alpha and a
//             running product, half the instructions of the full walk) -> P_g per pixel;
//   T_in      sub-batch g starts from T_start * P_0 ... P_{g-1}; T only falls, so the pixel has stopped in front of g
//             exactly if that running product fell below 1e-4 (forward.cu:336-340), up to the rounding of the product;
//   pass B    every group walks its sub-batch in full from its own T_in (fast_walk, the same code as K5): colour,
//             stop test, median depth, last contributor -- all four sub-batches at the same time;
//   combine   group 0 (which owns the pixels' state) adds the groups' results in list order and writes the per-batch
//             hand-over records of the backward pass (bstate) exactly as K5 does.
// The deep tile's chain shrinks from G x (stage + walk) to stage + A + B per G batches; the price is pass A, ~1/3 more
// work on the deep tiles.  Results differ from K5's by the association of T_in only (stated tolerance of the fast mode).
#ifndef FNX_DEEP_GROUPS
#define FNX_DEEP_GROUPS 4
#endif
template <int C, bool SPLIT>
__global__ void __launch_bounds__(256 * FNX_DEEP_GROUPS) __attribute__((amdgpu_waves_per_eu(FNX_DEEP_GROUPS, FNX_DEEP_GROUPS)))
blend_forward_deep_kernel(int T, int gx, const uint32_t *__restrict__ ranges_all, uint32_t *__restrict__ point_list_all,
                          int W, int H, const float4 *__restrict__ blend_rec_all, const float *__restrict__ bg,
                          float *__restrict__ final_T_all, uint32_t *__restrict__ n_contrib_all,
                          float *__restrict__ out_color_all, float *__restrict__ out_depth_all,
                          uint32_t *__restrict__ header_all, uint32_t capacity, const uint32_t *__restrict__ tile_count_all,
                          const uint32_t *__restrict__ dyn_start_all, float *__restrict__ acc_final_all,
                          const uint32_t *__restrict__ tile_order_all, uint32_t *__restrict__ depth_hint_all,
                          const StaticRef st, int materialize_all, const ViewBatch vb, int n_views) {
    constexpr int G = FNX_DEEP_GROUPS, NB = 256 * G, kGroup = 4;
    constexpr int kListStride = (256 + kGroup + 7) & ~7;
    constexpr uint32_t kNullOff = (uint32_t)NB * 16u;  // LDS offset of the NULL record (slot NB)
    static_assert(NB * 16 <= 0xFFFF, "list entries are 16-bit LDS offsets");
    __shared__ float4 s_ra[NB + 1];
    __shared__ float4 s_rb[NB + 1];
    __shared__ float4 s_rc[NB + 1];
    __shared__ __attribute__((aligned(16))) uint16_t s_list[16 * G][kListStride];  // lists 16 g + 4 w .. + 3: wave (g, w)
    __shared__ uint16_t s_mask[NB];
    __shared__ uint32_t s_wk[SPLIT ? 2 : 1][SPLIT ? NB : 1];
    __shared__ uint32_t s_wi[SPLIT ? 2 : 1][SPLIT ? NB : 1];
    __shared__ uint32_t s_adv;
    __shared__ uint32_t s_done[4], s_qmax[4], s_cnt[kMaxViews];
    __shared__ float s_pT[256];      // working transmittance of every pixel at the start of the super-batch (0: stopped)
    __shared__ float s_P[G][256];    // pass A: transmittance of sub-batch g per pixel
    __shared__ float4 s_pa[G][256];  // pass B of groups >= 1: colour taken in the sub-batch, pixel T behind it
    __shared__ float4 s_pb[G][256];  //                         working T behind it, last contributor, median depth, its flag
    const int tid = threadIdx.x, g = tid >> 8, gt = tid & 255, lane = tid & 63, w = gt >> 6, row = lane >> 4;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    if (tid < n_views) {
        const uint32_t *h = view_at(header_all, vb.img, tid);
        s_cnt[tid] = (h[HDR_NUM_RENDERED] > capacity || h[HDR_STATUS] != 0u) ? 0u : (h[HDR_DEEP_COUNT] & 0xFFFFFFu);
    }
    if (tid == 0) {
        s_ra[NB] = make_float4(0.f, 0.f, 0.f, 0.f);
        s_rb[NB] = make_float4(0.f, -200.0f, 0.f, 0.f);
        s_rc[NB] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    uint32_t most = 0;
    for (int v = 0; v < n_views; v++) most = max(most, s_cnt[v]);
    // deep tiles of all views, deepest first within a view (tile_scan_kernel), the views interleaved
    for (uint32_t slot = blockIdx.x; slot < most * (uint32_t)n_views; slot += gridDim.x) {
        __syncthreads();  // the previous tile is done with the LDS arrays
        const int vw = (int)(slot % (uint32_t)n_views);
        const uint32_t rank = slot / (uint32_t)n_views;
        if (rank >= s_cnt[vw]) continue;
        const uint32_t *ranges = view_at(ranges_all, vb.img, vw);
        float *final_T = view_at(final_T_all, vb.img, vw);
        uint32_t *n_contrib = view_at(n_contrib_all, vb.img, vw);
        uint32_t *header = view_at(header_all, vb.img, vw);
        uint32_t *point_list = view_at(point_list_all, vb.bin, vw);
        const float4 *blend_rec = view_at(blend_rec_all, vb.geom, vw);
        float *out_color = out_color_all + (size_t)vw * C * H * W;
        float *out_depth = out_depth_all + (size_t)vw * H * W;
        float *acc_final = view_at(acc_final_all, vb.img, vw);
        const int tile = (int)view_at(tile_order_all, vb.img, vw)[rank];
        const int tx = tile % gx, ty = tile / gx;
        const int px = tx * FNX_TILE_X + blend_pixel_x(w, lane), py = ty * FNX_TILE_Y + blend_pixel_y(w, lane);
        const bool inside = px < W && py < H;
        const uint32_t pix_id = (uint32_t)W * py + px;
        const float pxf = (float)px, pyf = (float)py;
        const float tile_x0 = (float)(tx * FNX_TILE_X), tile_y0 = (float)(ty * FNX_TILE_Y);
        const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        // pixel state: meaningful in group 0 (the other groups only ever hold the state of their own sub-batch)
        float alive = inside ? 1.0f : 0.0f, Tr = 1.0f, Dm = 15.0f;
        uint32_t last_contributor = 0;
        float acc[C];
#pragma unroll
        for (int ch = 0; ch < C; ch++) acc[ch] = 0.f;
        float4 pa = make_float4(0.f, 0.f, 0.f, 0.f), pb = pa, pc = pa;
        float pd = 0.f;
        uint32_t id_ahead = 0;
        const uint2 *sp = nullptr, *fp = nullptr;
        const float4 *rec_s = nullptr;
        uint32_t ns = 0, nf = 0, si = 0, fj = 0, my_id = 0;
        uint2 ws = make_uint2(0u, 0u), wf = ws;
        auto record_of = [&](uint32_t id) -> const float4 * {
            return (SPLIT && id >= st.id0) ? rec_s + 4 * (size_t)(id - st.id0) : blend_rec + 4 * (size_t)id;
        };
        auto load_windows = [&]() {
            ws = (si + (uint32_t)tid < ns) ? sp[si + tid] : make_uint2(0xFFFFFFFFu, 0u);
            wf = (fj + (uint32_t)tid < nf) ? fp[fj + tid] : make_uint2(0xFFFFFFFFu, 0u);
        };
        auto store_windows = [&]() {
            s_wk[0][SPLIT ? tid : 0] = ws.x;
            s_wi[0][SPLIT ? tid : 0] = ws.y;
            s_wk[SPLIT ? 1 : 0][SPLIT ? tid : 0] = wf.x;
            s_wi[SPLIT ? 1 : 0][SPLIT ? tid : 0] = wf.y;
        };
        auto merge_batch = [&](uint32_t cnt_next) -> uint32_t {  // merge path over the two windows (K5, NB entries wide)
            uint32_t id = 0;
            if ((uint32_t)tid < cnt_next) {
                const uint32_t nsw = min((uint32_t)NB, ns - si), nfw = min((uint32_t)NB, nf - fj);
                const uint32_t *ks = s_wk[0], *kf = s_wk[SPLIT ? 1 : 0];
                const uint32_t t = (uint32_t)tid;
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
        auto fetch = [&](uint32_t id) {
            const float4 *rec = record_of(id);
            pa = rec[0];
            pb = rec[1];
            pc = rec[2];
            if (C > 2) pd = rec[3].x;
        };
        if (SPLIT) {
            const char *static_blob = st.base + st.stride * vw;
            const uint32_t *starts = reinterpret_cast<const uint32_t *>(static_blob + st.starts);
            const uint32_t s0 = starts[tile];
            ns = starts[tile + 1] - s0;
            sp = reinterpret_cast<const uint2 *>(static_blob + st.pairs) + s0;
            rec_s = reinterpret_cast<const float4 *>(static_blob + st.rec);
            nf = view_at(tile_count_all, vb.img, vw)[tile];
            fp = reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(point_list) + vb.bin_pairs) +
                 view_at(dyn_start_all, vb.img, vw)[tile];
            load_windows();
            store_windows();
            __syncthreads();
            const uint32_t cnt0 = min((uint32_t)NB, r1 - r0);
            my_id = merge_batch(cnt0);
            __syncthreads();
            if (cnt0) {
                const uint32_t a = s_adv;
                si += a;
                fj += cnt0 - a;
            }
            if ((uint32_t)tid < cnt0) fetch(my_id);
            load_windows();
        } else {
            if (r0 + (uint32_t)tid < r1) fetch(point_list[r0 + tid]);
            if (r0 + (uint32_t)NB + (uint32_t)tid < r1) id_ahead = point_list[r0 + NB + tid];
        }
        float4 *bstate = reinterpret_cast<float4 *>(reinterpret_cast<char *>(point_list) + vb.bin_bstate) +
                         (size_t)(r0 >> 8) * 256 + gt;
        uint16_t *masks_out = reinterpret_cast<uint16_t *>(reinterpret_cast<char *>(point_list) + vb.bin_masks);
        bool blending = true;
        for (uint32_t base = r0; base < r1; base += NB) {
            // group 0 publishes the pixels' working T; "has every pixel stopped?" through LDS as in K5
            const uint32_t wave_done = __all(alive == 0.0f) ? 1u : 0u;
            if (g == 0) {
                if (lane == 0) s_done[w] = wave_done;
                s_pT[gt] = alive;
            }
            FNX_LOOP_BARRIER();
            const bool all_done = (s_done[0] & s_done[1] & s_done[2] & s_done[3]) != 0u;
            if (all_done) {
                if (!SPLIT || !materialize_all) break;
                blending = false;
            }
            // hand-over record in front of the super-batch's first sub-batch (the others: combine, below)
            if (g == 0 && blending && base != r0)
                bstate[(size_t)(((base - r0) >> 8) - 1) * 256] = make_float4(Tr, acc[0], acc[C > 1 ? 1 : 0], acc[C > 2 ? 2 : 0]);
            const uint32_t cnt = min((uint32_t)NB, r1 - base);
            uint32_t qm = 0;
            if ((uint32_t)tid < cnt && blending) {
                qm = block_mask_exact(pa.x, pa.y, pa.z, pa.w, pb.x, pb.z, pc.x, pc.y, tile_x0, tile_y0);
                constexpr float kL2e = 1.44269504088896341f;
                s_ra[tid] = make_float4(pa.x, pa.y, (-0.5f * kL2e) * pa.z, (-kL2e) * pa.w);
                const float lo = __builtin_amdgcn_logf(fmaxf(pb.y, 0.0f));
                if (C == 3) {
                    s_rb[tid] = make_float4((-0.5f * kL2e) * pb.x, lo, pc.z, pc.w);
                    s_rc[tid] = make_float4(pd, pb.w, 0.f, 0.f);
                } else {
                    s_rb[tid] = make_float4((-0.5f * kL2e) * pb.x, lo, pc.z, pb.w);
                }
            }
            if (SPLIT) {
                if ((uint32_t)tid < cnt) point_list[base + tid] = my_id;
                store_windows();
            } else {
                if (base + (uint32_t)NB + (uint32_t)tid < r1) fetch(id_ahead);
                if (base + 2u * NB + (uint32_t)tid < r1) id_ahead = point_list[base + 2u * NB + tid];
            }
            s_mask[tid] = (uint16_t)qm;
            if ((uint32_t)tid < cnt && blending) masks_out[base + tid] = (uint16_t)qm;
            {  // this wave's four lists start out as NULL pointers from end to end
                const uint32_t n2 = kNullOff | (kNullOff << 16);
                const uint4 nul = make_uint4(n2, n2, n2, n2);
                uint4 *mine = reinterpret_cast<uint4 *>(&s_list[16 * g + 4 * w][0]);
                for (int i = lane; i < 4 * kListStride / 8; i += 64) mine[i] = nul;
            }
            FNX_LOOP_BARRIER();
            uint32_t len[4] = {0u, 0u, 0u, 0u};
            const unsigned long long live = __ballot(s_pT[gt] != 0.0f);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int slot_e = 256 * g + 64 * k + lane;
                const uint32_t mk = (uint32_t)s_mask[slot_e] >> (4 * w);
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const bool bit = ((mk >> b) & 1u) && ((live >> (16 * b)) & 0xFFFFull) != 0ull;
                    const unsigned long long m = __ballot(bit);
                    if (bit) s_list[16 * g + 4 * w + b][len[b] + (uint32_t)__popcll(m & lt_mask)] = (uint16_t)(slot_e * 16);
                    len[b] += (uint32_t)__popcll(m);
                }
            }
            uint32_t next_cnt = 0, next_id = 0;
            if (SPLIT) {
                next_cnt = base + (uint32_t)NB < r1 ? min((uint32_t)NB, r1 - base - (uint32_t)NB) : 0u;
                next_id = merge_batch(next_cnt);
            }
            FNX_LOOP_BARRIER();
            if (SPLIT) {
                if (next_cnt) {
                    const uint32_t a = s_adv;
                    si += a;
                    fj += next_cnt - a;
                }
                my_id = next_id;
                if ((uint32_t)tid < next_cnt) fetch(my_id);
                load_windows();
                if (!blending) continue;
            }
            const uint32_t n_w = max(max(len[0], len[1]), max(len[2], len[3]));
            const uint16_t *mylist = s_list[16 * g + 4 * w + row];
            // pass A: transmittance of the sub-batches that have a successor in this super-batch
            if (g < G - 1) s_P[g][gt] = fast_walk_transmittance<C>(mylist, n_w, s_ra, s_rb, pxf, pyf);
            FNX_LOOP_BARRIER();
            // T in front of sub-batch g: the running product, with the stop rule applied between sub-batches
            float T_in = s_pT[gt];
#pragma unroll
            for (int h = 0; h < G - 1; h++)
                if (h < g) {
                    const float tn = T_in * s_P[h][gt];
                    T_in = tn < 0.0001f ? 0.0f : tn;
                }
            // pass B: the full walk of every sub-batch, from its own T_in (group 0: from the pixel state itself)
            uint32_t hit_off = 0xFFFFFFFFu;
            {
                // one copy of the walk for all groups: group 0 runs it on the pixel state, the others on a fresh state
                float acc_g[C], Tr_g = g == 0 ? Tr : T_in, alive_g = g == 0 ? alive : T_in, Dm_g = g == 0 ? Dm : 0.f;
#pragma unroll
                for (int ch = 0; ch < C; ch++) acc_g[ch] = g == 0 ? acc[ch] : 0.f;
                const float T_before = Tr_g;
                fast_walk<C>(mylist, n_w, s_ra, s_rb, s_rc, pxf, pyf, acc_g, Tr_g, alive_g, Dm_g, hit_off);
                const uint32_t hit = hit_off != 0xFFFFFFFFu ? (base - r0) + 1u + (hit_off >> 4) : 0u;
                if (g == 0) {
#pragma unroll
                    for (int ch = 0; ch < C; ch++) acc[ch] = acc_g[ch];
                    Tr = Tr_g;
                    alive = alive_g;
                    Dm = Dm_g;
                    last_contributor = hit ? hit : last_contributor;
                } else {
                    const bool crossed = T_before >= 0.5f && Tr_g < 0.5f;
                    s_pa[g][gt] = make_float4(acc_g[0], acc_g[C > 1 ? 1 : 0], acc_g[C > 2 ? 2 : 0], Tr_g);
                    s_pb[g][gt] = make_float4(alive_g, __uint_as_float(hit), Dm_g, crossed ? 1.0f : 0.0f);
                }
            }
            FNX_LOOP_BARRIER();
            // combine in list order: group 0 owns the pixel state and writes the backward pass's hand-over records
            if (g == 0) {
#pragma unroll
                for (int h = 1; h < G; h++) {
                    if (base + 256u * h >= r1) break;  // no such sub-batch
                    bstate[(size_t)(((base - r0) >> 8) + h - 1) * 256] =
                        make_float4(Tr, acc[0], acc[C > 1 ? 1 : 0], acc[C > 2 ? 2 : 0]);
                    const float4 ra_ = s_pa[h][gt], rb_ = s_pb[h][gt];
                    if (alive != 0.0f) {  // a pixel that stopped in front of sub-batch h takes nothing from it
                        acc[0] += ra_.x;
                        if (C > 1) acc[C > 1 ? 1 : 0] += ra_.y;
                        if (C > 2) acc[C > 2 ? 2 : 0] += ra_.z;
                        Tr = rb_.x != 0.0f || ra_.w != 0.0f ? ra_.w : Tr;
                        alive = rb_.x;
                        const uint32_t hit = __float_as_uint(rb_.y);
                        last_contributor = hit ? hit : last_contributor;
                        Dm = rb_.w != 0.0f ? rb_.z : Dm;
                    }
                }
            }
        }
        if (g == 0) {
            if (inside) {
                final_T[pix_id] = Tr;
                n_contrib[pix_id] = last_contributor;
                n_contrib[(size_t)W * H + pix_id] = last_contributor;  // (this kernel does not track the dynamic limit: the conservative value)
#pragma unroll
                for (int ch = 0; ch < C; ch++) {
                    out_color[(size_t)ch * H * W + pix_id] = acc[ch] + Tr * bg[ch];
                    acc_final[(size_t)ch * H * W + pix_id] = acc[ch];
                }
                out_depth[pix_id] = Dm;
            }
            uint32_t m = last_contributor;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off));
            if (lane == 0) s_qmax[w] = m;
        }
        __syncthreads();
        const uint32_t qmax = max(max(s_qmax[0], s_qmax[1]), max(s_qmax[2], s_qmax[3]));
        const uint32_t nb = (qmax + 255u) >> 8;
        if (nb) {
            if (tid == 0) s_adv = atomicAdd(&header[HDR_BWD_ITEMS], nb);
            __syncthreads();
            uint32_t *items = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(point_list) + vb.bin_items) + s_adv;
            for (uint32_t k = tid; k < nb; k += NB) items[k] = (uint32_t)tile | (k << kItemTileBits);
        }
        if (depth_hint_all && tid == 0) (depth_hint_all + (size_t)vw * T)[tile] = qmax;
    }
}
