// LAB switches of the rasteriser's translation units, in one place.  The production build (fluidnexus_amd/build.py)
// defines none of them: every switch below then has the value given here and the instrumented code is compiled out.
// tools/build_variant.py builds a variant library with -D flags for tools/lab_*.sh / tools/coh_ab.sh; a variant is a
// MEASURING instrument, several of them produce wrong results on purpose (noted per switch).
#pragma once

// raster_binning.hip -- temporal-coherence sort (sort_repair_kernel); bit mask, results WRONG, verification off:
//   2 no rectangle staging / per-block counts, 4 no window sort
#ifndef FNX_EXP_COH
#define FNX_EXP_COH 0
#endif
// raster_binning.hip -- emit_kernel; results WRONG except 0: 10 loads only, 11 no bitmask / output, 12 no store,
//   13 lane-contiguous store
#ifndef FNX_EXP_EMIT
#define FNX_EXP_EMIT 0
#endif
// raster_backward.hip -- blend_backward_kernel; results WRONG except 0: 1 no global flush, 2 no cross-lane fold,
//   3 staging only
#ifndef FNX_ABLATE
#define FNX_ABLATE 0
#endif
// raster_forward.hip -- barrier probe; results WRONG except 0: 1 = the staging and list barriers of the blend forward's batch
//   loop become wave-local waits (an upper bound for what a barrier-free batch structure could buy; round 5, config 3:
//   296.6 -> 295.5 us, i.e. nothing -- the forward does not wait at its barriers).
//   The backward's counterpart lives in lab/bwd_async_flush.patch (tools/build_variant.py --patch): bits 2 / 4 / 8 = its
//   barriers A / B / C.  Its numbers (8: 316 -> 254 us, 14: 229 us) are NOT a bound: without barrier C the waves of a
//   workgroup read different tickets and stop working on the same item.  The legitimate forms the same patch holds --
//   the flush behind the next item's barrier A, and the double-buffered flush that early waves take in chunks while the
//   slowest quadrant still walks -- measured 318.8 and 319.9 us against 316.4 (parity-green): the backward does not wait
//   at its barriers either.
#ifndef FNX_EXP_NOBAR
#define FNX_EXP_NOBAR 0
#endif
// Defined-or-not switches (all off in production):
//   FNX_EXP_CLOCK      per-phase / per-workgroup clocks of emit_kernel and blend_forward_kernel (tools/kernel_lab.py,
//                      tools/deep_probe.py read them through fnx_debug_* exports)
//   FNX_EXP_BCLK       per-phase clocks of blend_backward_kernel (tools/bwd_phases.py)
//   FNX_EXP_STATS      lane / row occupancy statistics of the blend forward's inner loop (tools/deep_probe.py)
//   FNX_EXP_WG_ATOMICS blend backward flush with workgroup-scope atomics -- WRONG sums across XCDs (timing only)
//   FNX_EXP_COLDREC    one more cold 16-byte gather per entry in the blend backward (timing only)
// Tuning constants with production defaults next to their use (not experiments): FNX_FWD_WAVES, FNX_FWD_GROUP,
// FNX_BWD_WAVES, FNX_BWD_GROUP, FNX_EMIT_THREADS, FNX_EMIT_CHUNK,
// FNX_EMIT_MASK_WORDS, FNX_EMIT_BANDS, FNX_EMIT_BAND_TARGET, FNX_DEEP_GROUPS, FNX_DEEP_PRIO, FNX_COH_THREADS.
