"""Autograd boundary of the MI355X-native Gaussian rasteriser.

Mirrors, name for name, the Python interface of the reference's two rasteriser packages
  FluidDynamics/submodules/gaussian_rasterization_ch3/diff_gaussian_rasterization_ch3/__init__.py
  (rasterize_gaussians :9-30, _RasterizeGaussians :33-140, GaussianRasterizationSettings :143-154,
  GaussianRasterizer :157-215) and its ch1 twin (NUM_CHANNELS = 1),
on top of the C ABI in include/fnx_raster.h (loaded with ctypes; no torch C++ glue).

Differences from the reference that callers can observe:
  * one implementation parameterised by `channels`; the ch3 / ch1 packages are thin aliases;
  * work is enqueued on torch's current HIP stream (reference: legacy default stream);
  * `set_host_sync(False)` removes the per-forward device->host read of num_rendered
    (rasterizer_impl.cu:264) by sizing the binning scratch from a running high-water mark; an
    overflow is reported by `check_status()` / the next forward.
There is no CPU path: tensors must live on a HIP device and the HIP library must be built.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _lib

_HOST_SYNC = True          # True = reference behaviour (exact-size binning buffer, one sync per forward)
_CAP_SLACK = 1.3           # head-room over the high-water mark when host sync is off
_capacity_hwm: dict = {}   # (device, W, H, channels) -> capacity in instances
_pending_status: list = []  # (ring slot, key) of sync-free forwards whose status has not been read yet
_captured_status: list = []  # (device, slot, key) of forwards recorded into a hipGraph: re-read on every check_status()
_status_ring: dict = {}     # device index -> persistent int32[_RING + _CAPTURED, 8] copy of each forward's header words
_RING = 256                 # rotating slots of eager forwards
_CAPTURED = 8192            # slots owned by captured forwards (rows _RING ... of the same tensor), never recycled
_ring_next = 0
_captured_next = 0
last_num_rendered = -1      # updated by check_status(): instance count of the most recent forward
max_sort_span_bits = 0      # updated by check_status(): largest bit length of a view's depth-key span seen so far


_DEEP_VARIANT = True  # depth hints on: tiles whose lists went deep in the previous forward are scheduled first


def set_deep_variant(enabled: bool, min_depth: int | None = None):
    """Switch the depth-hint mechanism (ViewBatch.depth_hint) on / off; `min_depth`: list depth from which a tile
    counts as deep (library-wide, default 1024)."""
    global _DEEP_VARIANT
    _DEEP_VARIANT = bool(enabled)
    if min_depth is not None:
        _lib.check(_lib.raster().fnx_set_deep_threshold(int(min_depth)))


# Module-level DEFAULTS of the per-call options (include/fnx_raster.h fnx_raster_opts_t).  The view-batched path hands
# every call its own options struct -- these values overlaid with the rasteriser instance's `options` dict -- and the
# backward of a render runs with the options its forward ran with (saved in the autograd context).  The setters also
# forward to the library's deprecated process-wide setters, which only the single-view entry points still read.
SORT_NARROW_MAX_BITS = 25   # widest depth-key span (bits) for which callers switch to the three-pass sort: two below the
                            # 27 bits three 9-bit passes order
_OPTS = dict(blend_math=0, lean_geometry=0, sort_narrow=0, deep_kernel=5, coherent_sort=0, sort_key=None,
             segments=1 if os.environ.get("FNX_SEG_FORWARD", "0") == "1" else 0)


def set_blend_math(mode: str):
    """Arithmetic of the blend kernels: "exact" = the bit-reproducible sequence the oracle repeats (default), "fast" =
    fused multiply-adds + v_exp_f32, stated tolerance (include/fnx_raster.h fnx_set_blend_math)."""
    _OPTS["blend_math"] = {"exact": 0, "fast": 1}[mode]
    _lib.check(_lib.raster().fnx_set_blend_math(_OPTS["blend_math"]))


def set_sort_narrow(enabled: bool):
    """True: the fourth pass of the depth sort is not launched (FNX_SORT_NARROW).  Only after check_status() has shown
    `max_sort_span_bits` <= SORT_NARROW_MAX_BITS on the scene at hand; a view that needs the pass after all raises
    FNX_ERR_SORT_SPAN at the next check_status()."""
    _OPTS["sort_narrow"] = 1 if enabled else 0
    _lib.check(_lib.raster().fnx_set_sort_narrow(_OPTS["sort_narrow"]))


def set_coherent_sort(enabled: bool):
    """View batches: after the first call of a (camera batch, channel count, splat count) the depth order of the
    PREVIOUS call is repaired in one launch instead of running the radix passes (FNX_SORT_COHERENT; the state lives on
    the ViewBatch).  Exact by construction: the repaired order is verified on the device and a view that fails is
    sorted from scratch inside the same launch (`ViewBatch.sort_counters()` reports how often).  True / 1: where it pays
    (coherent_sort_pays); 2: always."""
    _OPTS["coherent_sort"] = int(enabled) if enabled in (0, 1, 2) else (1 if enabled else 0)


_KEEP_LAST = False   # keep_last_blobs(): remember the image blobs of the most recent view-batched forward
_last_blobs = None


def keep_last_blobs(enabled: bool):
    """Measurement aid (bench.py): keep a reference to the image blobs of the latest view-batched forward so that
    walked_entries() can read the blend kernels' own counters after an eager iteration."""
    global _KEEP_LAST, _last_blobs
    _KEEP_LAST = bool(enabled)
    if not enabled:
        _last_blobs = None


def walked_entries():
    """(list entries the blend forward staged, per view; list entries the blend backward walked, all views) of the most
    recent view-batched forward / backward pair -- header words 10 / 11 of the image blobs (include/fnx_raster.h),
    counted by the kernels themselves, one atomic per workgroup.  Blocking.  None without keep_last_blobs(True)."""
    if _last_blobs is None:
        return None
    img, ibytes, V = _last_blobs
    torch.cuda.synchronize()
    al = (-img.data_ptr()) % 256
    words = [img[v * ibytes + al: v * ibytes + al + 64].view(torch.int32).cpu().tolist() for v in range(V)]
    return [w[10] & 0xFFFFFFFF for w in words], sum(w[11] & 0xFFFFFFFF for w in words)


_VIEW_BATCHES = None  # weak set of the ViewBatch objects that hold a sort state


def coherent_sort_counters(P=None):
    """(calls in coherent mode, of those: in-launch full sorts) summed over every live sort state (P: only the states of
    this splat count) -- blocking.  A caller
    that sees the second number grow (a scene whose splats jump further than the repair window between calls: e.g.
    particles at the fringe of the velocity field's support) switches back with set_coherent_sort(False): the mode is
    exact either way, a fallback only costs time (~2 ms per view and call)."""
    calls = falls = 0
    for vb in list(_VIEW_BATCHES or ()):
        for key in list(vb._sort_state):
            ch, kP = key[0], key[1]
            if P is not None and kP != int(P):
                continue
            for c in vb.sort_counters(ch, kP, sort_key=key[2] if len(key) > 2 else None):
                calls += c[0]
                falls += c[1]
    return calls, falls


COHERENT_MAX_WORKGROUPS = 640


def coherent_sort_pays(P, V):
    """The repair launch is one 512-thread workgroup per 2 048 ranks and view, two resident per compute unit (LDS): up to
    ~2.5 per compute unit it is one round of work and beats the nine radix launches (config 3: 490 workgroups, 58 against
    88 us); BASELINE config 5 (350 k splats x 8 views = 1 368 workgroups, twice per iteration) runs it in rounds and
    measured 320 against 341 it/s.  set_coherent_sort(2) / options=dict(coherent_sort=2) force the mode."""
    return ((int(P) + 2047) // 2048) * int(V) <= COHERENT_MAX_WORKGROUPS


def coherent_sort_states(P=None):
    """How many (view, channel count, splat count) sort states are alive (P: of this splat count only): a state's FIRST
    repair call may need the full sort without that saying anything about the scene (the radix passes that seeded it order
    culled splats last, the repair calls by depth), so a caller that judges the mode by coherent_sort_counters allows that
    many."""
    return sum(vb.V * sum(1 for k in vb._sort_state if P is None or k[1] == int(P)) for vb in list(_VIEW_BATCHES or ()))


def set_lean_geometry(enabled: bool):
    """View batches only: do not write the per-view GeometryState copies nothing reads back, one world covariance for
    all views (include/fnx_raster.h fnx_set_lean_geometry)."""
    _OPTS["lean_geometry"] = 1 if enabled else 0
    _lib.check(_lib.raster().fnx_set_lean_geometry(_OPTS["lean_geometry"]))


def set_segmented_forward(enabled: bool):
    """View batches, fast arithmetic, one image: the blend forward cuts the lists of deep tiles into segments that are
    blended at the same time and put together per pixel (include/fnx_raster.h fnx_raster_opts_t.segment_scratch)."""
    _OPTS["segments"] = 1 if enabled else 0


def segmented_forward_counters():
    """Per view batch and view: (work items of the last blend launch, of those segments, tiles cut, running total of tiles
    blended on as one list behind some segment, running total of the batches that took) -- blocking read-back."""
    out = []
    for vb in list(_VIEW_BATCHES or ()):
        for ch, t in vb._seg_scratch.items():
            rs = vb.settings[0]
            for v in range(vb.V):
                w = (C.c_uint32 * 16)()
                _lib.check(_lib.raster().fnx_segment_scratch_read(t.data_ptr(), int(rs.image_width), int(rs.image_height), v,
                                                                  _lib.raw_stream(), w))
                out.append(tuple(int(x) for x in w[:5]) + ((tuple(int(x) for x in w[6:11]),) if any(w[6:11]) else ()))
    return out


def set_deep_kernel(mode: int):
    _OPTS["deep_kernel"] = int(mode)
    _lib.check(_lib.raster().fnx_set_deep_kernel(int(mode)))


def set_backward_form(form: str | int):
    """Which form of the blend backward runs: "lanes" / 1 = a lane per list entry (default), "rows" / 0 = a lane per pixel
    (include/fnx_raster.h fnx_set_backward_form; the dual mode always runs the latter)."""
    f = {"rows": 0, "lanes": 1}.get(form, form)
    _lib.check(_lib.raster().fnx_set_backward_form(int(f)))


def get_backward_form() -> str:
    return ("rows", "lanes")[_lib.raster().fnx_get_backward_form()]


def get_blend_math() -> str:
    return ("exact", "fast")[_OPTS["blend_math"]]


def _call_options(vbatch, channels, P, overrides, zero3=None, grad_splat_limit=None, dual=None):
    """(fnx_raster_opts_t, dict) of one view-batched forward: module defaults overlaid with the instance's overrides."""
    o = dict(_OPTS)
    if overrides:
        unknown = set(overrides) - set(o)
        if unknown:
            raise ValueError(f"unknown rasteriser options: {sorted(unknown)}")
        o.update(overrides)
    sort_mode, state_ptr = (_lib.FNX_SORT_NARROW if o["sort_narrow"] else _lib.FNX_SORT_FULL), None
    if o["coherent_sort"] and P > 0 and (o["coherent_sort"] == 2 or coherent_sort_pays(P, vbatch.V)):
        state, seeded = vbatch.sort_state(channels, P, o.get("sort_key"))
        if state is not None:
            state_ptr = state.data_ptr()
            if seeded:
                sort_mode = _lib.FNX_SORT_COHERENT  # else: this call's radix passes leave the state seeded
    seg_ptr = None
    if o["segments"] and o["blend_math"] == 1 and dual is None and P > 0:
        seg_ptr = vbatch.segment_scratch(channels).data_ptr()
    opts = _lib.make_opts(blend_math=o["blend_math"], lean_geometry=o["lean_geometry"], sort_mode=sort_mode,
                          deep_kernel=o["deep_kernel"],
                          grad_splat_limit=-1 if grad_splat_limit is None else int(grad_splat_limit),
                          zero3=zero3, sort_state=state_ptr, dual=dual, segment_scratch=seg_ptr)
    return opts, o


_between_stages_hook = None  # called (no arguments) between the binning stage and the emit / blend stage of a view batch


def set_between_stages_hook(fn):
    """`fn()` runs on the host right after the view-batched forward has enqueued its first stage (preprocess, depth
    sort, counts, scans) and before it enqueues emission + blending: a caller can record an event there to hang a
    side branch under the throughput-bound blend kernels instead of under the latency-bound sort.  None removes it."""
    global _between_stages_hook
    _between_stages_hook = fn


def set_host_sync(enabled: bool, initial_capacity: int | None = None):
    """enabled=False: never read num_rendered back inside forward (see module docstring)."""
    global _HOST_SYNC
    _HOST_SYNC = bool(enabled)
    if initial_capacity is not None:
        _capacity_hwm["default"] = int(initial_capacity)


def _ring(dev: torch.device) -> torch.Tensor:
    """Header words of sync-free forwards are copied (device to device, on the forward's stream) into
    this persistent buffer and read back from here: it lives outside any hipGraph memory pool, so
    reading it never touches memory a captured graph owns."""
    r = _status_ring.get(dev.index)
    if r is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("run one sync-free forward eagerly before capturing a graph")
        r = torch.zeros(_RING + _CAPTURED, 8, dtype=torch.int32, device=dev)
        _status_ring[dev.index] = r
    return r


def _status_slots(dev: torch.device, n: int, key):
    """n consecutive ring rows for the header words of one sync-free forward (one per view), registered for
    check_status().  An eager forward takes rotating rows and is checked once; a forward that is being recorded into
    a hipGraph takes rows of its own that every replay rewrites, and check_status() re-reads them on every call --
    replays run no Python, so a popped entry would never be looked at again."""
    global _ring_next, _captured_next
    ring = _ring(dev)
    if torch.cuda.is_current_stream_capturing():
        if _captured_next + n > _CAPTURED:
            raise RuntimeError("status slots of captured forwards exhausted: rasterizer.release_captured_status()")
        slot = _RING + _captured_next
        _captured_next += n
        for v in range(n):
            _captured_status.append((dev.index, slot + v, key))
        return ring, slot
    if _ring_next % _RING + n > _RING:  # keep the n rows contiguous
        _ring_next += _RING - _ring_next % _RING
    slot = _ring_next % _RING
    _ring_next += n
    for v in range(n):
        _pending_status.append((dev.index, slot + v, key))
    while len(_pending_status) > _RING:
        del _pending_status[0]
    return ring, slot


def release_captured_status():
    """Forget the status rows of captured forwards (call when the graphs that own them are gone)."""
    global _captured_next
    _captured_status.clear()
    _captured_next = 0
    # ... and un-pin the temporal-coherence sort states those graphs referred to by raw pointer: a sequence that re-captures
    # frame after frame with a growing splat count otherwise keeps every frame's state (~60 MB per frame of config 3)
    for vb in list(_VIEW_BATCHES or ()):
        vb.release_captured_sort_states()


def check_status():
    """Blocking: raise if any sync-free forward since the last call overflowed its binning capacity;
    also refreshes the binning high-water marks and `last_num_rendered`."""
    global last_num_rendered, max_sort_span_bits
    if not _pending_status and not _captured_status:
        return
    host = {d: r.cpu() for d, r in _status_ring.items()}  # one small D2H copy per device, synchronising
    err = None
    if _pending_status:  # the most recent eager forward (per-call instances; the static ones of a split forward on top)
        dev_index, slot, _ = _pending_status[-1]
        last_num_rendered = int(host[dev_index][slot][0]) + int(host[dev_index][slot][3])
    entries = list(reversed(_pending_status)) + list(_captured_status)
    _pending_status.clear()
    for dev_index, slot, key in entries:
        n, status, cap = (int(x) for x in host[dev_index][slot][:3])
        _capacity_hwm[key] = max(_capacity_hwm.get(key, 0), int(n * _CAP_SLACK) + 1024)
        max_sort_span_bits = max(max_sort_span_bits, (int(host[dev_index][slot][5]) >> 24) & 0xFF)
        if status == _lib.FNX_ERR_CAPACITY and err is None:
            if n > cap:
                err = _lib.FnxError(status, f"binning capacity {cap} < num_rendered {n}")
            else:  # written by a backward that found the blob laid out for another capacity than its own
                err = _lib.FnxError(status, f"a backward pass ran with another binning capacity than its forward's ({cap}): "
                                            "its gradients are zero")
        if status == _lib.FNX_ERR_INVALID_ARG and err is None:
            err = _lib.FnxError(status, "a backward pass asked for gradients beyond its forward's gradient limit "
                                        "(grad_splat_limit / dual mode): refused, its gradients are zero")
        if status == _lib.FNX_ERR_SORT_SPAN and err is None:
            err = _lib.FnxError(status, "set_sort_narrow(True) but a view's depth keys span 2^27 ulps or more: the fourth "
                                        "sort pass was needed (set_sort_narrow(False) and render again)")
    if err is not None:
        raise err


def _ptr(t: torch.Tensor):
    """0-element tensor == 'not provided' == NULL (rasterize_points.cu:95-101)."""
    return t.data_ptr() if t.numel() else None


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings, channels=3, grad_splat_limit=None):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings, channels, grad_splat_limit)



def _gradient_mode(need, M):
    """`geometry_only` of fnx_rasterize_backward_ex from autograd's needs_input_grad (means3D, means2D, sh, colors,
    opacities, scales, rotations, cov3D, ...): 3 = only the 3D positions, 1 = nobody asked for opacity / colour / SH
    gradients, 2 = nobody asked for the gradient of the positions (3D or screen-space), 0 = everything."""
    if not (need[2] or need[3] or need[4]) and M == 0:
        if need[0] and not (need[1] or need[5] or need[6] or need[7]):
            return 3
        return 1
    if not (need[0] or need[1]):
        return 2
    return 0


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings, channels, grad_splat_limit=None):
        lib = _lib.raster()
        if means3D.dim() != 2 or means3D.shape[1] != 3:
            raise RuntimeError("means3D must have dimensions (num_points, 3)")  # rasterize_points.cu:56-58
        if not means3D.is_cuda:
            raise RuntimeError("fluidnexus_amd rasteriser: tensors must be on a HIP device (no CPU path)")
        dev = means3D.device
        rs = raster_settings
        P = means3D.shape[0]
        H, W = int(rs.image_height), int(rs.image_width)
        Cn = int(channels)
        means3D = _f32c(means3D)
        sh, colors_precomp, opacities = _f32c(sh.to(dev)), _f32c(colors_precomp.to(dev)), _f32c(opacities)
        scales, rotations, cov3Ds_precomp = _f32c(scales.to(dev)), _f32c(rotations.to(dev)), _f32c(cov3Ds_precomp.to(dev))
        bg, view, proj, campos = _f32c(rs.bg), _f32c(rs.view_matrix), _f32c(rs.proj_matrix), _f32c(rs.campos)
        M = sh.shape[1] if sh.numel() else 0
        stream = _lib.raw_stream()
        u8 = dict(dtype=torch.uint8, device=dev)
        geom = torch.empty(lib.fnx_geom_bytes(P, W, H), **u8)
        img = torch.empty(lib.fnx_image_bytes(W, H), **u8)
        if P == 0:  # rasterize_points.cu:81: zeros, no kernels
            color = torch.zeros(Cn, H, W, dtype=torch.float32, device=dev)
            depth = torch.zeros(1, H, W, dtype=torch.float32, device=dev)
            radii = torch.zeros(0, dtype=torch.int32, device=dev)
            binning = torch.empty(0, **u8)
            num_rendered = 0
        else:
            color = torch.empty(Cn, H, W, dtype=torch.float32, device=dev)
            depth = torch.empty(1, H, W, dtype=torch.float32, device=dev)
            radii = torch.empty(P, dtype=torch.int32, device=dev)
            _lib.check(lib.fnx_forward_stage1(
                Cn, geom.data_ptr(), img.data_ptr(), P, int(rs.sh_degree), M, W, H, means3D.data_ptr(), _ptr(sh),
                _ptr(colors_precomp), opacities.data_ptr(), _ptr(scales), float(rs.scale_modifier), _ptr(rotations),
                _ptr(cov3Ds_precomp), view.data_ptr(), proj.data_ptr(), campos.data_ptr(), float(rs.tan_fov_x),
                float(rs.tan_fov_y), int(bool(rs.prefiltered)), radii.data_ptr(), stream))
            key = (dev.index, W, H, Cn, P)
            known = _capacity_hwm.get(key) or _capacity_hwm.get("default")
            if _HOST_SYNC or not known:
                # reference behaviour; also the first sync-free call of a shape (seeds the high-water mark)
                n = C.c_int(0)
                _lib.check(lib.fnx_read_num_rendered(img.data_ptr(), W, H, stream, C.byref(n)))
                num_rendered = cap = int(n.value)
                global last_num_rendered
                last_num_rendered = num_rendered
                if not _HOST_SYNC:
                    _capacity_hwm[key] = cap = int(num_rendered * _CAP_SLACK) + 1024
            else:
                cap = known
                _capacity_hwm[key] = cap
                num_rendered = -1
            binning = torch.empty(lib.fnx_binning_bytes(cap), **u8)
            if grad_splat_limit is not None:  # the backward will stop behind every pixel's last splat below the limit
                _lib.check(lib.fnx_request_gradient_limit(int(grad_splat_limit)))
            _lib.check(lib.fnx_forward_stage2(Cn, geom.data_ptr(), binning.data_ptr(), cap, img.data_ptr(), P, W, H,
                                              bg.data_ptr(), _ptr(colors_precomp), radii.data_ptr(),
                                              color.data_ptr(), depth.data_ptr(), stream))
            if num_rendered < 0:
                ring, slot = _status_slots(dev, 1, key)
                al = (-img.data_ptr()) % 256  # the header sits at the first 256-byte boundary of the blob
                ring[slot].copy_(img[al:al + 32].view(torch.int32))
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.capacity = cap if P else 0  # the binning blob's layout depends on it
        ctx.channels = Cn
        ctx.grad_splat_limit = -1 if grad_splat_limit is None else int(grad_splat_limit)
        ctx.aux = (bg, view, proj, campos)
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom, binning, img)
        ctx.mark_non_differentiable(radii, depth)  # the reference ignores their grads (__init__.py:83)
        ctx.set_materialize_grads(False)  # no zero-filled stand-ins for the unused grads of radii / depth
        return color, radii, depth

    @staticmethod
    def backward(ctx, grad_out_color, _grad_radii, _grad_depth):
        if grad_out_color is None:
            return (None,) * 11
        lib = _lib.raster()
        rs = ctx.raster_settings
        Cn = ctx.channels
        colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom, binning, img = ctx.saved_tensors
        bg, view, proj, campos = ctx.aux
        dev = means3D.device
        P = means3D.shape[0]
        H, W = int(rs.image_height), int(rs.image_width)
        M = sh.shape[1] if sh.numel() else 0
        # one zero-filled slab for all nine gradient arrays (torch::zeros x9, rasterize_points.cu:150-158)
        widths = (3, 3, Cn, 4, 1, 6, 3 * M, 3, 4)
        flat = torch.zeros(P * sum(widths), dtype=torch.float32, device=dev)
        parts, off = [], 0
        for w in widths:
            parts.append(flat[off:off + P * w])
            off += P * w
        g_means3D, g_means2D, g_colors, g_conic, g_opacity, g_cov3D, g_sh, g_scales, g_rot = parts
        if P != 0:
            dL = _f32c(grad_out_color)
            stream = _lib.raw_stream()
            # skip what autograd would throw away: opacity / colour / SH gradients nobody asked for, and
            # splats the caller declared gradient-free (GaussianRasterizer.grad_splat_limit)
            need = ctx.needs_input_grad  # (means3D, means2D, sh, colors, opacities, scales, rotations, cov3D, ...)
            geometry_only = _gradient_mode(need, M)
            _lib.check(lib.fnx_rasterize_backward_ex(
                Cn, P, int(rs.sh_degree), M, int(ctx.capacity), bg.data_ptr(), W, H, means3D.data_ptr(),
                _ptr(sh), _ptr(colors_precomp), _ptr(scales), float(rs.scale_modifier), _ptr(rotations),
                _ptr(cov3Ds_precomp), view.data_ptr(), proj.data_ptr(), campos.data_ptr(), float(rs.tan_fov_x),
                float(rs.tan_fov_y), radii.data_ptr(), geom.data_ptr(), _ptr(binning), img.data_ptr(), dL.data_ptr(),
                g_means2D.data_ptr(), g_conic.data_ptr(), g_opacity.data_ptr(), g_colors.data_ptr(),
                g_means3D.data_ptr(), g_cov3D.data_ptr(), g_sh.data_ptr() if M else None, g_scales.data_ptr(),
                g_rot.data_ptr(), ctx.grad_splat_limit, geometry_only, stream))
        # same order and shapes as ch3 __init__.py:128-138 / rasterize_points.cu:150-158
        return (g_means3D.view(P, 3), g_means2D.view(P, 3), g_sh.view(P, M, 3), g_colors.view(P, Cn),
                g_opacity.view(P, 1), g_scales.view(P, 3), g_rot.view(P, 4), g_cov3D.view(P, 6), None, None, None)


class ViewBatch:
    """The cameras of one training batch, stacked once for the view-batched entry points
    (include/fnx_raster.h, fnx_*_views): V raster settings that share image size, background, scale
    modifier, SH degree and the prefiltered flag."""

    def __init__(self, settings_list):
        if not settings_list:
            raise ValueError("ViewBatch needs at least one view")
        if len(settings_list) > _lib.FNX_MAX_VIEWS:
            raise ValueError(f"at most {_lib.FNX_MAX_VIEWS} views per batch (got {len(settings_list)})")
        rs0 = settings_list[0]
        for rs in settings_list[1:]:
            same = (int(rs.image_height) == int(rs0.image_height) and int(rs.image_width) == int(rs0.image_width)
                    and float(rs.scale_modifier) == float(rs0.scale_modifier) and int(rs.sh_degree) == int(rs0.sh_degree)
                    and bool(rs.prefiltered) == bool(rs0.prefiltered)
                    and (rs.bg is rs0.bg or torch.equal(rs.bg, rs0.bg)))
            if not same:
                raise ValueError("the views of a batch must share image size, bg, scale_modifier, sh_degree, prefiltered")
        self.settings = list(settings_list)
        self.V = len(settings_list)
        self.view = torch.stack([_f32c(rs.view_matrix).reshape(16) for rs in settings_list]).contiguous()
        self.proj = torch.stack([_f32c(rs.proj_matrix).reshape(16) for rs in settings_list]).contiguous()
        self.campos = torch.stack([_f32c(rs.campos).reshape(3) for rs in settings_list]).contiguous()
        self.bg = _f32c(rs0.bg)
        self.tan_x = (C.c_float * self.V)(*[float(rs.tan_fov_x) for rs in settings_list])
        self.tan_y = (C.c_float * self.V)(*[float(rs.tan_fov_y) for rs in settings_list])
        self._depth_hint = {}
        self._seg_scratch = {}
        self._sort_state = {}
        self._sort_pinned = set()  # keys handed out while a stream was capturing: a hipGraph holds their raw pointers
        self._sort_tick = {}

    def segment_scratch(self, channels):
        """u8 tensor [V * fnx_segment_scratch_bytes(W, H)], zero-filled once: the segmented blend forward's work list and
        per-segment records (fnx_raster_opts_t.segment_scratch).  Like the depth hints, one per channel count: it lives as
        long as the camera batch, so a captured graph's raw pointer stays valid."""
        t = self._seg_scratch.get(int(channels))
        if t is None:
            rs = self.settings[0]
            n = _lib.raster().fnx_segment_scratch_bytes(int(rs.image_width), int(rs.image_height))
            t = self._seg_scratch[int(channels)] = torch.zeros(self.V * n + 256, dtype=torch.uint8, device=self.view.device)
            global _VIEW_BATCHES
            if _VIEW_BATCHES is None:
                import weakref
                _VIEW_BATCHES = weakref.WeakSet()
            _VIEW_BATCHES.add(self)
        return t

    def sort_state(self, channels, P, sort_key=None):
        """(u8 tensor [V * fnx_sort_state_bytes(P)], seeded) -- the persistent state of the temporal-coherence depth sort
        for this camera batch, channel count, splat count and `sort_key` (include/fnx_raster.h fnx_raster_opts_t.sort_state):
        zero-filled once; `seeded` is False for the one call whose radix passes seed it.  (None, False) while a graph is
        being captured before any eager call allocated it.
        ONE splat set per state: the state holds the previous call's depth order, so two different splat sets of the same
        size rendered through the same cameras must not share it (each call would repair the other set's order, fail its
        verification and pay the in-launch full sort: exact, but slow) -- give each its own `sort_key`
        (GaussianRasterizerViews.options["sort_key"], any hashable).
        The kernels receive the state as a RAW pointer, and a captured hipGraph keeps that pointer without a tensor
        reference: states handed out during a capture stay until release_captured_sort_states(); otherwise the
        least recently used states go once more than eight exist (splat counts of earlier frames)."""
        key = (int(channels), int(P), sort_key)
        self._sort_tick[key] = max(self._sort_tick.values(), default=0) + 1
        ent = self._sort_state.get(key)
        if ent is not None and torch.cuda.is_current_stream_capturing():
            self._sort_pinned.add(key)
        if ent is None:
            if torch.cuda.is_current_stream_capturing():
                return None, False
            for old in sorted((k for k in self._sort_state if k not in self._sort_pinned),
                              key=lambda k: self._sort_tick.get(k, 0))[:max(0, len(self._sort_state) - len(self._sort_pinned) - 8)]:
                del self._sort_state[old]
                self._sort_tick.pop(old, None)
            n = _lib.raster().fnx_sort_state_bytes(int(P))
            ent = self._sort_state[key] = torch.zeros(self.V * n, dtype=torch.uint8, device=self.view.device)
            global _VIEW_BATCHES
            if _VIEW_BATCHES is None:
                import weakref
                _VIEW_BATCHES = weakref.WeakSet()
            _VIEW_BATCHES.add(self)
            return ent, False  # this call's radix passes seed it
        return ent, True

    def release_captured_sort_states(self):
        """Un-pin the sort states a destroyed hipGraph referred to."""
        self._sort_pinned.clear()

    def sort_counters(self, channels, P, why=False, outliers=False, sort_key=None):
        """Per view (calls in coherent mode, of those: in-launch full sorts[, why: fnx_sort_state_read's bit mask][, splats
        taken as outliers so far: fnx_sort_state_outliers]) -- blocking read-back."""
        ent = self._sort_state.get((int(channels), int(P), sort_key))
        if ent is None:
            return []
        # ONE device-to-host copy for all views (fnx_sort_state_read / _outliers copy and synchronise once per view: 1.4 ms
        # per call at five views, twice per frame of a sequence): the header block is the first thing in a view's state
        # (csrc/fnx_state.h sort_state_layout; words COH_REPAIRS 5, COH_FALLBACKS 4, COH_WHY 6, COH_OUTLIERS 9)
        n = ent.numel() // self.V
        if ent.data_ptr() % 256 == 0 and n % 4 == 0:
            h = ent.view(self.V, n)[:, :64].contiguous().view(torch.int32).cpu().numpy().view("uint32")
            return [(int(r[5]), int(r[4])) + ((int(r[6]),) if why else ()) + ((int(r[9]),) if outliers else ()) for r in h]
        out, lib = [], _lib.raster()
        stream = _lib.raw_stream()
        for v in range(self.V):
            pair = (C.c_uint32 * 3)()
            _lib.check(lib.fnx_sort_state_read(ent.data_ptr(), int(P), v, stream, pair))
            row = (int(pair[0]), int(pair[1])) + ((int(pair[2]),) if why else ())
            if outliers:
                n_o = C.c_uint32(0)
                _lib.check(lib.fnx_sort_state_outliers(ent.data_ptr(), int(P), v, stream, C.byref(n_o)))
                row += (int(n_o.value),)
            out.append(row)
        return out

    def depth_hint(self, channels):
        """u32 [V, T]: how deep every tile of every view went in the previous forward of this batch (per channel
        count: the 1- and 3-channel renders of a frame see different splat sets).  The forward keeps it up to date
        and uses it to start long, non-saturating tiles first, at raised wave priority (include/fnx_raster.h)."""
        if not _DEEP_VARIANT:
            return None
        h = self._depth_hint.get(channels)
        if h is None:
            if torch.cuda.is_current_stream_capturing():
                return None  # allocated by the first eager forward; a capture that comes first simply runs without it
            rs = self.settings[0]
            T = ((int(rs.image_width) + 15) // 16) * ((int(rs.image_height) + 15) // 16)
            h = self._depth_hint[channels] = torch.zeros(self.V, T, dtype=torch.int32, device=self.view.device)
        return h


class StaticBin:
    """The LAST P_static splats of a view batch's arrays, binned once (include/fnx_raster.h, static-split
    extension): the frozen background Gaussians of render_dynamics seen from the fixed cameras of a frame.
    Arguments are the STATIC subset's arrays ([P_static, ...], same conventions as GaussianRasterizer.forward);
    `id_offset` = number of per-call splats in front of them in the full arrays.  One host sync (instance counts).
    Pass the object as `static_bin=` to GaussianRasterizerViews.forward with the FULL arrays: only the leading
    id_offset splats are then preprocessed / sorted / binned per call, the result is bit-identical."""

    materialize_all = False  # debug: every forward writes the whole merged point_list (parity tests)

    def __init__(self, view_batch, means3D, opacities, id_offset, shs=None, colors_precomp=None, scales=None,
                 rotations=None, cov3D_precomp=None, channels=3):
        lib = _lib.raster()
        if not means3D.is_cuda:
            raise RuntimeError("fluidnexus_amd rasteriser: tensors must be on a HIP device (no CPU path)")
        dev = means3D.device
        vb, rs = view_batch, view_batch.settings[0]
        V, P = vb.V, means3D.shape[0]
        if P == 0:
            raise ValueError("StaticBin needs at least one static splat")
        H, W = int(rs.image_height), int(rs.image_width)
        empty = torch.empty(0, dtype=torch.float32, device=dev)
        f = lambda t: empty if t is None else _f32c(t.detach())  # noqa: E731
        means3D, opacities = f(means3D), f(opacities)
        shs, colors_precomp, scales, rotations, cov3D_precomp = f(shs), f(colors_precomp), f(scales), f(rotations), f(cov3D_precomp)
        M = shs.shape[1] if shs.numel() else 0
        stream = _lib.raw_stream()
        u8 = dict(dtype=torch.uint8, device=dev)
        gbytes, ibytes = lib.fnx_geom_bytes(P, W, H), lib.fnx_image_bytes(W, H)
        geom, img = torch.empty(V * gbytes, **u8), torch.empty(V * ibytes, **u8)
        radii = torch.empty(V, P, dtype=torch.int32, device=dev)
        _lib.check(lib.fnx_forward_stage1_views(
            int(channels), V, geom.data_ptr(), img.data_ptr(), P, int(rs.sh_degree), M, W, H, means3D.data_ptr(), _ptr(shs),
            _ptr(colors_precomp), opacities.data_ptr(), _ptr(scales), float(rs.scale_modifier), _ptr(rotations),
            _ptr(cov3D_precomp), vb.view.data_ptr(), vb.proj.data_ptr(), vb.campos.data_ptr(), vb.tan_x, vb.tan_y,
            int(bool(rs.prefiltered)), radii.data_ptr(), stream))
        n, self.R = C.c_int(0), []
        for v in range(V):
            _lib.check(lib.fnx_read_num_rendered(img.data_ptr() + v * ibytes, W, H, stream, C.byref(n)))
            self.R.append(int(n.value))
        self.R_cap = max(self.R)
        scratch = torch.empty(V * lib.fnx_binning_bytes(self.R_cap), **u8)
        self.blob = torch.empty(V * lib.fnx_static_bytes(P, W, H, self.R_cap), **u8)
        _lib.check(lib.fnx_static_finalize_views(V, geom.data_ptr(), scratch.data_ptr(), img.data_ptr(), P, W, H,
                                                 int(id_offset), self.R_cap, radii.data_ptr(), self.blob.data_ptr(), stream))
        self.view_batch, self.P, self.id_offset, self.channels, self.W, self.H = vb, P, int(id_offset), int(channels), W, H


def rasterize_gaussians_views(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                              view_batch, channels=3, grad_splat_limit=None, static_bin=None, options=None, dual_bg=None):
    return _RasterizeGaussiansViews.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                          cov3Ds_precomp, view_batch, channels, grad_splat_limit, static_bin, options,
                                          dual_bg)


class _RasterizeGaussiansViews(torch.autograd.Function):
    """All views of a batch through one launch sequence.  Outputs [V,C,H,W] colour, [V,P] radii,
    [V,1,H,W] depth, each slice bit-identical to _RasterizeGaussians with that view's settings; the
    gradients of the shared inputs are the sums over the views, `means2D` ([V,P,3]) receives the
    per-view screen-space gradients."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, vbatch,
                channels, grad_splat_limit=None, static_bin=None, options=None, dual_bg=None):
        lib = _lib.raster()
        if means3D.dim() != 2 or means3D.shape[1] != 3:
            raise RuntimeError("means3D must have dimensions (num_points, 3)")
        if not means3D.is_cuda:
            raise RuntimeError("fluidnexus_amd rasteriser: tensors must be on a HIP device (no CPU path)")
        dev = means3D.device
        rs = vbatch.settings[0]
        V, P = vbatch.V, means3D.shape[0]
        H, W = int(rs.image_height), int(rs.image_width)
        Cn = int(channels)
        ctx.dual = None
        if static_bin is not None:
            return _RasterizeGaussiansViews._forward_split(ctx, means3D, sh, colors_precomp, opacities, scales, rotations,
                                                           cov3Ds_precomp, vbatch, Cn, grad_splat_limit, static_bin, options,
                                                           dual_bg)
        if dual_bg is not None:
            raise ValueError("dual mode (dual_bg) needs a static_bin: the second image is the per-call splats'")
        means3D = _f32c(means3D)
        sh, colors_precomp, opacities = _f32c(sh.to(dev)), _f32c(colors_precomp.to(dev)), _f32c(opacities)
        scales, rotations, cov3Ds_precomp = _f32c(scales.to(dev)), _f32c(rotations.to(dev)), _f32c(cov3Ds_precomp.to(dev))
        M = sh.shape[1] if sh.numel() else 0
        stream = _lib.raw_stream()
        u8 = dict(dtype=torch.uint8, device=dev)
        gbytes, ibytes = lib.fnx_geom_bytes(P, W, H), lib.fnx_image_bytes(W, H)
        geom = torch.empty(V * gbytes, **u8)
        img = torch.empty(V * ibytes, **u8)
        cap = 0
        ctx.status_ptr = None
        opts, ctx.options = _call_options(vbatch, Cn, P, options, grad_splat_limit=grad_splat_limit)
        if P == 0:
            color = torch.zeros(V, Cn, H, W, dtype=torch.float32, device=dev)
            depth = torch.zeros(V, 1, H, W, dtype=torch.float32, device=dev)
            radii = torch.zeros(V, 0, dtype=torch.int32, device=dev)
            binning = torch.empty(0, **u8)
        else:
            color = torch.empty(V, Cn, H, W, dtype=torch.float32, device=dev)
            depth = torch.empty(V, 1, H, W, dtype=torch.float32, device=dev)
            radii = torch.empty(V, P, dtype=torch.int32, device=dev)
            hint = vbatch.depth_hint(Cn)
            hint_ptr = hint.data_ptr() if hint is not None else None
            _lib.check(lib.fnx_forward_stage1_views_split_opts(
                Cn, V, geom.data_ptr(), img.data_ptr(), P, int(rs.sh_degree), M, W, H, means3D.data_ptr(), _ptr(sh),
                _ptr(colors_precomp), opacities.data_ptr(), _ptr(scales), float(rs.scale_modifier), _ptr(rotations),
                _ptr(cov3Ds_precomp), vbatch.view.data_ptr(), vbatch.proj.data_ptr(), vbatch.campos.data_ptr(),
                vbatch.tan_x, vbatch.tan_y, int(bool(rs.prefiltered)), radii.data_ptr(), None, 0, 0, hint_ptr,
                C.byref(opts), stream))
            key = (dev.index, W, H, Cn, P)
            known = _capacity_hwm.get(key) or _capacity_hwm.get("default")
            synced = _HOST_SYNC or not known
            if synced:
                # one capacity for all views = the largest instance count (reference: exact size per call)
                n, cap = C.c_int(0), 0
                for v in range(V):
                    _lib.check(lib.fnx_read_num_rendered(img.data_ptr() + v * ibytes, W, H, stream, C.byref(n)))
                    cap = max(cap, int(n.value))
                if not _HOST_SYNC:
                    cap = int(cap * _CAP_SLACK) + 1024
                    _capacity_hwm[key] = max(_capacity_hwm.get(key, 0), cap)
            else:
                cap = known
                _capacity_hwm[key] = cap
            bbytes = lib.fnx_binning_bytes(cap)
            binning = torch.empty(V * bbytes, **u8)
            if _between_stages_hook is not None:
                _between_stages_hook()
            status_ptr = None
            ctx.status_rows = None
            if not synced:  # deferred status check: the forward's last kernel writes the headers into ring slots
                ring, slot = _status_slots(dev, V, key)
                status_ptr = ring[slot:slot + V].data_ptr()
                ctx.status_rows = (dev.index, slot, V, key)
            ctx.status_ptr = status_ptr
            # (opts.grad_splat_limit: the backward will stop behind every pixel's last splat below the limit)
            _lib.check(lib.fnx_forward_stage2_views_split_opts(Cn, V, geom.data_ptr(), binning.data_ptr(), cap,
                                                               img.data_ptr(), P, W, H, vbatch.bg.data_ptr(),
                                                               color.data_ptr(), depth.data_ptr(), status_ptr, None, 0, 0,
                                                               0, hint_ptr, C.byref(opts), stream))
        if _KEEP_LAST and P and not torch.cuda.is_current_stream_capturing():
            global _last_blobs
            _last_blobs = (img, ibytes, V)
        ctx.vbatch = vbatch
        ctx.capacity = cap
        ctx.channels = Cn
        ctx.static_bin = None
        ctx.grad_splat_limit = -1 if grad_splat_limit is None else int(grad_splat_limit)
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom, binning, img)
        ctx.mark_non_differentiable(radii, depth)
        ctx.set_materialize_grads(False)  # no zero-filled stand-ins for the unused grads of radii / depth
        return color, radii, depth

    @staticmethod
    def _forward_split(ctx, means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, vbatch, Cn,
                       grad_splat_limit, sb, options=None, dual_bg=None):
        """Static-split forward: the trailing sb.P splats were binned once (StaticBin); this call preprocesses, sorts
        and bins only the leading ones and the blend kernel merges the two streams of every tile.
        dual_bg (f32[1]): DUAL mode (include/fnx_raster.h fnx_raster_dual_t) -- the same pass also blends a single-channel
        image of the per-call splats alone (value = channel 0 of their colours, background dual_bg); two more outputs,
        colour1 [V,1,H,W] and depth1 [V,1,H,W], and the backward differentiates both images at once."""
        lib = _lib.raster()
        dev = means3D.device
        rs = vbatch.settings[0]
        V, P_all = vbatch.V, means3D.shape[0]
        P = P_all - sb.P
        H, W = int(rs.image_height), int(rs.image_width)
        if sb.view_batch is not vbatch or sb.id_offset != P or sb.channels != Cn or P <= 0:
            raise ValueError("static_bin was built for another view batch / splat split / channel count")
        if grad_splat_limit is not None and int(grad_splat_limit) > P:
            raise ValueError("static splats take no gradients: grad_splat_limit must not exceed the per-call splat count")
        means3D = _f32c(means3D)
        sh, colors_precomp, opacities = _f32c(sh.to(dev)), _f32c(colors_precomp.to(dev)), _f32c(opacities)
        scales, rotations, cov3Ds_precomp = _f32c(scales.to(dev)), _f32c(rotations.to(dev)), _f32c(cov3Ds_precomp.to(dev))
        M = sh.shape[1] if sh.numel() else 0
        stream = _lib.raw_stream()
        u8 = dict(dtype=torch.uint8, device=dev)
        gbytes, ibytes = lib.fnx_geom_bytes(P, W, H), lib.fnx_image_bytes(W, H)
        geom = torch.empty(V * gbytes, **u8)
        img = torch.empty(V * ibytes, **u8)
        color = torch.empty(V, Cn, H, W, dtype=torch.float32, device=dev)
        depth = torch.empty(V, 1, H, W, dtype=torch.float32, device=dev)
        radii = torch.empty(V, P_all, dtype=torch.int32, device=dev)
        hint = vbatch.depth_hint(Cn)
        hint_ptr = hint.data_ptr() if hint is not None else None
        # positions-only backward ahead (geometry_only = 3): its accumulator is zero-filled by this forward's per-splat
        # kernel instead of a fill launch between the image loss and the backward
        ctx.g_means3D_zeroed = None
        if _gradient_mode(ctx.needs_input_grad, M) == 3:
            ctx.g_means3D_zeroed = torch.empty(P_all, 3, dtype=torch.float32, device=dev)
        dual = None
        if dual_bg is not None:
            if Cn != 3 or (grad_splat_limit is not None and int(grad_splat_limit) != P):
                raise ValueError("dual mode: 3 channels, and every per-call splat takes gradients (grad_splat_limit = their count)")
            if _gradient_mode(ctx.needs_input_grad, M) != 3 and any(ctx.needs_input_grad):
                raise ValueError("dual mode: the backward is the positions-only one (means3D the only leaf, no screen-space gradient)")
            img1 = torch.empty(V * ibytes, **u8)
            color1 = torch.empty(V, 1, H, W, dtype=torch.float32, device=dev)
            depth1 = torch.empty(V, 1, H, W, dtype=torch.float32, device=dev)
            bg1 = _f32c(dual_bg).reshape(-1)[:1]
            dual = _lib.make_dual(img1.data_ptr(), bg1.data_ptr(), color1.data_ptr(), depth1.data_ptr())
            grad_splat_limit = P
        opts, ctx.options = _call_options(
            vbatch, Cn, P, options, grad_splat_limit=grad_splat_limit,
            zero3=ctx.g_means3D_zeroed.data_ptr() if ctx.g_means3D_zeroed is not None else None, dual=dual)
        ctx.status_ptr = None
        _lib.check(lib.fnx_forward_stage1_views_split_opts(
            Cn, V, geom.data_ptr(), img.data_ptr(), P, int(rs.sh_degree), M, W, H, means3D.data_ptr(), _ptr(sh),
            _ptr(colors_precomp), opacities.data_ptr(), _ptr(scales), float(rs.scale_modifier), _ptr(rotations),
            _ptr(cov3Ds_precomp), vbatch.view.data_ptr(), vbatch.proj.data_ptr(), vbatch.campos.data_ptr(),
            vbatch.tan_x, vbatch.tan_y, int(bool(rs.prefiltered)), radii.data_ptr(), sb.blob.data_ptr(), sb.P, sb.R_cap,
            hint_ptr, C.byref(opts), stream))
        key = (dev.index, W, H, Cn, P, "split")
        known = _capacity_hwm.get(key) or _capacity_hwm.get("default")
        synced = _HOST_SYNC or not known
        if synced:
            n, cap = C.c_int(0), 0
            for v in range(V):
                _lib.check(lib.fnx_read_num_rendered(img.data_ptr() + v * ibytes, W, H, stream, C.byref(n)))
                cap = max(cap, int(n.value))
            if not _HOST_SYNC:
                cap = int(cap * _CAP_SLACK) + 1024
                _capacity_hwm[key] = max(_capacity_hwm.get(key, 0), cap)
        else:
            cap = known
            _capacity_hwm[key] = cap
        binning = torch.empty(V * (lib.fnx_binning_bytes_dual if dual is not None else lib.fnx_binning_bytes_split)(cap, sb.R_cap), **u8)
        if _between_stages_hook is not None:
            _between_stages_hook()
        status_ptr = None
        ctx.status_rows = None
        if not synced:
            ring, slot = _status_slots(dev, V, key)
            status_ptr = ring[slot:slot + V].data_ptr()
            ctx.status_rows = (dev.index, slot, V, key)
        ctx.status_ptr = status_ptr
        # (opts.grad_splat_limit; in split mode the library already stops at the first static id)
        _lib.check(lib.fnx_forward_stage2_views_split_opts(
            Cn, V, geom.data_ptr(), binning.data_ptr(), cap, img.data_ptr(), P, W, H, vbatch.bg.data_ptr(),
            color.data_ptr(), depth.data_ptr(), status_ptr, sb.blob.data_ptr(), sb.P, sb.R_cap,
            int(bool(StaticBin.materialize_all)), hint_ptr, C.byref(opts), stream))
        if _KEEP_LAST and not torch.cuda.is_current_stream_capturing():
            global _last_blobs
            _last_blobs = (img, ibytes, V)
        ctx.vbatch = vbatch
        ctx.capacity = cap
        ctx.channels = Cn
        ctx.static_bin = sb
        ctx.grad_splat_limit = P if grad_splat_limit is None else int(grad_splat_limit)
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom, binning, img)
        ctx.set_materialize_grads(False)
        if dual is not None:
            ctx.dual = (img1, bg1)
            ctx.mark_non_differentiable(radii, depth, depth1)
            return color, radii, depth, color1, depth1
        ctx.mark_non_differentiable(radii, depth)
        return color, radii, depth

    @staticmethod
    def backward(ctx, grad_out_color, _grad_radii, _grad_depth, grad_out_color1=None, _grad_depth1=None):
        if grad_out_color is None and grad_out_color1 is None:
            return (None,) * 14
        lib = _lib.raster()
        vbatch, Cn = ctx.vbatch, ctx.channels
        if ctx.dual is not None:  # either image's gradient may be missing: zeros
            rs0 = vbatch.settings[0]
            shape = (vbatch.V, 1, int(rs0.image_height), int(rs0.image_width))
            dev0 = ctx.saved_tensors[1].device
            if grad_out_color is None:
                grad_out_color = torch.zeros((vbatch.V, 3) + shape[2:], dtype=torch.float32, device=dev0)
            if grad_out_color1 is None:
                grad_out_color1 = torch.zeros(shape, dtype=torch.float32, device=dev0)
        # the backward runs with the options its forward ran with (blend arithmetic, lean geometry state)
        # a backward that refuses to run (capacity / gradient-limit mismatch) leaves its reason in the forward's status rows:
        # look at them again at the next check_status(), even if the forward's own entry has been consumed since
        rows = getattr(ctx, "status_rows", None)
        if rows is not None and ctx.status_ptr is not None and not torch.cuda.is_current_stream_capturing():
            for v in range(rows[2]):
                _pending_status.append((rows[0], rows[1] + v, rows[3]))
        dual = None
        if ctx.dual is not None:
            dL1 = _f32c(grad_out_color1)
            dual = _lib.make_dual(ctx.dual[0].data_ptr(), ctx.dual[1].data_ptr(), dL_dpix1=dL1.data_ptr())
        bopts = _lib.make_opts(blend_math=ctx.options["blend_math"], lean_geometry=ctx.options["lean_geometry"], dual=dual)
        static_args = (lambda sb: (sb.blob.data_ptr(), sb.P, sb.R_cap) if sb is not None else (None, 0, 0))
        rs = vbatch.settings[0]
        colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom, binning, img = ctx.saved_tensors
        dev = means3D.device
        V, P = vbatch.V, means3D.shape[0]
        sb = ctx.static_bin
        H, W = int(rs.image_height), int(rs.image_width)
        M = sh.shape[1] if sh.numel() else 0
        need = ctx.needs_input_grad
        geometry_only = _gradient_mode(need, M)
        if geometry_only == 3 and P != 0:
            # positions only: the blend backward's flush runs the geometry backward itself and adds into this one array
            g_means3D = getattr(ctx, "g_means3D_zeroed", None)  # zero-filled by the forward (one use only)
            ctx.g_means3D_zeroed = None
            if g_means3D is None:
                g_means3D = torch.zeros(P, 3, dtype=torch.float32, device=dev)
            dL = _f32c(grad_out_color)
            args = (Cn, V, P - (sb.P if sb is not None else 0), int(rs.sh_degree), M, vbatch.bg.data_ptr(), W, H,
                    means3D.data_ptr(), None, _ptr(colors_precomp), _ptr(scales), float(rs.scale_modifier), _ptr(rotations),
                    _ptr(cov3Ds_precomp), vbatch.view.data_ptr(), vbatch.proj.data_ptr(), vbatch.campos.data_ptr(),
                    vbatch.tan_x, vbatch.tan_y, radii.data_ptr(), geom.data_ptr(), _ptr(binning), ctx.capacity,
                    img.data_ptr(), dL.data_ptr(), None, None, None, None, None, None,
                    g_means3D.data_ptr(), None, None, None, None, ctx.grad_splat_limit, 3)
            stream = _lib.raw_stream()
            _lib.check(lib.fnx_rasterize_backward_views_split_opts(*args, *static_args(sb), ctx.status_ptr,
                                                                   C.byref(bopts), stream))
            return (g_means3D,) + (None,) * 13
        if ctx.dual is not None:
            raise RuntimeError("dual mode: only the positions-only backward is implemented (means3D the only leaf)")
        # per-view accumulators first, then the arrays summed over the views; one zero-filled slab
        widths = [V * 3, V * 4] + ([] if geometry_only == 1 else [V, V * Cn]) + [3, Cn, 1, 6, 3 * M, 3, 4]
        flat = torch.zeros(P * sum(widths), dtype=torch.float32, device=dev)
        parts, off = [], 0
        for w in widths:
            parts.append(flat[off:off + P * w])
            off += P * w
        if geometry_only == 1:
            g_means2D, g_conic, g_means3D, g_colors, g_opacity, g_cov3D, g_sh, g_scales, g_rot = parts
            g_opacity_v, g_colors_v = g_opacity, g_colors  # never written in this mode
        else:
            g_means2D, g_conic, g_opacity_v, g_colors_v, g_means3D, g_colors, g_opacity, g_cov3D, g_sh, g_scales, g_rot = parts
        if P != 0:
            dL = _f32c(grad_out_color)
            stream = _lib.raw_stream()
            args = (Cn, V, P - (sb.P if sb is not None else 0), int(rs.sh_degree), M, vbatch.bg.data_ptr(), W, H,
                    means3D.data_ptr(), _ptr(sh),
                    _ptr(colors_precomp), _ptr(scales), float(rs.scale_modifier), _ptr(rotations), _ptr(cov3Ds_precomp),
                    vbatch.view.data_ptr(), vbatch.proj.data_ptr(), vbatch.campos.data_ptr(), vbatch.tan_x, vbatch.tan_y,
                    radii.data_ptr(), geom.data_ptr(), _ptr(binning), ctx.capacity, img.data_ptr(), dL.data_ptr(),
                    g_means2D.data_ptr(), g_conic.data_ptr(), g_opacity_v.data_ptr(), g_colors_v.data_ptr(),
                    g_opacity.data_ptr(), g_colors.data_ptr(), g_means3D.data_ptr(), g_cov3D.data_ptr(),
                    g_sh.data_ptr() if M else None, g_scales.data_ptr(), g_rot.data_ptr(), ctx.grad_splat_limit,
                    geometry_only)
            # (split mode: the gradient arrays span all splats; rows of the static ones stay zero)
            _lib.check(lib.fnx_rasterize_backward_views_split_opts(*args, *static_args(sb), ctx.status_ptr,
                                                                   C.byref(bopts), stream))
        if V == 1 and geometry_only != 1:  # a single view accumulates straight into its per-view arrays
            g_opacity, g_colors = g_opacity_v, g_colors_v
        return (g_means3D.view(P, 3), g_means2D.view(V, P, 3), g_sh.view(P, M, 3), g_colors.view(P, Cn),
                g_opacity.view(P, 1), g_scales.view(P, 3), g_rot.view(P, 4), g_cov3D.view(P, 6), None, None, None, None, None,
                None)


class GaussianRasterizationSettings(NamedTuple):
    """Field names as in ch3 __init__.py:143-154 (not upstream 3DGS's tanfovx/viewmatrix/...)."""
    image_height: int
    image_width: int
    tan_fov_x: float
    tan_fov_y: float
    bg: torch.Tensor
    scale_modifier: float
    view_matrix: torch.Tensor
    proj_matrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool


class GaussianRasterizer(nn.Module):
    """ch3 __init__.py:157-215.  `channels` selects the ch3 (3) or ch1 (1) behaviour."""

    channels = 3
    # Optional hint (not in the reference): only splats with index < grad_splat_limit need gradients,
    # e.g. when static background Gaussians are concatenated behind the optimised ones.
    grad_splat_limit = None

    def __init__(self, raster_settings, channels=None):
        super().__init__()
        self.raster_settings = raster_settings
        if channels is not None:
            self.channels = int(channels)

    def mark_visible(self, positions):
        with torch.no_grad():
            rs = self.raster_settings
            lib = _lib.raster()
            if not positions.is_cuda:
                raise RuntimeError("fluidnexus_amd rasteriser: tensors must be on a HIP device (no CPU path)")
            pos = _f32c(positions)
            P = pos.shape[0]
            present = torch.zeros(P, dtype=torch.bool, device=pos.device)
            _lib.check(lib.fnx_mark_visible(P, _ptr(pos), _f32c(rs.view_matrix).data_ptr(),
                                            _f32c(rs.proj_matrix).data_ptr(), _ptr(present),
                                            _lib.raw_stream()))
        return present

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        raster_settings = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide exactly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (
                (scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        if self.channels != 3 and colors_precomp is None:
            raise RuntimeError("For non-RGB, provide precomputed Gaussian colors!")  # rasterizer_impl.cu:226-228
        empty = torch.empty(0, dtype=torch.float32, device=means3D.device)
        if shs is None:
            shs = empty
        if colors_precomp is None:
            colors_precomp = empty
        if scales is None:
            scales = empty
        if rotations is None:
            rotations = empty
        if cov3D_precomp is None:
            cov3D_precomp = empty
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                   raster_settings, self.channels, self.grad_splat_limit)


class GaussianRasterizerViews(nn.Module):
    """GaussianRasterizer over all views of a training batch at once (extension, see ViewBatch):
    forward(...) takes the same arguments, `means2D` being [V,P,3], and returns
    (color [V,C,H,W], radii [V,P], depth [V,1,H,W])."""

    channels = 3
    grad_splat_limit = None
    static_bin = None  # a StaticBin over the trailing splats of the arrays passed to forward (static-split mode)
    options = None     # per-instance overrides of the module-level option defaults (keys of rasterizer._OPTS)
    # f32[1]: DUAL mode (static_bin set, channels 3) -- forward also returns the single-channel image of the per-call
    # splats alone (value = channel 0 of their colours, this background): (color, radii, depth, color1, depth1)
    dual_bg = None

    def __init__(self, raster_settings_list, channels=None, options=None):
        super().__init__()
        self.view_batch = raster_settings_list if isinstance(raster_settings_list, ViewBatch) else ViewBatch(raster_settings_list)
        if channels is not None:
            self.channels = int(channels)
        if options is not None:
            self.options = dict(options)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide exactly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (
                (scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        if self.channels != 3 and colors_precomp is None:
            raise RuntimeError("For non-RGB, provide precomputed Gaussian colors!")
        empty = torch.empty(0, dtype=torch.float32, device=means3D.device)
        shs = empty if shs is None else shs
        colors_precomp = empty if colors_precomp is None else colors_precomp
        scales = empty if scales is None else scales
        rotations = empty if rotations is None else rotations
        cov3D_precomp = empty if cov3D_precomp is None else cov3D_precomp
        return rasterize_gaussians_views(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                         cov3D_precomp, self.view_batch, self.channels, self.grad_splat_limit,
                                         self.static_bin, self.options, self.dual_bg)
