"""CPU tests of the host side: the C-ABI libraries load and export every declared symbol (no compute
calls without a GPU), scratch-size/layout helpers, the Python interface's argument validation
(ch3 __init__.py:184-190), and the plug-in selectors."""
import ctypes as C
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fnx_[a-z0-9_]+)\s*\(", src)) - {"fnx_alloc_fn"})


def test_raster_library_exports_every_declared_symbol():
    from fluidnexus_amd import _lib
    lib = _lib.raster()
    names = _declared("fnx_raster.h")
    assert set(names) == set(_lib.SYMBOLS), set(names) ^ set(_lib.SYMBOLS)
    for n in names:
        getattr(lib, n)
    assert lib.fnx_abi_version() == _lib.ABI_VERSION == 7  # include/fnx_raster.h FNX_ABI_VERSION


def test_physics_and_losses_libraries_export_every_declared_symbol():
    from fluidnexus_amd import _physics_lib, losses
    pl = _physics_lib.physics()
    assert set(_declared("fnx_physics.h")) == set(_physics_lib.SYMBOLS)
    for n in _physics_lib.SYMBOLS:
        getattr(pl, n)
    ll = losses.lib()
    assert set(_declared("fnx_losses.h")) == set(losses.SYMBOLS)
    for n in losses.SYMBOLS:
        getattr(ll, n)
    assert ll.fnx_l1_ssim_tiles(3, 512, 512, 0) == 3 * 16 * 16 and ll.fnx_l1_ssim_tiles(3, 512, 512, 1) == 256  # 32x32 tiles


def test_scratch_layouts():
    from fluidnexus_amd import _lib
    lib = _lib.raster()
    g = _lib.geom_layout(300000, 512, 512)
    offs = [g.depths, g.clamped, g.radii, g.means2D, g.cov3D, g.conic_opacity, g.rgb, g.tiles_touched, g.sort_key0,
            g.sort_key1, g.sort_val0, g.sort_val1, g.rect, g.rect_sorted, g.krec, g.sort_hist, g.blk_hist, g.blk_rel, g.total]
    assert offs == sorted(offs) and all(o % 256 == 0 for o in offs)
    assert lib.fnx_geom_bytes(300000, 512, 512) == g.total
    assert lib.fnx_geom_bytes(0, 512, 512) <= lib.fnx_geom_bytes(1000, 512, 512) < lib.fnx_geom_bytes(2000, 512, 512)
    im = _lib.image_layout(512, 512)
    assert im.n_contrib - im.final_T >= 512 * 512 * 4 and lib.fnx_image_bytes(512, 512) == im.total
    b = _lib.binning_layout(1000)
    assert b.point_list == 0 and b.total >= 4000 and lib.fnx_binning_bytes(1000) == b.total


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from fluidnexus_amd import _lib
    monkeypatch.setattr(_lib, "_RASTER", None)
    monkeypatch.setattr(_lib, "_HERE", str(tmp_path))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.raster()


def _settings(pkg):
    z = torch.zeros
    return pkg.GaussianRasterizationSettings(image_height=16, image_width=16, tan_fov_x=0.4, tan_fov_y=0.4, bg=z(3),
                                             scale_modifier=1.0, view_matrix=torch.eye(4), proj_matrix=torch.eye(4),
                                             sh_degree=0, campos=z(3), prefiltered=False)


@pytest.mark.parametrize("name", ["diff_gaussian_rasterization_ch3", "diff_gaussian_rasterization_ch1"])
def test_rasterizer_argument_validation(name):
    pkg = __import__(name)
    assert pkg.GaussianRasterizationSettings._fields == (
        "image_height", "image_width", "tan_fov_x", "tan_fov_y", "bg", "scale_modifier", "view_matrix", "proj_matrix",
        "sh_degree", "campos", "prefiltered")
    r = pkg.GaussianRasterizer(raster_settings=_settings(pkg))
    x, o = torch.zeros(4, 3), torch.ones(4, 1)
    with pytest.raises(Exception, match="exactly one of either SHs or precomputed colors"):
        r(x, x, o, scales=x, rotations=torch.ones(4, 4))
    with pytest.raises(Exception, match="exactly one of either SHs or precomputed colors"):
        r(x, x, o, shs=torch.zeros(4, 16, 3), colors_precomp=x, scales=x, rotations=torch.ones(4, 4))
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(x, x, o, colors_precomp=x)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(x, x, o, colors_precomp=x, scales=x, rotations=torch.ones(4, 4), cov3D_precomp=torch.zeros(4, 6))
    with pytest.raises(RuntimeError, match="no CPU path"):  # CPU tensors never fall back
        r(x, x, o, colors_precomp=x[:, :pkg.NUM_CHANNELS], scales=x, rotations=torch.ones(4, 4))
    if pkg.NUM_CHANNELS == 1:
        with pytest.raises(RuntimeError, match="non-RGB"):
            r(x, x, o, shs=torch.zeros(4, 16, 3), scales=x, rotations=torch.ones(4, 4))


def test_selectors():
    from fluidnexus_amd.helpers.helper_gaussian import get_model
    from fluidnexus_amd.helpers.helper_pipe import get_render_pipe
    import diff_gaussian_rasterization_ch1 as c1
    import diff_gaussian_rasterization_ch3 as c3
    fn, S_, R_ = get_render_pipe("render_dynamics")
    assert fn.__name__ == "render_dynamics" and R_ is c3.GaussianRasterizer
    fn, S_, R_ = get_render_pipe("render_fluid")
    assert fn.__name__ == "render_fluid" and R_ is c1.GaussianRasterizer
    assert get_render_pipe("render_background")[0].__name__ == "render_background"
    with pytest.raises(NotImplementedError):
        get_render_pipe("nope")
    gm = get_model("gm_dynamics")()
    gm.setup_constants()
    assert abs(gm.poly6_term1 - 315.0 / (64.0 * 3.141592653589793 * 2.0 ** 9)) < 1e-12
    for grp in ("visual", "gs", "rigid"):
        for a, shape in (("xyz", (4, 3)), ("color", (4, 3)), ("scales", (4, 3)), ("rotation", (4, 4)), ("opacity", (4, 1))):
            setattr(gm, f"_{grp}_{a}", torch.ones(*shape))
    gm._opacity_dummy = torch.zeros(4, 1)
    assert torch.allclose(gm.get_gs_rotation.norm(dim=1), torch.ones(4))  # F.normalize activation
    assert torch.allclose(gm.get_visual_scaling, torch.full((4, 3), 2.718281828))
    assert torch.allclose(gm.get_opacity_dummy, torch.full((4, 1), 0.5))
    for p in ("get_visual_xyz", "get_visual_opacity", "get_visual_scaling", "get_visual_rotation", "get_visual_color",
              "get_gs_xyz", "get_gs_opacity", "get_gs_scaling", "get_gs_rotation", "get_gs_color", "get_xyz",
              "get_rigid_xyz", "get_opacity_dummy"):
        getattr(gm, p)


def test_view_sharding():
    from fluidnexus_amd.harness import shard_views
    assert [shard_views(5, r, 4) for r in range(4)] == [[0, 4], [1], [2], [3]]  # 2/1/1/1
    assert sorted(sum((shard_views(8, r, 8) for r in range(8)), [])) == list(range(8))
    assert shard_views(5, 0, 1) == [0, 1, 2, 3, 4]
