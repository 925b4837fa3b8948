"""Drop-in for FluidDynamics/submodules/gaussian_rasterization_ch1/diff_gaussian_rasterization_ch1
(selected by helpers/helper_pipe.py:14-23 for render_fluid): 1 channel; SH input is refused (rasterizer_impl.cu:242)."""
from fluidnexus_amd.rasterizer import GaussianRasterizationSettings, _RasterizeGaussians
from fluidnexus_amd.rasterizer import GaussianRasterizer as _Base
from fluidnexus_amd.rasterizer import rasterize_gaussians as _rasterize

NUM_CHANNELS = 1  # cuda_rasterizer/config.h:15


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings):
    return _rasterize(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                      raster_settings, NUM_CHANNELS)


class GaussianRasterizer(_Base):
    channels = NUM_CHANNELS

    def __init__(self, raster_settings):
        super().__init__(raster_settings, NUM_CHANNELS)


__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "_RasterizeGaussians"]
