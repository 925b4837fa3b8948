"""String -> (render fn, Settings class, Rasterizer class): the plug-in seam of
FluidDynamics/helpers/helper_pipe.py:1-48, resolved onto the MI355X rasteriser packages."""


def get_render_pipe(option="render_gs"):
    from .. import renderer
    if option == "render_fluid":
        import diff_gaussian_rasterization_ch1 as pkg
        fn = renderer.render_fluid
    elif option in ("render_background", "render_dynamics", "render_gs"):
        import diff_gaussian_rasterization_ch3 as pkg
        fn = {"render_background": renderer.render_background, "render_dynamics": renderer.render_dynamics,
              "render_gs": renderer.render}[option]
    else:
        raise NotImplementedError(f"Render {option} not implemented")
    return fn, pkg.GaussianRasterizationSettings, pkg.GaussianRasterizer
