"""FluidDynamics/helpers/helper_gaussian.py:4-26: model name -> class (same names, same default, same error)."""


def get_model(model="gm_gs"):
    if model == "gm_fluid":
        # the fluid model of the ScalarReal scenes: fluid and background are not separated
        from ..gaussian_splatting.gm_fluid import GaussianModel
    elif model == "gm_background":
        from ..gaussian_splatting.gm_background import GaussianModel
    elif model == "gm_dynamics":
        from ..gaussian_splatting.gm_dynamics import GaussianModel
    elif model == "gm_gs":
        # vanilla 3DGS with SH colour (gaussian_splatting/gaussian_model.py): the model behind the SH pipe (render_gs)
        from ..gaussian_splatting.gaussian_model import GaussianModel
    else:
        raise ValueError(f"Model {model} not found")
    return GaussianModel
