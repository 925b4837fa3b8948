"""FluidDynamics/helpers/helper_gaussian.py:4-26: model name -> class."""


def get_model(model="gm_dynamics"):
    if model == "gm_dynamics":
        from ..gaussian_splatting.gm_dynamics import GaussianModel
        return GaussianModel
    raise NotImplementedError(f"model {model} is outside this round's hot-path scope (SURVEY 8(f))")
