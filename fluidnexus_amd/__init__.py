"""fluidnexus_amd -- MI355X-native (gfx950) implementation of FluidNexus' per-frame optimisation
hot path: the differentiable 3D-Gaussian rasteriser (ch3 / ch1) and the physics-informed
particle losses, behind the reference's own Python plug-in interface.  See DESIGN.md.
"""
__version__ = "0.1.0"
