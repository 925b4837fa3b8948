"""Render pipes with the call signatures and return dicts of FluidDynamics/renderer/."""
from .pipes import render, render_background, render_dynamics, render_fluid  # noqa: F401
from .pipes import render_dynamics_views, render_fluid_views  # noqa: F401  (view-batched extensions)
