"""The Python surface the reference's entry scripts drive: every `gaussians.<attr>` that
entries_fluid_nexus/train_physical_particle.py, train_visual_particle.py and train_background.py (and their
entries_scalar_real twins) touch exists on the model the script's configuration selects, the position dumps write the
reference's files, and the level-two initialisation reproduces the reference's class.
Fixtures: tests/golden/entry_script_names.json (names only), save_particles.npz, level_two_init.npz, all produced by
tests/golden/gen_reference_golden.py from the reference's own sources in the build container."""
import json
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from fluidnexus_amd.helpers.helper_gaussian import get_model

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# script -> model class its shipped configurations select (configs/*.json "model")
SCRIPT_MODEL = {
    "entries_fluid_nexus/train_background.py": "gm_background",
    "entries_fluid_nexus/train_physical_particle.py": "gm_dynamics",
    "entries_fluid_nexus/train_visual_particle.py": "gm_dynamics",
    "entries_scalar_real/train_physical_particle.py": "gm_fluid",
    "entries_scalar_real/train_visual_particle.py": "gm_fluid",
}


@pytest.mark.parametrize("script", sorted(SCRIPT_MODEL))
def test_every_attribute_the_entry_script_touches_exists(script):
    names = json.load(open(os.path.join(G, "entry_script_names.json")))[script]
    gm = get_model(SCRIPT_MODEL[script])(device="cpu")
    missing = [n for n in names if not hasattr(gm, n)]
    assert not missing, f"{script} touches gaussians.{missing} which {SCRIPT_MODEL[script]} does not have"


def _state_model(z):
    gm = get_model("gm_dynamics")(device="cpu")
    gm.scale_factor = 100.0
    for k in ("_xyz", "_estimate_xyz", "_visual_xyz", "_estimate_xyz_nn", "_visual_color", "_visual_scales",
              "_visual_rotation", "_visual_opacity"):
        setattr(gm, k, torch.from_numpy(z[k].copy()))
    return gm


def test_position_dumps_write_the_reference_files(tmp_path):
    z = np.load(os.path.join(G, "save_particles.npz"))
    gm = _state_model(z)
    d = str(tmp_path / "q")
    gm.save_particles_frame(d, 3)
    gm.save_particles_simulation(d, 4)
    gm.save_particles_simulation_guess(d, 5)
    gm.save_particles_optimization_first(d, 0, 120)
    gm.save_particles_optimization(d, torch.from_numpy(z["other_visual"].copy()), 6, 250)
    gm.save_particles_optimization_level_two(d, 7, 999)
    assert sorted(os.listdir(d)) == [str(f) for f in z["files"]]
    for fn in z["files"]:
        got, want = np.load(os.path.join(d, str(fn))), z["file:" + str(fn)]
        assert got.dtype == want.dtype and got.shape == want.shape and (got == want).all(), fn


@pytest.mark.gpu
@pytest.mark.parametrize("case", [0, 1])
def test_level_two_initialisation_matches_the_reference(case):
    """init_quantities_current_level_two (gm_dynamics.py:399-414): scales from fnx_knn_mean_dist2 where the reference
    calls simple-knn's distCUDA2 (fp32 rounding of the mean of three squared distances), inheritance exact."""
    z = np.load(os.path.join(G, "level_two_init.npz"))
    gm = get_model("gm_dynamics")(device="cuda")
    for k in ("xyz", "color", "opacity", "scales", "rotation"):
        setattr(gm, f"_visual_{k}", torch.from_numpy(z[f"c{case}_in_{k}"].copy()).cuda())
    prev = {k: torch.from_numpy(z[f"c{case}_prev_{k}"].copy()).cuda() for k in ("color", "opacity", "scales", "rotation")}
    fl = [bool(x) for x in z[f"c{case}_flags"]]
    oa = SimpleNamespace(init_scales_w_xyz_dist=fl[0], inherit_prev_color=fl[1], inherit_prev_opacity=fl[2],
                         inherit_prev_scales=fl[3], inherit_prev_rotation=fl[4])
    gm.init_quantities_current_level_two(oa, prev["color"], prev["opacity"], prev["scales"], prev["rotation"])
    for k in ("color", "opacity", "rotation"):
        assert (getattr(gm, f"_visual_{k}").cpu().numpy() == z[f"c{case}_out_{k}"]).all(), k
    got, want = gm._visual_scales.cpu().numpy(), z[f"c{case}_out_scales"]
    assert got.shape == want.shape and np.abs(got - want).max() <= 2e-6, np.abs(got - want).max()
