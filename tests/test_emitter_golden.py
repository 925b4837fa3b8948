"""Frame-boundary bookkeeping of gm_dynamics.GaussianModel against vectors produced by the reference's own class
(tests/golden/gen_reference_golden.py gen_emitter -> emitter.npz): first-frame clouds, nozzle lattices,
emit_new_particles (fractional ratios, extra visual particles, the first-future-frame stacks) and the constant render
attributes.  Host-side logic: runs on the CPU (device="cpu"), bit-exact (same generators, same draw order)."""
import os
from types import SimpleNamespace

import numpy as np
import torch

from fluidnexus_amd.gaussian_splatting.gm_dynamics import GaussianModel

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "emitter.npz"))
HIDDEN = ("xyz", "estimate_xyz", "buoyancy", "force", "velocity", "imass", "counts", "particle_id")


def _args():
    return SimpleNamespace(**eval(str(G["optim"]))), SimpleNamespace(**eval(str(G["model"])))  # repr() of plain dicts


def _same(a, b):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.array_equal(a, b)


def test_first_frame_clouds_emitter_and_emission_match_reference():
    optim, model = _args()
    gm = GaussianModel(device="cpu")
    gm.setup_constants(optim)
    np.random.seed(11)
    gm.create_particles_visual(model)
    _same(gm._visual_xyz, G["visual_xyz0"])
    gm.create_particles_hidden(model)
    for n in HIDDEN:
        _same(getattr(gm, f"_{n}"), G[f"hidden0_{n}"])
    assert gm._particle_id_max == int(G["hidden0_id_max"])

    gm.prepare_emitter_points(model, is_future=True)
    _same(gm.visual_emitter_points, G["emit_future_visual"])
    gm.prepare_emitter_points(model)
    _same(gm.visual_emitter_points, G["emit_visual"])
    _same(gm.hidden_emitter_points, G["emit_hidden"])
    gm.prepare_emitter_future_first_points(model)
    _same(gm.visual_emitter_first_points, G["emit_first_visual"])
    _same(gm.hidden_emitter_first_points, G["emit_first_hidden"])

    gm.detach_visual_and_scale()
    gm.prepare_visual_particles_for_rendering()
    for n in ("color", "scales", "rotation", "opacity"):
        _same(getattr(gm, f"_visual_{n}"), G[f"render0_{n}"])

    torch.manual_seed(5)
    gm.emit_new_particles()
    _same(gm._visual_xyz, G["visual_xyz1"])
    for n in HIDDEN:
        _same(getattr(gm, f"_{n}"), G[f"hidden1_{n}"])
    gm.prepare_future_visual_particles_for_rendering(use_level_two_future=True)
    _same(np.array([getattr(gm, f"_visual_{n}").shape[0] for n in ("color", "scales", "rotation", "opacity")]), G["render1_shapes"])
    _same(gm._visual_opacity[-5:], G["render1_opacity_tail"])

    torch.manual_seed(6)
    gm.emit_new_particles(future_time_index=1)
    _same(gm._visual_xyz, G["visual_xyz2"])
    _same(gm._xyz, G["hidden2_xyz"])
    assert gm._particle_id_max == int(G["hidden2_id_max"]) and gm.emit_counter == int(G["emit_counter"])


def test_spiky_grad_matches_reference():
    optim, _ = _args()
    gm = GaussianModel(device="cpu")
    gm.setup_constants(optim)
    r = torch.from_numpy(G["spiky_r"])
    _same(gm.spiky_grad(r, torch.norm(r, dim=1)), G["spiky_grad"])
