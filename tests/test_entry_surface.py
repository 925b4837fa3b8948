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


@pytest.mark.parametrize("script", sorted(SCRIPT_MODEL))
def test_every_name_is_the_same_kind_of_thing_with_the_same_signature(script):
    """hasattr() would accept a stub: every name must be what it is on the reference's class -- a method taking the
    reference's parameters (same names, same order, defaults where the reference has them; extra parameters of this build
    must be optional), a property, or an instance attribute (tests/golden/entry_script_signatures.json, generated from the
    reference's classes by gen_reference_golden.py: names and parameter lists only)."""
    import inspect
    want = json.load(open(os.path.join(G, "entry_script_signatures.json")))[script]
    gm = get_model(SCRIPT_MODEL[script])(device="cpu")
    bad = []
    for name, w in sorted(want.items()):
        static = inspect.getattr_static(type(gm), name, None)
        if w["kind"] == "method":
            if not callable(getattr(gm, name, None)) or isinstance(static, property):
                bad.append(f"{name}: a method in the reference, not callable here")
                continue
            ps = [p for p in inspect.signature(getattr(gm, name)).parameters.values()]
            if any(p.kind is p.VAR_POSITIONAL for p in ps):
                bad.append(f"{name}: *args hides the signature")
                continue
            # (**kwargs of this build = optional keyword extras, e.g. setup_constants(optim_args=None, **overrides): a
            # reference-style call is unaffected; the reference's own **kwargs must be accepted too)
            ref_kw = any(r.startswith("**") for r in w["params"])
            if ref_kw and not any(p.kind is p.VAR_KEYWORD for p in ps):
                bad.append(f"{name}: the reference accepts **kwargs, this build does not")
                continue
            ps = [p for p in ps if p.kind is not p.VAR_KEYWORD]
            ref = [r for r in w["params"] if not r.startswith("*")]
            for i, rp in enumerate(ref):
                rn, rdef = rp.rstrip("="), rp.endswith("=")
                if i >= len(ps) or ps[i].name != rn:
                    bad.append(f"{name}: parameter {i} is {ps[i].name if i < len(ps) else None!r}, the reference's is {rn!r}")
                    break
                if rdef and ps[i].default is inspect.Parameter.empty:
                    bad.append(f"{name}: {rn} has a default in the reference, none here")
                    break
            else:
                extra = [p.name for p in ps[len(ref):] if p.default is inspect.Parameter.empty]
                if extra:
                    bad.append(f"{name}: extra required parameters {extra}")
        elif w["kind"] == "property":
            if not isinstance(static, property):
                bad.append(f"{name}: a property in the reference, {type(static).__name__} here")
        else:
            if not hasattr(gm, name) or inspect.isfunction(static):
                bad.append(f"{name}: an instance attribute in the reference, missing / a method here")
    assert not bad, f"{script} against {SCRIPT_MODEL[script]}:\n  " + "\n  ".join(bad)


def test_the_getters_the_scripts_read_evaluate_with_the_reference_activations():
    """The properties among the touched names, evaluated on a small state: exp for scales, sigmoid for opacities, the raw
    tensor for positions (gm_dynamics.py:32-40, gm_background.py:36-44)."""
    t = lambda *s: torch.randn(*s, generator=torch.Generator().manual_seed(3))  # noqa: E731
    bg = get_model("gm_background")(device="cpu")
    bg._xyz, bg._opacity, bg._scaling = t(5, 3), t(5, 1), t(5, 3)
    assert torch.equal(bg.get_xyz, bg._xyz)
    assert torch.equal(bg.get_opacity, torch.sigmoid(bg._opacity)) and torch.equal(bg.get_scaling, torch.exp(bg._scaling))
    for name in ("gm_dynamics", "gm_fluid"):
        gm = get_model(name)(device="cpu")
        gm._visual_xyz, gm._visual_scales = t(4, 3), t(4, 3)
        assert torch.equal(gm.get_visual_xyz, gm._visual_xyz)
        assert torch.equal(gm.get_visual_scaling, torch.exp(gm._visual_scales))


def test_scalar_real_emitter_points_match_the_reference():
    """gm_fluid.prepare_emitter_points() -- no arguments, hard-coded nozzle (gm_fluid.py:594-632) -- against the lattices the
    reference's own class builds (tests/golden/emitter_fluid.npz)."""
    z = np.load(os.path.join(G, "emitter_fluid.npz"))
    gm = get_model("gm_fluid")(device="cpu")
    gm.prepare_emitter_points()
    assert (gm.visual_emitter_points.numpy() == z["emit_visual"]).all() and gm.visual_emitter_points.shape == z["emit_visual"].shape
    assert (gm.hidden_emitter_points.numpy() == z["emit_hidden"]).all() and gm.hidden_emitter_points.shape == z["emit_hidden"].shape


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
