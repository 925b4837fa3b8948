"""-m gpu: two frames of a FluidNexus-style sequence through every stage, in the reference's own call order
(entries_fluid_nexus/train_physical_particle.py:82-98,190-302,456-470 and train_visual_particle.py:133-222), at a size
that runs in seconds: first-frame fit of the visual particles -> hidden particles + stabilising solver steps ->
per frame [remove / emit / predict / project] -> physical-particle optimisation -> confirm + advect -> visual-particle
stage -> next frame with MORE particles (the emitter).  Checks what has to hold across the hand-overs: particle
counts and ids grow as the emitter says, every stage moves its own leaves and nothing else, everything stays finite,
and the renders of consecutive stages are consistent."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _finite(*ts):
    return all(bool(torch.isfinite(t).all()) for t in ts)


def test_two_frames_through_every_stage():
    from fluidnexus_amd import harness as Hn
    from fluidnexus_amd import synthetic as S
    from fluidnexus_amd.helpers.helper_gaussian import get_model
    from fluidnexus_amd.helpers.helper_pipe import get_render_pipe
    dev = "cuda"
    size, V = 96, 3
    model = SimpleNamespace(init_visual_num_pts=6000, init_thick_visual_num_pts=1500, init_x_mid=0.34, init_z_mid=-0.225,
                            init_visual_y_min=0.0, init_visual_y_max=0.45, init_visual_y_thick_min=0.15,
                            init_visual_radius_small_max=0.04, init_visual_radius_max=0.08,
                            init_hidden_radius_max=0.08, init_hidden_y_min=0.0, init_hidden_y_max=0.45,
                            init_hidden_delta=0.0125, emitter_hidden_delta=0.0125, emitter_visual_delta=0.004,
                            emitter_center_y_hidden=-0.01, emitter_center_y_visual=-0.005,
                            emitter_center_y_hidden_max=0.0, emitter_center_y_visual_max=0.0,
                            emitter_visual_radius_ratio=9.0, emitter_hidden_radius_ratio=5.0)
    optim = SimpleNamespace(position_lr_init=1.6e-4, position_lr_final=1.6e-6, position_lr_delay_mult=0.01,
                            position_lr_max_steps=30000, H=2.0, KNN_K=100, p0=1.5, secs=0.033, k=3, alpha=-1.5,
                            emit_ratio_hidden=1.0, emit_ratio_visual=1.5)
    np.random.seed(3)
    torch.manual_seed(3)
    gm = get_model("gm_dynamics")()
    gm.setup_constants(optim)
    bgd = S.backdrop_gaussians(2500, seed=1, channels=3)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)  # noqa: E731
    gm._gs_xyz, gm._gs_color, gm._gs_rotation = t(bgd["means3D"]), t(bgd["colors"]), t(bgd["rotations"])
    gm._gs_scales = t(np.log(bgd["scales"]))
    gm._gs_opacity = t(np.log(bgd["opacities"] / (1 - bgd["opacities"])))
    cams = S.arc_cameras(V, size, size, device=dev)

    # ---- frame 0: fit the visual particles' positions (tpp:82-163) ------------------------------------------------
    gm.create_particles_visual(model)
    gm.prepare_visual_particles_for_rendering()
    n_vis0 = gm._visual_xyz.shape[0]
    first = Hn.FirstFrameLoop(gm, cams, rd_pipe="render_dynamics", cfg=dict(Hn.SCALAR_REAL, distance_threshold_visual=0.004),
                              capturable=False)
    first.make_targets()
    x0 = gm._visual_xyz.detach().clone()
    for _ in range(4):
        first.iteration()
    assert _finite(gm._visual_xyz) and float((gm._visual_xyz.detach() - x0).abs().max()) > 0

    # ---- frame 0: hidden particles, stabilising solver steps (tpp:190-228) ---------------------------------------
    gm.detach_visual_and_scale()
    assert not gm._visual_xyz.requires_grad and float(gm._visual_xyz.abs().max()) > 10  # scaled units now
    gm.create_particles_hidden(model)
    n_hid0 = gm._xyz.shape[0]
    assert n_hid0 > 500 and gm._particle_id_max == n_hid0
    for _ in range(2):
        gm.remove_invalid_particles()
        gm.guess_hidden_particles(stable=True)
        for _ in range(3):
            gm.update_solver_counts()
        for _ in range(3):
            gm.project_gas_constraints()
        gm.confirm_guess_hidden_particles()
    assert _finite(gm._xyz, gm._velocity)
    gm.prepare_emitter_points(model)
    n_emit_h, n_emit_v = gm.hidden_emitter_points.shape[0], gm.visual_emitter_points.shape[0]
    assert n_emit_h > 0 and n_emit_v > 0

    render_dynamics, GRsetting, GRzer = get_render_pipe("render_dynamics")
    bg = torch.zeros(3, device=dev)
    counts = []
    for frame in (1, 2):
        # ---- simulation step into the frame (tpp:283-302) --------------------------------------------------------
        gm.remove_invalid_particles()
        n_h, n_v, id_max = gm._xyz.shape[0], gm._visual_xyz.shape[0], gm._particle_id_max
        gm.emit_new_particles()
        assert gm._xyz.shape[0] == n_h + n_emit_h and gm._particle_id_max == id_max + n_emit_h
        assert gm._visual_xyz.shape[0] == n_v + n_emit_v + int(0.5 * n_emit_v)
        assert int(gm._particle_id.max()) == gm._particle_id_max - 1 and gm._particle_id.unique().numel() == gm._xyz.shape[0]
        gm.guess_hidden_particles()
        for _ in range(3):
            gm.update_solver_counts()
        for _ in range(3):
            gm.project_gas_constraints()
        gm.training_setup_current(optim)
        gm.prepare_visual_particles_for_rendering()
        assert gm._visual_color.shape[0] == gm._visual_xyz.shape[0]
        counts.append((gm._xyz.shape[0], gm._visual_xyz.shape[0]))

        # ---- physical-particle stage (tpp:329-432) ---------------------------------------------------------------
        loop = Hn.HotLoop(gm, cams, image_loss="fused", fused_physics=True, defer_visual_backward=True, batched_views=True,
                          cfg=dict(Hn.SMOKE, distance_threshold_visual=0.004))
        loop.make_targets()
        e0, vis0 = gm._estimate_xyz_nn.detach().clone(), gm._visual_xyz.clone()
        for _ in range(4):
            loop.iteration()
        assert _finite(gm._estimate_xyz_nn) and float((gm._estimate_xyz_nn.detach() - e0).abs().max()) > 0
        assert torch.equal(gm._visual_xyz, vis0)  # the stage optimises the hidden particles only
        assert gm.knn_k_report()["within_cap"]

        # ---- accept the frame (tpp:456-458) ----------------------------------------------------------------------
        gm.confirm_guess_hidden_particles_from_nn()
        gm.update_visual_xyz_from_nn()
        gm.confirm_guess_hidden_particles_wo_velocity()
        assert _finite(gm._xyz, gm._velocity, gm._visual_xyz) and not torch.equal(gm._visual_xyz, vis0)

        # ---- visual-particle stage of the frame (tvp:133-222) ----------------------------------------------------
        two = Hn.HotLoopLevelTwo(gm, cams, batched_views=True)
        two.make_targets()
        before = {n: getattr(gm, f"_visual_{n}").detach().clone() for n in gm._L2}
        xyz_before = gm._visual_xyz.clone()
        for _ in range(3):
            two.iteration()
        assert all(float((getattr(gm, f"_visual_{n}").detach() - before[n]).abs().max()) > 0 for n in gm._L2)
        assert torch.equal(gm._visual_xyz, xyz_before)  # positions are fixed in this stage
        with torch.no_grad():
            img = render_dynamics(cams[0], gm, None, bg, GRsetting=GRsetting, GRzer=GRzer, pos_type="visual", scale=True)["render"]
        assert _finite(img) and float(img.max()) > 0.05
        for n in gm._L2:  # the next frame's set-up starts from plain tensors again
            setattr(gm, f"_visual_{n}", getattr(gm, f"_visual_{n}").detach())
    assert counts[1][0] == counts[0][0] + n_emit_h and counts[1][1] > counts[0][1]
    assert n_vis0 < counts[0][1] and n_hid0 < counts[0][0]
