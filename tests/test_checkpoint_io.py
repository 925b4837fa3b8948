"""CPU: checkpoint / scene formats.  tests/golden/checkpoint/ was written by the reference's own
save_hidden / save_visual (gm_dynamics.py:1834-1923) from the state in checkpoint_state.npz
(tests/golden/gen_reference_golden.py::gen_checkpoint): the loaders must reproduce that state and the savers must
reproduce those files byte for byte."""
import filecmp
import json
import os

import numpy as np
import torch

from fluidnexus_amd.gaussian_splatting.gm_dynamics import GaussianModel
from fluidnexus_amd.utils.ply_io import read_vertex_ply, write_vertex_ply

HERE = os.path.dirname(__file__)
CK = os.path.join(HERE, "golden", "checkpoint")
STATE = np.load(os.path.join(HERE, "golden", "checkpoint_state.npz"))


def test_load_reference_checkpoint_and_save_it_back(tmp_path):
    gm = GaussianModel()
    assert gm.load_hidden(CK, 7, device="cpu") is True
    assert gm.load_visual(CK, 7, device="cpu") == STATE["_visual_xyz"].shape[0]
    for k in STATE.files:
        got = getattr(gm, k).numpy()
        ref = STATE[k]
        assert got.shape == ref.shape, k
        # positions went through / scale_factor (save) and * scale_factor (load) in fp32
        assert np.allclose(got, ref, rtol=3e-7, atol=0), k
    assert gm._particle_id.dtype == torch.int32
    sv = json.load(open(os.path.join(CK, "frame_007_scalar_values.json")))
    assert (gm.scale_factor, gm._secs, gm.alpha, gm.k, gm.p0) == (sv["scale_factor"], sv["secs"], sv["alpha"], sv["k"], sv["p0"])
    assert (gm.buoyancy_decay_rate, gm.buoyancy_max_y, gm.min_neighbors, gm.remove_out_boundary) == (0.98, 0.6, 1, False)
    assert (gm.emit_ratio_hidden, gm.emit_ratio_visual, gm.emit_counter) == (0.5, 2.0, 7)
    assert (gm.total_iterations, gm.total_sim_iterations, gm.total_tb_log_iterations, gm.particle_id_max) == (1234, 56, 78, 40)
    # save from the exact original state: identical files
    for k in STATE.files:
        setattr(gm, k, torch.tensor(STATE[k]))
    gm._particle_id_max = 40
    out = str(tmp_path / "ck")
    gm.save_all(out, 7)
    names = sorted(os.listdir(CK))
    assert sorted(os.listdir(out)) == names
    for n in names:
        if n.endswith(".npy"):
            assert filecmp.cmp(os.path.join(CK, n), os.path.join(out, n), shallow=False), n
    assert json.load(open(os.path.join(out, "frame_007_scalar_values.json"))) == sv
    assert list(json.load(open(os.path.join(out, "frame_007_scalar_values.json")))) == list(sv)  # key order too


def test_background_ply_roundtrip_and_layout(tmp_path):
    rng = np.random.RandomState(0)
    gm = GaussianModel()
    N = 17
    gm._gs_xyz = torch.tensor(rng.normal(size=(N, 3)).astype(np.float32))
    gm._gs_color = torch.tensor(rng.uniform(size=(N, 3)).astype(np.float32))
    gm._gs_opacity = torch.tensor(rng.normal(size=(N, 1)).astype(np.float32))
    gm._gs_scales = torch.tensor(rng.normal(size=(N, 3)).astype(np.float32))
    gm._gs_rotation = torch.tensor(rng.normal(size=(N, 4)).astype(np.float32))
    path = str(tmp_path / "bg" / "point_cloud.ply")
    gm.save_background_ply(path)
    raw = open(path, "rb").read()
    header, body = raw.split(b"end_header\n", 1)
    lines = header.decode().strip().split("\n")
    assert lines[:3] == ["ply", "format binary_little_endian 1.0", f"element vertex {N}"]
    props = [l.split()[-1] for l in lines[3:]]
    assert props == (["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2", "f_rest_0", "f_rest_1", "f_rest_2",
                      "opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3", "color_0", "color_1",
                      "color_2"])                                         # gm_background.py:184-203
    assert all(l.startswith("property float ") for l in lines[3:]) and len(body) == N * len(props) * 4
    names, col = read_vertex_ply(path)
    assert np.allclose(col["x"], -gm._gs_xyz[:, 0].numpy()) and np.allclose(col["y"], -gm._gs_xyz[:, 1].numpy())
    assert np.allclose(col["f_dc_1"], (gm._gs_color[:, 1].numpy() - 0.5) / 0.28209479177387814, rtol=1e-6)
    assert np.all(col["nx"] == 0) and np.all(col["f_rest_2"] == 0)
    other = GaussianModel()
    other.load_ply(path, device="cpu")
    for k in ("_gs_xyz", "_gs_color", "_gs_opacity", "_gs_scales", "_gs_rotation"):
        assert torch.equal(getattr(other, k), getattr(gm, k)), k


def test_ply_reader_handles_ascii_and_mixed_types(tmp_path):
    p = tmp_path / "a.ply"
    p.write_text("ply\nformat ascii 1.0\ncomment made by hand\nelement vertex 2\nproperty float x\nproperty double y\n"
                 "property uchar red\nelement face 0\nproperty list uchar int vertex_indices\nend_header\n"
                 "1.5 2.5 255\n-1 0.25 7\n")
    names, col = read_vertex_ply(str(p))
    assert names == ["x", "y", "red"] and np.allclose(col["x"], [1.5, -1]) and np.allclose(col["red"], [255, 7])
    q = str(tmp_path / "b.ply")
    write_vertex_ply(q, ["a", "b"], np.array([[1, 2], [3, 4], [5, 6]], np.float32))
    names, col = read_vertex_ply(q)
    assert names == ["a", "b"] and np.allclose(col["b"], [2, 4, 6])
