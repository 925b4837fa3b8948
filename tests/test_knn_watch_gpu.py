"""K-cap watch on the fused path (fnx_knn_watch, round 5): the benchmark's loop takes every pair within H, the reference
truncates neighbour lists at KNN_K (torch_cluster max_num_neighbors, gm_dynamics.py:1276,1302,1463).  A cloud whose lists
exceed K must raise the device flag through HotLoop -- and a cloud that stays below must not."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _loop(knn_k):
    from fluidnexus_amd import harness as Hn
    gm, cams = Hn.build_smoke_frame(P_fluid=6000, P_background=2000, hidden_dims=(8, 16, 8), n_views=2, size=96, seed=2)
    gm.KNN_K = knn_k
    cfg = dict(Hn.SMOKE, distance_threshold_visual=0.004)
    loop = Hn.HotLoop(gm, cams, image_loss="fused", fused_physics=True, defer_visual_backward=True, capturable=True,
                      batched_views=True, fused_step=True, cfg=cfg)
    loop.make_targets()
    return gm, loop


def test_lists_below_the_cap_raise_nothing():
    gm, loop = _loop(100)  # the jittered unit lattice has ~33 neighbours within H = 2
    for _ in range(3):
        loop.iteration()
    assert gm.check_knn_k() == 0
    rep = gm.knn_k_report()
    assert rep["within_cap"]


@pytest.mark.parametrize("graph", [False, True])
def test_a_list_beyond_the_cap_is_flagged_without_a_host_sync_in_the_loop(graph):
    from fluidnexus_amd import rasterizer
    gm, loop = _loop(20)  # below the lattice's neighbour count: every search overflows
    rasterizer.set_host_sync(False)
    try:
        loop.iteration()
        rasterizer.check_status()
        if graph:
            gm._knn_flags.zero_()
            loop.capture(warmup=1, iterations=2)
            gm._knn_flags.zero_()
            loop.iteration()  # a replay: the kernels inside the graph raise the flag
        torch.cuda.synchronize()
        assert int(gm._knn_flags[0].item()) == 7  # all three searches
        with pytest.raises(RuntimeError, match="KNN_K = 20"):
            gm.check_knn_k()
        assert int(gm._knn_flags[0].item()) == 0  # cleared by the check
    finally:
        rasterizer.set_host_sync(True)
        gm.disarm_knn_watch()
