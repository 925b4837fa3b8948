"""bench.py's multi-rank code path on ONE device: no multi-GPU node is available to the builder, so
the N-rank launch sequence -- self-spawn under torch.distributed.run, view sharding, all-reduce, max over the ranks, one
JSON line from rank 0 -- is driven with every rank on cuda:0 (FNX_SINGLE_DEVICE=1) through gloo, and the RCCL path with
the all-reduce inside the hipGraph with one rank (FNX_FORCE_DIST=1).  The numbers mean nothing; the records' shape does."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(env, *args, timeout=600):
    e = dict(os.environ, **env)
    e.setdefault("MASTER_ADDR", "127.0.0.1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--repeats", "1",
                        "--no-cpu-baseline", "--frames", "0", "--no-drop-in", "--no-exact-leg", *args],
                       capture_output=True, text=True, timeout=timeout, env=e, cwd=ROOT)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, (p.returncode, p.stdout[-2000:], p.stderr[-3000:])
    return json.loads(lines[0])


def test_eight_ranks_config5_prints_one_record():
    d = _bench({"FNX_SINGLE_DEVICE": "1", "FNX_DIST_BACKEND": "gloo"}, "--gpus", "8", "--config", "5")
    assert d["n_gpus"] == 8 and d["ranks"] == 8 and d["config"]["baseline_config"] == 5
    assert d["config"]["views_this_rank"] == 1 and d["config"]["global_views_per_step"] == 8
    assert d["value"] > 0 and d["scaling"] == "strong"
    assert "eager" in d["config"]["all_reduce"]  # gloo cannot be captured: the self-test says no, the eager collective runs


def test_four_ranks_config4_shards_five_views():
    d = _bench({"FNX_SINGLE_DEVICE": "1", "FNX_DIST_BACKEND": "gloo"}, "--gpus", "4")
    assert d["ranks"] == 4 and d["config"]["baseline_config"] == 4 and d["config"]["views_this_rank"] == 2


def test_one_rank_through_rccl_takes_the_all_reduce_into_the_graph():
    d = _bench({"FNX_FORCE_DIST": "1", "MASTER_PORT": "29533"})
    assert d["ranks"] == 1 and d["config"]["all_reduce"].startswith("inside the hipGraph"), d["config"]["all_reduce"]
    d0 = _bench({"FNX_FORCE_DIST": "1", "FNX_GRAPH_ALLREDUCE": "0", "MASTER_PORT": "29534"})
    assert d0["config"]["all_reduce"].startswith("eager")


def test_a_failed_capture_of_the_collective_falls_back_to_the_eager_one():
    """The capture with the all-reduce inside raises (injected, HotLoop.capture): the run goes on with the local phase
    replayed and the collective issued eagerly, and still prints its one record."""
    d = _bench({"FNX_FORCE_DIST": "1", "FNX_TEST_GRAPH_AR_FAIL": "raise", "MASTER_PORT": "29535"})
    assert d["ranks"] == 1 and d["config"]["all_reduce"].startswith("eager"), d["config"]["all_reduce"]
    assert d["value"] > 0 and d["config"]["launch"].startswith("hipGraph")
