"""Developer probe (GPU): how much of the blend backward's walk lies BEHIND a pixel's last dynamic (fluid) contributor?
The positions-only backward needs nothing from entries behind it: static splats get no gradient, and a dynamic entry's
gradient uses what lies behind only through total - prefix (stored by the forward)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_full_size_gpu import _Scene, _run, SIZE  # noqa: E402

sc = _Scene(sys.argv[1] if len(sys.argv) > 1 else "smoke_ch3")
n_dyn = {"smoke_ch3": 200_000, "ball_ch3": 350_000}[sc.name]
for view in (0, 2):
    h = _run(sc, sc.cams[view], np.array([0.1, 0.2, 0.3], np.float32))
    it = h.intermediates()
    pl, rg, nc = it["point_list"].astype(np.int64), it["ranges"].astype(np.int64).reshape(-1, 2), it["n_contrib"].astype(np.int64).reshape(SIZE, SIZE)
    gx = SIZE // 16
    tot = tail = tot_b = tail_b = 0
    batches = batches_dyn = 0
    for t in range(gx * gx):
        r0, r1 = rg[t]
        if r1 <= r0:
            continue
        ids = pl[r0:r1]
        dyn_pos = np.flatnonzero(ids < n_dyn)  # list positions of dynamic entries
        ty, tx = divmod(t, gx)
        n = nc[ty * 16:(ty + 1) * 16, tx * 16:(tx + 1) * 16]  # 1-based position of the last contributor, 0 = none
        # last dynamic entry in front of (or at) the pixel's last contributor
        k = np.searchsorted(dyn_pos, n.ravel(), side="left")  # number of dynamic entries with position < n
        dp = np.concatenate([dyn_pos, [0]])  # (a tile without dynamic entries: k = 0 everywhere)
        last_dyn = np.where(k > 0, dp[np.maximum(k - 1, 0)] + 1, 0)
        tot += n.sum()
        tail += (n.ravel() - last_dyn).sum()
        # at 4x4-block granularity (what the kernel walks: every pixel of a block walks to the block's max)
        nb = n.reshape(4, 4, 4, 4).transpose(0, 2, 1, 3).reshape(16, 16).max(1)
        lb = last_dyn.reshape(4, 4, 4, 4).transpose(0, 2, 1, 3).reshape(16, 16).max(1)
        tot_b += nb.sum() * 16
        tail_b += (nb - lb).sum() * 16
        batches += (n.max() + 255) // 256
        batches_dyn += (last_dyn.max() + 255) // 256
    print(f"{sc.name} view {view}: pixel-entries walked {tot}, behind the last dynamic contributor {tail} ({100 * tail / tot:.1f} %); "
          f"per 4x4 block {tot_b} / {tail_b} ({100 * tail_b / tot_b:.1f} %); batches {batches} -> {batches_dyn}")
