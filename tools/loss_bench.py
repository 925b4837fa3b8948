"""Developer tool: image-loss kernels alone (5 views, 3x512x512, grey and RGB), HIP-event timed."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidnexus_amd.losses import image_loss_value_and_grad  # noqa: E402

torch.manual_seed(0)
img = torch.rand(5, 3, 512, 512, device="cuda")
gt = torch.rand(5, 3, 512, 512, device="cuda")
for grey in (True, False):
    for _ in range(3):
        image_loss_value_and_grad(img, gt, 0.2, 1.0, grey=grey)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        image_loss_value_and_grad(img, gt, 0.2, 1.0, grey=grey)
    e1.record()
    e1.synchronize()
    print(f"grey={grey}: forward + backward {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per call (includes 2 torch.empty)")
