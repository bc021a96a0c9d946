"""Run by tests/test_gpu_edge_cases.py::test_static_deal_fallback in a child process with L3K_FAST_STATIC=1 (the switch
is read once per process): the single-wave kernel with the static deal of batches to the persistent waves, against the
oracle."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, HERE)
assert os.environ.get("L3K_FAST_STATIC") == "1"
os.environ["L3K_GENERIC_BELOW"] = "0"
import numpy as np  # noqa: E402
import torch  # noqa: E402

import helpers  # noqa: E402
import oracle_lib as O  # noqa: E402
from l3ster_amd import system  # noqa: E402

ctx = system.Context(0, torch.cuda.current_stream().cuda_stream)
assert ctx.get_tuning()["static_deal"] == 1 and ctx.get_tuning()["generic_below"] == 0  # from the environment, at creation
worst = 0.0
for p, ne in ((6, (5, 4, 3)), (4, (7, 5, 3)), (2, (9, 4, 4))):
    U = 4
    part = system.CubePartition(ne, p, perturb=0.1)
    mask = part.dirichlet_mask(U)
    mf = system.MatrixFreeSystem(system.DeviceMesh(ctx, part, U, mask), system.KERNEL_DIFFUSION3D, [0.7, 1.0])
    assert "static batches" in mf.route(), mf.route()
    x = part.synthetic_vector(U)
    y0 = part.synthetic_vector(U, seed=3)
    Y = torch.as_tensor(y0.copy(), device="cuda")
    mf.apply(torch.as_tensor(x, device="cuda"), Y, 1.25, -0.5)
    want = O.mf_apply(helpers.oracle_mesh(part, p + 1, U, np.arange(U), mask), O.KERNEL_DIFFUSION3D, x.T,
                      np.asfortranarray(y0.T.copy()), alpha=1.25, beta=-0.5, kparams=[0.7, 1.0])
    worst = max(worst, helpers.rel_err(Y.cpu().numpy().T, want))
print(f"static deal: worst relative error {worst:.2e}")
print("static deal", "ok" if worst < 1e-12 else "FAILED")
