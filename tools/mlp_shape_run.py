"""N commands of the dense-MLP workload at one (nx, nu, hidden) -- K = 65536, T = 64, rng = philox, the split-operand matrix-core K1 --
for a rocprofv3 --pmc / --kernel-trace pass over a shape other than C4's (bench.py --workload c4 is (16, 4, 256)):
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python tools/mlp_shape_run.py 12 6 128 [n=12]
matrix-pipe busy share = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), as profiles/pmc_c4_mfma.json computes it."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import pytorch_mppi_amd as pm  # noqa: E402
from pytorch_mppi_amd import _native as N  # noqa: E402

nx, nu, H = (int(v) for v in sys.argv[1:4])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 12
K, T = 65536, 64
torch.manual_seed(0)
model = pm.models.MLPResidual.random(nx, nu, H, seed=2)
c = pm.MPPI(model.dynamics, model.running_cost, nx, torch.eye(nu), num_samples=K, horizon=T, device="cuda", lambda_=1.0,
            U_init=torch.randn(T, nu) * 0.02, rng="philox", seed=1234, auto_jit=False)
x = torch.randn(nx).cuda()
c.command(x)
c.lambda_ = float(c.cost_total.float().std())
n0 = int(N.lib().mppi_stat_mlp_split_launches())
for _ in range(n):
    c.command(x)
torch.cuda.synchronize()
print(f"({nx},{nu},{H}): {n} commands, split-operand launches {int(N.lib().mppi_stat_mlp_split_launches()) - n0}")
