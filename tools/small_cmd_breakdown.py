"""Dev helper (GPU): device time of the single-launch command of a small problem (C2: pendulum 8192 x 32)
against K1 alone on the same problem -- i.e. what the appended K3 / combine / K4 phases cost."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import bench
import pytorch_mppi_amd as pm
from pytorch_mppi_amd import _native as N

lib = N.lib()
ctrl, x0, _ = bench.make_controller(pm, "c2", torch.device("cuda"), "philox", None, 8192)
for _ in range(20):
    ctrl.command(x0)
p = ctrl._last
st = ctrl._stream()


def dev_us(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    lib.mppi_profile_enable(1 << 30)
    for _ in range(n):
        fn()
    a, b, cn, ce = C.c_double(0), C.c_double(0), C.c_int64(0), C.c_int64(0)
    N.check(lib.mppi_profile_read2(C.byref(a), C.byref(b), C.byref(cn), C.byref(ce)), "read2")
    lib.mppi_profile_enable(0)
    return b.value / max(1, cn.value) * 1e3


n0 = lib.mppi_stat_single_launch_commands()
t_cmd = dev_us(lambda: N.check(lib.mppi_command(C.byref(p), 1, st), "cmd"))
n1 = lib.mppi_stat_single_launch_commands()
p.noise_src = N.NOISE_PHILOX                       # K1 generating + storing its rows, as inside the command
t_k1 = dev_us(lambda: N.check(lib.mppi_rollout_cost(C.byref(p), st), "k1"))
p.noise_src = N.NOISE_TNK4
t_k1_read = dev_us(lambda: N.check(lib.mppi_rollout_cost(C.byref(p), st), "k1"))
print(f"C2 single-launch command {t_cmd:.2f} us (single-launch form used: {n1 - n0 > 0}) | K1 alone, generating its rows {t_k1:.2f} us | "
      f"K1 alone, reading stored rows {t_k1_read:.2f} us")
