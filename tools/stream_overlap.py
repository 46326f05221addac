"""Do kernels on two HIP streams of this stack run concurrently?  (The question behind `overlap_collective`, DESIGN 5 /
6.10.)  A: one long single-thread kernel (mppi_kmppi_trajectory with a huge inner dimension: one workgroup, ~ms);
B: the Philox generator fill of C3's draw (every CU, ~35 us) x N.  Timed alone, then A on a side stream with B on
the current stream.  Overlap <=> t(both) ~ max(tA, tB)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
import pytorch_mppi_amd as pm
from pytorch_mppi_amd import _native as N

lib = N.lib()
dev = torch.device("cuda", 0)
ctrl, x0, _ = bench.make_controller(pm, "c3", dev, "philox", None, 65536)
ctrl.command(x0)
p = ctrl._last
z = torch.empty(ctrl._zelems(ctrl.T), device=dev)
S = 300000
W = torch.full((S,), 1e-6, device=dev)
th = torch.ones(S, device=dev)
out = torch.empty(1, device=dev)


def raw(stream):
    return C.c_void_p(stream.cuda_stream)


def A(stream):
    N.check(lib.mppi_kmppi_trajectory(0, 1, S, 1, W.data_ptr(), th.data_ptr(), out.data_ptr(), raw(stream)), "A")


def B(stream, n=60):
    for _ in range(n):
        N.check(lib.mppi_noise_fill_philox(C.byref(p), z.data_ptr(), raw(stream)), "B")


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


main = torch.cuda.current_stream()
A(main); B(main)
tA = timed(lambda: A(main))
tB = timed(lambda: B(main))
print(f"A alone {tA:.3f} ms | B alone {tB:.3f} ms | serial on one stream {timed(lambda: (A(main), B(main))):.3f} ms")
for name, side in (("torch pool stream", torch.cuda.Stream()), ("second pool stream", torch.cuda.Stream()),
                   ("high-priority stream", torch.cuda.Stream(priority=-1))):
    t = timed(lambda: (A(side), B(main)))
    print(f"A on a {name:22s} + B on the current stream: {t:.3f} ms  ->  {'OVERLAP' if t < 0.85 * (tA + tB) else 'serialised'}")
