"""Dev helper (GPU): what would the command -> action-on-the-host round trip cost if the update kernel wrote the action straight into
pinned host memory and the stream wrote a sequence number behind it (hipStreamWriteValue32), the host spinning on that word --
against command() + torch.cuda.synchronize() (the reference benchmark's protocol) and command().cpu()?"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import pytorch_mppi_amd as pm

hip = C.CDLL("libamdhip64.so")
hip.hipStreamWriteValue32.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint]
hip.hipStreamWriteValue32.restype = C.c_int
for name, (K, T, nx, nu, kind) in (("C2 pendulum 8192x32", (8192, 32, 2, 1, "pendulum")), ("C3 quadtoy 65536x64", (65536, 64, 16, 12, "integrator")),
                                   ("integrator 1024x20", (1024, 20, 8, 4, "integrator"))):
    m = pm.models.Integrator(nx, nu) if kind == "integrator" else pm.models.Pendulum()
    sigma = torch.eye(nu) if nu > 1 else torch.tensor(1.0)
    c = pm.MPPI(m.dynamics, m.running_cost, nx, sigma, num_samples=K, horizon=T, device="cuda", lambda_=50.0, rng="philox", seed=3)
    x = torch.randn(nx, device="cuda")
    host = torch.zeros(64, dtype=torch.float32).pin_memory()
    flag = torch.zeros(16, dtype=torch.int32).pin_memory()
    fword = C.c_uint32.from_address(flag.data_ptr())
    orig = c._problem
    def patched(Tn=None, U=None, _o=orig):
        p = _o(Tn, U); p.action_out = host.data_ptr(); return p
    for _ in range(20): c.command(x)
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for _ in range(n):
        c.command(x); torch.cuda.synchronize()
    a = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n):
        c.command(x).cpu()
    b = (time.perf_counter() - t0) / n
    c._problem = patched
    seq = 0
    for _ in range(20): c.command(x)
    torch.cuda.synchronize()
    st = torch.cuda.current_stream().cuda_stream
    rc = 0
    t0 = time.perf_counter()
    for _ in range(n):
        dev_action = c.command(x)
        seq += 1
        rc |= hip.hipStreamWriteValue32(st, flag.data_ptr(), seq, 0)
        while fword.value != seq:
            pass
    d = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    ok = torch.allclose(host[:nu], dev_action.reshape(-1)[:nu].cpu())
    print(f"{name}: command + synchronize {a * 1e6:6.1f} us | command().cpu() {b * 1e6:6.1f} us | action in pinned memory + stream word {d * 1e6:6.1f} us (rc {rc}, same action {ok})", flush=True)
    del c
