"""From which K on does the on-chip command (two waves per sample group: its launch costs the same ~60 us for any K <= 65536) beat the
streaming form (generator launch -> K1 -> K3 -> K4, which scales with K)?  C3's shape, rng="philox"; ms per command, pipelined."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, pytorch_mppi_amd as pm
m = pm.models.Integrator(16, 12)
x = torch.zeros(16).cuda()


def ms(K, onchip, n=200):
    c = pm.MPPI(m.dynamics, m.running_cost, 16, torch.eye(12) * 0.5, num_samples=K, horizon=64, device="cuda", lambda_=3000.0, rng="philox", seed=3,
                U_init=torch.zeros(64, 12))
    c.philox_onchip = onchip
    for _ in range(20):
        c.command(x)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            c.command(x)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n * 1e3)
    return best, c.last_draw


print("# K        on chip (forced)      streaming (forced)     the rule's choice")
for K in (16384, 24576, 28672, 32768, 36864, 40960, 49152, 65536, 98304, 131072):
    a, da = ms(K, True)
    b, db = ms(K, False)
    c, dc = ms(K, None)
    print(f"  {K:7d}   {a:.4f} ms ({da})   {b:.4f} ms ({db})   {c:.4f} ms ({dc})", flush=True)
