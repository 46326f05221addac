"""Fuzz of the matrix-core execution of traced networks (csrc/mlp_wide.hpp): random nn.Sequential dynamics (1-3 hidden layers, widths
8..40, 10 activations, some layers without bias, outputs wider than the state, a smooth cost) traced in fp32,
    python tools/wide_fuzz.py build [n]     here: compiles the functors (hipcc cross-compiles; the objects travel with the snapshot)
    python tools/wide_fuzz.py run [n]       on the GPU box: wide kernel vs the callback loop on the same injected noise
Smooth dynamics on purpose (no wrap, no clamp inside): every sample has to agree, not all but a few."""
import os, random, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, torch.nn as nn

ACTS = [nn.Tanh, nn.ReLU, nn.GELU, nn.ELU, nn.SiLU, nn.Sigmoid, nn.Softplus, lambda: nn.LeakyReLU(0.2), nn.Hardtanh, nn.Identity]


def program(seed):
    rng = random.Random(seed); torch.manual_seed(1000 + seed)
    nx, nu = rng.choice([(2, 1), (3, 2), (4, 2), (6, 3), (8, 4)])
    widths = [nx + nu] + [rng.choice([8, 12, 16, 24, 32, 33, 40]) for _ in range(rng.randint(1, 3))] + [rng.choice([nx, nx, nx + 5])]
    layers = []
    for i in range(len(widths) - 1):
        layers.append(nn.Linear(widths[i], widths[i + 1], bias=rng.random() < 0.8))
        if i < len(widths) - 2:
            layers.append(rng.choice(ACTS)())
    net = nn.Sequential(*layers).float()
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(0.6)
    f = lambda s, a: 0.9 * s + 0.2 * net(torch.cat((s, a), 1))[:, :nx]
    q = lambda s, a: (s ** 2).sum(1) + 0.1 * (a ** 2).sum(1)
    return f, q, net, nx, nu, widths


def main():
    mode = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    if mode == "build":
        import concurrent.futures as cf
        from pytorch_mppi_amd import jit
        with cf.ThreadPoolExecutor(max_workers=8) as ex:
            futs = {}
            for seed in range(n):
                f, q, net, nx, nu, widths = program(seed)
                futs[seed] = (widths, ex.submit(jit.from_torch, f, q, nx, nu))
            for seed, (widths, fu) in futs.items():
                m = fu.result()
                print(f"seed {seed}: widths {widths} wide={m.wide} dense layers {len(m._code.get('dense') or [])}", flush=True)
        return
    import pytorch_mppi_amd as pm
    worst = 0.0
    for seed in range(n):
        f, q, net, nx, nu, widths = program(seed)
        net.cuda()
        K, T = [(1000, 12), (4096, 20), (333, 7)][seed % 3]
        sigma = torch.eye(nu) * 0.5 if nu > 1 else torch.tensor(0.5)
        mk = lambda auto: pm.MPPI(f, q, nx, sigma, num_samples=K, horizon=T, device="cuda", U_init=torch.zeros(T, nu), lambda_=5.0, auto_jit=auto)
        a, b = mk("sync"), mk(False)
        assert a._model is not None and not a._needs_generic(), a.jit_note
        x0 = torch.linspace(-0.5, 0.5, nx).cuda()
        gen = torch.Generator().manual_seed(seed)
        errs = []
        for rnd in range(2):
            z = torch.randn(K, T, nu, generator=gen)
            for c in (a, b):
                c.U = torch.zeros(T, nu, device="cuda") if rnd == 0 else b.U.clone()
                c.inject_noise(z)
            ua, ub = a.command(x0), b.command(x0)
            sc = max(1.0, float(b.cost_total.abs().max()))
            errs.append((float((a.cost_total - b.cost_total).abs().max()) / sc, float((ua - ub).abs().max())))
            if rnd == 0:
                with torch.no_grad():
                    for p in net.parameters():
                        p.add_(torch.randn_like(p) * 0.05)        # run-time parameters: no recompilation
        e = max(max(x) for x in errs)
        worst = max(worst, e)
        print(f"seed {seed}: widths {widths} nx {nx} nu {nu} K {K} T {T} wide={a._model.wide}: cost / action error {errs}  {'ok' if e <= 2e-4 else 'FAIL'}", flush=True)
    print("worst", worst)


if __name__ == "__main__":
    main()
