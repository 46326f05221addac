"""GPU: edge semantics of the drop-in boundary that the main suites do not touch -- initial-state
shapes on the fused path and under sharding, HIP-graph capture of the controller family, the
sharded torch-generator modes."""
import pytest
import torch

import pytorch_mppi_amd as pm
import gpu_util

pytestmark = pytest.mark.gpu


def _lin():
    return pm.models.LinearGoal(torch.tensor([[1.0, 0.0], [0.0, -1.0]]), torch.tensor([2.0, 2.0]))


def test_capture_command_is_refused_for_smppi_and_kmppi():
    """SMPPI / KMPPI re-bind `action_sequence` / `theta` per command; a captured graph would replay
    stale pointers, so capture is plain-MPPI only."""
    lin = _lin()
    x = torch.tensor([-3.0, -2.0]).cuda()
    for cls, extra in ((pm.SMPPI, dict(delta_t=0.5)), (pm.KMPPI, dict(num_support_pts=4))):
        c = cls(lin.dynamics, lin.running_cost, 2, torch.eye(2), num_samples=256, horizon=8, device="cuda",
                lambda_=5.0, rng="torch", **extra)
        with pytest.raises(NotImplementedError):
            c.capture_command(x)
        c.command(x)           # the controller itself keeps working


def test_graph_replay_hides_the_lazy_attributes():
    m = pm.models.Integrator(6, 4)
    c = pm.MPPI(m.dynamics, m.running_cost, 6, torch.eye(4), num_samples=2048, horizon=16, device="cuda",
                lambda_=20.0, rng="torch")
    x = torch.ones(6, device="cuda")
    g = c.capture_command(x)
    a = g(x).clone()
    assert torch.isfinite(a).all() and c.cost_total is not None and abs(float(c.omega.sum()) - 1) < 1e-5
    assert c.noise is None and c.perturbed_action is None      # not derivable after the replay overwrote U


def test_fused_path_refuses_states_it_cannot_hold():
    m = pm.models.Integrator(6, 4)
    c = pm.MPPI(m.dynamics, m.running_cost, 6, torch.eye(4), num_samples=128, horizon=5, device="cuda", lambda_=5.0)
    c.command(torch.ones(1, 6))                                 # (1,nx): the reference's view(1,-1) case
    with pytest.raises(ValueError):
        c.command(torch.ones(3, 6))                             # neither (nx,) nor (K,nx)


def test_per_sample_states_of_the_global_problem_are_sliced_per_shard():
    """mppi.py:302 under sharding: a (K_global, nx) state selects per-sample initial states; each
    shard must take ITS rows (not row 0 for everybody)."""
    K, T, nx, nu, world = 1024, 9, 6, 4, 4
    g = torch.Generator().manual_seed(2)
    m = pm.models.Integrator(nx, nu)
    U0 = torch.randn(T, nu, generator=g) * 0.05
    states = torch.randn(K, nx, generator=g).cuda()
    z = torch.randn(K, T, nu, generator=g)
    kw = dict(num_samples=K, horizon=T, device="cuda", lambda_=15.0, U_init=U0.clone(), rng="torch")
    full = pm.MPPI(m.dynamics, m.running_cost, nx, torch.eye(nu), **kw)
    full.inject_noise(z)
    a_full = full.command(states)
    ctrls = [pm.MPPI(m.dynamics, m.running_cost, nx, torch.eye(nu), shard=(r, world), **kw) for r in range(world)]
    ps = []
    for c in ctrls:
        c.inject_noise(z)
        ps.append(c._begin(states, True))
        assert int(ps[-1].state_per_sample) == 1
    records = torch.stack([p._keep["record"] for p in ps])
    for c, p in zip(ctrls, ps):
        c._combine(p, records)
    acts = [c._end(p) for c, p in zip(ctrls, ps)]
    for c, a in zip(ctrls, acts):
        lo, hi = c._shard.bounds(c._shard.rank)
        assert torch.allclose(c.cost_total, full.cost_total[lo:hi], rtol=1e-5, atol=1e-4)
        assert torch.allclose(a, a_full, rtol=1e-5, atol=1e-6)


def test_sharded_torch_modes_draw_rank_distinct_samples_without_a_process_group():
    """rng='torch' under `shard=`: each shard's sample draws come from its own (seed, rank) generator
    (identically seeded ranks must not duplicate the perturbations)."""
    m = pm.models.Integrator(6, 4)
    cs = [pm.MPPI(m.dynamics, m.running_cost, 6, torch.eye(4), num_samples=512, horizon=8, device="cuda", lambda_=10.0,
                  U_init=torch.zeros(8, 4), rng="torch", seed=7, shard=(r, 2)) for r in range(2)]
    x = torch.ones(6, device="cuda")
    torch.manual_seed(0)
    ps = [c._begin(x, True) for c in cs]
    assert all(c.last_draw == "torch-rows" for c in cs)           # the shard's generator, the engine's launch
    draw = lambda c, p: gpu_util.consumed_normals(c, p)
    assert not torch.equal(draw(cs[0], ps[0]), draw(cs[1], ps[1]))
    # and the draw is reproducible for a given (seed, rank)
    c0 = pm.MPPI(m.dynamics, m.running_cost, 6, torch.eye(4), num_samples=512, horizon=8, device="cuda", lambda_=10.0,
                 U_init=torch.zeros(8, 4), rng="torch", seed=7, shard=(0, 2))
    p0 = c0._begin(x, True)
    assert torch.equal(draw(c0, p0), draw(cs[0], ps[0]))
    # ... and it is the draw torch.randn makes from that generator
    g = torch.Generator(device="cuda")
    g.manual_seed(c0._shard_gen.initial_seed())
    assert torch.equal(draw(c0, p0), torch.randn(256, 8, 4, device="cuda", generator=g).cpu())


@pytest.mark.parametrize("path", ["fused", "generic"])
@pytest.mark.parametrize("rng", ["inject", "philox"])
def test_set_noise_takes_effect_like_a_controller_built_with_the_new_noise(path, rng):
    """SURVEY 8f-4 (reference autotune.py:140-189 rewrites noise_sigma / noise_mu between commands):
    after `set_noise(Sigma2, mu2)` a command equals the fp64 oracle built with Sigma2 / mu2 -- sampler
    factors, action-cost inverse and (Philox) generator-side colouring all refreshed -- diag -> full."""
    import numpy as np
    from oracle import mppi_oracle as orc, dynamics as dyn, philox as oph
    K, T, nx, nu = 1200, 20, 8, 4
    dt = torch.float64
    g = torch.Generator().manual_seed(11)
    m = pm.models.Integrator(nx, nu)
    f, q = (m.dynamics, m.running_cost) if path == "fused" else ((lambda s, a: m.dynamics(s, a)), (lambda s, a: m.running_cost(s, a)))
    f64, q64 = dyn.make_quadtoy(nx, nu)
    sig1 = torch.diag(torch.tensor([1.0, 0.5, 2.0, 1.5], dtype=dt))
    A = torch.randn(nu, nu, generator=g, dtype=dt) * 0.4
    sig2 = A @ A.T + 0.3 * torch.eye(nu, dtype=dt)
    mu2 = torch.tensor([0.1, -0.05, 0.0, 0.2], dtype=dt)
    U0 = torch.randn(T, nu, generator=g, dtype=dt) * 0.05
    x0 = torch.randn(nx, generator=g, dtype=dt)
    umax = torch.full((nu,), 1.5, dtype=dt)
    c = pm.MPPI(f, q, nx, sig1, num_samples=K, horizon=T, device="cuda", lambda_=8.0, U_init=U0.clone(), u_max=umax,
                rng="philox" if rng == "philox" else "torch", seed=5)
    assert (not c._needs_generic()) == (path == "fused")
    U = U0
    for call, (sig, mu) in enumerate(((sig1, torch.zeros(nu, dtype=dt)), (sig2, mu2), (sig1 * 0.5, mu2 * 0.0)), start=1):
        if call > 1:
            c.set_noise(sig, mu)
        if rng == "philox":
            z = torch.from_numpy(oph.normals_ktn(5, call, K, T, nu)).to(dt)
        else:
            z = torch.randn(K, T, nu, generator=g, dtype=dt)
            c.inject_noise(z)
        a = c.command(x0.cuda())
        p = orc.Problem(dynamics=f64, running_cost=q64, nx=nx, noise_sigma=sig, K=K, T=T, lambda_=8.0, noise_mu=mu, u_max=umax)
        r = orc.command(p, U, x0, z, True)
        tol = 1e-9 if rng == "inject" else 1e-5               # Philox: hardware log/sin/cos vs numpy in Box-Muller
        for got, key in ((a, "action"), (c.U, "U"), (c.cost_total, "cost_total"), (c.omega, "omega"),
                         (c.perturbed_action, "perturbed_action")):
            ref = r[key].numpy()
            err = float(np.abs(got.cpu().numpy() - ref).max())
            assert err <= tol * max(1.0, float(np.abs(ref).max())), (call, key, err)
        U = r["U"]
        c.U = U.cuda()                                        # same nominal sequence on both sides


@pytest.mark.parametrize("rng", ["philox", "torch-native", "inject"])
@pytest.mark.parametrize("cls", ["mppi", "kmppi"])
def test_padded_noise_row_pitch_changes_nothing(rng, cls):
    """K = 131072 fp32: a dense row-of-4 array would have 2 MiB rows, so the engine pads the row pitch
    by 1 MiB (mppi_noise_pitch).  Same commands as with the dense layout forced, through every producer
    and consumer of the rows (generator / torch draw / layout conversion, K1, K3, KMPPI interpolation,
    lazy noise materialisation)."""
    from pytorch_mppi_amd import _native as N
    K, T, nx, nu = 131072, 6, 6, 4
    assert N.noise_pitch(K, N.F32) == K + 65536
    m = pm.models.Integrator(nx, nu)
    g = torch.Generator().manual_seed(3)
    U0 = torch.randn(T, nu, generator=g) * 0.1
    x = torch.randn(nx, generator=g).cuda()
    S = 4
    zs = [torch.randn(K, S if cls == "kmppi" else T, nu, generator=g) for _ in range(2)]
    outs = []
    for dense in (False, True):
        torch.manual_seed(11)
        kw = dict(num_samples=K, horizon=T, device="cuda", lambda_=8.0, U_init=U0.clone(), u_max=torch.ones(nu),
                  rng="torch" if rng == "inject" else rng, seed=9)
        c = pm.KMPPI(m.dynamics, m.running_cost, nx, torch.eye(nu), num_support_pts=S, **kw) if cls == "kmppi" else \
            pm.MPPI(m.dynamics, m.running_cost, nx, torch.eye(nu), **kw)
        if rng == "inject":
            c.ktn_direct = False                     # force the (K,T,nu) -> rows conversion pass
        if dense:
            c._zpitch_cache = ((c.K_local, c.dtype), K)
        res = []
        for z in zs:
            if rng == "inject":
                c.inject_noise(z)
            a = c.command(x)
            assert int(c._last.noise_pitch) == (K if dense else K + 65536)
            res.append((a.clone(), c.cost_total.clone(), c.omega.clone(), c.noise.clone()))
        outs.append(res)
    for r_pad, r_dense in zip(*outs):
        for u, v in zip(r_pad, r_dense):
            if rng == "torch-native":                # the draw fills the padded array: other samples get other normals
                assert u.shape == v.shape and torch.isfinite(u).all()
            else:
                assert torch.equal(u, v)


@pytest.mark.parametrize("case", ["pendulum-philox", "pendulum-inject", "integrator-null-bounds", "smppi", "sharded", "f64"])
def test_single_launch_command_for_small_problems(case):
    """Small problems (K <= 16384, T*nu <= 256, diagonal Sigma) run the whole command -- rollout, weights,
    weighted sums, combine over workgroups by the last one to arrive, update -- as ONE launch.  Checked:
    the single-launch form is what ran, several consecutive commands (the arrival ticket must be back at
    zero each time) agree with the fp64 oracle, omega / cost_total_non_zero (derived on first read) too,
    and a problem just outside the envelope still takes the three-launch path with the same results."""
    import numpy as np
    from pytorch_mppi_amd import _native as N
    from oracle import mppi_oracle as orc, dynamics as dyn, philox as oph
    lib = N.lib()
    g = torch.Generator().manual_seed(13)
    dt = torch.float64 if case == "f64" else torch.float32
    if case.startswith("pendulum"):
        m, nx, nu, K, T = pm.models.Pendulum(), 2, 1, 8192, 32
        sigma = torch.tensor(10.0, dtype=dt)
        kw = dict(u_min=torch.tensor(-2.0, dtype=dt), u_max=torch.tensor(2.0, dtype=dt), lambda_=1.0)
        f64, q64 = dyn.pendulum_dynamics, dyn.pendulum_cost
        x0 = torch.tensor([3.141592653589793, 1.0], dtype=dt)
    else:
        nx, nu, K, T = 6, 4, 3000, 20                      # ragged K: the last workgroup is partly empty
        m = pm.models.Integrator(nx, nu)
        sigma = torch.diag(torch.tensor([1.0, 0.5, 2.0, 1.5], dtype=dt))
        kw = dict(lambda_=12.0, sample_null_action=True, u_max=torch.full((nu,), 1.2, dtype=dt), noise_mu=torch.tensor([0.1, 0.0, -0.1, 0.0], dtype=dt))
        f64, q64 = dyn.make_quadtoy(nx, nu)
        x0 = torch.randn(nx, generator=g, dtype=dt)
    U0 = torch.randn(T, nu, generator=g, dtype=dt) * 0.1
    rng = "torch" if case == "pendulum-inject" else "philox"
    cls, extra = (pm.SMPPI, dict(w_action_seq_cost=0.5, delta_t=0.5)) if case == "smppi" else (pm.MPPI, {})
    shard = (0, 1) if case == "sharded" else None
    c = cls(m.dynamics, m.running_cost, nx, sigma, num_samples=K, horizon=T, device="cuda", U_init=U0.clone(), rng=rng, seed=3,
            shard=shard, **kw, **extra)
    if case == "pendulum-inject":
        c.ktn_direct = False                               # rows through the layout conversion -> TNK4 in memory
    if case == "sharded":
        c._force_collective = True
    p64 = orc.Problem(dynamics=f64, running_cost=q64, nx=nx, noise_sigma=sigma.double(), K=K, T=T,
                      **{k: (v.double() if torch.is_tensor(v) else v) for k, v in kw.items()})
    n0 = lib.mppi_stat_single_launch_commands()
    U = U0.double()
    A = U0.double().clone()
    Ud = torch.zeros_like(U)
    for call in (1, 2, 3):
        import gpu_util
        # rng="philox": the draw of command `call` as the device generates it (gpu_util: bitwise what the kernel consumes)
        z = gpu_util.device_philox_normals(c, call).double() if rng == "philox" else torch.randn(K, T, nu, generator=g, dtype=torch.float64)
        if rng != "philox":
            c.inject_noise(z.to(dt))
        a = c.command(x0.cuda())
        if case == "smppi":
            r = orc.smppi_command(p64, Ud, A, x0.double(), z, torch.tensor(float("-inf")), torch.tensor(float("inf")), 0.5, 0.5, True)
            Ud, A = r["U"], r["action_sequence"]
        else:
            r = orc.command(p64, U, x0.double(), z, True)
            U = r["U"]
        tol = 1e-9 if dt == torch.float64 else 1e-5
        for got, key in ((a, "action"), (c.cost_total, "cost_total"), (c.omega, "omega")):
            ref = r[key].numpy()
            err = float(np.abs(got.cpu().numpy().astype(np.float64) - ref).max())
            assert err <= tol * max(1.0, float(np.abs(ref).max())), (case, call, key, err)
        assert abs(float(c.omega.double().sum()) - 1.0) < 1e-5
        if case != "smppi":
            c.U = U.to(dt).cuda()
        else:
            c.U, c.action_sequence = Ud.to(dt).cuda(), A.to(dt).cuda()
    ran_single = lib.mppi_stat_single_launch_commands() - n0
    assert ran_single == (0 if case == "sharded" else 3), ran_single       # sharded commands keep omega eager -> 3 launches


def test_host_resident_states_take_the_kernarg_upload_and_match_device_states():
    """numpy / CPU-tensor / list states (what a simulator returns, mppi.py:262-264) travel inside a launch packet
    (mppi_upload_small); same results as the device-tensor state, a different state every command; larger host
    states (per-sample) keep the ordinary copy."""
    g = torch.Generator().manual_seed(2)
    nx, nu, K, T = 6, 4, 512, 12
    m = pm.models.Integrator(nx, nu)
    mk = lambda: pm.MPPI(m.dynamics, m.running_cost, nx, torch.eye(nu), num_samples=K, horizon=T, device="cuda",
                         rng="philox", seed=4, U_init=torch.zeros(T, nu))
    a, b, c = mk(), mk(), mk()
    for i in range(11):
        x = torch.randn(nx, generator=g)
        ua = a.command(x.cuda())
        ub = b.command(x.numpy() if i % 2 else x)           # numpy array / CPU tensor
        uc = c.command([float(v) for v in x])               # python list
        assert torch.equal(ua, ub) and torch.equal(ua, uc), i
        assert b.state.is_cuda and torch.equal(b.state.cpu(), x)
    # per-sample initial states from the host as well
    xs = torch.randn(K, nx, generator=g)
    assert torch.equal(a.command(xs.cuda()), b.command(xs.numpy()))
    x64 = torch.randn(nx, generator=g, dtype=torch.float64)          # a dtype the controller has to cast
    assert torch.equal(a.command(x64.float().cuda()), b.command(x64.numpy()))


def test_pendulum_with_an_angle_far_outside_the_usual_range():
    """The fused pendulum prunes sinf's huge-argument path from its hot loop and reduces such angles modulo 2 pi in
    fp64 instead (csrc/common.hpp m_sin_moderate); an initial angle of tens of thousands of radians must still roll
    out like the torch callables do (same fp32 states, sin to fp32 rounding)."""
    m = pm.models.Pendulum()
    K, T = 512, 20
    z = torch.randn(K, T, 1, generator=torch.Generator().manual_seed(4))
    kw = dict(num_samples=K, horizon=T, device="cuda", lambda_=1.0, u_min=torch.tensor(-2.0), u_max=torch.tensor(2.0),
              U_init=torch.zeros(T, 1))
    for th0 in (9000.0, 3.0e5, -2.5e6):
        x0 = torch.tensor([th0, 0.5]).cuda()
        fused = pm.MPPI(m.dynamics, m.running_cost, 2, torch.tensor(4.0), **kw)
        generic = pm.MPPI(lambda s, a: m.dynamics(s, a), lambda s, a: m.running_cost(s, a), 2, torch.tensor(4.0), **kw)
        assert not fused._needs_generic() and generic._needs_generic()
        outs = []
        for c in (fused, generic):
            c.inject_noise(z)
            c.command(x0)
            outs.append(c.cost_total)
        scale = float(outs[1].abs().max())
        assert torch.isfinite(outs[0]).all()
        assert float((outs[0] - outs[1]).abs().max()) <= 2e-5 * scale, th0


def test_two_problem_shapes_share_one_workspace_on_the_single_launch_path():
    """the arrival ticket of the single-launch command used to sit at a shape-dependent offset of the
    workspace, so a small problem's ticket lay inside a larger problem's scratch -- after one command of the larger shape
    no workgroup of the smaller one was ever elected last and its outputs were silently never written.  Now the ticket is
    the last 4 elements of the caller's buffer: two controllers of different shapes alternate on ONE zero-filled workspace
    (same (workspace, workspace_elems) pair for both) and must reproduce, bit for bit, what they compute alone."""
    from pytorch_mppi_amd import _native as N
    lib = N.lib()

    def make(kind):
        if kind == "pendulum":
            m = pm.models.Pendulum()
            return pm.MPPI(m.dynamics, m.running_cost, 2, torch.tensor(10.0), num_samples=8192, horizon=32, device="cuda", lambda_=1.0,
                           u_min=torch.tensor(-2.0), u_max=torch.tensor(2.0), U_init=torch.zeros(32, 1), rng="philox", seed=5), \
                torch.tensor([3.0, 1.0]).cuda()
        m = pm.models.Integrator(6, 4)
        return pm.MPPI(m.dynamics, m.running_cost, 6, torch.eye(4), num_samples=3000, horizon=20, device="cuda", lambda_=9.0,
                       U_init=torch.zeros(20, 4), rng="philox", seed=6, sample_null_action=True), torch.linspace(-1, 1, 6).cuda()

    alone = {}
    for kind in ("pendulum", "integrator"):
        c, x = make(kind)
        alone[kind] = [c.command(x).clone() for _ in range(6)]
        need = c._ws.numel()
        alone[kind + "_need"] = need
    shared = torch.zeros(max(alone["pendulum_need"], alone["integrator_need"]) + 64, device="cuda")
    ctrls = {k: make(k) for k in ("pendulum", "integrator")}
    for c, _ in ctrls.values():
        c._ws = shared                                    # one buffer, one (pointer, size) pair for both shapes
    n0 = lib.mppi_stat_single_launch_commands()
    for i in range(6):
        for kind in (("pendulum", "integrator") if i % 2 == 0 else ("integrator", "pendulum")):
            c, x = ctrls[kind]
            a = c.command(x)
            assert c._ws is shared
            assert torch.equal(a, alone[kind][i]), (kind, i)
    assert lib.mppi_stat_single_launch_commands() == n0 + 12, "both shapes must have run as ONE launch each time"
    torch.cuda.synchronize()
    assert int(shared[-4:].view(torch.int32)[0]) == 0, "the ticket is left at zero"


def test_soak_single_launch_commands_on_two_streams_under_load():
    """VERDICT r02 'weak' 8: the single-launch command hands partial records between workgroups with write-through
    (sc1) stores, a drained vmcnt and an arrival ticket, and reads them back with sc1 loads (the R1 form of the guide's
    inter-workgroup recipe) -- a protocol whose failure mode is a RARE stale read under UNEVEN load.  10 000 commands of
    each of two small controllers, issued from two host threads on two streams while a third stream keeps every XCD busy
    with C3-sized commands; every action must equal, bit for bit, the action the same controller computes alone."""
    import threading
    n = 10000

    def make(seed, kind):
        if kind == 0:
            m = pm.models.Pendulum()
            return pm.MPPI(m.dynamics, m.running_cost, 2, torch.tensor(10.0), num_samples=8192, horizon=32, device="cuda", lambda_=1.0,
                           u_min=torch.tensor(-2.0), u_max=torch.tensor(2.0), U_init=torch.zeros(32, 1), rng="philox", seed=seed), \
                torch.tensor([3.0, 1.0]).cuda()
        m = pm.models.Integrator(6, 4)
        return pm.MPPI(m.dynamics, m.running_cost, 6, torch.eye(4), num_samples=4096 + 37, horizon=24, device="cuda", lambda_=9.0,
                       U_init=torch.zeros(24, 4), rng="philox", seed=seed, u_max=torch.ones(4)), torch.linspace(-1, 1, 6).cuda()

    def run(c, x, out, stream=None):
        ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
        with ctx:
            for i in range(n):
                out[i].copy_(c.command(x).reshape(-1))

    from pytorch_mppi_amd import _native as N
    lib = N.lib()
    ref = []
    for kind in (0, 1):
        c, x = make(100 + kind, kind)
        out = torch.empty(n, c.nu, device="cuda")
        run(c, x, out)
        ref.append(out)
    torch.cuda.synchronize()
    n0 = lib.mppi_stat_single_launch_commands()
    outs, threads, stop = [], [], threading.Event()
    for kind in (0, 1):
        c, x = make(100 + kind, kind)
        out = torch.empty(n, c.nu, device="cuda")
        outs.append(out)
        threads.append(threading.Thread(target=run, args=(c, x, out, torch.cuda.Stream())))

    def load():                                            # uneven background load: C3-sized commands on a third stream
        m = pm.models.Integrator(16, 12)
        big = pm.MPPI(m.dynamics, m.running_cost, 16, torch.eye(12), num_samples=65536, horizon=64, device="cuda", lambda_=50.0,
                      rng="philox", seed=9)
        xb = torch.zeros(16, device="cuda")
        with torch.cuda.stream(torch.cuda.Stream()):
            while not stop.is_set():
                for _ in range(20):
                    big.command(xb)
                torch.cuda.current_stream().synchronize()

    bg = threading.Thread(target=load)
    bg.start()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    stop.set()
    bg.join()
    torch.cuda.synchronize()
    assert lib.mppi_stat_single_launch_commands() >= n0 + 2 * n
    for kind in (0, 1):
        bad = (outs[kind] != ref[kind]).any(dim=1).nonzero()
        assert bad.numel() == 0, (kind, "first differing command", int(bad[0]) if bad.numel() else None, "of", int(bad.numel()))


def test_mlp_weights_outside_the_fp16_operand_range_take_the_exact_kernel():
    """ADVICE r02 / VERDICT r02 item 5: |W2| >= 3e4 used to raise in MLPResidual.__init__; now the problem block carries
    MPPI_MODEL_FLAG_EXACT_FP32 and the exact fp32 MFMA kernel runs -- also when the weights change later (invalidate())."""
    import pytorch_mppi_amd as pm
    from pytorch_mppi_amd import _native as N
    nx, nu, H, K, T = 16, 4, 256, 4096, 12
    m = pm.models.MLPResidual.random(nx, nu, H, seed=3, res_scale=1e-6)
    z = torch.randn(K, T, nu, generator=torch.Generator().manual_seed(1))
    x0 = torch.linspace(-1, 1, nx).cuda()

    def run(model, generic):
        f, q = (model.dynamics, model.running_cost) if not generic else (lambda s, a: model.dynamics(s, a), lambda s, a: model.running_cost(s, a))
        c = pm.MPPI(f, q, nx, torch.eye(nu), num_samples=K, horizon=T, device="cuda", lambda_=5.0, U_init=torch.zeros(T, nu))
        c.inject_noise(z)
        return c, c.command(x0)
    c0, a0 = run(m, False)
    assert int(c0._last.model_flags) == 0
    m.W2[0, 0] = 4.0e4                      # in-place update behind the controller's back ...
    m.invalidate()                          # ... announced the documented way
    c1, a1 = run(m, False)
    assert int(c1._last.model_flags) == N.MODEL_FLAG_EXACT_FP32
    c2, a2 = run(m, True)
    assert torch.isfinite(c1.cost_total).all()
    assert float((c1.cost_total - c2.cost_total).abs().max()) <= 1e-5 * float(c2.cost_total.abs().max())
    assert float((a1 - a2).abs().max()) <= 1e-5 * max(1.0, float(a2.abs().max()))


def test_model_with_process_noise_and_one_rollout_runs_the_callables():
    """with_process_noise() makes the torch callables stochastic for any M; the fused M = 1 kernels are
    deterministic, so such a model takes the callback path unless the fused multi-rollout kernel (1 < M <= 4) applies."""
    import pytorch_mppi_amd as pm
    m = pm.models.Integrator(6, 4).with_process_noise(0.1)
    c = pm.MPPI(m.dynamics, m.running_cost, 6, torch.eye(4), num_samples=512, horizon=8, device="cuda")
    assert c._needs_generic()
    c3 = pm.MPPI(m.dynamics, m.running_cost, 6, torch.eye(4), num_samples=512, horizon=8, device="cuda", rollout_samples=3)
    assert not c3._needs_generic()
    c.command(torch.zeros(6).cuda())


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["pendulum", "integrator"])
def test_builtin_models_stay_fused_with_step_dependent_dynamics(name):
    """mppi.py:147-154: step_dependent_dynamics=True calls dynamics(state, action, t).  The built-in models are
    time-invariant and accept t, so the controller keeps the fused kernels and commands exactly what the same controller
    commands without the flag (VERDICT r02 missing 6)."""
    m = pm.models.Pendulum() if name == "pendulum" else pm.models.Integrator(6, 4)
    nx, nu = m.nx, m.nu
    kw = dict(nx=nx, noise_sigma=torch.eye(nu) * 0.5, num_samples=2048, horizon=12, device="cuda", lambda_=1.0,
              U_init=torch.zeros(12, nu), rng="philox", seed=11)
    a = pm.MPPI(m.dynamics, m.running_cost, **kw)
    b = pm.MPPI(m.dynamics, m.running_cost, step_dependent_dynamics=True, **kw)
    assert b._model is m and not b._needs_generic()
    x = torch.full((nx,), 0.3, device="cuda")
    for _ in range(3):
        assert torch.equal(a.command(x), b.command(x))
    # and the callback form of the same problem (the reference's own loop, with t passed) agrees with the fused one
    g = pm.MPPI(lambda s, u, t: m.dynamics(s, u, t), lambda s, u, t: m.running_cost(s, u, t), step_dependent_dynamics=True,
                **dict(kw, rng="torch"))
    assert g._model is None
    z = torch.randn(2048, 12, nu, device="cuda")
    b2 = pm.MPPI(m.dynamics, m.running_cost, step_dependent_dynamics=True, **dict(kw, rng="torch"))
    for c in (g, b2):
        c.inject_noise(z)
    assert torch.allclose(g.command(x), b2.command(x), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("kind", ["mppi-philox", "mppi-torch", "smppi", "kmppi", "kmppi-two-launch", "group"])
def test_commands_leave_no_garbage_for_the_cycle_collector(kind):
    """a command's buffers must die by reference count: with the cycle collector OFF (a control loop may well run that way, and
    bench.py's timed region does) device memory must not grow from command to command.  Round 5's KMPPI once tied its two problem
    blocks into a cycle: 200 MB of raw actions per command stayed alive until the collector ran, and the allocator answered with a
    fresh hipMalloc per command (0.7-0.9 ms, intermittently)."""
    import gc
    import pytorch_mppi_amd as pm
    nx, nu, K, T = 8, 4, 32768, 32
    m = pm.models.Integrator(nx, nu)
    kw = dict(num_samples=K, horizon=T, device="cuda", lambda_=20.0)
    if kind == "mppi-philox":
        c = pm.MPPI(m.dynamics, m.running_cost, nx, torch.eye(nu), rng="philox", **kw)
    elif kind == "mppi-torch":
        c = pm.MPPI(m.dynamics, m.running_cost, nx, torch.eye(nu), **kw)
    elif kind == "smppi":
        c = pm.SMPPI(m.dynamics, m.running_cost, nx, torch.eye(nu), action_max=torch.ones(nu), delta_t=0.1, **kw)
    elif kind == "group":
        c = pm.MPPI(m.dynamics, m.running_cost, nx, torch.eye(nu), rng="philox", devices=[0, 0], **kw)
    else:
        c = pm.KMPPI(m.dynamics, m.running_cost, nx, torch.eye(nu), num_support_pts=8, rng="philox", **kw)
        if kind == "kmppi-two-launch":
            c.fuse_interpolation = False
            c.onchip_update = False
    x = torch.zeros(nx, device="cuda")
    for _ in range(5):
        c.command(x)
    torch.cuda.synchronize()
    gc.collect()
    gc.disable()
    try:
        base = torch.cuda.memory_allocated()
        for _ in range(40):
            c.command(x)
        torch.cuda.synchronize()
        grown = torch.cuda.memory_allocated() - base
    finally:
        gc.enable()
    assert grown <= 4 * K * 4, f"{kind}: {grown / 1e6:.1f} MB of device memory held by garbage after 40 commands"
