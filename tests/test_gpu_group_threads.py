"""The engine's device group (C-ABI 22, csrc/group.hip): one worker thread per device issues that device's
K1 / K3 / K4, the record exchange and K5 -- against the one-thread form of round 5 (MPPI_GROUP_THREADS=0), bit for bit, on the
one-GPU rig (every shard on device 0, staged exchange), plus the error paths of the hand-over protocol.  What needs two real
devices (ncclAllGather from the workers, peer copies of the state) is in test_two_real_devices_*, skipped on this pool's boxes."""
import ctypes as C

import pytest
import torch

import pytorch_mppi_amd as pm
from pytorch_mppi_amd import _native as N

pytestmark = pytest.mark.gpu


def _mk(cls, K, T, nx, nu, rng, devices, **kw):
    m = pm.models.Integrator(nx, nu)
    g = torch.Generator().manual_seed(3)
    extra = {}
    if cls is pm.MPPI:
        extra["U_init"] = torch.randn(T, nu, generator=g) * 0.05
    if cls is pm.KMPPI:
        extra["num_support_pts"] = 8
    if cls is pm.SMPPI:
        extra.update(action_min=-torch.ones(nu), action_max=torch.ones(nu), w_action_seq_cost=0.5, delta_t=0.2)
    torch.manual_seed(11)
    return cls(m.dynamics, m.running_cost, nx, torch.eye(nu) * 0.5, num_samples=K, horizon=T, device="cuda", lambda_=30.0,
               rng=rng, seed=99, devices=devices, **extra, **kw)


def _pair(monkeypatch, cls, K, T, nx, nu, rng, devices, **kw):
    monkeypatch.setenv("MPPI_GROUP_THREADS", "1")
    a = _mk(cls, K, T, nx, nu, rng, devices, **kw)
    monkeypatch.setenv("MPPI_GROUP_THREADS", "0")
    b = _mk(cls, K, T, nx, nu, rng, devices, **kw)
    assert "worker thread" in a.issue and "calling thread" in b.issue
    return a, b


@pytest.mark.parametrize("cls", [pm.MPPI, pm.SMPPI, pm.KMPPI])
@pytest.mark.parametrize("rng", ["philox", "torch"])
def test_worker_threads_command_the_bits_of_the_one_thread_form(monkeypatch, cls, rng):
    K, T, nx, nu = 24000, 24, 8, 4
    a, b = _pair(monkeypatch, cls, K, T, nx, nu, rng, [0, 0, 0])
    x = torch.linspace(-1, 1, nx, device="cuda")
    for i in range(4):
        # (rng="torch": the shards draw from their own generators, keyed by (seed, shard) -- the same in both groups)
        ua, ub = a.command(x, shift_nominal_trajectory=i != 2), b.command(x, shift_nominal_trajectory=i != 2)
        assert torch.equal(ua, ub), i
        assert all(torch.equal(a.shards[0].U, s.U) for s in a.shards[1:]), "every device must hold bit-identical U"
        assert torch.equal(a.U, b.U) and torch.equal(a.cost_total, b.cost_total) and torch.equal(a.omega, b.omega)
    assert torch.equal(a.noise, b.noise) and torch.equal(a.perturbed_action, b.perturbed_action)


def test_large_commands_run_on_chip_through_the_workers(monkeypatch):
    """C3-sized shards take the on-chip form (no row array) inside the worker's mppi_command; the form is reported back per device"""
    K, T, nx, nu = 2 * 49152, 32, 16, 12
    a, b = _pair(monkeypatch, pm.MPPI, K, T, nx, nu, "philox", [0, 0])
    x = torch.linspace(-1, 1, nx, device="cuda")
    before = int(N.lib().mppi_stat_onchip_commands())
    for _ in range(3):
        assert torch.equal(a.command(x), b.command(x))
    assert int(N.lib().mppi_stat_onchip_commands()) - before == 12 and all(s.last_draw == "philox-onchip" for s in a.shards)
    assert torch.equal(a.cost_total, b.cost_total)


def test_host_states_per_sample_states_and_attribute_writes_through_the_workers(monkeypatch):
    K, T, nx, nu = 4096, 10, 6, 4
    a, b = _pair(monkeypatch, pm.MPPI, K, T, nx, nu, "philox", [0, 0])
    assert torch.equal(a.command([0.1] * nx), b.command([0.1] * nx))                       # a host list (mppi.py:262-264)
    X = torch.randn(K, nx, generator=torch.Generator().manual_seed(4))
    assert torch.equal(a.command(X), b.command(X))                                         # per-sample states: every shard its rows
    for g in (a, b):
        g.lambda_ = 3.0
        g.u_min, g.u_max = -torch.ones(nu) * 0.2, torch.ones(nu) * 0.2
    x = torch.zeros(nx, device="cuda")
    assert torch.equal(a.command(x), b.command(x)) and float(a.perturbed_action.abs().max()) <= 0.2 + 1e-6
    z = torch.randn(K, T, nu, generator=torch.Generator().manual_seed(1))
    a.inject_noise(z), b.inject_noise(z)
    assert torch.equal(a.command(x), b.command(x))


def test_callback_path_groups_still_command(monkeypatch):
    """plain torch callables without tracing: nothing to hand to the workers, the shards issue from the calling thread"""
    K, T, nx, nu = 2048, 8, 4, 2

    def dyn(x, u):
        return x + 0.1 * torch.cat((u, u), dim=1)

    def cost(x, u):
        return (x * x).sum(dim=1)
    outs = []
    for threads in ("1", "0"):
        monkeypatch.setenv("MPPI_GROUP_THREADS", threads)
        torch.manual_seed(2)
        c = pm.MPPI(dyn, cost, nx, torch.eye(nu), num_samples=K, horizon=T, device="cuda", lambda_=5.0, rng="philox", seed=5,
                    auto_jit=False, devices=[0, 0], U_init=torch.zeros(T, nu))
        x = torch.ones(nx, device="cuda")
        outs.append(torch.stack([c.command(x).clone() for _ in range(2)]))
    assert torch.equal(*outs)


def test_module_callables_are_replicated_per_device_and_follow_retraining(monkeypatch):
    """the reference's learned-dynamics pattern (tests/pendulum_approximate.py:47-67): the dynamics is a bound method of an
    nn.Module on cuda:0; further devices get a copy that follows the original's parameters"""
    from pytorch_mppi_amd import group
    net = torch.nn.Linear(3, 2).cuda()

    class Dyn(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.net = net

        def forward(self, x, u):
            return x + 0.05 * self.net(torch.cat((x, u), dim=1))
    d = Dyn()
    c = pm.MPPI(d, lambda x, u: (x * x).sum(dim=1), 2, torch.eye(1), num_samples=1024, horizon=6, device="cuda", lambda_=1.0,
                rng="philox", seed=1, auto_jit=False, devices=[0, 0])
    a0 = c.command(torch.ones(2, device="cuda")).clone()
    # (both shards live on device 0 here: nothing was copied -- the replica bookkeeping itself is tests/test_group_plumbing.py)
    assert len(object.__getattribute__(c, "_replicas").items) == 0 and torch.isfinite(a0).all()
    assert isinstance(c, group.DeviceGroup)


def test_an_unsupported_one_call_form_falls_back_to_the_shards_own_launches(monkeypatch):
    """KMPPI with more support points than the fused-interpolation K1 takes: mppi_command_kmppi refuses on every worker, nothing
    is exchanged, the shards run the two-launch form themselves -- same bits as without workers"""
    K, T, nx, nu = 8192, 40, 8, 4
    monkeypatch.setenv("MPPI_GROUP_THREADS", "1")
    a = _mk(pm.KMPPI, K, T, nx, nu, "philox", [0, 0])
    monkeypatch.setenv("MPPI_GROUP_THREADS", "0")
    b = _mk(pm.KMPPI, K, T, nx, nu, "philox", [0, 0])
    for g in (a, b):
        g.fuse_interpolation = False
    x = torch.linspace(-1, 1, nx, device="cuda")
    for _ in range(2):
        assert torch.equal(a.command(x), b.command(x))
    assert torch.equal(a.theta, b.theta)


def test_records_read_in_place_and_gathered_by_copies_combine_to_the_same_bits(monkeypatch):
    """the staged exchange: K5 reads every shard's record where its K4 left it (mppi_combine_ptrs) -- against the form that gathers
    them by copies first (MPPI_GROUP_IN_PLACE=0)"""
    K, T, nx, nu = 12000, 16, 8, 4
    monkeypatch.setenv("MPPI_GROUP_THREADS", "1")
    a = _mk(pm.MPPI, K, T, nx, nu, "philox", [0, 0, 0])
    monkeypatch.setenv("MPPI_GROUP_IN_PLACE", "0")
    b = _mk(pm.MPPI, K, T, nx, nu, "philox", [0, 0, 0])
    x = torch.linspace(-1, 1, nx, device="cuda")
    for _ in range(3):
        assert torch.equal(a.command(x), b.command(x))
    assert torch.equal(a.U, b.U) and torch.equal(a.omega, b.omega)
    rb = b.shards[1]._last._keep["records"]
    assert torch.equal(rb[0], b.shards[0]._last._keep["record"]) and torch.equal(rb[2], b.shards[2]._last._keep["record"])


def test_hand_over_protocol_errors(monkeypatch):
    """mppi_group_submit / _wait / _abort on a staged two-slot group: a part missing -> the command is abandoned (the worker that
    started skips the exchange), a failing device reports its own error and the other MPPI_E_GROUP_PEER, and the group stays
    usable afterwards"""
    lib = N.lib()
    grp = C.c_void_p()
    monkeypatch.setenv("MPPI_GROUP_IN_PLACE", "0")      # (gathered records: what the assertions below look at)
    assert lib.mppi_group_create(2, (C.c_int32 * 2)(0, 0), None, C.byref(grp)) == 0 and lib.mppi_group_size(grp) == 2
    try:
        c = _mk(pm.MPPI, 2048, 8, 8, 4, "philox", [0, 0])
        x = torch.zeros(8, device="cuda")
        c.command(x)
        torch.cuda.synchronize()
        ps = [s._prepare(x, True) for s in c.shards]
        assert all(p._deferred for p in ps)
        recs = [torch.zeros(2, 2 + 8 * 4, device="cuda") for _ in ps]
        st = torch.cuda.current_stream().cuda_stream
        assert lib.mppi_group_wait(grp, None, None) == -1 and b"nothing submitted" in lib.mppi_last_error()
        # one part only, then wait: abandoned
        assert lib.mppi_group_submit(grp, 0, C.byref(ps[0]), None, recs[0].data_ptr(), st) == 0
        assert lib.mppi_group_submit(grp, 0, C.byref(ps[0]), None, recs[0].data_ptr(), st) == -1      # twice the same device
        assert lib.mppi_group_wait(grp, None, None) == -1 and b"abandoned" in lib.mppi_last_error()
        torch.cuda.synchronize()
        assert float(recs[0].abs().max()) == 0.0, "an abandoned command must not exchange"
        # a failing device (no state bound): its own error comes back, not the peer's
        bad = N.MppiProblem.from_buffer_copy(ps[1])
        bad.state = None
        assert lib.mppi_group_submit(grp, 0, C.byref(ps[0]), None, recs[0].data_ptr(), st) == 0
        assert lib.mppi_group_submit(grp, 1, C.byref(bad), None, recs[1].data_ptr(), st) == 0
        forms, nds = (C.c_int32 * 2)(), (C.c_int32 * 2)()
        rc = lib.mppi_group_wait(grp, forms, nds)
        assert rc == -1 and b"device 0: " in lib.mppi_last_error() and b"state" in lib.mppi_last_error(), lib.mppi_last_error()
        torch.cuda.synchronize()
        assert float(recs[0].abs().max()) == 0.0
        # abort with nothing pending is a no-op; then a complete command goes through
        assert lib.mppi_group_abort(grp) == 0
        for g in range(2):
            assert lib.mppi_group_submit(grp, g, C.byref(ps[g]), None, recs[g].data_ptr(), st) == 0
        assert lib.mppi_group_wait(grp, forms, nds) == 0, lib.mppi_last_error()
        torch.cuda.synchronize()
        assert torch.equal(recs[0], recs[1]) and float(recs[0].abs().max()) > 0.0
        assert torch.equal(recs[0][0], ps[0]._keep["record"]) and torch.equal(recs[0][1], ps[1]._keep["record"])
    finally:
        assert lib.mppi_group_destroy(grp) == 0


def test_group_objects_can_be_dropped_and_rebuilt():
    """workers are joined when the group object goes away (weakref.finalize): no thread leak across many groups"""
    import gc
    import threading
    x = torch.zeros(8, device="cuda")
    n0 = threading.active_count()
    for _ in range(6):
        c = _mk(pm.MPPI, 1024, 6, 8, 4, "philox", [0, 0, 0, 0])
        c.command(x)
        del c
        gc.collect()
    torch.cuda.synchronize()
    assert threading.active_count() == n0


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two real devices (this pool's boxes have one)")
@pytest.mark.parametrize("cls", [pm.MPPI, pm.KMPPI])
def test_two_real_devices_hold_identical_U_and_match_the_unsharded_controller(cls):
    """the real multi-device path -- ncclCommInitAll, one ncclAllGather per worker thread, the state's peer
    copy, per-device K5 -- asserted bit-identical across devices and against the unsharded controller's command"""
    K, T, nx, nu = 32768, 24, 8, 4
    n = min(torch.cuda.device_count(), 8)
    grp = _mk(cls, K, T, nx, nu, "philox", list(range(n)))
    one = _mk(cls, K, T, nx, nu, "philox", None)
    assert "RCCL" in grp.exchange and "worker thread" in grp.issue
    x = torch.linspace(-1, 1, nx, device="cuda:0")
    for i in range(5):
        a, b = grp.command(x + 0.01 * i), one.command(x + 0.01 * i)
        U0 = grp.shards[0].U.cpu()
        assert all(torch.equal(U0, s.U.cpu()) for s in grp.shards[1:]), "every device must hold bit-identical U"
        assert a.device == torch.device("cuda", 0)
        assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max()))
        assert float((grp.cost_total - one.cost_total).abs().max()) <= 1e-5 * float(one.cost_total.abs().max())


@pytest.mark.parametrize("K,T,upc", [(3 * 49152, 32, 1), (6000, 4, 2)])       # on chip | rows generated inside K1
def test_the_rearmed_steady_state_command_commands_the_bits_of_the_ordinary_one(monkeypatch, K, T, upc):
    """a plain MPPI group on the engine's generator re-arms the previous command's blocks (group.DeviceGroup._command_rearmed) --
    against the same group with MPPI_GROUP_REARM=0, over a loop with everything a caller may do in between: another state, a host
    state, attribute writes, a method call, a read of the lazily derived arrays, shift off, injected noise"""
    nx, nu = 16, 12
    monkeypatch.setenv("MPPI_GROUP_THREADS", "1")
    a = _mk(pm.MPPI, K, T, nx, nu, "philox", [0, 0, 0], u_per_command=upc)
    monkeypatch.setenv("MPPI_GROUP_REARM", "0")
    b = _mk(pm.MPPI, K, T, nx, nu, "philox", [0, 0, 0], u_per_command=upc)
    monkeypatch.delenv("MPPI_GROUP_REARM")
    x = torch.linspace(-1, 1, nx, device="cuda")
    kept = []
    fast = 0
    for i in range(14):
        if i == 4:
            for g in (a, b):
                g.lambda_ = 11.0                                     # an attribute write: the next command is an ordinary one
        if i == 7:
            for g in (a, b):
                g.change_horizon(T)                                  # a method call, likewise
        if i == 9:
            z = torch.randn(K, T, nu, generator=torch.Generator().manual_seed(i))
            a.inject_noise(z), b.inject_noise(z)
        st = x * (1.0 + 0.1 * i) if i % 3 else [0.05 * i] * nx        # a device tensor / a host list
        shift = i != 5
        ua, ub = a.command(st, shift_nominal_trajectory=shift), b.command(st, shift_nominal_trajectory=shift)
        fast += object.__getattribute__(a, "_fast") is not None and not object.__getattribute__(a, "_dirty")
        assert torch.equal(ua, ub), i
        assert ua.shape == ((nu,) if upc == 1 else (upc, nu))
        assert all(torch.equal(a.shards[0].U, s.U) for s in a.shards[1:])
        assert torch.equal(a.U, b.U) and torch.equal(a.cost_total, b.cost_total) and torch.equal(a.omega, b.omega), i
        kept.append((ua, ub.clone()))
        if i in (2, 11):
            assert torch.equal(a.noise, b.noise) and torch.equal(a.perturbed_action, b.perturbed_action)
    assert fast >= 8, "most commands of the loop must have gone the re-armed way"
    assert object.__getattribute__(b, "_fast") is None
    for ua, ub in kept:
        assert torch.equal(ua, ub), "an action returned earlier must never change (mppi.py:270-275)"
