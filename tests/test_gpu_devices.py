"""One Python process, N devices: `MPPI(..., devices=[...])` (pytorch_mppi_amd/group.py).
The GPU box has ONE device: `devices=[0]` is the unsharded controller on cuda:0, `devices=[0, 0]` runs two shards back to back on
the one device with the records staged through device copies (RCCL takes one rank per device) -- the shard plan, the per-shard
K1 / K3 / K4, the record layout, K5 and the whole attribute / method plumbing of the group object are the product code; the grouped
RCCL all-gather of real multi-device groups (mppi_exchange_combine_all) can only be reached on a multi-GPU node."""
import numpy as np
import pytest
import torch

import gpu_util
import pytorch_mppi_amd as pm
from oracle import dynamics as odyn
from oracle import mppi_oracle as orc
from pytorch_mppi_amd.group import DeviceGroup

pytestmark = pytest.mark.gpu


def _mk(cls, K, T, nx, nu, rng="philox", devices=None, **kw):
    m = pm.models.Integrator(nx, nu)
    g = torch.Generator().manual_seed(3)
    extra = {}
    if cls is pm.MPPI:
        extra["U_init"] = torch.randn(T, nu, generator=g) * 0.05
    if cls is pm.KMPPI:
        extra["num_support_pts"] = 8
    torch.manual_seed(11)
    return cls(m.dynamics, m.running_cost, nx, torch.eye(nu) * 0.5, num_samples=K, horizon=T, device="cuda", lambda_=30.0,
               rng=rng, seed=99, devices=devices, **extra, **kw)


@pytest.mark.parametrize("rng", ["philox", "torch"])
def test_one_device_is_the_unsharded_controller_bit_for_bit(rng):
    K, T, nx, nu = 8192, 16, 8, 4
    x = torch.linspace(-1, 1, nx, device="cuda")
    outs = []
    for devices in (None, [0]):
        c = _mk(pm.MPPI, K, T, nx, nu, rng, devices)
        assert type(c) is pm.MPPI and c.d == torch.device("cuda", 0) or devices is None
        torch.manual_seed(5)
        acts = torch.stack([c.command(x).clone() for _ in range(3)])
        outs.append((acts, c.U.clone(), c.cost_total.clone(), c.omega.clone()))
    for u, v in zip(*outs):
        assert torch.equal(u, v)


@pytest.mark.parametrize("cls", [pm.MPPI, pm.SMPPI, pm.KMPPI])
def test_two_shards_on_one_device_command_what_the_unsharded_controller_commands(cls):
    """rng="philox": the normals are a function of the GLOBAL sample index, so a group draws what the unsharded controller draws"""
    K, T, nx, nu = 24000, 24, 8, 4
    x = torch.linspace(-1, 1, nx, device="cuda")
    kw = dict(sample_null_action=True) if cls is pm.MPPI else {}
    if cls is pm.SMPPI:
        kw = dict(action_min=-torch.ones(nu), action_max=torch.ones(nu), w_action_seq_cost=0.5, delta_t=0.2)
    grp = _mk(cls, K, T, nx, nu, "philox", [0, 0], **kw)
    one = _mk(cls, K, T, nx, nu, "philox", None, **kw)
    assert isinstance(grp, DeviceGroup) and isinstance(grp, cls) and "TEST RIG" in grp.exchange
    assert [s.K_local for s in grp.shards] == [12000, 12000] and [s.k_offset for s in grp.shards] == [0, 12000]
    assert torch.equal(grp.U, one.U) and grp.K == K and grp.T == T
    for i in range(3):
        a, b = grp.command(x), one.command(x)
        assert a.shape == b.shape and a.device == torch.device("cuda", 0)
        assert torch.equal(grp.shards[0].U, grp.shards[1].U), "every device must hold bit-identical U"
        scale = max(1.0, float(b.abs().max()))
        assert float((a - b).abs().max()) <= 2e-5 * scale, (i, float((a - b).abs().max()))
        assert float((grp.U - one.U).abs().max()) <= 2e-5 * max(1.0, float(one.U.abs().max()))
        ct = grp.cost_total
        assert ct.shape == (K,) and float((ct - one.cost_total).abs().max()) <= 1e-5 * float(one.cost_total.abs().max())
        om = grp.omega
        assert om.shape == (K,) and abs(float(om.double().sum()) - 1.0) < 1e-5
        assert float((om - one.omega).abs().max()) <= 1e-5 * float(one.omega.max()) + 1e-9
    assert grp.noise.shape == one.noise.shape == (K, T, nu)
    if cls is not pm.KMPPI:
        assert torch.equal(grp.noise, one.noise) or float((grp.noise - one.noise).abs().max()) < 1e-6     # (bounded noise: clamp(U + eps) - U)


def test_group_against_the_fp64_oracle_on_the_global_draw():
    K, T, nx, nu = 16384, 20, 8, 4
    grp = _mk(pm.MPPI, K, T, nx, nu, "philox", [0, 0], u_min=torch.tensor([-0.8] * nu), u_max=torch.tensor([0.9] * nu))
    x = torch.linspace(-1, 1, nx, device="cuda")
    U0 = grp.U.clone()
    a = grp.command(x)
    z = torch.cat([gpu_util.consumed_normals(s) for s in grp.shards], dim=0)
    assert z.shape == (K, T, nu)
    f, q = odyn.make_quadtoy(nx, nu)
    outs = {}
    for dt in (torch.float64, torch.float32):
        p = orc.Problem(dynamics=f, running_cost=q, nx=nx, noise_sigma=(torch.eye(nu) * 0.5).to(dt), K=K, T=T, lambda_=30.0,
                        u_min=torch.tensor([-0.8] * nu, dtype=dt), u_max=torch.tensor([0.9] * nu, dtype=dt))
        outs[dt] = orc.command(p, U0.cpu().to(dt), x.cpu().to(dt), z.to(dt), True)
    r64, r32 = outs[torch.float64], outs[torch.float32]
    for name, got in (("action", a), ("U", grp.U), ("cost_total", grp.cost_total), ("omega", grp.omega)):
        ref = r64[name].numpy().astype(np.float64)
        g = got.cpu().numpy().astype(np.float64).reshape(ref.shape)
        scale = max(1.0, float(np.abs(ref).max())) if name != "omega" else float(np.abs(ref).max())
        floor = float(np.abs(r32[name].numpy().astype(np.float64) - ref).max())
        err = float(np.abs(g - ref).max())
        assert err <= max(1e-5 * scale, 2 * floor), (name, err, floor, scale)


def test_attribute_writes_reach_every_shard_and_methods_run_on_all_of_them():
    K, T, nx, nu = 4096, 12, 6, 4
    grp = _mk(pm.MPPI, K, T, nx, nu, "philox", [0, 0])
    x = torch.zeros(nx, device="cuda")
    grp.lambda_ = 7.5
    grp.U = torch.full((T, nu), 0.25, device="cuda")
    assert all(s.lambda_ == 7.5 and float(s.U[0, 0]) == 0.25 for s in grp.shards)
    grp.command(x)
    torch.manual_seed(0)
    grp.reset()                                       # draws per shard; shard 0's sequence is the group's
    assert torch.equal(grp.shards[0].U, grp.shards[1].U) and not torch.equal(grp.U, torch.full((T, nu), 0.25, device="cuda"))
    grp.change_horizon(T + 3)
    assert all(s.T == T + 3 and s.U.shape == (T + 3, nu) for s in grp.shards)
    a = grp.command(x)
    assert a.shape == (nu,) and grp.cost_total.shape == (K,)
    assert "K=4096" in grp.get_params() and grp.devices == [0, 0]
    z = torch.randn(K, T + 3, nu, generator=torch.Generator().manual_seed(1))
    grp.inject_noise(z)                               # a global draw: every shard takes its rows
    grp.command(x)
    assert float((grp.noise.cpu() - z * 0.5 ** 0.5).abs().max()) < 1e-6


def test_per_sample_initial_states_are_split_by_global_index():
    K, T, nx, nu = 2048, 8, 6, 4
    grp = _mk(pm.MPPI, K, T, nx, nu, "philox", [0, 0])
    one = _mk(pm.MPPI, K, T, nx, nu, "philox", None)
    X = torch.randn(K, nx, generator=torch.Generator().manual_seed(4)).cuda()
    a, b = grp.command(X), one.command(X)
    assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max()))
    assert float((grp.cost_total - one.cost_total).abs().max()) <= 1e-5 * float(one.cost_total.abs().max())


def test_the_grouped_exchange_refuses_a_device_listed_twice():
    """RCCL takes one rank per device: the C entry says so before touching RCCL (the group object then stages the records)"""
    import ctypes as C
    from pytorch_mppi_amd import _native as N
    comms = (C.c_void_p * 2)()
    assert N.lib().mppi_dist_init_all(2, (C.c_int32 * 2)(0, 0), comms) == N.E_UNSUPPORTED
    assert N.lib().mppi_dist_init_all(0, (C.c_int32 * 2)(0, 0), comms) == -1


def test_communicators_of_a_one_device_group_from_ncclcomminitall():
    """what CAN be run of the RCCL path on one GPU: ncclCommInitAll over [0] and the grouped all-gather + K5 at world size 1"""
    import ctypes as C
    from pytorch_mppi_amd import _native as N
    lib = N.lib()
    if not lib.mppi_dist_available():
        pytest.skip("no RCCL in this process")
    comms = (C.c_void_p * 1)()
    rc = lib.mppi_dist_init_all(1, (C.c_int32 * 1)(0), comms)
    assert rc == 0, lib.mppi_last_error()
    try:
        c = _mk(pm.MPPI, 4096, 12, 6, 4, "philox", None)
        c._force_collective = True
        from pytorch_mppi_amd.dist import ShardPlan
        c._shard = ShardPlan(c.K, 0, 1)
        p = c._begin(torch.zeros(6, device="cuda"), True)
        records = torch.empty(1, 2 + 12 * 4, device="cuda")
        p._keep["records"] = records
        st = torch.cuda.current_stream().cuda_stream
        rc = lib.mppi_exchange_combine_all(1, (C.c_int32 * 1)(0), (C.POINTER(N.MppiProblem) * 1)(C.pointer(p)), comms,
                                           (C.c_void_p * 1)(records.data_ptr()), (C.c_void_p * 1)(st))
        assert rc == 0, lib.mppi_last_error()
        a = c._end(p)
        torch.cuda.synchronize()
        assert torch.equal(records[0], p._keep["record"]) and torch.isfinite(a).all()
    finally:
        lib.mppi_dist_destroy(comms[0])


@pytest.mark.parametrize("K,G", [(30011, 3), (8192, 4), (1000, 7)])
def test_odd_splits_over_several_shards(K, G):
    """ragged K over 3 / 4 / 7 shards of the one device: the contiguous split by global index, a sampler's rows on shard 0, every
    shard's U bit-identical, the command the unsharded controller's"""
    T, nx, nu = 12, 6, 4
    sa = torch.randn(2, T, nu, generator=torch.Generator().manual_seed(5)) * 0.3

    class S(pm.SpecificActionSampler):
        def sample_trajectories(self, state, info):
            return sa.clone()
    kw = dict(sample_null_action=True, specific_action_sampler=S())
    grp = _mk(pm.MPPI, K, T, nx, nu, "philox", [0] * G, **kw)
    kw = dict(sample_null_action=True, specific_action_sampler=S())
    one = _mk(pm.MPPI, K, T, nx, nu, "philox", None, **kw)
    assert sum(s.K_local for s in grp.shards) == K and [s.k_offset for s in grp.shards] == [sum(t.K_local for t in grp.shards[:i]) for i in range(G)]
    x = torch.linspace(-1, 1, nx, device="cuda")
    for _ in range(2):
        a, b = grp.command(x), one.command(x)
        assert all(torch.equal(grp.shards[0].U, s.U) for s in grp.shards[1:])
        assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max()))
        assert float((grp.cost_total - one.cost_total).abs().max()) <= 1e-5 * float(one.cost_total.abs().max())
    assert torch.equal(grp.perturbed_action[0], torch.zeros(T, nu, device="cuda"))
    assert torch.allclose(grp.perturbed_action[1:3].cpu(), sa, atol=0)
