"""CPU: the attribute / method plumbing of a device group (pytorch_mppi_amd/group.py) on stand-in shards -- reads go to shard 0,
per-sample results are concatenated in global sample order, writes reach every shard, methods run on all of them and the replicated
sequences are re-copied from shard 0, `command` stitches prepare / issue + exchange / end; nn.Module callables are copied per device and follow the original.  (The real thing -- shard controllers on the GPU --
is tests/test_gpu_devices.py.)"""
import pytest
import torch

import pytorch_mppi_amd as pm
from pytorch_mppi_amd import group


class FakeShard:
    def __init__(self, g, G, K):
        self.d = torch.device("cpu")
        self.dtype = torch.float32
        self.K, self.K_local, self.k_offset = K, K // G, g * (K // G)
        self.U = torch.full((3, 2), float(g))
        self.lambda_ = 1.0
        self.cost_total = torch.arange(self.K_local, dtype=torch.float32) + 100 * g
        self.states = torch.zeros(1, self.K_local, 3, 4) + g
        self.omega = None
        self.calls = []
        self.info = None
        self._jit_pending = None
        self._model = None
        self.rng, self.last_draw, self._injected, self.M, self.specific_action_sampler = "torch", None, [], 1, None

    def reset(self):
        self.calls.append("reset")
        self.U = torch.rand(3, 2)                       # every shard draws for itself ...
        return "r%d" % self.k_offset

    def scale(self, t, gain=1.0):
        self.calls.append(("scale", t.device, gain))
        return float(t.sum()) * gain

    def _prepare(self, state, shift):
        # (a shard on the callback path: everything local to it has been issued when _prepare returns -- MPPI._prepare)
        self.calls.append(("begin", tuple(torch.as_tensor(state).shape), shift))
        p = type("P", (), {})()
        p._keep = {"record": torch.tensor([float(self.k_offset), 1.0, 2.0])}
        p._deferred = False
        return p

    def _combine(self, p, records):
        self.calls.append(("combine", tuple(records.shape)))
        p._keep["records"] = records

    def _end(self, p):
        self.U = p._keep["records"].sum(0)[:2].reshape(1, 2)
        return self.U[0]


def _group(G=3, K=12):
    cls = group.group_class(pm.MPPI)
    g = object.__new__(cls)
    object.__setattr__(g, "_shards", [FakeShard(i, G, K) for i in range(G)])
    object.__setattr__(g, "_devs", [0] * G)
    object.__setattr__(g, "_comms", None)
    object.__setattr__(g, "_staged", True)
    object.__setattr__(g, "exchange", "staged")
    object.__setattr__(g, "_engine", None)
    object.__setattr__(g, "_replicas", group._Replicas())
    object.__setattr__(g, "_state_bufs", None)
    object.__setattr__(g, "_rec_bufs", {})
    object.__setattr__(g, "_fast", None)
    object.__setattr__(g, "_dirty", True)
    object.__setattr__(g, "_rearm", True)
    object.__setattr__(g, "wait_seconds", 0.0)
    return g


def test_group_class_is_a_subclass_of_what_was_asked_for():
    for cls in (pm.MPPI, pm.SMPPI, pm.KMPPI):
        gc_ = group.group_class(cls)
        assert issubclass(gc_, cls) and issubclass(gc_, group.DeviceGroup) and group.group_class(cls) is gc_
        assert gc_.__name__ == cls.__name__ + "OnDevices" and gc_._base is cls
    assert group._dev_index(3) == 3 and group._dev_index("cuda:2") == 2
    with pytest.raises(ValueError):
        group._dev_index("cpu")
    with pytest.raises(ValueError):
        pm.MPPI(None, None, 2, torch.eye(2), devices=[])            # a group needs devices; one device is a plain controller


def test_reads_writes_and_methods():
    g = _group()
    assert isinstance(g, pm.MPPI) and g.lambda_ == 1.0 and g.K == 12 and g.devices == [0, 0, 0] and len(g.shards) == 3
    assert torch.equal(g.U, torch.zeros(3, 2))                      # plain reads: shard 0
    ct = g.cost_total                                               # per-sample results: the shards' parts in global order
    assert ct.shape == (12,) and ct.tolist() == [0, 1, 2, 3, 100, 101, 102, 103, 200, 201, 202, 203]
    assert g.states.shape == (1, 12, 3, 4) and float(g.states[0, 5, 0, 0]) == 1.0           # (M, K, T, nx): the sample axis is 1
    assert g.omega is None                                          # not there yet on some shard: not there
    g.lambda_ = 2.5                                                 # writes: every shard
    g.U = torch.ones(3, 2)
    assert all(s.lambda_ == 2.5 and torch.equal(s.U, torch.ones(3, 2)) for s in g.shards)
    assert "lambda_" not in g.__dict__ and "U" not in g.__dict__    # the group object holds no controller state of its own
    out = g.reset()                                                 # methods: on every shard, shard 0's result ...
    assert out == "r0" and all(s.calls[-1] == "reset" for s in g.shards)
    assert all(torch.equal(s.U, g.shards[0].U) for s in g.shards[1:])          # ... and shard 0's sequence is everybody's afterwards
    assert g.scale(torch.ones(4), gain=2.0) == 8.0


def test_command_is_begin_on_every_shard_one_exchange_end_on_every_shard():
    g = _group(G=2, K=8)
    a = g.command([0.5, 1.0, 1.5], shift_nominal_trajectory=False, info={"k": 1})
    for s in g.shards:
        assert s.info == {"k": 1}
        assert [c[0] for c in s.calls if isinstance(c, tuple)] == ["begin", "combine"]
        assert s.calls[0] == ("begin", (3,), False) and s.calls[1] == ("combine", (2, 3))
    # the records of BOTH shards, in shard order, reached every shard: {k_offset, 1, 2} summed = {0 + 4, 2, ...}
    assert a.tolist() == [4.0, 2.0] and torch.equal(g.shards[0].U, g.shards[1].U)


def test_module_callables_travel_with_the_shard_and_follow_the_original():
    """pendulum_approximate.py:47-67's pattern: the dynamics is a network (here: the module itself, and a bound method of one)"""
    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = torch.nn.Linear(3, 2)

        def forward(self, x, u):
            return self.lin(torch.cat((x, u), dim=1))

        def cost(self, x, u):
            return (x * self.lin.bias).sum(dim=1)
    net = Net()
    r = group._Replicas()
    d = torch.device("cpu")                                  # (what matters here is the copy / sync bookkeeping, not the device)
    f, c = r.on(net, d), r.on(net.cost, d)
    assert isinstance(f, Net) and f is not net and c.__self__ is f          # ONE copy serves both callables
    assert r.on(net, d) is f and r.on(len, d) is len                         # cached; non-modules pass through
    x, u = torch.ones(4, 2), torch.ones(4, 1)
    assert torch.equal(f(x, u), net(x, u))
    with torch.no_grad():
        net.lin.weight.mul_(2.0)                              # "retraining" between two commands (mppi.py:890-893)
    assert not torch.equal(f(x, u), net(x, u))
    r.sync()
    assert torch.equal(f(x, u), net(x, u)) and torch.equal(c(x, u), net.cost(x, u))


def test_a_process_local_plan_never_touches_torch_distributed():
    from pytorch_mppi_amd.dist import LOCAL, ShardPlan
    sp = ShardPlan(10, 1, 3, LOCAL)
    assert sp.local and sp.group is None and (sp.k_offset, sp.K_local) == (4, 3) and sp.native_comm("cuda") is None
    assert not ShardPlan(10, 1, 3).local
