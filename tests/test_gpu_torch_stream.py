"""rng="torch" (the drop-in default): the values of the reference's own draw (mppi.py:203 `torch.randn(K, T, nu)`) are computed
by the engine's own launch straight into its sample-minor rows (csrc/noise_torch.hip, `mppi_noise_fill_torch`; MPPI._torch_stream_fill).
What has to hold, bit for bit because it is the SAME arithmetic:
  * the rows are torch.randn's values for the same generator state, for shapes on both sides of ATen's grid cap, ragged K, every
    control width with (T nu) % 4 == 0;
  * the generator ends where torch.randn would have left it: every later draw of the process is unchanged;
  * a controller on this path reports the reference's noise (`ctrl.noise`) for the seed, and commands the same actions as the same
    controller reading torch.randn's own array (torch_rows = False);
  * where it does not apply ((T nu) % 4 != 0, fp64, graph capture) the command draws with torch.randn as before."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(65536, 64, 12), (8192, 32, 4), (1000, 30, 4), (77, 9, 4), (100000, 16, 8), (4096, 32, 12), (3, 1, 4), (50000, 15, 12),
          (513, 10, 2), (2049, 4, 1), (30011, 20, 6)]


def _grid(numel):
    props = torch.cuda.get_device_properties(0)
    return min(props.multi_processor_count * (props.max_threads_per_multi_processor // 256), (numel + 255) // 256)


@pytest.mark.parametrize("K,T,nu", SHAPES)
def test_rows_are_torch_randn_bit_for_bit(K, T, nu):
    from pytorch_mppi_amd import _native as N
    dev = torch.device("cuda", 0)
    torch.cuda.init()
    gen = torch.cuda.default_generators[0]
    torch.manual_seed(1234 + K)
    torch.randn(7, device=dev)                               # the generator is somewhere in its stream
    seed, off = gen.initial_seed(), gen.get_offset()
    ref = torch.randn(K, T, nu, device=dev)
    moved = gen.get_offset() - off
    numel = K * T * nu
    grid = _grid(numel)
    assert moved == ((numel - 1) // (1024 * grid) + 1) * 4    # ATen's increment: what the caller of the C entry must add
    pitch = N.noise_pitch(K, N.F32)
    z = torch.full((T * nu // 4, pitch, 4), float("nan"), device=dev)
    rc = N.lib().mppi_noise_fill_torch(z.data_ptr(), K, T, nu, pitch, seed, off, grid, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    got = z[:, :K, :].permute(1, 0, 2).reshape(K, T, nu)
    assert torch.equal(got, ref), f"{int((got != ref).sum())} of {numel} values differ"


def test_unsupported_shapes_are_refused():
    from pytorch_mppi_amd import _native as N
    z = torch.empty(4096, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert N.lib().mppi_noise_fill_torch(z.data_ptr(), 10, 3, 3, 64, 1, 0, 1, st) == N.E_UNSUPPORTED      # (T nu) % 4 != 0
    assert N.lib().mppi_noise_fill_torch(z.data_ptr(), 100, 2, 2, 64, 1, 0, 1, st) == N.E_UNSUPPORTED     # pitch < K
    assert N.lib().mppi_noise_fill_torch(None, 10, 2, 2, 64, 1, 0, 1, st) == -1            # MPPI_E_BADARG


def _ctrl(cls, K, T, nx, nu, rows, sigma=None, **kw):
    import pytorch_mppi_amd as pm
    model = pm.models.Integrator(nx, nu)
    g = torch.Generator().manual_seed(3)
    sigma = torch.eye(nu) * 0.6 if sigma is None else sigma
    extra = dict(U_init=torch.randn(T, nu, generator=g) * 0.1) if cls == "MPPI" else {}
    if cls == "KMPPI":
        extra = dict(num_support_pts=kw.pop("num_support_pts", 8))
    c = getattr(pm, cls)(model.dynamics, model.running_cost, nx, sigma, num_samples=K, horizon=T, device="cuda", lambda_=2.0,
                         **extra, **kw)
    c.torch_rows = rows
    return c


@pytest.mark.parametrize("cls,K,T,nx,nu,kw", [
    ("MPPI", 65536, 64, 16, 12, {}),
    ("MPPI", 3000, 20, 8, 4, dict(sample_null_action=True, u_min=torch.tensor([-0.4] * 4), u_max=torch.tensor([0.5] * 4))),
    ("MPPI", 5000, 16, 8, 4, dict(noise_sigma_full=True)),
    ("SMPPI", 4000, 24, 8, 4, {}),
    ("KMPPI", 4000, 32, 8, 4, {}),
])
def test_controller_on_the_engines_rows_is_the_controller_on_torch_randn(cls, K, T, nx, nu, kw):
    kw = dict(kw)
    if kw.pop("noise_sigma_full", False):
        A = torch.randn(nu, nu, generator=torch.Generator().manual_seed(8)) * 0.2
        kw["sigma"] = A @ A.T + 0.3 * torch.eye(nu)
    x = torch.linspace(-1, 1, nx, device="cuda")
    out = {}
    for rows in (True, False):
        c = _ctrl(cls, K, T, nx, nu, rows, **kw)
        torch.manual_seed(77)
        acts = []
        for _ in range(3):
            acts.append(c.command(x).clone())
            if rows:
                assert c.last_draw in ("torch-rows", "torch-rows-ahead"), c.last_draw
            else:
                assert c.last_draw is None
        tail = torch.randn(9, device="cuda")                 # where did the command leave the generator?
        out[rows] = (torch.stack(acts), c.noise.clone(), c.cost_total.clone(), tail)
    a, b = out[True], out[False]
    assert torch.equal(a[3], b[3]), "the generator must end where torch.randn leaves it"
    # the same standard normals; with bounds `noise` is clamp(U + eps) - U, evaluated by two different kernels (one rounding apart)
    assert float((a[1] - b[1]).abs().max()) <= (1e-6 if "u_min" in kw else 0.0), "ctrl.noise must be the colouring of the same standard normals"
    scale = float(b[2].abs().max())
    assert float((a[2] - b[2]).abs().max()) <= 1e-5 * scale
    assert float((a[0] - b[0]).abs().max()) <= 1e-5 * max(1.0, float(b[0].abs().max()))


def test_the_draw_is_the_references_draw_for_the_seed():
    """mppi.py:203 through the reference's own distribution object is torch.randn(K, T, nu) @ chol + mu: for a diagonal
    Sigma, `ctrl.noise` / sigma must be torch.randn(K, T, nu) of the same seed."""
    K, T, nx, nu = 2048, 12, 8, 4
    c = _ctrl("MPPI", K, T, nx, nu, True)
    torch.manual_seed(2024)
    c.command(torch.zeros(nx, device="cuda"))
    assert c.last_draw == "torch-rows"
    torch.manual_seed(2024)
    ref = torch.randn(K, T, nu, device="cuda") * (0.6 ** 0.5)
    assert float((c.noise - ref).abs().max()) <= 1e-6


@pytest.mark.parametrize("why", ["shape", "fp64", "capture"])
def test_where_it_does_not_apply_the_command_draws_with_torch_randn(why):
    import pytorch_mppi_amd as pm
    nx, nu, T = 6, (3 if why == "shape" else 4), (5 if why == "shape" else 8)
    model = pm.models.Integrator(nx, nu)
    dt = torch.float64 if why == "fp64" else torch.float32
    c = pm.MPPI(model.dynamics, model.running_cost, nx, torch.eye(nu, dtype=dt) * 0.5, num_samples=1024, horizon=T, device="cuda")
    x = torch.zeros(nx, device="cuda", dtype=dt)
    if why == "capture":
        g = c.capture_command(x)
        torch.manual_seed(4)
        a = g(x).clone()
        b = g(x).clone()
        assert not torch.equal(a, b), "replays must advance the generator"
        assert c.last_draw is None
        c.command(x)
        assert c.last_draw in ("torch-rows", "torch-rows-ahead")     # outside the graph the controller is back on its rows
        return
    torch.manual_seed(4)
    c.command(x)
    assert c.last_draw is None
    torch.manual_seed(4)
    ref = torch.randn(1024, T, nu, device="cuda", dtype=dt) * (0.5 ** 0.5)
    assert float((c.noise - ref).abs().max()) <= 1e-6


def test_a_users_own_graph_capture_is_noticed():
    """`with torch.cuda.graph(g): ctrl.command(x)` written by the user: the offset argument of the engine's launch would be frozen
    in the graph, torch.randn's registered generator is not -- the path must step aside by itself."""
    c = _ctrl("MPPI", 1024, 8, 6, 4, True)
    x = torch.zeros(6, device="cuda")
    c.command(x)
    assert c.last_draw == "torch-rows"
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            took = c._torch_stream_fill(c._problem(), 1024, 8, 4)
    assert took is False


# ---- ABI 21: the NEXT command's draw inside this command's K3 launch (csrc/noise_torch.hip weights_partial_diag_next_kernel) ----
def _pair(cls, K, T, nx, nu, **kw):
    torch.manual_seed(1)                      # (KMPPI draws its initial control points at construction)
    a = _ctrl(cls, K, T, nx, nu, True, **kw)
    torch.manual_seed(1)
    b = _ctrl(cls, K, T, nx, nu, True, **kw)
    b.draw_ahead = False
    return a, b


@pytest.mark.parametrize("cls,K,T,nx,nu,kw,ahead", [
    ("MPPI", 65536, 64, 16, 12, {}, True),
    ("MPPI", 20000, 32, 8, 4, dict(sample_null_action=True, u_min=torch.tensor([-0.4] * 4), u_max=torch.tensor([0.5] * 4)), True),
    ("MPPI", 30011, 20, 8, 6, {}, True),                     # ragged K, a control width that is no multiple of 4
    ("SMPPI", 24000, 24, 8, 4, {}, True),
    ("KMPPI", 24000, 32, 8, 4, dict(num_support_pts=16), None),      # (its fused control-point update has no K3 launch: either way)
    ("MPPI", 2048, 8, 6, 4, {}, False),                      # single-launch command: no K3 to carry the draw
])
def test_draw_ahead_changes_no_bit(cls, K, T, nx, nu, kw, ahead):
    """A controller whose K3 launches generate the next draw commands the SAME bits as one that draws at the start of every command:
    the rows are a pure function of (seed, offset, shape), K3's arithmetic is the same code in both launches, and the generator is
    advanced by the same amounts at the same commands."""
    x = torch.linspace(-1, 1, nx, device="cuda")
    out = []
    for c in _pair(cls, K, T, nx, nu, **kw):
        torch.manual_seed(99)
        acts, draws = [], []
        for i in range(5):
            acts.append(c.command(x).clone())
            draws.append(c.last_draw)
        tail = torch.randn(9, device="cuda")
        out.append((torch.stack(acts), c.cost_total.clone(), c.U.clone(), c.noise.clone(), tail, draws, c._next_hits))
    a, b = out
    for i in range(5):
        assert torch.equal(a[i], b[i]), f"output {i} differs between draw-ahead and draw-at-start"
    assert b[6] == 0 and all(d == "torch-rows" for d in b[5])
    if ahead is True:
        assert a[5] == ["torch-rows"] + ["torch-rows-ahead"] * 4 and a[6] == 4, (a[5], a[6])
    elif ahead is False:
        assert a[6] == 0


def test_a_draw_by_somebody_else_between_two_commands_is_noticed():
    """the user calls torch.randn between two commands -- the draw generated ahead assumed an offset that is no
    longer the generator's: it is dropped, the command draws at the generator's present state, and the stream is the reference's"""
    K, T, nx, nu = 32768, 32, 8, 4             # (a streaming command: the single-launch form of small problems has no K3 launch)
    x = torch.zeros(nx, device="cuda")
    out = []
    for c in _pair("MPPI", K, T, nx, nu):
        torch.manual_seed(5)
        seq = []
        for i in range(6):
            if i == 2:
                seq.append(torch.randn(5, device="cuda"))            # the user's own draw
            if i == 5:
                torch.manual_seed(5)                                 # ... and a re-seed: (seed, offset) of the FIRST command again
            seq.append(c.command(x).clone())
            seq.append(c.noise[:3].clone())
            seq.append(c.last_draw)
        out.append((seq, c._next_hits))
    (a, ha), (b, hb) = out
    assert [v for v in a if isinstance(v, str)] == ["torch-rows", "torch-rows-ahead", "torch-rows", "torch-rows-ahead", "torch-rows-ahead", "torch-rows"]
    assert ha == 3 and hb == 0
    for u, v in zip(a, b):
        if not isinstance(u, str):
            assert torch.equal(u, v)
    # the first and the last command drew from the same (seed, offset): the same rows, whoever generated them
    assert float((a[1] - a[-2]).abs().max()) < 1e-6          # (`noise` is (U + eps) - U in fp32, and U has moved: one rounding apart)


def test_useless_draws_ahead_are_given_up():
    """the reference's benchmark protocol calls reset() -- a draw from the generator -- before every command: after two draws
    generated for nothing the controller stops asking for them (and tries again every 64th command)"""
    c = _ctrl("MPPI", 32768, 32, 8, 4, True)
    x = torch.zeros(8, device="cuda")
    asked = []
    for i in range(8):
        c.reset()
        c.command(x)
        asked.append(c._last.next_z is not None and c._last.next_z != 0)
    assert c._next_hits == 0 and asked[:2] == [True, True] and not any(asked[2:]), asked


def test_engine_stream_rows_generated_ahead_are_the_rows_of_the_generator_launch():
    """rng="philox", rows in memory (below the on-chip threshold): the rows of command n+1 come out of command n's K3 launch
    (MPPI_NEXT_DRAW_PHILOX) -- bit for bit what mppi_noise_fill_philox writes for that command, and the controller commands the
    same bits as one that runs the generator launch in front of every command"""
    import gpu_util
    K, T, nx, nu = 32768, 32, 8, 4
    x = torch.linspace(-1, 1, nx, device="cuda")
    out = []
    for ahead in (True, False):
        import pytorch_mppi_amd as pm
        m = pm.models.Integrator(nx, nu)
        g = torch.Generator().manual_seed(3)
        c = pm.MPPI(m.dynamics, m.running_cost, nx, torch.eye(nu) * 0.6, num_samples=K, horizon=T, device="cuda", lambda_=2.0,
                    U_init=torch.randn(T, nu, generator=g) * 0.1, rng="philox", seed=77, sample_null_action=True)
        c.draw_ahead = ahead
        c.draw_ahead_philox = True        # (off by default: that generator is store-bound, the fused launch gains nothing -- DESIGN.md)
        acts, draws = [], []
        for i in range(4):
            acts.append(c.command(x).clone())
            draws.append(c.last_draw)
        assert torch.equal(gpu_util.consumed_normals(c), gpu_util.device_philox_normals(c, c._call))
        out.append((torch.stack(acts), c.U.clone(), c.cost_total.clone(), c.noise.clone(), draws, c._pf_hits))
    a, b = out
    for i in range(4):
        assert torch.equal(a[i], b[i]), i
    assert a[4] == ["philox-fill"] + ["philox-rows-ahead"] * 3 and a[5] == 3, (a[4], a[5])
    assert b[4] == ["philox-fill"] * 4 and b[5] == 0


def test_rows_generated_ahead_are_torch_randn_bit_for_bit_over_random_shapes():
    """the draw a K3 launch generated for the NEXT command (slices of ATen's (block, call) units spread over generator workgroups,
    arbitrary first call per slice) against torch.randn itself from the generator state the adopting command found: 24 random shapes
    on both sides of ATen's grid cap, ragged K, every control width with (T nu) % 4 == 0"""
    import random
    import gpu_util
    import pytorch_mppi_amd as pm
    rnd = random.Random(20250925)
    gen = torch.cuda.default_generators[0]
    done = 0
    for trial in range(60):
        nu = rnd.choice([1, 2, 3, 4, 6, 8, 12, 16])
        T = rnd.randint(5, 80)
        if (T * nu) % 4:
            continue
        K = rnd.choice([rnd.randint(20000, 140000), 16384 * rnd.randint(2, 6), rnd.randint(20000, 40000) | 1])
        if K * T * nu < (1 << 19) or K * T * nu > 60_000_000 or (K <= 16384 and T * nu <= 256):
            continue
        nx = max(nu, 4)
        m = pm.models.Integrator(nx, nu)
        c = pm.MPPI(m.dynamics, m.running_cost, nx, torch.eye(nu) * 0.5, num_samples=K, horizon=T, device="cuda", lambda_=10.0)
        x = torch.zeros(nx, device="cuda")
        torch.manual_seed(1000 + trial)
        torch.randn(rnd.randint(1, 9), device="cuda")            # somewhere in the stream
        c.command(x)
        if c._next_draw is None:
            continue                                             # (a shape whose command has no streaming K3: nothing generated ahead)
        state = gen.get_state()
        ref = torch.randn(K, T, nu, device="cuda")               # what the reference's next command would draw
        after = gen.get_offset()
        gen.set_state(state)
        c.command(x)
        assert c.last_draw == "torch-rows-ahead", (K, T, nu, c.last_draw)
        assert gen.get_offset() == after, (K, T, nu)
        got = gpu_util.consumed_normals(c)
        assert torch.equal(got, ref.cpu()), (K, T, nu, int((got != ref.cpu()).sum()))
        done += 1
        del c
        if done >= 24:
            break
    assert done >= 12, done
