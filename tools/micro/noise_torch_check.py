"""is the torch.randn stream reproducible bit for bit from outside ATen, and at what price?  (csrc/noise_torch.hip through the product library)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from pytorch_mppi_amd import _native as N
lib = N.lib()
dev = torch.device("cuda", 0)
props = torch.cuda.get_device_properties(dev)
print("CUs", props.multi_processor_count, "max threads per CU", props.max_threads_per_multi_processor)
gen = torch.cuda.default_generators[0]
for (K, T, nu) in ((65536, 64, 12), (8192, 32, 4), (1000, 30, 4), (77, 9, 4), (100000, 16, 8), (4096, 32, 12)):
    torch.manual_seed(1234 + K)
    _ = torch.randn(7, device=dev)                      # some earlier consumption
    seed, off0 = gen.initial_seed(), gen.get_offset()
    state = gen.get_state()
    ref = torch.randn(K, T, nu, device=dev)
    off1 = gen.get_offset()
    nxt = torch.randn(5, device=dev)
    numel = K * T * nu
    grid = min(props.multi_processor_count * (props.max_threads_per_multi_processor // 256), (numel + 255) // 256)
    inc = ((numel - 1) // (256 * grid * 4) + 1) * 4
    p = N.MppiProblem(); p.K, p.T, p.nu, p.dtype = K, T, nu, 0
    pitch = N.noise_pitch(K, 0); p.noise_pitch = pitch
    J4 = T * nu // 4
    z = torch.full((J4, pitch, 4), float("nan"), device=dev)
    rc = lib.mppi_noise_fill_torch(z.data_ptr(), K, T, nu, pitch, seed, off0, grid, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = z[:, :K, :].permute(1, 0, 2).reshape(K, T, nu)
    same = torch.equal(got, ref)
    nd = int((got != ref).sum()) if not same else 0
    print(f"K {K} T {T} nu {nu}: rc {rc} grid {grid} offset {off0} -> {off1} (predicted +{inc}: {'ok' if off1 - off0 == inc else 'MISMATCH'}) bitwise equal {same} ({nd} of {numel} differ; nan left {int(torch.isnan(got).sum())})")
    if not same:
        d = (got - ref).abs(); print("   max abs diff", float(d[~torch.isnan(d)].max()) if (~torch.isnan(d)).any() else None, got.flatten()[:4].tolist(), ref.flatten()[:4].tolist())
# timing
K, T, nu = 65536, 64, 12
p = N.MppiProblem(); p.K, p.T, p.nu, p.dtype = K, T, nu, 0
pitch = N.noise_pitch(K, 0); p.noise_pitch = pitch
z = torch.empty(T * nu // 4 * pitch * 4, device=dev)
grid = 2048
st = torch.cuda.current_stream().cuda_stream
for name, fn in (("engine kernel (torch stream, row layout)", lambda: lib.mppi_noise_fill_torch(z.data_ptr(), K, T, nu, pitch, 1, 0, grid, st)),
                 ("torch.randn(K,T,nu)", lambda: torch.randn(K, T, nu, device=dev))):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): fn()
    torch.cuda.synchronize(); print(f"{name}: {(time.perf_counter() - t0) / 50 * 1e6:.1f} us")
