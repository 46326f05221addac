import time, torch
torch.cuda.init(); x = torch.zeros(1, device="cuda")
g = torch.cuda.default_generators[0]
for name, fn in (("is_current_stream_capturing", torch.cuda.is_current_stream_capturing), ("get_offset", g.get_offset), ("initial_seed", g.initial_seed), ("set_offset", lambda: g.set_offset(4))):
    for _ in range(100): fn()
    t0 = time.perf_counter()
    for _ in range(10000): fn()
    print(name, (time.perf_counter() - t0) / 10000 * 1e6, "us")
