"""Dev helper (GPU): per-output max errors of the engine vs a golden fixture and vs the fp64 oracle."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import golden_util as gu

for name in sys.argv[1:]:
    cfg, d = gu.load(name)
    dtype = gu.TDT[cfg["dtype"]]
    for native in (True, False):
        ctrl = gu.engine_controller(cfg, d, native=native)
        state = gu.t(d, "state", dtype).cuda()
        for s in range(cfg["steps"]):
            ctrl.inject_noise(gu.t(d, f"z{s}", dtype))
            act = ctrl.command(state, shift_nominal_trajectory=bool(d[f"shift{s}"]))
            got = dict(action=act, U=ctrl.U, cost_total=ctrl.cost_total, omega=ctrl.omega, noise=ctrl.noise,
                       perturbed_action=ctrl.perturbed_action)
            line = []
            for k, v in got.items():
                ref = np.asarray(d[f"{k}{s}"], dtype=np.float64)
                err = np.abs(v.cpu().numpy().astype(np.float64) - ref).max() / max(1.0, np.abs(ref).max())
                line.append(f"{k}={err:.2e}")
            print(name, "native" if native else "generic", s, " ".join(line))
    if cfg["model"] == "mlp":
        x = torch.randn(1000, dtype=dtype) * 2
        print("tanh dev-vs-cpu", float((torch.tanh(x.cuda()).cpu() - torch.tanh(x)).abs().max()))
