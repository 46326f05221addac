"""Do the bodies the tracer writes compile for the DEVICE, in float and in double?  Every case of tests/test_trace_coverage.py and
tests/test_trace_programs.py is generated and compiled (hipcc --cuda-device-only, a bare kernel around step / cost / terminal
with csrc/common.hpp's helpers: seconds per case, no rollout machinery).  The host verification of the tests covers the
arithmetic; this covers helper names, overloads and literal types on the device side.  No GPU needed.
    python tools/trace_device_compile_check.py"""
import sys, os, subprocess, tempfile, random
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from pytorch_mppi_amd import trace
TEMPLATE = r'''
#include "common.hpp"
namespace mppi {
template <typename T>
struct M {
  static constexpr int NX = %(nx)d, NU = %(nu)d;
  const T* p;
  __device__ void step(T (&x)[NX], const T (&u)[NU], int t) const { %(step)s }
  __device__ T cost(const T (&x)[NX], const T (&u)[NU], int t) const { %(cost)s }
  __device__ T terminal(const T (&x)[NX]) const { %(terminal)s }
};
template <typename T> __global__ void k(const T* p, T* out) {
  M<T> m{p}; T x[M<T>::NX], u[M<T>::NU];
  for (int i = 0; i < M<T>::NX; ++i) x[i] = out[i]; for (int i = 0; i < M<T>::NU; ++i) u[i] = out[8 + i];
  T c = m.cost(x, u, 3); m.step(x, u, 3); c += m.terminal(x);
  for (int i = 0; i < M<T>::NX; ++i) out[i] = x[i]; out[15] = c;
}
template __global__ void k<float>(const float*, float*);
template __global__ void k<double>(const double*, double*);
}
'''
def devcompile(code, nx, nu):
    src = TEMPLATE % dict(nx=nx, nu=nu, step=code['step'], cost=code['cost'], terminal=code['terminal'] or 'return T(0);')
    with tempfile.TemporaryDirectory() as d:
        f = os.path.join(d, 'm.hip'); open(f, 'w').write(src)
        r = subprocess.run(['hipcc', '--offload-arch=gfx950', '-O1', '-std=c++17', '-I', os.path.join(ROOT, 'pytorch_mppi_amd', 'csrc'), '-I', os.path.join(ROOT, 'include'),
                            '--cuda-device-only', '-c', f, '-o', os.path.join(d, 'm.o')], capture_output=True, text=True)
        return r.returncode == 0, r.stderr
if __name__ == '__main__':
    import test_trace_coverage as tc, test_trace_programs as tp
    bad = 0
    cases = [(n, tc._seq(a), tc.Q, 3, 2, None, False) for n, a in tc.ACTS.items()] + [(n, f, q, 3, 2, None, False) for n, (f, q) in tc.CASES.items()] \
        + [(n, *v[:6]) for n, v in tp.PROGRAMS.items() if n not in tp.UNTRACEABLE]
    for name, f, q, nx, nu, term, sd in cases:
        code = trace.generate(f, q, nx, nu, term, sd)
        ok, err = devcompile(code, nx, nu)
        if not ok:
            bad += 1; print('DEVICE COMPILE FAIL', name, err[-600:], flush=True)
    print('checked', len(cases), 'bad', bad)
