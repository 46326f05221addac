// Where the streaming K1 (csrc/rollout.hpp, rollout_cost_kernel<Integrator<16,12>, TNK4, diag>) stands against the pure row
// stream of tools/micro/loader_stream.hip (32.8-33.0 us HBM-cold incl. launch gaps): the product kernel as-is and with
// -DMPPI_K1_STREAM_ONLY (its memory pipeline alone: rows summed, no arithmetic), rows HBM-cold (8 arrays cycled), events
// over 64 back-to-back launches -- the same clock as loader_stream.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast [-DMPPI_K1_STREAM_ONLY] [-DMPPI_K1_ROWS=n] -I include -I pytorch_mppi_amd/csrc
//         tools/micro/k1_parts.hip -o tools/micro/k1_parts_x
#include <cstdio>
#include <vector>
#include "dispatch.hpp"
#include "rollout.hpp"
namespace mppi {
bool profile_next_events(hipEvent_t* a, hipEvent_t* b, unsigned long long** t) { *a = *b = nullptr; if (t) *t = nullptr; return false; }
}
using namespace mppi;
int main(int argc, char** argv) {
  const int K = 65536, T = 64, nx = 16, nu = 12, J = T * nu, NBUF = 8;
  KArgs<float> a{};
  a.K = K; a.Tn = T; a.nx = nx; a.nu = nu; a.J = J; a.J4 = J / 4; a.Jpad = J; a.zp = K; a.diag = 1; a.noise_src = MPPI_NOISE_TNK4;
  a.lambda_ = 40.f; a.u_scale = 1.f; a.e_scale = 1.f; a.M = 1; a.n_env = 1; a.fuse = -1; a.u_per_command = 1;
  auto dev = [](size_t n, float v) { std::vector<float> h(n, v); float* d; (void)hipMalloc(&d, n * 4); (void)hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); return d; };
  a.state = dev(nx, 0.1f); a.U = dev(J, 0.01f); a.u_init = dev(nu, 0.f); a.mu = dev(nu, 0.f);
  std::vector<float> hL(nu * nu, 0.f);
  for (int n = 0; n < nu; ++n) hL[n * nu + n] = 1.f;
  float *L, *Si; (void)hipMalloc(&L, nu * nu * 4); (void)hipMalloc(&Si, nu * nu * 4);
  (void)hipMemcpy(L, hL.data(), nu * nu * 4, hipMemcpyHostToDevice); (void)hipMemcpy(Si, hL.data(), nu * nu * 4, hipMemcpyHostToDevice);
  a.L = L; a.sinv = Si; a.umin = dev(nu, -2.5f); a.umax = dev(nu, 2.5f);
  a.cost = dev(K, 0.f); a.block_min = dev(K / 64 + 4, 0.f);
  const size_t elems = (size_t)(J / 4) * K * 4;
  std::vector<float> h(elems);
  for (size_t i = 0; i < elems; ++i) h[i] = (float)((i * 2654435761u >> 22) & 255) * (1.0f / 128.0f) - 1.0f;
  std::vector<float*> bufs(NBUF);
  for (int b = 0; b < NBUF; ++b) { (void)hipMalloc(&bufs[b], elems * 4); (void)hipMemcpy(bufs[b], h.data(), elems * 4, hipMemcpyHostToDevice); }
  hipStream_t st; (void)hipStreamCreate(&st);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int n = 64;
  for (int i = 0; i < 10; ++i) { a.z = bufs[i % NBUF]; int rc = launch_rollout<IntegratorModel<float, 16, 12>, float>(a, st); if (rc) { printf("launch failed %d\n", rc); return 1; } }
  (void)hipEventRecord(e0, st);
  for (int i = 0; i < n; ++i) { a.z = bufs[i % NBUF]; launch_rollout<IntegratorModel<float, 16, 12>, float>(a, st); }
  (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  float c0; (void)hipMemcpy(&c0, a.cost, 4, hipMemcpyDeviceToHost);
  const double us = ms / n * 1e3, mb = elems * 4 / 1e6;
#ifdef MPPI_K1_STREAM_ONLY
  const char* what = "memory pipeline only (MPPI_K1_STREAM_ONLY)";
#else
  const char* what = "as shipped";
#endif
  printf("K1 streaming, HBM-cold, %s, ring rows %d: %.1f us per launch (back to back) = %.2f TB/s = %.1f %% of 8 TB/s   cost[0] = %g\n", what, MPPI_K1_ROWS, us,
         mb / us, mb / us / 8.0 * 100, c0);
  return 0;
}
