// Where the time of the on-chip K1 (csrc/rollout_onchip.hpp) goes at C3 (K = 65536, T = 64, nx = 16, nu = 12): the product
// kernel as-is and with phases knocked out (-DMPPI_ONCHIP_EXP=bits: 1 no weighting phase | 2 no second generation |
// 4 nothing kept | 8 no rollout arithmetic), on synthetic arguments, HIP events over 200 back-to-back launches.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast [-DMPPI_ONCHIP_EXP=n] -I include -I pytorch_mppi_amd/csrc
//         tools/micro/onchip_parts.hip -o tools/micro/onchip_parts_n
#include <cstdio>
#include <vector>
#include "dispatch.hpp"
#include "rollout.hpp"
namespace mppi {
static unsigned long long* g_ts = nullptr;     // MPPI_MICRO_STAMPS=1: the measurement hook's stamp slots (what do they cost?)
bool profile_next_events(hipEvent_t* a, hipEvent_t* b, unsigned long long** t) { *a = *b = nullptr; if (t) *t = g_ts; return false; }
}
using namespace mppi;
#ifndef MPPI_ONCHIP_EXP
#define MPPI_ONCHIP_EXP 0
#endif
int main(int argc, char** argv) {
  const int K = argc > 1 ? atoi(argv[1]) : 65536, T = 64, nx = 16, nu = 12, J = T * nu;
  KArgs<float> a{};
  a.K = K; a.Tn = T; a.nx = nx; a.nu = nu; a.J = J; a.J4 = J / 4; a.Jpad = J; a.zp = K; a.diag = 1; a.noise_src = MPPI_NOISE_PHILOX;
  a.lambda_ = argc > 2 ? (float)atof(argv[2]) : 40.f; a.u_scale = 1.f; a.e_scale = 1.f; a.M = 1; a.n_env = 1; a.fuse = 1; a.seed = 1234; a.call = 7; a.u_per_command = 1;
  auto dev = [](size_t n, float v) { std::vector<float> h(n, v); float* d; (void)hipMalloc(&d, n * 4); (void)hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); return d; };
  a.state = dev(nx, 0.1f); a.U = dev(J, 0.01f); a.u_init = dev(nu, 0.f); a.mu = dev(nu, 0.f);
  std::vector<float> hL(nu * nu, 0.f);
  for (int n = 0; n < nu; ++n) hL[n * nu + n] = 1.f;
  float *L, *Si; (void)hipMalloc(&L, nu * nu * 4); (void)hipMalloc(&Si, nu * nu * 4);
  (void)hipMemcpy(L, hL.data(), nu * nu * 4, hipMemcpyHostToDevice); (void)hipMemcpy(Si, hL.data(), nu * nu * 4, hipMemcpyHostToDevice);
  a.L = L; a.sinv = Si; const float bnd = argc > 3 ? __builtin_huge_valf() : 2.5f;      // any third argument: no bounds
  a.umin = dev(nu, -bnd); a.umax = dev(nu, bnd);
  a.cost = dev(K, 0.f); a.block_min = dev(K / 64 + 4, 0.f); a.record = dev(2 + J, 0.f); a.U_out = dev(J, 0.f);
  const int nb = (K + 255) / 256;
  a.eta_part = dev((size_t)nb + (size_t)nb * J, 0.f); a.nkc = nb; a.R = 1;
  if (getenv("MPPI_MICRO_SPILL")) {           // rows that fit neither registers nor LDS wait in memory instead of being generated twice
    const long long cap = (long long)nb * 256 * 4 * 192;
    (void)hipMalloc(&a.spill, cap * 4); a.spill_cap = cap;
  }
  if (getenv("MPPI_MICRO_STAMPS")) { (void)hipMalloc(&mppi::g_ts, 16 * STAMP_SLOTS); (void)hipMemset(mppi::g_ts, 0, 16 * STAMP_SLOTS); }
  hipStream_t st; (void)hipStreamCreate(&st);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int n = 200;
  for (int i = 0; i < 20; ++i) {
    int rc = launch_rollout_onchip<IntegratorModel<float, 16, 12>, float>(a, st);
    if (rc != MPPI_OK_ONCHIP) { printf("launch failed %d\n", rc); return 1; }
  }
  (void)hipEventRecord(e0, st);
  for (int i = 0; i < n; ++i) launch_rollout_onchip<IntegratorModel<float, 16, 12>, float>(a, st);
  (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  float c0; (void)hipMemcpy(&c0, a.cost, 4, hipMemcpyDeviceToHost);
  printf("%s%son-chip K1, K = %d, lambda %g, %s, knocked out = %2d (1 weighting | 2 second generation | 4 keeping | 8 rollout): %.1f us per launch (back to back)  cost[0] = %g\n",
         mppi::g_ts ? "[stamps] " : "", a.spill ? "[spill] " : "", K, a.lambda_, argc > 3 ? "no bounds" : "bounds +-2.5", MPPI_ONCHIP_EXP, ms / n * 1e3, c0);
  return 0;
}
