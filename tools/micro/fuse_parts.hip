// fuse_parts.hip -- where the single-launch command (csrc/rollout.hpp, FUSE) spends its time at C2 (pendulum, K = 8192, T = 32):
// the product kernel with -DMPPI_FUSE_STAMPS: every workgroup stamps the device clock at
//   0 entry | 1 tables in LDS | 2 rollout done | 3 weights known | 4 partial record stored | 5 ticket drawn | 6 combine done (last workgroup)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -DMPPI_FUSE_STAMPS -I include -I pytorch_mppi_amd/csrc tools/micro/fuse_parts.hip -o tools/micro/fuse_parts
#include <cstdio>
#include <vector>
#include <algorithm>
#include "dispatch.hpp"
#include "rollout.hpp"
namespace mppi {
bool profile_next_events(hipEvent_t* a, hipEvent_t* b, unsigned long long** t) { *a = *b = nullptr; if (t) *t = nullptr; return false; }
}
using namespace mppi;
int main(int argc, char** argv) {
  const int K = argc > 1 ? atoi(argv[1]) : 8192, T = argc > 2 ? atoi(argv[2]) : 32, nx = 2, nu = 1, J = T * nu;
  KArgs<float> a{};
  a.K = K; a.Tn = T; a.nx = nx; a.nu = nu; a.J = J; a.J4 = (J + 3) / 4; a.Jpad = ((a.J4 * 4 + 63) / 64) * 64; a.zp = K; a.diag = 1;
  a.noise_src = MPPI_NOISE_PHILOX; a.lambda_ = 1.f; a.u_scale = 1.f; a.e_scale = 1.f; a.M = 1; a.n_env = 1; a.fuse = 1; a.seed = 1234; a.call = 7;
  a.u_per_command = 1;
  auto dev = [](size_t n, float v) { std::vector<float> h(n, v); float* d; (void)hipMalloc(&d, n * 4); (void)hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); return d; };
  std::vector<float> hs = {3.0f, 1.0f};
  float* st0; (void)hipMalloc(&st0, 8); (void)hipMemcpy(st0, hs.data(), 8, hipMemcpyHostToDevice);
  a.state = st0; a.U = dev(J, 0.01f); a.u_init = dev(nu, 0.f); a.mu = dev(nu, 0.f);
  a.L = dev(1, 3.1622777f); a.sinv = dev(1, 0.1f); a.umin = dev(nu, -2.f); a.umax = dev(nu, 2.f);
  a.cost = dev(K, 0.f); a.record = dev(2 + J, 0.f); a.U_out = dev(J, 0.f);
  const int nb = (K + 255) / 256;
  a.nb1 = (K + 63) / 64; a.nkc = nb; a.R = 1;
  a.block_min = dev((size_t)a.nb1 + 8, 0.f); a.eta_part = dev(nb, 0.f); a.P_part = dev((size_t)nb * a.Jpad, 0.f);
  a.z = dev((size_t)a.J4 * K * 4, 0.f);                       // "generate once": K1 stores its rows, its own K3 part re-reads them
  unsigned* tk; (void)hipMalloc(&tk, 16); (void)hipMemset(tk, 0, 16); a.ticket = tk;
  hipStream_t st; (void)hipStreamCreate(&st);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  int rc = 0;
  for (int i = 0; i < 20; ++i) rc = launch_rollout<PendulumModel<float>, float>(a, st);
  if (rc != MPPI_OK_FUSED) { printf("not the single-launch form: rc %d\n", rc); return 1; }
  const int n = 200;
  (void)hipEventRecord(e0, st);
  for (int i = 0; i < n; ++i) launch_rollout<PendulumModel<float>, float>(a, st);
  (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("single-launch command, pendulum K = %d T = %d: %.2f us per launch (back to back)\n", K, T, ms / n * 1e3);
  std::vector<unsigned long long> h(64 * 8);
  (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_fuse_stamps), h.size() * 8);
  const char* names[] = {"tables in LDS", "rollout (+ in-kernel generation)", "block minimum, weights, eta", "own part of K3 (partial record)", "stores out + ticket", "combine by the last workgroup"};
  const double tick = 0.01;      // wall_clock64: 100 MHz
  unsigned long long t0 = ~0ull, tend = 0;
  for (int b = 0; b < nb && b < 64; ++b) { t0 = std::min(t0, h[b * 8]); for (int i = 0; i < 7; ++i) tend = std::max(tend, h[b * 8 + i]); }
  for (int ph = 0; ph < 5; ++ph) {
    std::vector<double> d;
    for (int b = 0; b < nb && b < 64; ++b) d.push_back((double)(h[b * 8 + ph + 1] - h[b * 8 + ph]) * tick);
    std::sort(d.begin(), d.end());
    printf("  %-36s median %6.2f us   max %6.2f\n", names[ph], d[d.size() / 2], d.back());
  }
  double cmb = 0;
  for (int b = 0; b < nb && b < 64; ++b) if (h[b * 8 + 6] > h[b * 8 + 5]) cmb = std::max(cmb, (double)(h[b * 8 + 6] - h[b * 8 + 5]) * tick);
  printf("  %-36s        %6.2f us\n  first entry -> last stamp            %6.2f us\n", names[5], cmb, (double)(tend - t0) * tick);
  return 0;
}
