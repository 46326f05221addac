// The two-waves-per-sample form of the on-chip K1 (csrc/rollout_onchip_pair.hpp) against the one-wave kernel on the same synthetic
// C3-sized problem: every output compared bit for bit (costs, workgroup minima, eta, partial sums), then both timed (HIP events over
// 200 back-to-back launches).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -I include -I pytorch_mppi_amd/csrc tools/micro/onchip_pair_check.hip -o tools/micro/onchip_pair_check
#include <cstdio>
#include <cstring>
#include <vector>
#include "dispatch.hpp"
#include "rollout.hpp"
namespace mppi {
bool profile_next_events(hipEvent_t* a, hipEvent_t* b, unsigned long long** t) { *a = *b = nullptr; if (t) *t = nullptr; return false; }
}
using namespace mppi;
#ifndef CHECK_MODEL
#define CHECK_MODEL IntegratorModel<float, 16, 12>
#define CHECK_NX 16
#define CHECK_NU 12
#endif
int main(int argc, char** argv) {
  const int K = argc > 1 ? atoi(argv[1]) : 65536, T = argc > 4 ? atoi(argv[4]) : 64, nx = CHECK_NX, nu = CHECK_NU, J = T * nu;
  KArgs<float> a{};
  a.K = K; a.Tn = T; a.nx = nx; a.nu = nu; a.J = J; a.J4 = J / 4; a.Jpad = J; a.zp = K; a.diag = 1; a.noise_src = MPPI_NOISE_PHILOX;
  a.lambda_ = argc > 2 ? (float)atof(argv[2]) : 40.f; a.u_scale = 1.f; a.e_scale = 1.f; a.M = 1; a.n_env = 1; a.fuse = 1; a.seed = 1234; a.call = 7; a.u_per_command = 1;
  a.null_action = argc > 3 ? atoi(argv[3]) : 0;
  a.use_terminal = 1;
  auto devv = [](const std::vector<float>& h) { float* d; (void)hipMalloc(&d, h.size() * 4); (void)hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice); return d; };
  auto dev = [&](size_t n, float v) { return devv(std::vector<float>(n, v)); };
  std::vector<float> hU(J), hs(nx), hmu(nu);
  const float uscale = argc > 5 ? (float)atof(argv[5]) : 0.05f;   // (lambda U Sigma^-1 eps: a large nominal sequence alone makes the softmax peaked)
  for (int j = 0; j < J; ++j) hU[j] = uscale * (float)((j * 37) % 23 - 11);
  for (int i = 0; i < nx; ++i) hs[i] = 0.1f * (float)(i - 5);
  for (int n = 0; n < nu; ++n) hmu[n] = 0.01f * (float)n;
#ifdef CHECK_MP_ELEMS      // a model with a parameter blob (LinearGoal: B (nx,nu) | goal (nx))
  {
    std::vector<float> hp(CHECK_MP_ELEMS);
    for (size_t i = 0; i < hp.size(); ++i) hp[i] = 0.02f * (float)((int)((i * 29) % 17) - 8);
    a.mp = devv(hp);
  }
#endif
  a.state = devv(hs); a.U = devv(hU); a.u_init = dev(nu, 0.f); a.mu = devv(hmu);
  std::vector<float> hL(nu * nu, 0.f), hSi(nu * nu, 0.f);
  for (int n = 0; n < nu; ++n) { hL[n * nu + n] = 0.8f + 0.05f * n; hSi[n * nu + n] = 1.f / (hL[n * nu + n] * hL[n * nu + n]); }
  a.L = devv(hL); a.sinv = devv(hSi);
  if (argc > 6) { a.umin = dev(nu, -__builtin_huge_valf()); a.umax = dev(nu, __builtin_huge_valf()); }      // any sixth argument: no bounds
  else { a.umin = dev(nu, -1.5f); a.umax = dev(nu, 2.0f); }
  const int nb = (K + 255) / 256;
  const long long cap = (long long)nb * 512 * 4 * 192;
  (void)hipMalloc(&a.spill, cap * 4); a.spill_cap = cap;
  a.record = dev(2 + J, 0.f); a.U_out = dev(J, 0.f);
  a.nkc = nb; a.R = 1;
  const size_t wsn = (size_t)nb + (size_t)nb * J;
  float* cost[2]; float* bmin[2]; float* eta[2];
  for (int v = 0; v < 2; ++v) { cost[v] = dev(K, -1.f); bmin[v] = dev(K / 64 + 4, -1.f); eta[v] = dev(wsn, -1.f); }
  hipStream_t st; (void)hipStreamCreate(&st);
  auto run = [&](int v) {
    KArgs<float> b = a;
    b.cost = cost[v]; b.block_min = bmin[v]; b.eta_part = eta[v];
    if (v == 0) return launch_rollout_onchip<CHECK_MODEL, float>(b, st);
    onchip_carve(b);
    return launch_rollout_onchip_pair<CHECK_MODEL>(b, st);
  };
  setenv("MPPI_ONCHIP_PAIR", "0", 1);
  for (int v = 0; v < 2; ++v) {
    const int rc = run(v);
    if (rc != MPPI_OK_ONCHIP) { printf("launch %d failed: %d (%s)\n", v, rc, hipGetErrorString(hipGetLastError())); return 1; }
    const hipError_t e = hipStreamSynchronize(st);
    if (e != hipSuccess) { printf("kernel %d: %s\n", v, hipGetErrorString(e)); return 1; }
  }
  auto cmp = [&](const char* what, float* p0, float* p1, size_t n) {
    std::vector<float> h0(n), h1(n);
    (void)hipMemcpy(h0.data(), p0, n * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(h1.data(), p1, n * 4, hipMemcpyDeviceToHost);
    size_t bad = 0, first = 0;
    for (size_t i = 0; i < n; ++i) if (memcmp(&h0[i], &h1[i], 4) != 0) { if (!bad) first = i; ++bad; }
    printf("  %-10s %zu values, %zu differ", what, n, bad);
    if (bad) printf(" (first at %zu: %.9g vs %.9g)", first, h0[first], h1[first]);
    printf("   [0] = %.9g\n", h0[0]);
    return bad;
  };
  size_t bad = cmp("cost", cost[0], cost[1], K) + cmp("block_min", bmin[0], bmin[1], nb) + cmp("eta", eta[0], eta[1], nb) + cmp("P_part", eta[0] + nb, eta[1] + nb, (size_t)nb * J);
  {
    std::vector<float> he(nb); (void)hipMemcpy(he.data(), eta[0], nb * 4, hipMemcpyDeviceToHost);
    double m = 0; for (float v : he) m += v;
    printf("  mean eta per 256-sample workgroup %.1f (256 = flat softmax, 1 = peaked: the weighting phase skips dead waves)\n", m / nb);
  }
  printf("K = %d T = %d lambda %g null_action %d: %s\n", K, T, a.lambda_, a.null_action, bad ? "MISMATCH" : "bit-identical");
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int v = 0; v < 2; ++v) {
    for (int i = 0; i < 20; ++i) run(v);
    (void)hipEventRecord(e0, st);
    const int n = 200;
    for (int i = 0; i < n; ++i) run(v);
    (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("  %s: %.1f us per launch (back to back)\n", v ? "two waves per sample group" : "one wave per SIMD          ", ms / n * 1e3);
  }
  return bad ? 2 : 0;
}
