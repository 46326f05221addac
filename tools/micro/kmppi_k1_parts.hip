// Where the time of the KMPPI-fused K1 (csrc/rollout_kmppi.hpp) goes, at C3-sized work (K = 65536, T = 64,
// nx = 16, nu = 12, S = 32): the kernel is built as-is and with one of its three phases knocked out
// (-DMPPI_KMPPI_EXP=1 no row loads | 2 no matrix instructions | 4 no rollout arithmetic), launched on
// synthetic arguments and timed with HIP events.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -mllvm -amdgpu-mfma-vgpr-form [-DMPPI_KMPPI_EXP=n]
//         [-DMPPI_KMPPI_NO_PIPELINE] -I include -I pytorch_mppi_amd/csrc tools/micro/kmppi_k1_parts.hip -o kmppi_parts_n
#include <cstdio>
#include <vector>
#include "dispatch.hpp"
#include "rollout.hpp"
namespace mppi {
bool profile_next_events(hipEvent_t* a, hipEvent_t* b, unsigned long long** t) { *a = *b = nullptr; if (t) *t = nullptr; return false; }
}
using namespace mppi;
int main() {
  const int K = 65536, T = 64, nx = 16, nu = 12, S = 32, J = T * nu;
  KArgs<float> a{};
  a.K = K; a.Tn = T; a.nx = nx; a.nu = nu; a.J = J; a.J4 = J / 4; a.zp = K; a.diag = 1; a.noise_src = MPPI_NOISE_TNK4;
  a.lambda_ = 1.f; a.u_scale = 1.f; a.e_scale = 1.f; a.M = 1; a.n_env = 1; a.S = S; a.fuse = -1;
  auto dev = [](size_t n, float v) { std::vector<float> h(n, v); float* d; hipMalloc(&d, n * 4); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); return d; };
  std::vector<float> hz((size_t)S * 3 * K * 4);
  for (size_t i = 0; i < hz.size(); ++i) hz[i] = 0.001f * (float)((i * 2654435761u >> 20) & 1023) - 0.5f;
  float* z; hipMalloc(&z, hz.size() * 4); hipMemcpy(z, hz.data(), hz.size() * 4, hipMemcpyHostToDevice);
  // rotate over 3 arrays (300 MB): the rows are not in the Infinity Cache when the kernel reads them
  float* zs[3] = {z, nullptr, nullptr};
  for (int i = 1; i < 3; ++i) { hipMalloc(&zs[i], hz.size() * 4); hipMemcpy(zs[i], z, hz.size() * 4, hipMemcpyDeviceToDevice); }
  a.state = dev(nx, 0.1f); a.U = dev(J, 0.01f); a.u_init = dev(nu, 0.f); a.mu = dev(nu, 0.f);
  std::vector<float> hL(nu * nu, 0.f), hS(nu * nu, 0.f);
  for (int n = 0; n < nu; ++n) hL[n * nu + n] = hS[n * nu + n] = 1.f;
  float *L, *Si; hipMalloc(&L, nu * nu * 4); hipMalloc(&Si, nu * nu * 4);
  hipMemcpy(L, hL.data(), nu * nu * 4, hipMemcpyHostToDevice); hipMemcpy(Si, hS.data(), nu * nu * 4, hipMemcpyHostToDevice);
  a.L = L; a.sinv = Si; a.umin = dev(nu, -1.f); a.umax = dev(nu, 1.f);
  a.cost = dev(K, 0.f); a.block_min = dev(K / 64, 0.f);
  a.W = dev((size_t)T * S, 1.f / S); a.theta = dev((size_t)S * nu, 0.f);
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f, sum = 0.f; const int n = 30;
  for (int i = 0; i < n + 5; ++i) {
    a.z = zs[i % 3];
    hipEventRecord(e0, st);
    int rc = launch_rollout_kmppi<IntegratorModel<float, 16, 12>, float>(a, st);
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    if (rc != 0) { printf("launch failed %d\n", rc); return 1; }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (i >= 5) { sum += ms; best = ms < best ? ms : best; }
  }
  float c0; hipMemcpy(&c0, a.cost, 4, hipMemcpyDeviceToHost);
#ifndef MPPI_KMPPI_EXP
#define MPPI_KMPPI_EXP 0
#endif
  printf("KMPPI-fused K1, knocked out = %d (1 loads | 2 MFMAs | 4 rollout)%s: avg %.1f us, min %.1f us (events, incl. dispatch)  cost[0] = %g\n",
         MPPI_KMPPI_EXP,
#ifdef MPPI_KMPPI_NO_PIPELINE
         ", tiles one after the other",
#else
         ", tile i+1 formed under the rollout of tile i",
#endif
         sum / n * 1e3, best * 1e3, c0);
  return 0;
}
