// dispatch.hpp -- per-model launchers (one translation unit per model so hipcc runs in parallel).
// Each returns 0 / hipError_t, or MPPI_E_UNSUPPORTED when no (nx,nu) instantiation exists.
#pragma once
#include "common.hpp"

namespace mppi {
// measurement hook (capi.hip): when profiling is on, hands out a start/stop event pair for the
// next K1 launch; returns false when profiling is off
bool profile_next_events(hipEvent_t* start, hipEvent_t* stop, unsigned long long** tstamp = nullptr);
#define MPPI_DECL_MODEL(name)                                          \
  int rollout_##name(const KArgs<float>& a, hipStream_t st);            \
  int rollout_##name(const KArgs<double>& a, hipStream_t st);           \
  bool supported_##name(int nx, int nu, int hidden);
MPPI_DECL_MODEL(pendulum)
MPPI_DECL_MODEL(integrator)
MPPI_DECL_MODEL(linear_goal)
MPPI_DECL_MODEL(mlp)
#undef MPPI_DECL_MODEL
// fp32 MFMA formulation of the MLP rollout (rollout_mlp_mfma.hip)
bool mlp_mfma_supported(int nx, int nu, int hidden);
int rollout_mlp_mfma(const KArgs<float>& a, hipStream_t st);
// bf16 matrix cores with three-piece operand splitting, fp32-level accuracy (rollout_mlp_split.hip)
bool mlp_split_supported(int nx, int nu, int hidden);
int rollout_mlp_split(const KArgs<float>& a, hipStream_t st);
long long mlp_split_launches();      // successful launches of it in this process (mppi_stat_mlp_split_launches)
// the on-chip command's two-waves-per-sample kernel (rollout_onchip_pair.hpp): counted by the library (capi.hip,
// mppi_stat_onchip_pair_launches); a weak reference, so that stand-alone tools including rollout.hpp link without it
void onchip_pair_launched() __attribute__((weak));
}  // namespace mppi
