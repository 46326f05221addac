// update.hpp -- launchers implemented in update.hip
#pragma once
#include "common.hpp"
namespace mppi {
template <typename T> int launch_noise_fill_philox(const KArgs<T>& a, T* out, hipStream_t st);
template <typename T> int launch_noise_fill_philox_coloured(const KArgs<T>& a, T* out, hipStream_t st);
template <typename T> int launch_noise_from_ktn(const KArgs<T>& a, const T* in, T* out, hipStream_t st);
template <typename T> int launch_kmppi_interp(const KArgs<T>& a, const T* W, int Thor, int J4out, T* out, hipStream_t st);
template <typename T> int launch_prepare(const KArgs<T>& a, hipStream_t st);
template <typename T> int launch_cost_block_min(const KArgs<T>& a, hipStream_t st);
template <typename T> int launch_weights_partial(const KArgs<T>& a, hipStream_t st);
// the diagonal K3 on fp32 TNK4 rows, with (next_z != NULL) or without the next command's torch-stream draw generated beside it
// (noise_torch.hip); MPPI_E_UNSUPPORTED: not that kind of problem / the launch cannot carry the draw
int launch_weights_partial_rows_f32(const KArgs<float>& a, void* next_z, int kind, uint64_t seed, uint64_t philox_offset, int32_t grid_blocks, hipStream_t st);
template <typename T> int launch_finalize(const KArgs<T>& a, int apply, hipStream_t st);
// K4 of the on-chip command: combines the per-workgroup partial records (a.nkc of them) in block order
template <typename T> int launch_finalize_blocks(const KArgs<T>& a, int apply, hipStream_t st);
template <typename T> int launch_combine(const KArgs<T>& a, const T* rec, int G, hipStream_t st, const T* const* ptrs = nullptr);
// KMPPI's small sequence operators: out = M (R x S) . x (S x nu), optionally plus the rolled nominal sequence
template <typename T> int launch_kmppi_sequences(int R, int S, int nu, const T* M, const T* x, T* out,
                                                   int Troll, const T* U, const T* u_init, T* U_out, hipStream_t st);
// SMPPI: shifted U / action sequence and the base sequence A + U*dt in one launch
template <typename T> int launch_kmppi_after_update(int Tn, int S, int nu, const T* W, const T* Ws, const T* theta, const T* u_init,
                                                    T* U_out, T* theta_s, T* U_s, hipStream_t st);
template <typename T> int launch_smppi_shift(int Tn, int nu, const T* U, const T* u_init, const T* A, T dt, T* U_out, T* A_out,
                                             T* B_out, hipStream_t st);
}  // namespace mppi
