// noise_torch.hip -- the values of `torch.randn(K, T, nu)` (fp32) written STRAIGHT into the engine's sample-minor rows-of-4.
//
// rng="torch" (the drop-in default) must give sample k, timestep t, control n the value the reference's own draw puts there
// (reference src/pytorch_mppi/mppi.py:203: one torch.randn(K, T, nu) per command on the controller's device).  Until round 4
// the engine called torch.randn and read the (K,T,nu) array in place through LDS transposition tiles (K1 45 us, K3 40 us at C3
// against 33.5 / 33 us on rows in the engine's layout).  This launch computes the SAME values -- ATen's normal_ kernel is
//     thread idx = blockIdx * 256 + threadIdx of G = 256 * grid threads, rocrand_init(seed, idx, philox_offset),
//     call m: rocrand_normal4 -> element li = idx + G * (4 m + ii) receives component ii          (ii = 0..3)
// (aten/src/ATen/native/cuda/DistributionTemplates.h: distribution_elementwise_grid_stride_kernel, unroll 4; Box-Muller from
// rocrand's own device header, included here, so the arithmetic is the library's; this unit is compiled with
// -ffp-contract=on like the library: with =fast one value in ~10^5 differs in its last bit) -- but hands the ATen threads to
// the lanes BY DESTINATION: thread idx owns elements idx, idx + G, ...; a quad of lanes takes the four threads whose first
// elements are one row-of-4 of one sample, a wave 16 consecutive samples, and every call's four components are transposed
// inside the quad so that each lane stores one whole row-of-4 (16 bytes) where it belongs in the row layout.  No Philox block is
// computed twice and none is wasted: the launch does exactly ATen's arithmetic (68 us at C3 against 73-75 us for torch.randn
// itself, whose 4-byte stores bind it; profiles/r04_noise_torch_stream.txt) and K1 / K3 then run their row kernels.
// The caller advances the generator's offset exactly as ATen would (pytorch_mppi_amd/mppi.py `_torch_stream_fill`) and checks
// the first draw of every shape bit for bit against torch.randn itself (a torch whose kernel enumerates its stream otherwise
// falls back to the in-place path by that check).
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_philox4x32_10.h>
#include <rocrand/rocrand_normal.h>
#include "common.hpp"
#include "dispatch.hpp"

namespace mppi {

// One thread = one ATen thread `idx`, looping over its calls m = 0 .. ncalls-1 like ATen's grid-stride loop does; what differs is
// WHICH idx a lane takes: block (x, y) = (64-sample chunk, row-of-4 jb), thread (kk, c) -> idx = k J + 4 jb + c (k = 64 x + kk), so that
// component 0 of call 0 of a wave is 16 samples x one row-of-4 of the row layout.  Element q = 4 m + ii of the thread is
// idx + q G = (k + q qG + carries, (j + q rG) mod J) with G = qG J + rG: advanced incrementally, no division; the carry is the same
// for the four c of a row-of-4 (J is a multiple of 4), so a wave's stores stay 256 contiguous bytes for every q.
__global__ void __launch_bounds__(256) noise_fill_torch_kernel(float* __restrict__ z, long long K, int J, long long pitch,
                                                               unsigned long long seed, unsigned long long offset, long long G,
                                                               int ncalls, int qG, int rG) {
  const int c = threadIdx.x & 3, kk = threadIdx.x >> 2;
  const long long k = (long long)blockIdx.x * 64 + kk;
  const int jq = 4 * (int)blockIdx.y;                     // first column of the quad's row-of-4
  const long long idx = k * J + jq + c;                   // this lane's ATen thread
  if (idx >= G) return;                                    // (G and jq are multiples of 4: a quad is in or out as a whole)
  // The four lanes of a quad are four consecutive ATen threads; component ii of their four calls are the four columns of ONE
  // row-of-4 of one sample.  After a 4 x 4 transpose inside the quad lane c holds all four columns of component ii = c and stores
  // them as ONE 16-byte word: a quarter of the store instructions (with 4-byte stores the launch was bound by them: 83 us against 68).
  // Lane c therefore follows the destination of element q = 4 m + c of the quad's first thread: (k2, j2), advanced by 4 G per call.
  const int q4 = 4 * qG + (4 * rG) / J, r4 = (4 * rG) % J;       // 4 G = q4 J + r4
  long long k2 = k;
  int j2 = jq;
  for (int i = 0; i < c; ++i) {                            // start at element c G
    j2 += rG;
    k2 += qG;
    if (j2 >= J) { j2 -= J; ++k2; }
  }
  const unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
  unsigned long long ctr = offset / 4ull;          // rocrand_init(seed, idx, offset): counter = (offset / 4, idx), key = seed; + 1 per call
  for (int m = 0; m < ncalls; ++m, ++ctr) {
    // the library's state object costs a second Philox evaluation per call: the block is computed directly (csrc/common.hpp: the
    // same Philox4x32-10) and handed to the library's own Box-Muller
    const U4 blk = philox4x32_10(U4{(unsigned)ctr, (unsigned)(ctr >> 32), (unsigned)idx, (unsigned)((unsigned long long)idx >> 32)}, k0, k1);
    const float2 n01 = rocrand_device::detail::box_muller(blk.x, blk.y);
    const float2 n23 = rocrand_device::detail::box_muller(blk.z, blk.w);
    float a0 = n01.x, a1 = n01.y, a2 = n23.x, a3 = n23.y;              // components 0..3 of THIS lane's thread (= column c)
    {   // transpose: lane c ends up with component c of the quad's four threads = columns 0..3
      const bool o1 = (c & 1) != 0, o2 = (c & 2) != 0;
      const float x = dpp_mov<0xB1>(o1 ? a0 : a1), y = dpp_mov<0xB1>(o1 ? a2 : a3);   // quad_perm [1,0,3,2]: lane ^ 1
      if (o1) { a0 = x; a2 = y; } else { a1 = x; a3 = y; }
      const float u = dpp_mov<0x4E>(o2 ? a0 : a2), w = dpp_mov<0x4E>(o2 ? a1 : a3);   // quad_perm [2,3,0,1]: lane ^ 2
      if (o2) { a0 = u; a1 = w; } else { a2 = u; a3 = w; }
    }
    if (k2 < K) *reinterpret_cast<float4*>(z + ((long long)(j2 >> 2) * pitch + k2) * 4) = make_float4(a0, a1, a2, a3);   // (li < numel <=> k2 < K)
    j2 += r4;
    k2 += q4;
    if (j2 >= J) { j2 -= J; ++k2; }
  }
}

}  // namespace mppi

extern "C" int mppi_noise_fill_torch(void* z, int64_t K, int32_t T, int32_t nu, int64_t pitch, uint64_t seed, uint64_t philox_offset,
                                     int32_t grid_blocks, void* stream) {
  if (z == nullptr || grid_blocks <= 0 || K <= 0 || T <= 0 || nu <= 0) return MPPI_E_BADARG;
  const long long J = (long long)T * nu;
  if (J % 4 != 0 || J / 4 > 65535 || pitch < K) return MPPI_E_UNSUPPORTED;
  const long long G = 256ll * grid_blocks, numel = K * J;
  const long long ncalls = (numel - 1) / (G * 4) + 1;         // ATen: every thread makes the same number of calls
  const long long kchunks = (G / J + 1 + 63) / 64;            // the samples whose rows hold the first G elements, in chunks of 64
  if (ncalls > 0x7fffffff || kchunks > 0x7fffffff || G / J > 0x3fffffff) return MPPI_E_UNSUPPORTED;
  hipLaunchKernelGGL(mppi::noise_fill_torch_kernel, dim3((unsigned)kchunks, (unsigned)(J / 4)), dim3(256), 0, (hipStream_t)stream, (float*)z,
                     (long long)K, (int)J, (long long)pitch, (unsigned long long)seed, (unsigned long long)philox_offset, G, (int)ncalls,
                     (int)(G / J), (int)(G % J));
  return (int)hipGetLastError();
}
