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
//
// Round 5 (ABI 21): the NEXT command's draw inside THIS command's K3 launch (`weights_partial_rows_kernel`).  The generator is
// VALU-bound (171 instructions per four normals: 56 us at C3) and leaves the memory system idle; K3 is HBM-bound (201 MB in 34 us)
// and leaves the VALU idle; two kernels on two streams do not overlap on this stack (profiles/r02_stream_overlap.txt), so the two
// share ONE launch of two kinds of workgroups: K3's (csrc/weights.hpp `k3_diag_block`) and generator workgroups that each take a
// contiguous slice of the generator's (block, call) units.  The
// values depend on (seed, philox offset, launch grid of ATen's kernel) only -- all known when command n is issued, if nobody else
// draws from the generator before command n+1; the host checks exactly that at command n+1 (pytorch_mppi_amd/mppi.py
// `_torch_stream_fill`) and draws again with the stand-alone launch when it does not hold.
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_philox4x32_10.h>
#include <rocrand/rocrand_normal.h>
// (this unit is compiled with -ffp-contract=on for rocrand's Box-Muller above; everything below -- the engine's own headers, K3's
// arithmetic -- is compiled as in every other unit of the library, so that k3_diag_block gives the bits it gives in update.hip)
#pragma clang fp contract(fast)
#include "common.hpp"
#include "dispatch.hpp"
#include "update.hpp"
#include "weights.hpp"

namespace mppi {

// ATen's launch as the engine's lanes see it: G = 256 * grid threads, every thread makes `ncalls` calls; G = qG J + rG.
// One "generator block" (x, y) = 256 lanes = (64-sample chunk x of the samples whose rows hold the first G elements, row-of-4 y).
struct TorchDraw {
  float* z;
  long long K, pitch, G;
  int J, ncalls, qG, rG;
  unsigned kchunks, nrows4;
  unsigned nk3, ngen;            // the draw beside K3 (weights_partial_rows_kernel): workgroups of either kind
  unsigned long long seed, offset;
  // kind 1 (the ENGINE's stream, rng="philox"): the rows of command `offset` (= this command's call + 1) for the samples
  // [k_offset, k_offset + K) -- what noise_fill_philox_kernel (update.hip) writes; unit = (256-sample chunk, row-of-4)
  int kind, seven;
  long long k_offset;
};

// units [u0, u1) of the engine's own stream, lane t of 256: unit u = row jb = u / nchunks of chunk u % nchunks (one division per
// slice, then counted up); two rows at a time -- two independent Philox chains in the instruction stream, as below
__device__ __forceinline__ void philox_stream_units(const TorchDraw& d, unsigned long long u, const unsigned long long u1, const int t) {
  if (u >= u1) return;
  const unsigned nchunks = d.kchunks;
  unsigned jb = (unsigned)(u / nchunks), ch = (unsigned)(u - (unsigned long long)jb * nchunks);
  unsigned n = (unsigned)(u1 - u);
  auto row = [&](unsigned jb_, unsigned ch_, float (&r)[4]) {
    philox_normal4<float>(d.seed, d.offset, d.k_offset + (long long)ch_ * BLOCK + t, (long long)jb_, r, d.seven != 0);
  };
  auto put = [&](unsigned jb_, unsigned ch_, const float (&r)[4]) {
    const long long k = (long long)ch_ * BLOCK + t;
    if (k < d.K) *reinterpret_cast<float4*>(d.z + ((long long)jb_ * d.pitch + k) * 4) = make_float4(r[0], r[1], r[2], r[3]);
  };
  auto next = [&](unsigned& jb_, unsigned& ch_) { if (++ch_ == nchunks) { ch_ = 0; ++jb_; } };
  for (; n >= 2; n -= 2) {
    unsigned jb1 = jb, ch1 = ch;
    next(jb1, ch1);
    float r0[4], r1[4];
    row(jb, ch, r0);
    row(jb1, ch1, r1);
    put(jb, ch, r0);
    put(jb1, ch1, r1);
    jb = jb1; ch = ch1;
    next(jb, ch);
  }
  if (n) {
    float r0[4];
    row(jb, ch, r0);
    put(jb, ch, r0);
  }
}

// Calls [m0, m1) of the ATen threads of generator block (x, y), lane t of 256: thread (kk, c) = (t >> 2, t & 3) -> ATen thread
// idx = k J + 4 y + c (k = 64 x + kk), so that component 0 of call 0 of a wave is 16 samples x one row-of-4 of the row layout.
// Element q = 4 m + ii of the thread is idx + q G = (k + q qG + carries, (j + q rG) mod J): advanced incrementally, no division in the
// loop; the carry is the same for the four c of a row-of-4 (J is a multiple of 4), so a wave's stores stay 256 contiguous bytes.
__device__ __forceinline__ void torch_stream_calls(const TorchDraw& d, const unsigned x, const unsigned y, const int t, const int m0, const int m1) {
  const int c = t & 3, kk = t >> 2;
  const int J = d.J;
  const long long k = (long long)x * 64 + kk;
  const int jq = 4 * (int)y;                               // first column of the quad's row-of-4
  const long long idx = k * J + jq + c;                    // this lane's ATen thread
  if (idx >= d.G) return;                                   // (G and jq are multiples of 4: a quad is in or out as a whole)
  // The four lanes of a quad are four consecutive ATen threads; component ii of their four calls are the four columns of ONE
  // row-of-4 of one sample.  After a 4 x 4 transpose inside the quad lane c holds all four columns of component ii = c and stores
  // them as ONE 16-byte word: a quarter of the store instructions (with 4-byte stores the launch was bound by them: 83 us against 68).
  // Lane c therefore follows the destination of element q = 4 m + c of the quad's first thread: (k2, j2), advanced by 4 G per call.
  const int q4 = 4 * d.qG + (4 * d.rG) / J, r4 = (4 * d.rG) % J;       // 4 G = q4 J + r4
  long long k2 = k;
  int j2 = jq;
  if (m0 == 0) {
    for (int i = 0; i < c; ++i) {                          // start at element c G
      j2 += d.rG;
      k2 += d.qG;
      if (j2 >= J) { j2 -= J; ++k2; }
    }
  } else {                                                 // start at element (4 m0 + c) G: one division per slice
    const long long e = k * J + jq + (long long)(4 * m0 + c) * d.G;
    k2 = e / J;
    j2 = (int)(e - k2 * J);
  }
  const unsigned k0 = (unsigned)d.seed, k1 = (unsigned)(d.seed >> 32);
  unsigned long long ctr = d.offset / 4ull + (unsigned long long)m0;   // rocrand_init(seed, idx, offset): counter = (offset / 4, idx), key = seed; + 1 per call
  const unsigned ilo = (unsigned)idx, ihi = (unsigned)((unsigned long long)idx >> 32);
  // one call: the Philox block is computed directly (csrc/common.hpp: the same Philox4x32-10; the library's state object costs
  // a second evaluation per call) and handed to the library's own Box-Muller; transpose inside the quad; one 16-byte store
  auto finish = [&](const U4 blk) {
    const float2 n01 = rocrand_device::detail::box_muller(blk.x, blk.y);
    const float2 n23 = rocrand_device::detail::box_muller(blk.z, blk.w);
    float a0 = n01.x, a1 = n01.y, a2 = n23.x, a3 = n23.y;              // components 0..3 of THIS lane's thread (= column c)
    {   // transpose: lane c ends up with component c of the quad's four threads = columns 0..3
      const bool o1 = (c & 1) != 0, o2 = (c & 2) != 0;
      const float xx = dpp_mov<0xB1>(o1 ? a0 : a1), yy = dpp_mov<0xB1>(o1 ? a2 : a3);   // quad_perm [1,0,3,2]: lane ^ 1
      if (o1) { a0 = xx; a2 = yy; } else { a1 = xx; a3 = yy; }
      const float u = dpp_mov<0x4E>(o2 ? a0 : a2), w = dpp_mov<0x4E>(o2 ? a1 : a3);     // quad_perm [2,3,0,1]: lane ^ 2
      if (o2) { a0 = u; a1 = w; } else { a2 = u; a3 = w; }
    }
    if (k2 < d.K) *reinterpret_cast<float4*>(d.z + ((long long)(j2 >> 2) * d.pitch + k2) * 4) = make_float4(a0, a1, a2, a3);   // (li < numel <=> k2 < K)
    j2 += r4;
    k2 += q4;
    if (j2 >= J) { j2 -= J; ++k2; }
  };
  int m = m0;
  // two calls at a time: two independent Philox chains in one instruction stream.  Beside K3's waves (weights_partial_rows_kernel
  // below) only one or two generator waves share a SIMD, and a lone chain of dependent multiplies leaves the VALU idle between
  // its rounds; the stand-alone launch (eight waves per SIMD) does not care either way.  Same blocks, same order of stores.
  for (; m + 1 < m1; m += 2, ctr += 2) {
    const unsigned long long c1 = ctr + 1;
    const U4 b0 = philox4x32_10(U4{(unsigned)ctr, (unsigned)(ctr >> 32), ilo, ihi}, k0, k1);
    const U4 b1 = philox4x32_10(U4{(unsigned)c1, (unsigned)(c1 >> 32), ilo, ihi}, k0, k1);
    finish(b0);
    finish(b1);
  }
  if (m < m1) finish(philox4x32_10(U4{(unsigned)ctr, (unsigned)(ctr >> 32), ilo, ihi}, k0, k1));
}

// the stand-alone launch: block (x, y) = generator block (x, y), every thread loops over all its calls like ATen's grid-stride loop
__global__ void __launch_bounds__(256) noise_fill_torch_kernel(const TorchDraw d) {
  torch_stream_calls(d, blockIdx.x, blockIdx.y, (int)threadIdx.x, 0, d.ncalls);
}

// K3 of command n + the draw of command n+1 in ONE launch of two kinds of workgroups (d.z != NULL; a 1-D grid): K3 workgroups --
// k3_diag_block on tile (kc, jt), HBM-bound -- and generator workgroups -- a contiguous slice of the generator's units, unit =
// (generator block, call) in call-minor order, VALU-bound -- interleaved by index, so that the dispatcher keeps a mix of both
// resident on the chip and fills a slot that either kind frees with whatever comes next.  No barrier is shared between the two
// kinds and neither waits for the other: the launch ends when the slower resource (memory or VALU) has done its part.  (First
// form of this kernel, same round: four K3 waves + four generator waves PER workgroup, meeting at K3's barriers -- 81 us at C3
// against 34 + 59 for the two launches: a workgroup lives as long as its slower half, and the registers of the other half idle
// with it; profiles/r05_draw_ahead_forms.txt.)  d.z == NULL: the plain K3 on K3's own (nkc, tiles, environments) grid.
template <int R>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4)))
weights_partial_rows_kernel(const KArgs<float> a, const TorchDraw d) {
  int kc = (int)blockIdx.x, jt = (int)blockIdx.y;
  bool gen = false;
  unsigned i = 0;
  const unsigned ngen = d.ngen;
  if (d.z != nullptr) {
    // index pattern: one K3 workgroup, then r = ngen / nk3 generator workgroups, nk3 times over; what is left generates
    const unsigned nk3 = d.nk3, r = ngen / nk3, p = r + 1, b = blockIdx.x, q = b / p, rem = b - q * p;
    if (q < nk3) { gen = rem != 0; i = gen ? q * r + rem - 1 : q; } else { gen = true; i = nk3 * r + (b - nk3 * p); }
    kc = (int)(i % (unsigned)a.nkc);
    jt = (int)(i / (unsigned)a.nkc);
  }
  if (!gen) {
    k3_diag_block<float, MPPI_NOISE_TNK4, R>(a, kc, jt);
    return;
  }
  const unsigned long long units = (unsigned long long)d.kchunks * d.nrows4 * (unsigned)d.ncalls;
  unsigned long long u = units * i / ngen;
  const unsigned long long u1 = units * (i + 1) / ngen;
  if (d.kind == 1) {
    philox_stream_units(d, u, u1, (int)threadIdx.x);
    return;
  }
  while (u < u1) {
    const unsigned gb = (unsigned)(u / (unsigned)d.ncalls);
    const int m0 = (int)(u - (unsigned long long)gb * (unsigned)d.ncalls);
    const unsigned long long left = u1 - u;
    const int m1 = (unsigned long long)(d.ncalls - m0) <= left ? d.ncalls : m0 + (int)left;
    torch_stream_calls(d, gb % d.kchunks, gb / d.kchunks, (int)threadIdx.x, m0, m1);
    u += (unsigned long long)(m1 - m0);
  }
}

static int torch_draw_args(TorchDraw& d, void* z, int64_t K, int64_t J, int64_t pitch, uint64_t seed, uint64_t philox_offset, int32_t grid_blocks) {
  if (z == nullptr || grid_blocks <= 0 || K <= 0 || J <= 0) return MPPI_E_BADARG;
  if (J % 4 != 0 || J / 4 > 65535 || pitch < K) return MPPI_E_UNSUPPORTED;
  const long long G = 256ll * grid_blocks, numel = K * J;
  const long long ncalls = (numel - 1) / (G * 4) + 1;         // ATen: every thread makes the same number of calls
  const long long kchunks = (G / J + 1 + 63) / 64;            // the samples whose rows hold the first G elements, in chunks of 64
  if (ncalls > 0x7fffffff || kchunks > 0x7fffffff || G / J > 0x3fffffff) return MPPI_E_UNSUPPORTED;
  d.z = (float*)z; d.K = K; d.pitch = pitch; d.G = G; d.J = (int)J; d.ncalls = (int)ncalls; d.qG = (int)(G / J); d.rG = (int)(G % J);
  d.kchunks = (unsigned)kchunks; d.nrows4 = (unsigned)(J / 4); d.seed = seed; d.offset = philox_offset;
  return 0;
}

// THE diagonal K3 on fp32 rows in the engine's layout, with or without the next draw beside it: one kernel, so that a command's
// bits do not depend on whether its K3 launch also generated (this unit's floating-point contraction mode differs from
// update.hip's: the same source in the two units is not the same arithmetic to the last bit).  next_z == NULL: 256 threads per
// workgroup, K3 alone.  Otherwise the next draw must have the shape and pitch of the rows this K3 reads; MPPI_E_UNSUPPORTED when
// the launch cannot carry it (several environments on grid.z, (T nu) % 4 != 0, next_z == z): the caller runs K3 alone.
int launch_weights_partial_rows_f32(const KArgs<float>& a, void* next_z, int kind, uint64_t seed, uint64_t philox_offset, int32_t grid_blocks, hipStream_t st) {
  if (a.noise_src != MPPI_NOISE_TNK4 || !(a.diag || a.coloured)) return MPPI_E_UNSUPPORTED;
  TorchDraw d{};
  if (next_z != nullptr) {
    if (a.n_env > 1 || next_z == (const void*)a.z) return MPPI_E_UNSUPPORTED;
    if (kind == MPPI_NEXT_DRAW_PHILOX) {
      // the engine's own stream: rows of command `philox_offset` (the caller passes call + 1), uncoloured
      if (a.coloured) return MPPI_E_UNSUPPORTED;
      d.z = (float*)next_z; d.K = a.K; d.pitch = a.zp; d.J = a.J; d.ncalls = 1; d.kchunks = (unsigned)((a.K + BLOCK - 1) / BLOCK);
      d.nrows4 = (unsigned)a.J4; d.seed = seed; d.offset = philox_offset; d.kind = 1; d.k_offset = a.k_offset; d.seven = a.seven;
    } else if (kind == MPPI_NEXT_DRAW_TORCH) {
      if (a.J % 4 != 0) return MPPI_E_UNSUPPORTED;
      if (int e = torch_draw_args(d, next_z, a.K, a.J, a.zp, seed, philox_offset, grid_blocks)) return e;
    } else {
      return MPPI_E_BADARG;
    }
  }
  const unsigned njt = (unsigned)((a.J4 * 4 + UPD_TJ - 1) / UPD_TJ);
  dim3 grid(a.nkc, njt, a.n_env);
  if (next_z != nullptr) {
    // generator workgroups: four per K3 tile, a dozen calls per lane each (many short workgroups: the dispatcher balances them over
    // the CUs; with one long one per K3 tile the launch took as long as the CU that happened to get four of them -- C3, per command:
    // 0.140 ms at 1, 0.117 at 2, 0.1145 at 4, 0.118 at 8, 0.127 at 16 per tile; 0.127 without the draw-ahead: profiles/r05_draw_ahead_forms.txt)
    static const int ratio_env = [] { const char* e = getenv("MPPI_NEXT_GEN_PER_K3"); return e ? atoi(e) : 0; }();   // tools/ knob
    const unsigned long long units = (unsigned long long)d.kchunks * d.nrows4 * (unsigned)d.ncalls;
    d.nk3 = (unsigned)a.nkc * njt;
    unsigned long long ng = (unsigned long long)d.nk3 * (ratio_env > 0 ? ratio_env : 4);
    if (ng > units / 2) ng = units / 2 / d.nk3 * d.nk3;       // at least two calls (rows, for the engine's stream) per generator workgroup
    if (ng < d.nk3) ng = d.nk3;
    d.ngen = (unsigned)ng;
    grid = dim3(d.nk3 + d.ngen, 1, 1);
  }
  switch (a.R) {
    case 1: hipLaunchKernelGGL((weights_partial_rows_kernel<1>), grid, dim3(BLOCK), 0, st, a, d); break;
    case 2: hipLaunchKernelGGL((weights_partial_rows_kernel<2>), grid, dim3(BLOCK), 0, st, a, d); break;
    case 4: hipLaunchKernelGGL((weights_partial_rows_kernel<4>), grid, dim3(BLOCK), 0, st, a, d); break;
    default: return MPPI_E_UNSUPPORTED;      // (R = 8 is a tools/ knob: the plain kernel of update.hip)
  }
  return (int)hipGetLastError();
}

}  // namespace mppi

extern "C" int mppi_noise_fill_torch(void* z, int64_t K, int32_t T, int32_t nu, int64_t pitch, uint64_t seed, uint64_t philox_offset,
                                     int32_t grid_blocks, void* stream) {
  if (T <= 0 || nu <= 0) return MPPI_E_BADARG;
  mppi::TorchDraw d;
  if (int e = mppi::torch_draw_args(d, z, K, (int64_t)T * nu, pitch, seed, philox_offset, grid_blocks)) return e;
  hipLaunchKernelGGL(mppi::noise_fill_torch_kernel, dim3(d.kchunks, d.nrows4), dim3(256), 0, (hipStream_t)stream, d);
  return (int)hipGetLastError();
}
