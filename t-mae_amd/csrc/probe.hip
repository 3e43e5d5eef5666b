// Box calibration probes for bench.py ("measured achievable peaks on the box next to the spec
// peaks"): one MI355X differs from the next by several per cent in sustained HBM rate and by ~10 % in the clock it holds
// under matrix load (MI355X_MICROARCH.md, DVFS give-back), so a roofline fraction is only comparable across boxes next to
// what THIS box sustains.  Two kernels, no tuning knobs:
//   tmae_probe_copy  -- 16 bytes per lane streaming copy (the guide's "float4 copy": 6.29 TB/s on the reference box);
//   tmae_probe_mfma  -- back-to-back v_mfma_f32_16x16x32_bf16 on pseudo-random operands held in registers, two waves per
//                       SIMD, 16 independent accumulators (random, not zero, operands: the clock the chip holds depends on
//                       the data).
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(256) void probe_copy_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, int64_t n16) {
  // full tiles of 1024 x 16 bytes: four unconditional loads in flight per lane, then four stores (a load under a
  // predicate is waited for before the next one is issued); the tail tile clamps its loads and predicates its stores
  const int64_t tiles = (n16 + 1023) / 1024;
  for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int64_t i = t * 1024 + threadIdx.x;
    u32x4 v[4];
    if (t * 1024 + 1024 <= n16) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = NT ? __builtin_nontemporal_load(src + i + 256 * j) : src[i + 256 * j];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (NT) __builtin_nontemporal_store(v[j], dst + i + 256 * j);
        else dst[i + 256 * j] = v[j];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = src[i + 256 * j < n16 ? i + 256 * j : n16 - 1];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (i + 256 * j < n16) dst[i + 256 * j] = v[j];
    }
  }
}

int tmae_probe_copy(const void* src, void* dst, int64_t bytes, int nontemporal, void* stream_) {
  (void)hipGetLastError();
  if (!src || !dst || bytes <= 0 || (bytes & 15) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return TMAE_EARG;
  const int64_t n16 = bytes / 16;
  const int64_t want = (n16 + 1023) / 1024;
  const unsigned grid = (unsigned)(want < 256 * 16 ? want : 256 * 16);       // 16 workgroups per CU, grid-stride beyond that
  if (nontemporal)
    hipLaunchKernelGGL(probe_copy_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream_, (const u32x4*)src, (u32x4*)dst, n16);
  else
    hipLaunchKernelGGL(probe_copy_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream_, (const u32x4*)src, (u32x4*)dst, n16);
  return tmae_launch_status();
}

template <int NA, int NB>
__global__ __launch_bounds__(512, 2) void probe_mfma_kernel(int iters, float* __restrict__ sink) {
  // pseudo-random bf16 operands in [0.5, 2) with random signs (exponent bits fixed: no inf / nan, no denormals)
  unsigned s = (blockIdx.x * 512u + threadIdx.x) * 2654435761u + 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s; };
  u32x4 au[NA], bu[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) au[i][j] = (rnd() & 0x807F807Fu) | 0x3F803F80u;
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) bu[i][j] = (rnd() & 0x807F807Fu) | 0x3F003F00u;
  f32x4 acc[NA * NB];
#pragma unroll
  for (int i = 0; i < NA * NB; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int i = 0; i < NA; ++i)
        acc[i * NB + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, au[i]), __builtin_bit_cast(bf16x8, bu[j]),
                                                                  acc[i * NB + j], 0, 0, 0);
  }
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NA * NB; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (t == 123.456f) sink[0] = t;                 // keeps the loop alive; practically never taken
}

int tmae_probe_mfma(int iters, float* sink, int64_t* flops, void* stream_) {
  (void)hipGetLastError();
  if (iters <= 0 || !sink) return TMAE_EARG;
  const unsigned grid = 256;                       // one 512-thread workgroup per CU: two waves on every SIMD
  // (A/B build: TMAE_PROBE_ACC=36 runs 4 x 9 = 36 accumulator tiles per wave -- 144 registers, the dense weight-gradient kernel's count)
  static const int nacc = TMAE_AB_INT("TMAE_PROBE_ACC", 16);
  int per_iter = 16;
#ifdef TMAE_AB
  if (nacc == 36) {
    hipLaunchKernelGGL((probe_mfma_kernel<4, 9>), dim3(grid), dim3(512), 0, (hipStream_t)stream_, iters, sink);
    per_iter = 36;
  } else
#endif
  hipLaunchKernelGGL((probe_mfma_kernel<4, 4>), dim3(grid), dim3(512), 0, (hipStream_t)stream_, iters, sink);
  (void)nacc;
  if (flops) *flops = (int64_t)grid * 8 * (int64_t)iters * per_iter * (2LL * 16 * 16 * 32);
  return tmae_launch_status();
}
