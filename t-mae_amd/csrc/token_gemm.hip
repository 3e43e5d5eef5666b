// Token-list GEMM  y[m, N] = x[m, K] . W[N, K]^T (+ bias)  in bf16 with fp32 accumulation, for the Linear layers of
// the encoder (sst_basic_block.py:45-83: in-projections, out-projection, FFN) and their input gradients
// (dX = dY . W: the same kernel on W^T).  m is 1e5..1e6 tokens, K <= 256, N <= 768: the operation is one streaming
// pass over x and y, so the bound is HBM; the library GEMMs reach ~3 TB/s on these shapes.
//
// x-stationary: a workgroup owns 128 tokens (32 per wavefront).  Every lane loads its 16-byte pieces of the x rows
// straight from global memory into the MFMA operand layout (the contraction index is contiguous in both x and W, so
// no transposition is needed) and keeps them in registers for the whole kernel: x is read from HBM exactly once.
// W streams through LDS in chunks of 64 output columns (double buffered; W is at most a few hundred KB and stays in
// L2), each chunk is contracted against the resident x fragments with v_mfma_f32_16x16x32_bf16 in the swapped
// orientation (rows = output columns n, columns = tokens): a lane then holds 4 consecutive n of one token, i.e. one
// 8-byte store per tile, bias added in fp32 before the rounding.
//
// Measured (MI355X, m = 470 k): 4.3 TB/s at k = n = 128, 3.8 at k = 256 / n = 128, 3.4 at k = 128 / n = 256 -- 10-25 %
// faster than hipBLASLt there; for n >= 256 at k = 256 the library wins (the y stores of a chunk are not overlapped
// with the next chunk's MFMAs: a wave waits for its stores before it may reuse their registers), so the Python side
// routes only the shapes where this kernel is ahead (ops.token_gemm).
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned bf16_bits(float v) {
  return (unsigned)__builtin_bit_cast(unsigned short, __float2bfloat16(v));
}

#define TG_TOK 128   // tokens per workgroup
#define TG_NCH 64    // output columns per W chunk

template <int K, int TT, int NW>
__global__ __launch_bounds__(64 * NW, 2) void token_gemm_kernel(const __hip_bfloat16* __restrict__ x, int64_t ldx,
                                                           const __hip_bfloat16* __restrict__ W,
                                                           const __hip_bfloat16* __restrict__ bias,
                                                           __hip_bfloat16* __restrict__ y, int64_t ldy, int64_t m,
                                                           int N) {
  constexpr int KS = K / 32;             // MFMA k-steps
  constexpr int PITCH = K * 2 + 16;      // bytes per W row in LDS: the +16 spreads the 16 rows of a tile over all banks
  constexpr int CPR = K / 8;             // 16-byte chunks per W row
  constexpr int NTH = 64 * NW;           // threads per workgroup
  constexpr int WL = TG_NCH * CPR / NTH; // chunks per thread per W chunk
  __shared__ __attribute__((aligned(16))) char wl[2][TG_NCH * PITCH];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, i = lane & 15;
  const int64_t tok0 = (int64_t)blockIdx.x * (NW * TT * 16) + w * (TT * 16);

  // resident x fragments: token tile tt, k-step ks -> row tok0 + tt*16 + i, channels ks*32 + g*8 .. +7
  bf16x8 xf[TT][KS];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    const int64_t row = tok0 + tt * 16 + i;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      u32x4 u = u32x4{0u, 0u, 0u, 0u};
      if (row < m) u = *reinterpret_cast<const u32x4*>(x + row * ldx + ks * 32 + g * 8);
      xf[tt][ks] = __builtin_bit_cast(bf16x8, u);
    }
  }
  u32x4 wr[WL];
#define TG_WLOAD(chunk)                                                                               \
  _Pragma("unroll") for (int j = 0; j < WL; ++j) {                                                    \
    const int c_ = tid + NTH * j, row_ = c_ / CPR, ch_ = c_ % CPR;                                    \
    wr[j] = *reinterpret_cast<const u32x4*>(W + (int64_t)((chunk) * TG_NCH + row_) * K + ch_ * 8);     \
  }
#define TG_WSTORE(buf)                                                                                \
  _Pragma("unroll") for (int j = 0; j < WL; ++j) {                                                    \
    const int c_ = tid + NTH * j, row_ = c_ / CPR, ch_ = c_ % CPR;                                    \
    *reinterpret_cast<u32x4*>(&wl[buf][row_ * PITCH + ch_ * 16]) = wr[j];                             \
  }
  const int nch = N / TG_NCH;
  // Ordering inside a chunk: the loads of the NEXT chunk (W, bias) are issued before the MFMAs and CONSUMED (LDS
  // write / unpack) right after them, before this chunk's y stores are issued.  On gfx9 loads and stores share
  // vmcnt and may complete out of order, so a wait on a load while stores are pending is a wait for the stores too:
  // with the stores issued last, they drain during the next chunk's MFMAs instead of stalling this one (measured:
  // the naive order ran at one memory round trip per chunk).
  f32x4 bcur[4], bnext[4];
#define TG_BLOAD(chunk, dst)                                                                                \
  _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) {                                                        \
    const uint2 u_ = bias ? *reinterpret_cast<const uint2*>(bias + (chunk) * TG_NCH + nt * 16 + 4 * g)      \
                          : make_uint2(0u, 0u);                                                             \
    dst[nt] = f32x4{__uint_as_float(u_.x << 16), __uint_as_float(u_.x & 0xFFFF0000u),                       \
                    __uint_as_float(u_.y << 16), __uint_as_float(u_.y & 0xFFFF0000u)};                      \
  }
  TG_WLOAD(0)
  TG_BLOAD(0, bcur)
  TG_WSTORE(0)
  __syncthreads();
  for (int c = 0; c < nch; ++c) {
    const int buf = c & 1;
    const bool more = c + 1 < nch;
    uint2 braw[4];
    if (more) {
      TG_WLOAD(c + 1)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        braw[nt] = bias ? *reinterpret_cast<const uint2*>(bias + (c + 1) * TG_NCH + nt * 16 + 4 * g) : make_uint2(0u, 0u);
    }
    f32x4 acc[4][TT];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) acc[nt][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const bf16x8 a = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(&wl[buf][(nt * 16 + i) * PITCH + (ks * 4 + g) * 16]));
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
          acc[nt][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, xf[tt][ks], acc[nt][tt], 0, 0, 0);
      }
    }
    if (more) {                          // consume the prefetched loads BEFORE any store of this chunk is issued
      TG_WSTORE(buf ^ 1)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        bnext[nt] = f32x4{__uint_as_float(braw[nt].x << 16), __uint_as_float(braw[nt].x & 0xFFFF0000u),
                          __uint_as_float(braw[nt].y << 16), __uint_as_float(braw[nt].y & 0xFFFF0000u)};
    }
    // C layout: rows (= output columns) 4g + r, column (= token) i
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int n0 = c * TG_NCH + nt * 16 + 4 * g;
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        const int64_t row = tok0 + tt * 16 + i;
        if (row < m) {
          const f32x4 v = acc[nt][tt] + bcur[nt];
          uint2 o;
          o.x = bf16_bits(v[0]) | (bf16_bits(v[1]) << 16);
          o.y = bf16_bits(v[2]) | (bf16_bits(v[3]) << 16);
          *reinterpret_cast<uint2*>(y + row * ldy + n0) = o;
        }
      }
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) bcur[nt] = bnext[nt];
    __syncthreads();
  }
}

int tmae_token_gemm(const void* x, int64_t ldx, int64_t m, int k, const void* w, int n, const void* bias, void* y,
                    int64_t ldy, void* stream_) {
  (void)hipGetLastError();
  hipStream_t stream = (hipStream_t)stream_;
  if (m < 0 || (k != 128 && k != 256) || n <= 0 || (n % TG_NCH) || ldx < k || ldy < n || (ldx % 8) || (ldy % 4))
    return TMAE_EARG;
  if (m == 0) return TMAE_OK;
  if (!x || !w || !y) return TMAE_EARG;
  if (((uintptr_t)x & 15) || ((uintptr_t)w & 15) || ((uintptr_t)y & 7) || (bias && ((uintptr_t)bias & 7))) return TMAE_EARG;
#define TG_LAUNCH(KK, TT, NW)                                                                                       \
  hipLaunchKernelGGL((token_gemm_kernel<KK, TT, NW>), dim3(tmae_cdiv(m, NW * TT * 16)), dim3(64 * NW), 0, stream,    \
                     (const __hip_bfloat16*)x, ldx, (const __hip_bfloat16*)w, (const __hip_bfloat16*)bias,          \
                     (__hip_bfloat16*)y, ldy, m, n)
  // 4 waves x 32 tokens per workgroup; 8 waves x 16 tokens (twice the resident waves) measured no better
  if (k == 128) TG_LAUNCH(128, 2, 4); else TG_LAUNCH(256, 2, 4);
#undef TG_LAUNCH
  return tmae_launch_status();
}
