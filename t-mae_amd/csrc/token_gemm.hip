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
// 8-byte piece per tile; the W rows of a chunk are permuted in LDS so that a PAIR of tiles gives a lane 8 consecutive
// columns and the four lanes of a token (g = 0..3) 32 consecutive ones: every store instruction writes a contiguous
// 64-byte segment per token (round 3: with 16 consecutive columns per lane an instruction wrote 16-byte pieces with
// 16-byte holes, and the L2 request rate -- not bytes -- bounded the stores: csrc/token_gemm_wreg.hip).  Bias added in
// fp32 before the rounding.
//
// Measured (MI355X, m = 470 k): 5.4 TB/s at k = n = 128, 5.3 at k = 128 / n = 256, 4.1 at k = n = 256, 3.1 at
// k = 256 / n = 512 -- ahead of hipBLASLt (3.0-3.4 TB/s) on every shape it supports; a trivial copy kernel with the
// same 1 : 2 read : write mix moves the n = 512 traffic at 5.8 TB/s, which is the practical ceiling.
// What the ablations of the n = 512 case say (each term removed alone, 250 us total): x loads 100 us, stores 70 us,
// MFMAs 40 us, W loads 30 us, LDS reads 12 us -- nearly additive, i.e. the phases of a workgroup do not overlap, and
// the x loads of a starting workgroup queue behind the other workgroups' stores (without stores they cost 30 us).
// Hence, for the heavy shapes, csrc/token_gemm_wreg.hip (round 3: W in registers, x through an LDS-DMA ring, full-line
// stores: 5.0-6.1 TB/s); it replaced round 1's W-resident-in-LDS kernel (4.2-4.8 TB/s).  This kernel keeps the small
// token lists (< 32 k tokens), contraction 512 and the GELU-derivative epilogue.
// The store width also decides: with one 8-byte store per tile (32-byte segments per row) the kernel ran at
// 2.9 TB/s, the partial-line requests saturate the L2 request rate long before its bandwidth.
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned bf16_bits(float v) {
  return (unsigned)__builtin_bit_cast(unsigned short, __float2bfloat16(v));
}

// d/dx of the exact (erf) GELU: Phi(x) + x phi(x).  erf through Abramowitz-Stegun 7.1.26 (|error| < 1.5e-7, far
// below the bf16 rounding of the result), which shares its exponential with the density: one exp, one rcp, a few FMAs
// -- the library erff costs ~40 live registers per value in this fully unrolled epilogue.
__device__ __forceinline__ float dgelu(float x) {
  const float e = __expf(-0.5f * x * x);                          // = exp(-u^2), u = x / sqrt(2)
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * 0.70710678f * fabsf(x));
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erf_abs = 1.0f - poly * e;
  const float cdf = 0.5f * (1.0f + copysignf(erf_abs, x));
  return cdf + x * e * 0.39894228040143267794f;
}

// POS variants (in-projections of the attention layers, q = k = x + pos, sst_basic_block.py:45-47): the position
// embedding is a function of the token's cell inside its 8 x 8 window and separable -- pos[cell] = [ex[xc] | ey[yc]]
// (spt_backbone.py:186-224) -- so (x + pos) W^T = x W^T + Tx[xc] + Ty[yc] with two 8-row tables per layer.  The tables
// ride along as 32 extra contraction columns of W (hi and lo bf16 halves of Tx, Ty: the sum is exact to ~2^-17) and the
// kernel contracts them against a one-hot fragment it makes from the token's cell byte (xc | yc << 3): one more MFMA
// k-step on an idle matrix core instead of an elementwise pass that reads and writes [m, d].
__device__ __forceinline__ bf16x8 pos_onehot(unsigned cell, int g) {
  const unsigned sel = (g & 1) ? (cell >> 3) & 7u : cell & 7u;
  const unsigned v = (sel & 1u) ? 0x3F800000u : 0x00003F80u, wi = sel >> 1;
  const u32x4 u = {wi == 0u ? v : 0u, wi == 1u ? v : 0u, wi == 2u ? v : 0u, wi == 3u ? v : 0u};
  return __builtin_bit_cast(bf16x8, u);
}

#define TG_NT 2      // cache policy of the y stores: nt (written once, streamed; measured 7 % faster than the default)
#define TG_NCH (NTC * 16)   // output columns per W chunk (NTC 16-column tiles: 4, or 2 for K = 512)

template <int K, int TT, int NW, int NTC, int EPI, int NCHT, bool POS = false>
__global__ __launch_bounds__(64 * NW, 2) void token_gemm_kernel(const __hip_bfloat16* __restrict__ x, int64_t ldx,
                                                           const __hip_bfloat16* __restrict__ W,
                                                           const __hip_bfloat16* __restrict__ bias,
                                                           __hip_bfloat16* __restrict__ y, int64_t ldy, int64_t m,
                                                           int N, unsigned ybytes,
                                                           const __hip_bfloat16* __restrict__ aux,
                                                           const uint8_t* __restrict__ cells) {
  constexpr int KX = K / 32;             // k-steps read from x
  constexpr int KA = K + (POS ? 32 : 0); // W row length: the contraction incl. the position columns
  constexpr int KS = KA / 32;            // MFMA k-steps
  constexpr int PITCH = KA * 2 + 16;     // bytes per W row in LDS: the +16 spreads the 16 rows of a tile over all banks
  constexpr int CPR = KA / 8;            // 16-byte chunks per W row
  constexpr int NTH = 64 * NW;           // threads per workgroup
  constexpr int WL = TG_NCH * CPR / NTH; // chunks per thread per W chunk
  static_assert(TG_NCH * CPR % NTH == 0, "W chunk must divide over the threads");
  __shared__ __attribute__((aligned(16))) char wl[2][TG_NCH * PITCH];
  __shared__ __attribute__((aligned(16))) char bl[2][TG_NCH * 2];       // the chunk's bias values travel with its W rows
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, i = lane & 15;
  const int64_t tok0 = (int64_t)blockIdx.x * (NW * TT * 16) + w * (TT * 16);

  // resident x fragments: token tile tt, k-step ks -> row tok0 + tt*16 + i, channels ks*32 + g*8 .. +7
  bf16x8 xf[TT][KS];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    const int64_t row = tok0 + tt * 16 + i;
#pragma unroll
    for (int ks = 0; ks < KX; ++ks) {
      u32x4 u = u32x4{0u, 0u, 0u, 0u};
      if (row < m) u = *reinterpret_cast<const u32x4*>(x + row * ldx + ks * 32 + g * 8);
      xf[tt][ks] = __builtin_bit_cast(bf16x8, u);
    }
    if constexpr (POS) xf[tt][KX] = pos_onehot(row < m ? cells[row] : 0u, g);
  }
  u32x4 wr[WL], br;
#define TG_WLOAD(chunk)                                                                               \
  _Pragma("unroll") for (int j = 0; j < WL; ++j) {                                                    \
    const int c_ = tid + NTH * j, row_ = c_ / CPR, ch_ = c_ % CPR;                                    \
    wr[j] = *reinterpret_cast<const u32x4*>(W + (int64_t)((chunk) * TG_NCH + row_) * KA + ch_ * 8);    \
  }                                                                                                   \
  br = *reinterpret_cast<const u32x4*>(bias + (chunk) * TG_NCH + (tid & (TG_NCH / 8 - 1)) * 8);
#define TG_WSTORE(buf)                                                                                \
  _Pragma("unroll") for (int j = 0; j < WL; ++j) {                                                    \
    const int c_ = tid + NTH * j, row_ = c_ / CPR, ch_ = c_ % CPR;                                    \
    *reinterpret_cast<u32x4*>(&wl[buf][(16 * (2 * (row_ >> 5) + ((row_ >> 2) & 1)) + 4 * ((row_ >> 3) & 3) + (row_ & 3)) * PITCH + ch_ * 16]) = wr[j]; \
  }                                                                                                   \
  *reinterpret_cast<u32x4*>(&bl[buf][(tid & (TG_NCH / 8 - 1)) * 16]) = br;   /* every thread (same data): under `if (tid < 8)` the compiler sinks the LOAD into the branch and follows it with vmcnt(0) */
  const int nch = NCHT ? NCHT : N / TG_NCH;     // NCHT > 0: the chunk loop is fully unrolled (straight-line code)
  // Ordering inside a chunk.  On gfx9 loads and stores share vmcnt and retire in order, so a wait for a load also
  // waits for every store issued BEFORE it.  The loads of chunk c+2 (W, bias) are therefore issued at the end of
  // chunk c, right BEFORE chunk c's y stores: when chunk c+1 needs them (LDS write after its MFMAs) the wait is
  // vmcnt(#stores of chunk c) -- the stores keep draining for two chunks instead of having to reach L2 within one
  // MFMA phase (measured before: ~4 us per chunk per workgroup, the store round trip, for 0.4 us of MFMA work).
  // No branches around the loads / LDS writes (the last chunks re-load a clamped chunk index): the compiler's
  // wait-count bookkeeping turns conservative (vmcnt(0)) at control-flow joins.  Also:
  //  * the stores are buffer stores whose only VGPR sources are the packed data and a per-lane offset that never
  //    changes (the chunk offset travels in an SGPR): no address temporaries to recycle;
  //  * the chunk loop is unrolled by two with two sets of data registers, and a set is kept formally alive (empty
  //    asm) until the MFMAs of the following chunk are done: its stores drain under those MFMAs instead of
  //    stalling the wave at the top of the loop (measured before: one memory round trip per chunk).
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, (int)ybytes, 0x00020000);
  // EPI == 2 (DGELU): y = (x . W^T) * gelu'(aux), aux [m, N] bf16 with y's pitch (the pre-activation saved by the
  // forward): the lane's 16-byte pieces of the NEXT chunk are fetched right after this chunk's were consumed.
  const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(EPI == 2 ? aux : y), 0, (int)ybytes, 0x00020000);
  u32x4 auxr[NTC / 2][TT];
  int voff[TT];                                       // byte offset of (row, 16g) in y; >= ybytes drops the store
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    const int64_t row = tok0 + tt * 16 + i;
    voff[tt] = row < m ? (int)((row * ldy + 8 * g) * 2) : (int)ybytes;
  }
  u32x4 oA[NTC / 2][TT], oB[NTC / 2][TT];             // [16-byte piece of the lane's 4*NTC columns][token tile]
#pragma unroll
  for (int h = 0; h < NTC / 2; ++h)
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) { oA[h][tt] = u32x4{0u, 0u, 0u, 0u}; oB[h][tt] = u32x4{0u, 0u, 0u, 0u}; }
  TG_WLOAD(0)
  if constexpr (EPI == 2) {
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
      for (int h = 0; h < NTC / 2; ++h) auxr[h][tt] = __builtin_amdgcn_raw_buffer_load_b128(arsrc, voff[tt] + h * 64, 0, 0);
  }
  TG_WSTORE(0)
  {
    const int c1_ = nch > 1 ? 1 : 0;
    TG_WLOAD(c1_)
    // the compiler merges the wait-count state of the loop entry with that of the back edge and waits for the more
    // conservative one: give the entry the same shape -- the loads followed by TT * NTC / 2 stores (out of range:
    // dropped by the buffer bounds check) -- or every first chunk of the unrolled pair drains the previous stores
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
      for (int h = 0; h < NTC / 2; ++h)
        __builtin_amdgcn_raw_buffer_store_b128(oB[h][tt], yrsrc, (int)ybytes, 0, 0);
  }
  __syncthreads();

#define TG_KEEP(o)                                                                                          \
  _Pragma("unroll") for (int h = 0; h < NTC / 2; ++h)                                                       \
    _Pragma("unroll") for (int tt = 0; tt < TT; ++tt)                                                       \
      asm volatile("" ::"v"(o[h][tt].x), "v"(o[h][tt].y), "v"(o[h][tt].z), "v"(o[h][tt].w));

#define TG_CHUNK(c, buf, ocur, oprev)                                                                       \
  {                                                                                                         \
    const bool more = (c) + 1 < nch;                                                                        \
    const int c2_ = (c) + 2 < nch ? (c) + 2 : nch - 1;                                                      \
    f32x4 acc[NTC][TT];                                                                                     \
    uint2 bcur[NTC];           /* packed bf16 bias of the lane's 4 columns per tile (zeros if none) */     \
    _Pragma("unroll") for (int nt = 0; nt < NTC; ++nt)                                                      \
      bcur[nt] = *reinterpret_cast<const uint2*>(&bl[buf][(32 * (nt >> 1) + 8 * g + 4 * (nt & 1)) * 2]);                     \
    _Pragma("unroll") for (int nt = 0; nt < NTC; ++nt)                                                      \
      _Pragma("unroll") for (int tt = 0; tt < TT; ++tt) acc[nt][tt] = f32x4{0.f, 0.f, 0.f, 0.f};            \
    _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                                                     \
      _Pragma("unroll") for (int nt = 0; nt < NTC; ++nt) {                                                  \
        const bf16x8 a_ = __builtin_bit_cast(                                                               \
            bf16x8, *reinterpret_cast<const u32x4*>(&wl[buf][(nt * 16 + i) * PITCH + (ks * 4 + g) * 16]));   \
        _Pragma("unroll") for (int tt = 0; tt < TT; ++tt)                                                   \
          acc[nt][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_, xf[tt][ks], acc[nt][tt], 0, 0, 0);      \
      }                                                                                                     \
    }                                                                                                       \
    TG_KEEP(oprev) /* the previous chunk's store data stayed untouched while its stores drained */          \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    TG_WSTORE((buf) ^ 1) /* chunk c+1 (loaded before the previous chunk's stores) -> the other buffer */     \
    /* C layout: rows 4g + r of tile t = output columns 32 (t / 2) + 8g + 4 (t % 2) + r (W rows are permuted in LDS), */ \
    /* column = token i: a tile pair = 8 consecutive columns per lane, 64 contiguous bytes per token and store */      \
    _Pragma("unroll") for (int tt = 0; tt < TT; ++tt)                                                       \
      _Pragma("unroll") for (int h = 0; h < NTC / 2; ++h) {                                                 \
        unsigned w_[4];                                                                                     \
        _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                                     \
          const int nt = 2 * h + q;                                                                         \
          f32x4 v_ = acc[nt][tt] + f32x4{__uint_as_float(bcur[nt].x << 16), __uint_as_float(bcur[nt].x & 0xFFFF0000u), \
                                         __uint_as_float(bcur[nt].y << 16), __uint_as_float(bcur[nt].y & 0xFFFF0000u)}; \
          if constexpr (EPI == 2) {                                                                         \
            const unsigned a0_ = auxr[h][tt][2 * q], a1_ = auxr[h][tt][2 * q + 1];                          \
            v_[0] *= dgelu(__uint_as_float(a0_ << 16)); v_[1] *= dgelu(__uint_as_float(a0_ & 0xFFFF0000u));   \
            v_[2] *= dgelu(__uint_as_float(a1_ << 16)); v_[3] *= dgelu(__uint_as_float(a1_ & 0xFFFF0000u));   \
          }                                                                                                 \
          w_[2 * q] = bf16_bits(v_[0]) | (bf16_bits(v_[1]) << 16);                                           \
          w_[2 * q + 1] = bf16_bits(v_[2]) | (bf16_bits(v_[3]) << 16);                                       \
        }                                                                                                   \
        ocur[h][tt] = u32x4{w_[0], w_[1], w_[2], w_[3]};                                                    \
      }                                                                                                     \
    if constexpr (EPI == 2) {                                                                               \
      if (more) {                                                                                           \
        _Pragma("unroll") for (int tt = 0; tt < TT; ++tt)                                                   \
          _Pragma("unroll") for (int h = 0; h < NTC / 2; ++h)                                               \
            auxr[h][tt] = __builtin_amdgcn_raw_buffer_load_b128(arsrc, voff[tt] + h * 64, ((c) + 1) * (TG_NCH * 2), 0); \
      }                                                                                                     \
    }                                                                                                       \
    /* issue the loads of chunk c+2 BEFORE this chunk's stores */                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    TG_WLOAD(c2_)                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    _Pragma("unroll") for (int tt = 0; tt < TT; ++tt)                                                       \
      _Pragma("unroll") for (int h = 0; h < NTC / 2; ++h)                                                   \
        __builtin_amdgcn_raw_buffer_store_b128(ocur[h][tt], yrsrc, voff[tt] + h * 64, (c) * (TG_NCH * 2), (NCHT || K == 256) ? TG_NT : 0); \
    /* gfx950: a VALU write to the data registers of a 16-byte buffer store in the two issue slots after it corrupts the  \
       store (seen in round 3: one dword, lanes 12-15 of every 16, a few tiles per launch).  The compiler pads that hazard \
       only for stores WITHOUT an SGPR offset (GCNHazardRecognizer::createsVALUHazard), and these stores carry the chunk    \
       offset in an SGPR: pad by hand */                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    asm volatile("s_nop 1" ::: "memory");                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    __syncthreads();                                                                                        \
  }

  if constexpr (NCHT > 0) {
    // the frequent widths (N = 128, 256, 512): no loop, so the compiler's wait counts are exact -- across a back edge it
    // merges the states of loop entry and latch conservatively and every other chunk drains the stores before it
#pragma unroll
    for (int c = 0; c < NCHT; c += 2) {
      TG_CHUNK(c, 0, oA, oB)
      if (c + 1 < NCHT) TG_CHUNK(c + 1, 1, oB, oA)
    }
  } else {
    for (int c = 0; c < nch; c += 2) {
      TG_CHUNK(c, 0, oA, oB)
      if (c + 1 < nch) TG_CHUNK(c + 1, 1, oB, oA)
    }
  }
#undef TG_CHUNK
#undef TG_KEEP
}

static int token_gemm_launch(const void* x, int64_t ldx, int64_t m, int k, const void* w, int n, const void* bias, void* y,
                             int64_t ldy, const void* aux, const void* cells, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (m < 0 || (k != 32 && k != 64 && k != 128 && k != 256 && k != 512) || n <= 0 || (n % 64) || ldx < k || ldy < n || (ldx % 8) || (ldy % 8))
    return TMAE_EARG;
  if ((k == 32 || k == 64) && (cells || aux)) return TMAE_EARG;
  if (cells && (aux || k == 512)) return TMAE_EARG;
  if (m == 0) return TMAE_OK;
  if (!x || !w || !y || !bias) return TMAE_EARG;          // no bias: pass a zero vector (keeps the kernel branch-free)
  if (((uintptr_t)x & 15) || ((uintptr_t)w & 15) || ((uintptr_t)y & 15) || ((uintptr_t)bias & 7)) return TMAE_EARG;
  const int64_t ybytes = ((m - 1) * ldy + n) * 2;                 // buffer stores address y with 32-bit byte offsets
  if (ybytes >= (int64_t)1 << 31) return TMAE_EARG;
#define TG_ARGS (const __hip_bfloat16*)x, ldx, (const __hip_bfloat16*)w, (const __hip_bfloat16*)bias, (__hip_bfloat16*)y, \
                ldy, m, n, (unsigned)ybytes
#define TG_LAUNCH(KK, TT, NW, NTC, NCHT)                                                                            \
  do {                                                                                                              \
    const dim3 grid_(tmae_cdiv(m, NW * TT * 16)), blk_(64 * NW);                                                    \
    if (aux)                                                                                                        \
      hipLaunchKernelGGL((token_gemm_kernel<KK, TT, NW, NTC, 2, NCHT>), grid_, blk_, 0, stream, TG_ARGS,            \
                         (const __hip_bfloat16*)aux, (const uint8_t*)nullptr);                                      \
    else                                                                                                            \
      hipLaunchKernelGGL((token_gemm_kernel<KK, TT, NW, NTC, 0, NCHT>), grid_, blk_, 0, stream, TG_ARGS,            \
                         (const __hip_bfloat16*)nullptr, (const uint8_t*)nullptr);                                  \
  } while (0)
#define TG_LAUNCH_POS(KK, NCHT)                                                                                     \
  hipLaunchKernelGGL((token_gemm_kernel<KK, 2, 4, 4, 0, NCHT, true>), dim3(tmae_cdiv(m, 128)), dim3(256), 0, stream, \
                     TG_ARGS, (const __hip_bfloat16*)nullptr, (const uint8_t*)cells)
  // heavy shapes: the W-in-registers kernel of token_gemm_wreg.hip (x read once for all N columns, LDS-DMA ring)
  {
    static const int wreg = TMAE_AB_INT("TMAE_TG_WREG", 1);
    if (!aux && wreg != 0 && m >= (n > 512 ? 65536 : 32768)) {
      const int rc = tmae_token_gemm_wreg(x, ldx, m, k, w, n, bias, (const uint8_t*)cells, y, ldy, 0, stream_);
      if (rc != TMAE_EARG) return rc;
    }
  }
  // 4 waves x 32 tokens, 64-column chunks; contraction 512: 16 tokens per wave (64 x registers) and 32-column chunks
  // (the frequent widths run the fully unrolled chunk loop)
  if (cells) {
    if (k == 128) {
      if (n == 128) TG_LAUNCH_POS(128, 2); else if (n == 256) TG_LAUNCH_POS(128, 4);
      else if (n == 384) TG_LAUNCH_POS(128, 6); else TG_LAUNCH_POS(128, 0);
    } else {
      TG_LAUNCH_POS(256, 0);
    }
  } else if (k == 32) {
    // the VFE's first Linear on the hi | lo split point features (temporal_dyn_vfe.py:110-112; 16 + 16 columns -> 64): one
    // k-step, one 64-column chunk; the library ran this 174 MB pass at 1.4 TB/s
    if (n == 64) TG_LAUNCH(32, 2, 4, 4, 1); else TG_LAUNCH(32, 2, 4, 4, 0);
  } else if (k == 64) {
    // the VFE's second Linear (64 -> 128 over every point of the frame, temporal_dyn_vfe.py:110-112 with make_fc_layers):
    // 348 MB per frame that the library moved at 3.6 TB/s
    if (n == 128) TG_LAUNCH(64, 2, 4, 4, 2); else TG_LAUNCH(64, 2, 4, 4, 0);
  } else if (k == 128) {
    if (n == 128) TG_LAUNCH(128, 2, 4, 4, 2); else if (n == 256) TG_LAUNCH(128, 2, 4, 4, 4); else TG_LAUNCH(128, 2, 4, 4, 0);
  } else if (k == 256) {
    if (n == 256) TG_LAUNCH(256, 2, 4, 4, 4); else if (n == 512) TG_LAUNCH(256, 2, 4, 4, 8); else TG_LAUNCH(256, 2, 4, 4, 0);
  } else {
    TG_LAUNCH(512, 1, 4, 2, 0);
  }
#undef TG_LAUNCH
#undef TG_LAUNCH_POS
#undef TG_ARGS
  return tmae_launch_status();
}

int tmae_token_gemm(const void* x, int64_t ldx, int64_t m, int k, const void* w, int n, const void* bias, void* y,
                    int64_t ldy, void* stream_) {
  (void)hipGetLastError();
  return token_gemm_launch(x, ldx, m, k, w, n, bias, y, ldy, nullptr, nullptr, stream_);
}

int tmae_token_gemm_acc(const void* x, int64_t ldx, int64_t m, int k, const void* w, int n, const void* bias, void* y,
                        int64_t ldy, void* stream_) {
  (void)hipGetLastError();
  if (m < 0) return TMAE_EARG;
  if (m == 0) return TMAE_OK;
  // the in-place form exists in the W-in-registers kernel only (the y tile travels through its LDS ring):
  // [m, 256] += [m, 512] W^T and [m, 128] += [m, 256] W^T -- the FFN-1 input gradients of the d = 256 / d = 128 stages --, the
  // in-projections' (768 -> 256, 384 -> 128) and the square ones (256 -> 256, 128 -> 128)
  if (m < 32768 || !((k == 512 && n == 256) || (k == 256 && n == 128) || (k == 768 && n == 256) || (k == 384 && n == 128) ||
                     (k == 256 && n == 256) || (k == 128 && n == 128)))
    return TMAE_EARG;
  return tmae_token_gemm_wreg(x, ldx, m, k, w, n, bias, nullptr, y, ldy, 1, stream_);
}

int tmae_token_gemm_res(const void* x, int64_t ldx, int64_t m, int k, const void* w, int n, const void* bias, const void* res,
                        void* y, int64_t ldy, void* stream_) {
  (void)hipGetLastError();
  if (m < 0) return TMAE_EARG;
  if (m == 0) return TMAE_OK;
  // the residual tile travels through the W-in-registers kernel's LDS ring (the accumulate form, out of place): the two shapes of
  // the encoder FFN's second Linear
  if (m < 32768 || !((k == 512 && n == 256) || (k == 256 && n == 128))) return TMAE_EARG;
  return tmae_token_gemm_wreg_res(x, ldx, m, k, w, n, bias, res, y, ldy, stream_);
}

int tmae_token_gemm_gelu(const void* x, int64_t ldx, int64_t m, int k, const void* w, int n, const void* bias, void* y,
                         void* y_gelu, int64_t ldy, void* stream_) {
  (void)hipGetLastError();
  if (m < 0) return TMAE_EARG;
  if (m == 0) return TMAE_OK;
  // the dual-store form exists in the W-in-registers kernel only, on the two shapes of the encoder FFN's first Linear
  if (m < 32768 || !((k == 256 && n == 512) || (k == 128 && n == 256))) return TMAE_EARG;
  return tmae_token_gemm_wreg_gelu(x, ldx, m, k, w, n, bias, y, y_gelu, ldy, stream_);
}

int tmae_token_gemm_dgelu(const void* x, int64_t ldx, int64_t m, int k, const void* w, int n, const void* bias,
                          const void* aux, void* y, int64_t ldy, void* stream_) {
  (void)hipGetLastError();
  if (!aux || ((uintptr_t)aux & 15)) return TMAE_EARG;
  static const int wreg = TMAE_AB_INT("TMAE_TG_WREG_DGELU", 1);
  if (wreg != 0 && m >= 32768) {                        // heavy shapes: the W-in-registers kernel (h_pre rides in its LDS ring)
    const int rc = tmae_token_gemm_wreg_dgelu(x, ldx, m, k, w, n, bias, aux, y, ldy, stream_);
    if (rc != TMAE_EARG) return rc;
  }
  return token_gemm_launch(x, ldx, m, k, w, n, bias, y, ldy, aux, nullptr, stream_);
}

int tmae_token_gemm_pos(const void* x, int64_t ldx, int64_t m, int k, const void* w_aug, int n, const void* bias,
                        const uint8_t* cells, void* y, int64_t ldy, void* stream_) {
  (void)hipGetLastError();
  if (!cells && m > 0) return TMAE_EARG;
  return token_gemm_launch(x, ldx, m, k, w_aug, n, bias, y, ldy, nullptr, cells, stream_);
}
