// A9: the sparse convolution itself as an implicit GEMM (spconv's gather-GEMM-scatter, pcdet/utils/spconv_utils.py:37-56,
// spt_backbone.py:280-304), forward and input gradient:
//     out[o, n] = sum_{t < 9} sum_{c < CIN}  in[nbr[o, t], c] * W[n, t * CIN + c]          (nbr < 0: no such neighbour)
// The [m, 9 CIN] im2col matrix of the gather + library-GEMM formulation (2.1 GB written and read back for a d = 256
// stage) never exists: a workgroup owns a 128-row x 128-column output tile, walks over the 9 taps x CIN / 64 slices of
// the contraction and stages, per slice, the 128 gathered input rows (64 channels each: one 128-byte piece per row)
// and the 128 weight rows through LDS -- register-staged one slice ahead (issue the next slice's global loads, contract
// the current one, write the next one into the other LDS buffer, one barrier per slice).
// bf16 in, fp32 accumulate (v_mfma_f32_16x16x32_bf16), bf16 out.  The products are taken "swapped" (rows = output
// channels, column = token) over weight rows permuted in LDS so that a lane ends up with 16 CONSECUTIVE output channels
// of one token: two 16-byte stores per token, four lanes per 128-byte line (the store shape csrc/token_gemm.hip found
// decisive).  The input gradient is the same kernel on the transposed rulebook and the transposed weight.
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define IG_BM 128
#define IG_BN 128
#define IG_BK 64
#define IG_PITCH (IG_BK * 2 + 16)      // bytes per staged row: +16 spreads 16 consecutive rows over all banks

__device__ __forceinline__ unsigned ig_bf16_bits(float v) {
  return (unsigned)__builtin_bit_cast(unsigned short, __float2bfloat16(v));
}

template <int CIN>
__global__ __launch_bounds__(256, 2) void spconv_igemm_kernel(const __hip_bfloat16* __restrict__ feat, int64_t ldf,
                                                             const int32_t* __restrict__ nbr, int64_t m_out,
                                                             const __hip_bfloat16* __restrict__ W, int cout,
                                                             __hip_bfloat16* __restrict__ out, int64_t ldo) {
  constexpr int KC = CIN / IG_BK;                      // slices per tap
  constexpr int STEPS = 9 * KC;
  __shared__ __attribute__((aligned(16))) char As[2][IG_BM * IG_PITCH];
  __shared__ __attribute__((aligned(16))) char Bs[2][IG_BN * IG_PITCH];
  __shared__ int nb[IG_BM * 9];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, i = lane & 15;
  // column tiles of one row tile run on adjacent block ids of one XCD residue (blockIdx.x % 8 constant): they gather the
  // same input rows through that XCD's L2
  const int nct = cout / IG_BN;
  const int64_t bid = blockIdx.x;
  const int64_t rt = (bid / (8 * nct)) * 8 + (bid & 7);
  const int ct = (int)((bid >> 3) % nct);
  const int64_t row0 = rt * IG_BM;
  if (row0 >= m_out) return;
  const int n0 = ct * IG_BN;
  for (int e = tid; e < IG_BM * 9; e += 256) {
    const int64_t r = row0 + e / 9;
    nb[e] = r < m_out ? nbr[r * 9 + (e % 9)] : -1;
  }
  __syncthreads();
  const int piece = tid & 7, lrow = tid >> 3;          // 16-byte piece of the 128-byte slice row; rows lrow + 32 j
  int brow[4];                                         // LDS row of weight row lrow + 32 j (16 consecutive columns per lane)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = lrow + 32 * j, s = n & 63;
    brow[j] = (n & 64) + 16 * ((s >> 2) & 3) + 4 * (s >> 4) + (s & 3);
  }
  u32x4 ra[4], rb[4];
  int src[4];
  auto gload = [&](int step) {
    const int t = step / KC, kc = step % KC;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      src[j] = nb[(lrow + 32 * j) * 9 + t];
      // unconditional (clamped) loads, masked when they are written to LDS: a load under a predicate is waited for
      // before the next one is issued
      ra[j] = *reinterpret_cast<const u32x4*>(feat + (int64_t)(src[j] < 0 ? 0 : src[j]) * ldf + kc * IG_BK + piece * 8);
      rb[j] = *reinterpret_cast<const u32x4*>(W + (int64_t)(n0 + lrow + 32 * j) * (9 * CIN) + t * CIN + kc * IG_BK + piece * 8);
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32x4 z = {0u, 0u, 0u, 0u};
      *reinterpret_cast<u32x4*>(&As[buf][(lrow + 32 * j) * IG_PITCH + piece * 16]) = src[j] < 0 ? z : ra[j];
      *reinterpret_cast<u32x4*>(&Bs[buf][brow[j] * IG_PITCH + piece * 16]) = rb[j];
    }
  };
  const int wm = w & 1, wn = w >> 1;                   // wave tile: rows wm*64.., columns wn*64..
  f32x4 acc[4][4];                                     // [column tile nt][row tile mt]
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  gload(0);
  lstore(0);
  __syncthreads();
  for (int step = 0; step < STEPS; ++step) {
    const int buf = step & 1;
    if (step + 1 < STEPS) gload(step + 1);
#pragma unroll
    for (int ks = 0; ks < IG_BK / 32; ++ks) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        af[nt] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(&Bs[buf][(wn * 64 + nt * 16 + i) * IG_PITCH + (ks * 4 + g) * 16]));
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
        bfr[mt] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(&As[buf][(wm * 64 + mt * 16 + i) * IG_PITCH + (ks * 4 + g) * 16]));
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
          acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[nt], bfr[mt], acc[nt][mt], 0, 0, 0);
    }
    if (step + 1 < STEPS) lstore(buf ^ 1);
    __syncthreads();
  }
  // rows 4g + r of column tile nt = output channels wn*64 + 16g + 4nt + r; column i = token wm*64 + mt*16 + i
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int64_t r = row0 + wm * 64 + mt * 16 + i;
    if (r < m_out) {
      __hip_bfloat16* p = out + r * ldo + n0 + wn * 64 + 16 * g;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        u32x4 v;
        v[0] = ig_bf16_bits(acc[2 * h][mt][0]) | (ig_bf16_bits(acc[2 * h][mt][1]) << 16);
        v[1] = ig_bf16_bits(acc[2 * h][mt][2]) | (ig_bf16_bits(acc[2 * h][mt][3]) << 16);
        v[2] = ig_bf16_bits(acc[2 * h + 1][mt][0]) | (ig_bf16_bits(acc[2 * h + 1][mt][1]) << 16);
        v[3] = ig_bf16_bits(acc[2 * h + 1][mt][2]) | (ig_bf16_bits(acc[2 * h + 1][mt][3]) << 16);
        *reinterpret_cast<u32x4*>(p + 8 * h) = v;
      }
    }
  }
}

// out [m_out, cout] bf16 = sparse conv of feat [m_in, cin] bf16 through the rulebook nbr [m_out, 9] with the weight
// matrix W [cout, 9 * cin] bf16 (the spconv-2 layout [cout, 3, 3, cin] flattened).  cin in {128, 256, 384}, cout % 128 == 0.
static int igemm_launch(const void* feat, int64_t ldf, int64_t m_in, int cin, const int32_t* nbr, int64_t m_out,
                        const void* w, int cout, void* out, int64_t ldo, hipStream_t stream) {
  if (m_in < 0 || m_out < 0 || (cin != 128 && cin != 256 && cin != 384) || cout <= 0 || (cout % IG_BN) || ldf < cin || ldo < cout ||
      (ldf % 8) || (ldo % 8))
    return TMAE_EARG;
  if (m_out == 0) return TMAE_OK;
  if (!feat || !nbr || !w || !out || m_in == 0) return TMAE_EARG;
  if (((uintptr_t)feat & 15) || ((uintptr_t)w & 15) || ((uintptr_t)out & 15)) return TMAE_EARG;
  const int nct = cout / IG_BN;
  const int64_t rts = (m_out + IG_BM - 1) / IG_BM;
  const int64_t grid = ((rts + 7) / 8) * 8 * nct;       // row tiles padded to a multiple of 8 (the XCD-aware id map)
  if (cin == 128)
    hipLaunchKernelGGL((spconv_igemm_kernel<128>), dim3((unsigned)grid), dim3(256), 0, stream, (const __hip_bfloat16*)feat,
                       ldf, nbr, m_out, (const __hip_bfloat16*)w, cout, (__hip_bfloat16*)out, ldo);
  else if (cin == 256)
    hipLaunchKernelGGL((spconv_igemm_kernel<256>), dim3((unsigned)grid), dim3(256), 0, stream, (const __hip_bfloat16*)feat,
                       ldf, nbr, m_out, (const __hip_bfloat16*)w, cout, (__hip_bfloat16*)out, ldo);
  else
    hipLaunchKernelGGL((spconv_igemm_kernel<384>), dim3((unsigned)grid), dim3(256), 0, stream, (const __hip_bfloat16*)feat,
                       ldf, nbr, m_out, (const __hip_bfloat16*)w, cout, (__hip_bfloat16*)out, ldo);
  return tmae_launch_status();
}

int tmae_spconv_fwd(const void* feat, int64_t ldf, int64_t m_in, int cin, const int32_t* nbr, int64_t m_out,
                    const void* weight, int cout, void* out, int64_t ldo, void* stream_) {
  (void)hipGetLastError();
  return igemm_launch(feat, ldf, m_in, cin, nbr, m_out, weight, cout, out, ldo, (hipStream_t)stream_);
}

// din [m_in, cin] = sum_t dout[nbr_t[i, t], :] . W[:, t, :]: the forward kernel on the transposed rulebook with
// weight_t [cin, 9 * cout], weight_t[c, t * cout + n] = W[n, t * cin + c] (the caller keeps that copy).
int tmae_spconv_bwd_data(const void* dout, int64_t lddo, int64_t m_out, int cout, const int32_t* nbr_t, int64_t m_in,
                         const void* weight_t, int cin, void* din, int64_t lddi, void* stream_) {
  (void)hipGetLastError();
  return igemm_launch(dout, lddo, m_out, cout, nbr_t, m_in, weight_t, cin, din, lddi, (hipStream_t)stream_);
}
