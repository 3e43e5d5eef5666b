// 3x3 convolutions (padding 1) between a 64-channel channels-last map and a NARROW one (k <= 8 channels): the last convs of
// CenterHead's branches (pcdet/models/dense_heads/center_head.py:11-45: 64 -> 2 / 1 / 3 / 2 / num_class on the stride-1 BEV map),
// their input gradient and their weight gradient.  The library runs these shapes as implicit GEMMs padded to 32 output columns:
// 370 - 530 us forward, 130 us input gradient, 340 us weight gradient per branch on the [8, 468, 468] map -- ten times what the
// bytes need (the 64-channel map is 224 MB: 40 us at the copy rate).  Here every kernel is ONE pass over the 64-channel map:
//   * a workgroup (4 waves) owns a 16 x 16 block of cells at a time and walks over blocks (persistent: the weights are fetched
//     once); the block's 18 x 18 halo of 64-channel rows (or of the 8-channel gradient rows) goes through an LDS image, the next
//     block's rows are in flight in registers while the current block is contracted;
//   * the narrow side lives in registers (forward: all 9 x 64 x 16 weights = 18 MFMA fragments per lane) or in a few KB of LDS;
//   * products are taken with the narrow index on the accumulator ROWS wherever the result is narrow, so a lane ends up with
//     consecutive channels of one cell (contiguous stores).
// Layouts: in / din [B, Y, X, 64] bf16 with a channel pitch `ld` (elements per cell, so that a 64-channel slice of a wider map
// can be passed), out / dout [B, Y, X, k] bf16 contiguous, weight [k, 9, 64] bf16 (taps ky-major: the [k, 3, 3, 64] layout
// flattened), bias [k] fp32, dw [k, 9, 64] fp32.
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#define HN_T 16                      // cells per block edge
#define HN_HW 18                     // halo edge
#define HN_NH (HN_HW * HN_HW)        // 324 halo rows
#define HN_CHUNKS (HN_NH * 8)        // 16-byte chunks of a 64-channel halo image
#define HN_ITERS_OF(NTH) ((HN_CHUNKS + (NTH) - 1) / (NTH))
#define HN_IMG (HN_NH * 128)         // bytes of a 64-channel halo image
#define HN_OOB 0xFFFFFFFFu

__device__ __forceinline__ __amdgpu_buffer_rsrc_t hn_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 hn_mfma(const u32x4& a, const u32x4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ unsigned hn_pack(float a, float b) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}
__device__ __forceinline__ s16x4 hn_tr_read(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p);
}

struct HnBlock { int b, y0, x0; };
__device__ __forceinline__ HnBlock hn_block(int u, int bx, int by) {
  HnBlock r;
  const int tx = u % bx, t2 = u / bx;
  r.b = t2 / by; r.y0 = (t2 - r.b * by) * HN_T; r.x0 = tx * HN_T;
  return r;
}

// ---- the 64-channel halo image ------------------------------------------------------------------
// row h = hy * 18 + hx (cell (y0 - 1 + hy, x0 - 1 + hx)), 128 bytes; the 16-byte chunk c of a row sits at position c ^ (hx & 7):
// a swizzle by the COLUMN only, so that readers of neighbouring cells (consecutive hx, the lanes of an MFMA operand read) hit
// different banks and a tap shift changes hx by a constant.  Chunks outside the map read as zeros (buffer range check).
template <int NTH = 256> struct HnHaloT { u32x4 v[HN_ITERS_OF(NTH)]; };
typedef HnHaloT<256> HnHalo;
template <int NTH>
__device__ __forceinline__ void hn_halo_load(HnHaloT<NTH>& r, __amdgpu_buffer_rsrc_t rs, int64_t ld, const HnBlock& blk, int Y, int X,
                                             int tid) {
#pragma unroll
  for (int it = 0; it < HN_ITERS_OF(NTH); ++it) {
    const int idx = it * NTH + tid, h = idx >> 3, c = idx & 7;
    const int hy = (h * 3641) >> 16, hx = h - hy * HN_HW;            // h / 18 for h < 2 ^ 11
    const int y = blk.y0 - 1 + hy, x = blk.x0 - 1 + hx;
    const bool ok = idx < HN_CHUNKS && y >= 0 && y < Y && x >= 0 && x < X;
    const unsigned off = ok ? (unsigned)((((int64_t)blk.b * Y + y) * X + x) * ld * 2 + c * 16) : HN_OOB;
    r.v[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
  }
}
template <int NTH>
__device__ __forceinline__ void hn_halo_store(const HnHaloT<NTH>& r, char* img, int tid) {
#pragma unroll
  for (int it = 0; it < HN_ITERS_OF(NTH); ++it) {
    const int idx = it * NTH + tid, h = idx >> 3, c = idx & 7;
    const int hy = (h * 3641) >> 16, hx = h - hy * HN_HW;
    if (idx < HN_CHUNKS) *reinterpret_cast<u32x4*>(img + h * 128 + ((c ^ (hx & 7)) << 4)) = r.v[it];
  }
}
// byte offset of chunk c of the halo row of cell (y, x) of the block shifted by tap (ky, kx)
__device__ __forceinline__ int hn_row(int y, int x, int ky, int kx, int c) {
  const int hx = x + kx;
  return ((y + ky) * HN_HW + hx) * 128 + ((c ^ (hx & 7)) << 4);
}

// ---- the narrow image: 8 (NP = 16) or 16 (NP = 32) channels per cell, zero-padded ---------------------------------------------
// HALO = true: the 18 x 18 halo of the block (input gradient: a cell reads its neighbours), else its 16 x 16 cells.
template <bool HALO>
__device__ __forceinline__ void hn_narrow_load(unsigned short (&v)[2][8], __amdgpu_buffer_rsrc_t rs, int k, const HnBlock& blk, int Y,
                                               int X, int tid) {
  constexpr int NC = HALO ? HN_NH : HN_T * HN_T, W = HALO ? HN_HW : HN_T, O = HALO ? 1 : 0;
#pragma unroll
  for (int it = 0; it < (NC + 255) / 256; ++it) {
    const int h = it * 256 + tid;
    const int hy = HALO ? (h * 3641) >> 16 : h >> 4, hx = h - hy * W;
    const int y = blk.y0 - O + hy, x = blk.x0 - O + hx;
    const bool ok = h < NC && y >= 0 && y < Y && x >= 0 && x < X;
    const unsigned base = ok ? (unsigned)((((int64_t)blk.b * Y + y) * X + x) * k * 2) : HN_OOB;
#pragma unroll
    for (int n = 0; n < 8; ++n)
      v[it][n] = n < k ? (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rs, (int)(ok ? base + 2 * n : HN_OOB), 0, 0) : (unsigned short)0;
  }
}
template <bool HALO, int NP>
__device__ __forceinline__ void hn_narrow_store(const unsigned short (&v)[2][8], char* img, int tid) {
  constexpr int NC = HALO ? HN_NH : HN_T * HN_T;
#pragma unroll
  for (int it = 0; it < (NC + 255) / 256; ++it) {
    const int h = it * 256 + tid;
    if (h < NC) {
      const u32x4 u = {(unsigned)v[it][0] | ((unsigned)v[it][1] << 16), (unsigned)v[it][2] | ((unsigned)v[it][3] << 16),
                       (unsigned)v[it][4] | ((unsigned)v[it][5] << 16), (unsigned)v[it][6] | ((unsigned)v[it][7] << 16)};
      *reinterpret_cast<u32x4*>(img + h * NP) = u;
      if constexpr (NP == 32) *reinterpret_cast<u32x4*>(img + h * NP + 16) = u32x4{0u, 0u, 0u, 0u};
    }
  }
}

// ------------------------------------------------------------------------------------------------
// forward: out[b, y, x, n] = bias[n] + sum_{ky, kx, c} in[b, y + ky - 1, x + kx - 1, c] * W[n, ky * 3 + kx, c]
// Per 16 cells of a block row: 18 MFMAs (9 taps x 2 channel halves), A = weights (rows = n), B = halo rows (columns = cells):
// lane (g, i) ends with out[cell i][n = 4 g + r].
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void headconv_fwd_kernel(const __hip_bfloat16* __restrict__ in, int64_t ldi, unsigned in_bytes,
                                                           int B, int Y, int X, const __hip_bfloat16* __restrict__ W,
                                                           const float* __restrict__ bias, int k,
                                                           __hip_bfloat16* __restrict__ out, int nblocks) {
  __shared__ __attribute__((aligned(16))) char img[HN_IMG];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, i = lane & 15;
  const int bx = (X + HN_T - 1) / HN_T, by = (Y + HN_T - 1) / HN_T;
  const __amdgpu_buffer_rsrc_t rs = hn_rsrc(in, in_bytes);
  int u = blockIdx.x;
  if (u >= nblocks) return;
  HnBlock blk = hn_block(u, bx, by);
  HnHalo hal;
  hn_halo_load(hal, rs, ldi, blk, Y, X, tid);
  // weight fragments: lane (g, i) = row n = i, channels hf * 32 + 8 g .. + 7 of tap t
  u32x4 wf[9][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
      wf[t][hf] = i < k ? *reinterpret_cast<const u32x4*>(W + ((int64_t)i * 9 + t) * 64 + hf * 32 + 8 * g) : u32x4{0u, 0u, 0u, 0u};
  f32x4 b4;
#pragma unroll
  for (int r = 0; r < 4; ++r) b4[r] = (4 * g + r < k) ? bias[4 * g + r] : 0.f;
  for (;;) {
    hn_halo_store(hal, img, tid);
    __syncthreads();
    const HnBlock cur = blk;
    const int un = u + gridDim.x;
    if (un < nblocks) { blk = hn_block(un, bx, by); hn_halo_load(hal, rs, ldi, blk, Y, X, tid); }   // in flight under the MFMAs
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int y = 4 * w + rr;
      f32x4 acc = b4;
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const u32x4 a = *reinterpret_cast<const u32x4*>(img + hn_row(y, i, t / 3, t % 3, hf * 4 + g));
          acc = hn_mfma(wf[t][hf], a, acc);
        }
      const int gy = cur.y0 + y, gx = cur.x0 + i;
      if (gy < Y && gx < X) {
        __hip_bfloat16* o = out + (((int64_t)cur.b * Y + gy) * X + gx) * k;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (4 * g + r < k) o[4 * g + r] = __float2bfloat16(acc[r]);
      }
    }
    u = un;
    if (u >= nblocks) break;
    __syncthreads();                                   // every wave is done with the image before it is overwritten
  }
}

// ------------------------------------------------------------------------------------------------
// input gradient: din[b, y, x, c] = sum_{ky, kx, n} dout[b, y - ky + 1, x - kx + 1, n] * W[n, ky * 3 + kx, c]
// Contraction index = (tap, n padded to 8): 72 -> 3 MFMA steps of 32 (tap = 4 j + g per step j and lane group g; taps 9 .. 11 carry
// zero weights).  A = weights with the channel on the rows, PERMUTED: row m = 4 g' + r of column tile ct is channel
// 16 g' + 4 ct + r, so that lane (g', i) ends with the 16 consecutive channels 16 g' .. of cell i (two 16-byte stores).
// wpk [4 ct][3 j][64 lanes][8] bf16: the A fragments, made by headconv_pack_kernel.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void headconv_pack_kernel(const __hip_bfloat16* __restrict__ W, int k,
                                                            __hip_bfloat16* __restrict__ wpk) {
  const int e = blockIdx.x * 256 + threadIdx.x;        // (ct, j, lane, n)
  if (e >= 4 * 3 * 64 * 8) return;
  const int n = e & 7, lane = (e >> 3) & 63, j = (e >> 9) % 3, ct = e / (512 * 3);
  const int g = lane >> 4, i = lane & 15, t = 4 * j + g;
  const int c = 16 * (i >> 2) + 4 * ct + (i & 3);
  wpk[e] = (t < 9 && n < k) ? W[((int64_t)n * 9 + t) * 64 + c] : __float2bfloat16(0.f);
}

__global__ __launch_bounds__(256) void headconv_bwd_data_kernel(const __hip_bfloat16* __restrict__ dout, unsigned dout_bytes, int B,
                                                                int Y, int X, int k, const __hip_bfloat16* __restrict__ wpk,
                                                                __hip_bfloat16* __restrict__ din, int64_t ldo, int nblocks) {
  __shared__ __attribute__((aligned(16))) char img[HN_NH * 16];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, i = lane & 15;
  const int bx = (X + HN_T - 1) / HN_T, by = (Y + HN_T - 1) / HN_T;
  const __amdgpu_buffer_rsrc_t rs = hn_rsrc(dout, dout_bytes);
  int u = blockIdx.x;
  if (u >= nblocks) return;
  HnBlock blk = hn_block(u, bx, by);
  unsigned short nv[2][8];
  hn_narrow_load<true>(nv, rs, k, blk, Y, X, tid);
  u32x4 wf[4][3];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int j = 0; j < 3; ++j) wf[ct][j] = *reinterpret_cast<const u32x4*>(wpk + ((ct * 3 + j) * 64 + lane) * 8);
  // the gradient row this lane's taps read for cell (y, x = i): tap t = 4 j + g (clamped: taps past 8 meet zero weights),
  // source cell (y - ky + 1, x - kx + 1) = halo (y + 2 - ky, i + 2 - kx)
  int boff[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int t = (4 * j + g) < 9 ? 4 * j + g : 8;
    boff[j] = ((2 - t / 3) * HN_HW + i + 2 - t % 3) * 16;
  }
  for (;;) {
    hn_narrow_store<true, 16>(nv, img, tid);
    __syncthreads();
    const HnBlock cur = blk;
    const int un = u + gridDim.x;
    if (un < nblocks) { blk = hn_block(un, bx, by); hn_narrow_load<true>(nv, rs, k, blk, Y, X, tid); }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int y = 4 * w + rr;
      u32x4 bf[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) bf[j] = *reinterpret_cast<const u32x4*>(img + y * (HN_HW * 16) + boff[j]);
      f32x4 acc[4];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        acc[ct] = hn_mfma(wf[ct][0], bf[0], f32x4{0.f, 0.f, 0.f, 0.f});
        acc[ct] = hn_mfma(wf[ct][1], bf[1], acc[ct]);
        acc[ct] = hn_mfma(wf[ct][2], bf[2], acc[ct]);
      }
      const int gy = cur.y0 + y, gx = cur.x0 + i;
      if (gy < Y && gx < X) {
        char* o = reinterpret_cast<char*>(din + (((int64_t)cur.b * Y + gy) * X + gx) * ldo) + 32 * g;
        const u32x4 lo = {hn_pack(acc[0][0], acc[0][1]), hn_pack(acc[0][2], acc[0][3]), hn_pack(acc[1][0], acc[1][1]),
                          hn_pack(acc[1][2], acc[1][3])};
        const u32x4 hi = {hn_pack(acc[2][0], acc[2][1]), hn_pack(acc[2][2], acc[2][3]), hn_pack(acc[3][0], acc[3][1]),
                          hn_pack(acc[3][2], acc[3][3])};
        *reinterpret_cast<u32x4*>(o) = lo;
        *reinterpret_cast<u32x4*>(o + 16) = hi;
      }
    }
    u = un;
    if (u >= nblocks) break;
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// weight gradient: dw[n, t, c] = sum over cells of dout[cell, n] * in[cell + tap t, c]
// The contraction index is the CELL: both operands come out of their row-major LDS images through the transposing read
// (ds_read_b64_tr_b16, as in csrc/wgrad.hip).  A step = 32 cells = two block rows; wave w owns the channels 16 w .. 16 w + 15 and all
// nine taps (9 accumulator tiles [16 n][16 c]); persistent workgroups write one fp32 slab each, a second launch sums the slabs
// in a fixed order.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void headconv_wgrad_kernel(const __hip_bfloat16* __restrict__ dout, unsigned dout_bytes,
                                                             const __hip_bfloat16* __restrict__ in, int64_t ldi, unsigned in_bytes,
                                                             int B, int Y, int X, int k, float* __restrict__ slab, int nblocks) {
  __shared__ __attribute__((aligned(16))) char img[HN_IMG];
  __shared__ __attribute__((aligned(16))) char dimg[HN_T * HN_T * 32];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  const int bx = (X + HN_T - 1) / HN_T, by = (Y + HN_T - 1) / HN_T;
  const __amdgpu_buffer_rsrc_t rsi = hn_rsrc(in, in_bytes), rsd = hn_rsrc(dout, dout_bytes);
  f32x4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  int u = blockIdx.x;
  if (u < nblocks) {
    HnBlock blk = hn_block(u, bx, by);
    HnHalo hal;
    unsigned short nv[2][8];
    hn_halo_load(hal, rsi, ldi, blk, Y, X, tid);
    hn_narrow_load<false>(nv, rsd, k, blk, Y, X, tid);
    // this lane's part of the transposing reads: it supplies the address of cell row (8 g + q [+ 4]) of a step, 8-byte piece p, and
    // receives column i (n for dout, channel 16 w + i for in) of the 4 cells
    const int xq = 8 * (g & 1) + q, yq = g >> 1;      // cell (2 s + yq, xq [+ 4]) of step s
    for (;;) {
      hn_halo_store(hal, img, tid);
      hn_narrow_store<false, 32>(nv, dimg, tid);
      __syncthreads();
      const int un = u + gridDim.x;
      if (un < nblocks) {
        blk = hn_block(un, bx, by);
        hn_halo_load(hal, rsi, ldi, blk, Y, X, tid);
        hn_narrow_load<false>(nv, rsd, k, blk, Y, X, tid);
      }
#pragma unroll 2
      for (int s = 0; s < 8; ++s) {
        const int y = 2 * s + yq;
        const s16x4 alo = hn_tr_read(dimg + (y * HN_T + xq) * 32 + 8 * p);
        const s16x4 ahi = hn_tr_read(dimg + (y * HN_T + xq + 4) * 32 + 8 * p);
        const s16x8 a8 = {alo[0], alo[1], alo[2], alo[3], ahi[0], ahi[1], ahi[2], ahi[3]};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int ky = t / 3, kx = t % 3;
          const s16x4 blo = hn_tr_read(img + hn_row(y, xq, ky, kx, 2 * w + (p >> 1)) + 8 * (p & 1));
          const s16x4 bhi = hn_tr_read(img + hn_row(y, xq + 4, ky, kx, 2 * w + (p >> 1)) + 8 * (p & 1));
          const s16x8 b8 = {blo[0], blo[1], blo[2], blo[3], bhi[0], bhi[1], bhi[2], bhi[3]};
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a8), __builtin_bit_cast(bf16x8, b8), acc[t], 0, 0, 0);
        }
      }
      u = un;
      if (u >= nblocks) break;
      __syncthreads();
    }
  }
  // slab [k][9][64] of this workgroup: lane (g, i) holds dw[n = 4 g + r][t][c = 16 w + i]
  float* sl = slab + (int64_t)blockIdx.x * (k * 576);
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (4 * g + r < k) sl[((4 * g + r) * 9 + t) * 64 + 16 * w + i] = acc[t][r];
}

__global__ __launch_bounds__(256) void headconv_wgrad_reduce_kernel(const float* __restrict__ slab, int nslab, int count,
                                                                    float* __restrict__ dw) {
  // one wave per 64 outputs would leave the chip idle (count <= 4608): 4 waves split the slabs of 64 outputs and meet in LDS
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, e = blockIdx.x * 64 + lane;
  float s = 0.f;
  if (e < count) {
    float s4[4] = {0.f, 0.f, 0.f, 0.f};                  // four loads in flight per lane (fixed order: deterministic)
    int j = w;
    for (; j + 12 < nslab; j += 16) {
#pragma unroll
      for (int q = 0; q < 4; ++q) s4[q] += slab[(int64_t)(j + 4 * q) * count + e];
    }
    for (; j < nslab; j += 4) s4[0] += slab[(int64_t)j * count + e];
    s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
  }
  part[w][lane] = s;
  __syncthreads();
  if (w == 0 && e < count) dw[e] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

// ================================================================================================
// 64 -> 64 channels: the stem convs of CenterHead's branches (center_head.py:28-31: Conv2d(64, 64, 3, padding=1) in front of
// BatchNorm + ReLU), their input gradient (the same kernel on dout with flipped, transposed weights) and their weight gradient.
// The library's implicit GEMMs take 360 / 300 / 500 us per branch; the bytes (41 KB of halo in, 32 KB out per 256 cells) need 90.
// ================================================================================================
// A-fragment image of the weights, [t][hf][ct][lane] x 16 bytes = 73 728 bytes (made once per call by the pack kernel, copied
// into LDS once per persistent workgroup): lane (g, i) of fragment (t, hf, ct) = output channel 16 (i >> 2) + 4 ct + (i & 3)
// (the row permutation that leaves a lane with consecutive channels), input channels hf * 32 + 8 g .. + 7 of tap t.
// mode 0: weight [64 out][9][64 in] as it is; mode 1 (input gradient): out' = in, in' = out, tap 8 - t.
#define HN_WIMG (9 * 2 * 4 * 64 * 16)
__global__ __launch_bounds__(256) void headconv64_pack_kernel(const __hip_bfloat16* __restrict__ W, int mode,
                                                              __hip_bfloat16* __restrict__ wimg) {
  const int e = blockIdx.x * 256 + threadIdx.x;        // ((((t * 2 + hf) * 4 + ct) * 64 + lane) * 8 + j
  if (e >= HN_WIMG / 2) return;
  const int j = e & 7, lane = (e >> 3) & 63, ct = (e >> 9) & 3, hf = (e >> 11) & 1, t = e >> 12;
  const int g = lane >> 4, i = lane & 15;
  const int co = 16 * (i >> 2) + 4 * ct + (i & 3), ci = hf * 32 + 8 * g + j;
  wimg[e] = mode == 0 ? W[((int64_t)co * 9 + t) * 64 + ci] : W[((int64_t)ci * 9 + (8 - t)) * 64 + co];
}

// 512 threads: wave (rg = w & 3, h = w >> 2) owns the cell rows 4 rg .. 4 rg + 3 of a block and the column tiles 2 h, 2 h + 1: per
// (tap, channel half) 2 weight fragments + 4 row fragments from LDS feed 8 MFMAs; lane (g, i) ends with the 8 consecutive
// output channels 16 g + 8 h .. of cell i of each row (one 16-byte store).  post (optional, the shape of out): added to the result.
__global__ __launch_bounds__(512) void headconv64_kernel(const __hip_bfloat16* __restrict__ in, int64_t ldi, unsigned in_bytes, int B,
                                                         int Y, int X, const __hip_bfloat16* __restrict__ wimg_g,
                                                         __hip_bfloat16* __restrict__ out, int64_t ldo, int nblocks, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) char lds64[];
  char* img = lds64;
  char* wimg = lds64 + HN_IMG;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, i = lane & 15, rg = w & 3, h = w >> 2;
  const int bx = (X + HN_T - 1) / HN_T, by = (Y + HN_T - 1) / HN_T;
  const __amdgpu_buffer_rsrc_t rs = hn_rsrc(in, in_bytes);
  int u = blockIdx.x;
  if (u >= nblocks) return;
  HnBlock blk = hn_block(u, bx, by);
  HnHaloT<512> hal;
  hn_halo_load(hal, rs, ldi, blk, Y, X, tid);
#pragma unroll
  for (int it = 0; it < HN_WIMG / 16 / 512; ++it)
    reinterpret_cast<u32x4*>(wimg)[it * 512 + tid] = reinterpret_cast<const u32x4*>(wimg_g)[it * 512 + tid];
  for (;;) {
    hn_halo_store(hal, img, tid);
    __syncthreads();
    const HnBlock cur = blk;
    const int un = u + gridDim.x;
    if (un < nblocks) { blk = hn_block(un, bx, by); hn_halo_load(hal, rs, ldi, blk, Y, X, tid); }
    f32x4 acc[4][2];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      acc[rr][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[rr][1] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int gy = cur.y0 + 4 * rg + rr, gx = cur.x0 + i;
      if (accumulate && gy < Y && gx < X) {              // out += : the second half of a 128-channel contraction (one extra rounding)
        const u32x4 o = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(out + (((int64_t)cur.b * Y + gy) * X + gx) * ldo) +
                                                        32 * g + 16 * h);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[rr][0][2 * j] = __uint_as_float(o[j] << 16);      acc[rr][0][2 * j + 1] = __uint_as_float(o[j] & 0xFFFF0000u);
          acc[rr][1][2 * j] = __uint_as_float(o[2 + j] << 16);  acc[rr][1][2 * j + 1] = __uint_as_float(o[2 + j] & 0xFFFF0000u);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const u32x4 a0 = *reinterpret_cast<const u32x4*>(wimg + ((((t * 2 + hf) * 4 + 2 * h) * 64 + lane) << 4));
        const u32x4 a1 = *reinterpret_cast<const u32x4*>(wimg + ((((t * 2 + hf) * 4 + 2 * h + 1) * 64 + lane) << 4));
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const u32x4 b = *reinterpret_cast<const u32x4*>(img + hn_row(4 * rg + rr, i, t / 3, t % 3, hf * 4 + g));
          acc[rr][0] = hn_mfma(a0, b, acc[rr][0]);
          acc[rr][1] = hn_mfma(a1, b, acc[rr][1]);
        }
      }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int gy = cur.y0 + 4 * rg + rr, gx = cur.x0 + i;
      if (gy < Y && gx < X) {
        const u32x4 o = {hn_pack(acc[rr][0][0], acc[rr][0][1]), hn_pack(acc[rr][0][2], acc[rr][0][3]),
                         hn_pack(acc[rr][1][0], acc[rr][1][1]), hn_pack(acc[rr][1][2], acc[rr][1][3])};
        *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(out + (((int64_t)cur.b * Y + gy) * X + gx) * ldo) + 32 * g + 16 * h) = o;
      }
    }
    u = un;
    if (u >= nblocks) break;
    __syncthreads();
  }
}

// weight gradient dw[n, t, c] = sum over cells of dout[cell, n] * in[cell + tap t, c], n, c < 64: the transposing-read scheme of
// headconv_wgrad_kernel with a full 64-channel dout image (rows of 128 bytes, chunks swizzled by the cell's column).  512 threads: wave
// (nt = w & 3, ch = w >> 2) owns the rows n = 16 nt .. and the channels 32 ch .. 32 ch + 31 of all nine taps (18 accumulator tiles).
__global__ __launch_bounds__(512) void headconv64_wgrad_kernel(const __hip_bfloat16* __restrict__ dout, int64_t lddo, unsigned dout_bytes,
                                                               const __hip_bfloat16* __restrict__ in, int64_t ldi, unsigned in_bytes,
                                                               int B, int Y, int X, float* __restrict__ slab, int nblocks) {
  extern __shared__ __attribute__((aligned(16))) char lds64[];
  char* img = lds64;
  char* dimg = lds64 + HN_IMG;                            // [256 cells][128 bytes]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  const int nt = w & 3, ch = w >> 2;
  const int bx = (X + HN_T - 1) / HN_T, by = (Y + HN_T - 1) / HN_T;
  const __amdgpu_buffer_rsrc_t rsi = hn_rsrc(in, in_bytes), rsd = hn_rsrc(dout, dout_bytes);
  f32x4 acc[2][9];
#pragma unroll
  for (int cc = 0; cc < 2; ++cc)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[cc][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  int u = blockIdx.x;
  if (u < nblocks) {
    HnBlock blk = hn_block(u, bx, by);
    HnHaloT<512> hal;
    u32x4 dv[4];
    auto dload = [&]() {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int idx = it * 512 + tid, cell = idx >> 3, c = idx & 7;
        const int y = blk.y0 + (cell >> 4), x = blk.x0 + (cell & 15);
        const unsigned off = (y < Y && x < X) ? (unsigned)((((int64_t)blk.b * Y + y) * X + x) * lddo * 2 + c * 16) : HN_OOB;
        dv[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsd, (int)off, 0, 0));
      }
    };
    hn_halo_load(hal, rsi, ldi, blk, Y, X, tid);
    dload();
    const int xq = 8 * (g & 1) + q, yq = g >> 1;
    for (;;) {
      hn_halo_store(hal, img, tid);
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int idx = it * 512 + tid, cell = idx >> 3, c = idx & 7;
        *reinterpret_cast<u32x4*>(dimg + cell * 128 + ((c ^ (cell & 7)) << 4)) = dv[it];
      }
      __syncthreads();
      const int un = u + gridDim.x;
      if (un < nblocks) {
        blk = hn_block(un, bx, by);
        hn_halo_load(hal, rsi, ldi, blk, Y, X, tid);
        dload();
      }
#pragma unroll 1
      for (int s = 0; s < 8; ++s) {
        const int y = 2 * s + yq;
        const int c0 = y * HN_T + xq, c1 = c0 + 4;
        const s16x4 alo = hn_tr_read(dimg + c0 * 128 + (((2 * nt + (p >> 1)) ^ (c0 & 7)) << 4) + 8 * (p & 1));
        const s16x4 ahi = hn_tr_read(dimg + c1 * 128 + (((2 * nt + (p >> 1)) ^ (c1 & 7)) << 4) + 8 * (p & 1));
        const s16x8 a8 = {alo[0], alo[1], alo[2], alo[3], ahi[0], ahi[1], ahi[2], ahi[3]};
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int cc = 0; cc < 2; ++cc) {
            const int chunk = 2 * (2 * ch + cc) + (p >> 1);
            const s16x4 blo = hn_tr_read(img + hn_row(y, xq, t / 3, t % 3, chunk) + 8 * (p & 1));
            const s16x4 bhi = hn_tr_read(img + hn_row(y, xq + 4, t / 3, t % 3, chunk) + 8 * (p & 1));
            const s16x8 b8 = {blo[0], blo[1], blo[2], blo[3], bhi[0], bhi[1], bhi[2], bhi[3]};
            acc[cc][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a8), __builtin_bit_cast(bf16x8, b8),
                                                                 acc[cc][t], 0, 0, 0);
          }
      }
      u = un;
      if (u >= nblocks) break;
      __syncthreads();
    }
  }
  // slab [64 n][9][64 c]: lane (g, i) holds dw[16 nt + 4 g + r][t][32 ch + 16 cc + i]
  float* sl = slab + (int64_t)blockIdx.x * (64 * 576);
#pragma unroll
  for (int cc = 0; cc < 2; ++cc)
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) sl[((16 * nt + 4 * g + r) * 9 + t) * 64 + 32 * ch + 16 * cc + i] = acc[cc][t][r];
}

// ---- C ABI ---------------------------------------------------------------------------------------
static bool hn_args_ok(int batch, int ny, int nx, int k, int64_t ld, int64_t* cells) {
  if (batch <= 0 || ny <= 0 || nx <= 0 || k < 1 || k > 8 || ld < 64 || (ld % 8)) return false;
  *cells = (int64_t)batch * ny * nx;
  return *cells * ld * 2 < ((int64_t)1 << 31) && *cells * k * 2 < ((int64_t)1 << 31);   // 32-bit buffer offsets
}
static int hn_grid(int64_t nblocks) {
  const int64_t cap = (int64_t)tmae_num_cus() * 3;     // three workgroups per CU fit (LDS)
  return (int)(nblocks < cap ? nblocks : cap);
}
static int64_t hn_nblocks(int batch, int ny, int nx) { return (int64_t)batch * ((ny + HN_T - 1) / HN_T) * ((nx + HN_T - 1) / HN_T); }

int tmae_conv3x3_c64_narrow_fwd(const void* in, int64_t ldi, int batch, int ny, int nx, const void* weight, const float* bias, int k,
                                void* out, void* stream_) {
  (void)hipGetLastError();
  int64_t cells;
  if (!hn_args_ok(batch, ny, nx, k, ldi, &cells) || !in || !weight || !bias || !out || ((uintptr_t)in & 15) || ((uintptr_t)weight & 15))
    return TMAE_EARG;
  const int64_t nb = hn_nblocks(batch, ny, nx);
  hipLaunchKernelGGL(headconv_fwd_kernel, dim3((unsigned)hn_grid(nb)), dim3(256), 0, (hipStream_t)stream_,
                     (const __hip_bfloat16*)in, ldi, (unsigned)((cells - 1) * ldi * 2 + 128), batch, ny, nx,
                     (const __hip_bfloat16*)weight, bias, k, (__hip_bfloat16*)out, (int)nb);
  return tmae_launch_status();
}

size_t tmae_conv3x3_c64_narrow_bwd_data_workspace(void) { return tmae_align(4 * 3 * 64 * 8 * 2); }

int tmae_conv3x3_c64_narrow_bwd_data(const void* dout, int batch, int ny, int nx, int k, const void* weight, void* din, int64_t ldo,
                                     void* ws, size_t ws_bytes, void* stream_) {
  (void)hipGetLastError();
  int64_t cells;
  if (!hn_args_ok(batch, ny, nx, k, ldo, &cells) || !dout || !weight || !din || !ws || ((uintptr_t)din & 15) || ((uintptr_t)ws & 15) ||
      ((uintptr_t)dout & 1) || ws_bytes < tmae_conv3x3_c64_narrow_bwd_data_workspace())
    return TMAE_EARG;
  hipStream_t stream = (hipStream_t)stream_;
  hipLaunchKernelGGL(headconv_pack_kernel, dim3(4 * 3 * 64 * 8 / 256), dim3(256), 0, stream, (const __hip_bfloat16*)weight, k,
                     (__hip_bfloat16*)ws);
  const int64_t nb = hn_nblocks(batch, ny, nx);
  hipLaunchKernelGGL(headconv_bwd_data_kernel, dim3((unsigned)hn_grid(nb)), dim3(256), 0, stream, (const __hip_bfloat16*)dout,
                     (unsigned)(cells * k * 2), batch, ny, nx, k, (const __hip_bfloat16*)ws, (__hip_bfloat16*)din, ldo, (int)nb);
  return tmae_launch_status();
}

size_t tmae_conv3x3_c64_narrow_wgrad_workspace(int k) {
  if (k < 1 || k > 8) return 0;
  return tmae_align((size_t)tmae_num_cus() * 3 * k * 576 * sizeof(float));
}

int tmae_conv3x3_c64_narrow_wgrad(const void* dout, const void* in, int64_t ldi, int batch, int ny, int nx, int k, float* dw, void* ws,
                                  size_t ws_bytes, void* stream_) {
  (void)hipGetLastError();
  int64_t cells;
  if (!hn_args_ok(batch, ny, nx, k, ldi, &cells) || !dout || !in || !dw || !ws || ((uintptr_t)in & 15) || ((uintptr_t)dout & 1) ||
      ((uintptr_t)ws & 15) || ws_bytes < tmae_conv3x3_c64_narrow_wgrad_workspace(k))
    return TMAE_EARG;
  hipStream_t stream = (hipStream_t)stream_;
  const int64_t nb = hn_nblocks(batch, ny, nx);
  const int grid = hn_grid(nb), count = k * 576;
  hipLaunchKernelGGL(headconv_wgrad_kernel, dim3((unsigned)grid), dim3(256), 0, stream, (const __hip_bfloat16*)dout,
                     (unsigned)(cells * k * 2), (const __hip_bfloat16*)in, ldi, (unsigned)((cells - 1) * ldi * 2 + 128), batch, ny, nx,
                     k, (float*)ws, (int)nb);
  hipLaunchKernelGGL(headconv_wgrad_reduce_kernel, dim3((unsigned)((count + 63) / 64)), dim3(256), 0, stream, (const float*)ws, grid,
                     count, dw);
  return tmae_launch_status();
}

// ---- 64 -> 64 -------------------------------------------------------------------------------------
size_t tmae_conv3x3_c64_workspace(void) { return tmae_align(HN_WIMG); }

int tmae_conv3x3_c64(const void* in, int64_t ldi, int batch, int ny, int nx, const void* weight, int input_gradient, int accumulate,
                     void* out, int64_t ldo, void* ws, size_t ws_bytes, void* stream_) {
  (void)hipGetLastError();
  int64_t cells, cells2;
  if (!hn_args_ok(batch, ny, nx, 1, ldi, &cells) || !hn_args_ok(batch, ny, nx, 1, ldo, &cells2) || !in || !weight || !out || !ws ||
      ((uintptr_t)in & 15) || ((uintptr_t)out & 15) || ((uintptr_t)ws & 15) || ((uintptr_t)weight & 1) ||
      ws_bytes < tmae_conv3x3_c64_workspace() || (input_gradient != 0 && input_gradient != 1) || (accumulate != 0 && accumulate != 1))
    return TMAE_EARG;
  hipStream_t stream = (hipStream_t)stream_;
  hipLaunchKernelGGL(headconv64_pack_kernel, dim3(HN_WIMG / 2 / 256), dim3(256), 0, stream, (const __hip_bfloat16*)weight,
                     input_gradient, (__hip_bfloat16*)ws);
  const int lds = HN_IMG + HN_WIMG;
  static TmaeLdsAttr attr;
  if (int e = tmae_allow_lds(attr, (const void*)headconv64_kernel, lds)) return e;
  const int64_t nb = hn_nblocks(batch, ny, nx);
  const int64_t cap = tmae_num_cus();                  // one workgroup per CU (115 KB of LDS)
  hipLaunchKernelGGL(headconv64_kernel, dim3((unsigned)(nb < cap ? nb : cap)), dim3(512), lds, stream, (const __hip_bfloat16*)in, ldi,
                     (unsigned)((cells - 1) * ldi * 2 + 128), batch, ny, nx, (const __hip_bfloat16*)ws, (__hip_bfloat16*)out, ldo,
                     (int)nb, accumulate);
  return tmae_launch_status();
}

size_t tmae_conv3x3_c64_wgrad_workspace(void) { return tmae_align((size_t)tmae_num_cus() * 64 * 576 * sizeof(float)); }

int tmae_conv3x3_c64_wgrad(const void* dout, int64_t lddo, const void* in, int64_t ldi, int batch, int ny, int nx, float* dw, void* ws,
                           size_t ws_bytes, void* stream_) {
  (void)hipGetLastError();
  int64_t cells, cells2;
  if (!hn_args_ok(batch, ny, nx, 1, ldi, &cells) || !hn_args_ok(batch, ny, nx, 1, lddo, &cells2) || !dout || !in || !dw || !ws ||
      ((uintptr_t)in & 15) || ((uintptr_t)dout & 15) || ((uintptr_t)ws & 15) || ws_bytes < tmae_conv3x3_c64_wgrad_workspace())
    return TMAE_EARG;
  hipStream_t stream = (hipStream_t)stream_;
  const int lds = HN_IMG + HN_T * HN_T * 128;
  static TmaeLdsAttr attr;
  if (int e = tmae_allow_lds(attr, (const void*)headconv64_wgrad_kernel, lds)) return e;
  const int64_t nb = hn_nblocks(batch, ny, nx);
  const int64_t cap = tmae_num_cus();                  // one workgroup per CU (234 registers x 512 threads)
  const int grid = (int)(nb < cap ? nb : cap), count = 64 * 576;
  hipLaunchKernelGGL(headconv64_wgrad_kernel, dim3((unsigned)grid), dim3(512), lds, stream, (const __hip_bfloat16*)dout, lddo,
                     (unsigned)((cells - 1) * lddo * 2 + 128), (const __hip_bfloat16*)in, ldi, (unsigned)((cells - 1) * ldi * 2 + 128),
                     batch, ny, nx, (float*)ws, (int)nb);
  hipLaunchKernelGGL(headconv_wgrad_reduce_kernel, dim3((unsigned)((count + 63) / 64)), dim3(256), 0, stream, (const float*)ws, grid,
                     count, dw);
  return tmae_launch_status();
}
