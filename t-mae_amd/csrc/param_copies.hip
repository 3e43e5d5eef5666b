// Low-precision copies of the parameters after the optimizer step (the autocast path reads bf16 weights; the dX GEMMs
// of the token-list Linears want them transposed): ONE launch casts a whole list of fp32 matrices to bf16 and writes
// the transposed bf16 copy next to it.  torch needs a multi-tensor copy plus one transpose-copy launch per weight
// (~40 per step); the addresses never change, so the descriptor table is built once and lives on the device.
#include "common.h"

struct CastEntry {          // 5 x int64 in the host-built table
  const float* src;         // [n, k] fp32, contiguous
  __hip_bfloat16* dst;      // [n, k] bf16
  __hip_bfloat16* dstT;     // [k, n] bf16 (or NULL)
  int64_t nk;               // n | k << 32
  int64_t tile0;            // first 32 x 32 tile of this matrix in the grid
};

__global__ __launch_bounds__(256) void multi_cast_transpose_kernel(const CastEntry* __restrict__ tab, int count) {
  __shared__ float tile[32][33];
  int lo = 0, hi = count - 1;                       // the matrix of this block: last entry with tile0 <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].tile0 <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const CastEntry e = tab[lo];
  const int n = (int)(e.nk & 0xFFFFFFFF), k = (int)(e.nk >> 32);
  const int t = (int)((int64_t)blockIdx.x - e.tile0), tk = (k + 31) / 32;
  const int r0 = (t / tk) * 32, c0 = (t % tk) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = r0 + ty + 8 * j, c = c0 + tx;
    float v = 0.f;
    if (r < n && c < k) {
      v = e.src[(int64_t)r * k + c];
      e.dst[(int64_t)r * k + c] = __float2bfloat16(v);
    }
    tile[ty + 8 * j][tx] = v;
  }
  if (!e.dstT) return;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = c0 + ty + 8 * j, r = r0 + tx;     // dstT[c, r] = src[r, c]
    if (r < n && c < k) e.dstT[(int64_t)c * n + r] = __float2bfloat16(tile[tx][ty + 8 * j]);
  }
}

int tmae_multi_cast_transpose(const void* table, int count, int64_t total_tiles, void* stream_) {
  (void)hipGetLastError();
  if (count < 0 || total_tiles < 0 || total_tiles >= ((int64_t)1 << 31)) return TMAE_EARG;
  if (count == 0 || total_tiles == 0) return TMAE_OK;
  if (!table || ((uintptr_t)table & 7)) return TMAE_EARG;
  hipLaunchKernelGGL(multi_cast_transpose_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream_,
                     (const CastEntry*)table, count);
  return tmae_launch_status();
}
