// A3: MAE random masking from injected noise.  The reference argsorts the noise of every sample
// (common_utils.py:49-63); only the *set* of the len_keep smallest values is used, so this is a
// per-sample radix SELECT on the float bits (4 x 8-bit passes, histograms in LDS) -- no sort.
// Ties on the threshold value are resolved by ascending voxel index (= stable argsort).
#include "common.h"

#define MASK_THREADS 1024

__device__ __forceinline__ int block_excl_scan_1024(int v, int* lds_wave /*16*/) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  __syncthreads();
  if (lane == 63) lds_wave[w] = inc;
  __syncthreads();
  int woff = 0;
  for (int i = 0; i < w; ++i) woff += lds_wave[i];
  return woff + inc - v;
}

__global__ __launch_bounds__(MASK_THREADS) void mask_select_kernel(const float* __restrict__ noise,
                                                                  const int32_t* __restrict__ sample_offsets,
                                                                  double keep_frac, float* __restrict__ mask,
                                                                  int32_t* __restrict__ keepflag) {
  __shared__ int hist[256];
  __shared__ int wtot[16];
  __shared__ unsigned s_prefix;
  __shared__ int s_k, s_neq;
  const int b = blockIdx.x;
  const int lo = sample_offsets[b], hi = sample_offsets[b + 1];
  const int L = hi - lo;
  if (L <= 0) return;
  const int len_keep = (int)((double)L * keep_frac);     // python: int(L * (1 - mask_ratio))
  if (len_keep <= 0 || len_keep >= L) {
    const bool keep = len_keep >= L;
    for (int i = threadIdx.x; i < L; i += MASK_THREADS) {
      mask[lo + i] = keep ? 0.f : 1.f;
      keepflag[lo + i] = keep ? 1 : 0;
    }
    return;
  }
  if (threadIdx.x == 0) { s_prefix = 0u; s_k = len_keep; }
  // One workgroup per sample = 16 waves on one CU: every pass over the sample's noise used to be ~46 dependent load -> LDS-atomic
  // round trips per thread (5 passes x ~35 us for 190 KB that sit in L2; round 6).  Samples of up to MASK_REGS x 1024 voxels are
  // read ONCE, all loads in flight together, and the four digit passes run on registers.
  constexpr int MASK_REGS = 64;
  const bool in_regs = L <= MASK_REGS * MASK_THREADS;                    // workgroup-uniform
  unsigned vals[MASK_REGS];
  if (in_regs) {
#pragma unroll
    for (int q = 0; q < MASK_REGS; ++q) {
      const int i = q * MASK_THREADS + (int)threadIdx.x;
      vals[q] = (q * MASK_THREADS < L) ? __float_as_uint(noise[lo + min(i, L - 1)]) : 0u;
    }
  }
  for (int pass = 3; pass >= 0; --pass) {
    const int shift = pass * 8;
    const unsigned hi_mask = (pass == 3) ? 0u : (0xFFFFFFFFu << (shift + 8));
    for (int i = threadIdx.x; i < 256; i += MASK_THREADS) hist[i] = 0;
    __syncthreads();
    const unsigned prefix = s_prefix;
    if (in_regs) {
#pragma unroll
      for (int q = 0; q < MASK_REGS; ++q) {
        if (q * MASK_THREADS < L) {                                      // workgroup-uniform
          const bool in = q * MASK_THREADS + (int)threadIdx.x < L && (vals[q] & hi_mask) == (prefix & hi_mask);
          const unsigned digit = (vals[q] >> shift) & 255u;
          if (pass == 3) {
            // the top byte of uniform noise in [0, 1) is sign + 7 exponent bits: three values hold 7/8 of the elements -- a wave
            // counts its lanes per distinct digit with ballots and adds once per (wave, digit)
            unsigned long long todo = __ballot(in);
            while (todo) {
              const int leader = __ffsll((long long)todo) - 1;
              const unsigned dl = (unsigned)__shfl((int)digit, leader, 64);
              const unsigned long long same = __ballot(in && digit == dl) & todo;
              if ((int)(threadIdx.x & 63) == leader) atomicAdd(&hist[dl], __popcll(same));
              todo &= ~same;
            }
          } else if (in) {
            atomicAdd(&hist[digit], 1);
          }
        }
      }
    } else {
      for (int i = threadIdx.x; i < L; i += MASK_THREADS) {
        const unsigned bits = __float_as_uint(noise[lo + i]);
        if ((bits & hi_mask) == (prefix & hi_mask)) atomicAdd(&hist[(bits >> shift) & 255u], 1);
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int k = s_k, cum = 0, d = 0;
      for (; d < 255; ++d) {
        if (cum + hist[d] >= k) break;
        cum += hist[d];
      }
      s_k = k - cum;
      s_neq = hist[d];                 // after the last pass: how many elements equal the threshold value
      s_prefix = prefix | ((unsigned)d << shift);
    }
    __syncthreads();
  }
  const unsigned kth = s_prefix;   // bits of the len_keep-th smallest value
  const int k_eq = s_k;            // how many elements equal to it are kept (lowest indices first)
  if (in_regs && k_eq == s_neq) {
    // every element that equals the threshold is kept (with float noise: there is exactly one) -- no tie ranks: coalesced
    // stores from the registers.  (The ranked path below walks a contiguous chunk per thread: 64 cache lines per wave-load.)
#pragma unroll
    for (int q = 0; q < MASK_REGS; ++q) {
      const int i = q * MASK_THREADS + (int)threadIdx.x;
      if (q * MASK_THREADS < L && i < L) {
        const bool keep = vals[q] <= kth;
        mask[lo + i] = keep ? 0.f : 1.f;
        keepflag[lo + i] = keep ? 1 : 0;
      }
    }
    return;
  }
  // contiguous chunk per thread keeps index order for the tie ranks
  const int chunk = (L + MASK_THREADS - 1) / MASK_THREADS;
  const int c0 = min((int)threadIdx.x * chunk, L), c1 = min(c0 + chunk, L);
  int eq = 0;
  for (int i = c0; i < c1; ++i) eq += (__float_as_uint(noise[lo + i]) == kth) ? 1 : 0;
  int tie_rank = block_excl_scan_1024(eq, wtot);
  for (int i = c0; i < c1; ++i) {
    const unsigned bits = __float_as_uint(noise[lo + i]);
    bool keep = bits < kth;
    if (bits == kth) { keep = tie_rank < k_eq; ++tie_rank; }
    mask[lo + i] = keep ? 0.f : 1.f;
    keepflag[lo + i] = keep ? 1 : 0;
  }
}

__global__ __launch_bounds__(256) void mask_compact_kernel(const int32_t* __restrict__ keepflag,
                                                          const int32_t* __restrict__ pos, int64_t m,
                                                          int32_t* __restrict__ vis_index) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < m && keepflag[i]) vis_index[pos[i]] = (int32_t)i;
}

size_t tmae_random_mask_workspace(int64_t m, int batch) {
  (void)batch;
  return 2 * tmae_align((size_t)m * 4) + tmae_scan_i32_workspace(m) + 1024;
}

int tmae_random_mask(const float* noise, const int32_t* sample_offsets, int64_t m, int batch, double keep_frac,
                     float* mask, int32_t* vis_index, int32_t* n_vis, void* wsp, size_t ws_bytes, void* stream_) {
  (void)hipGetLastError();   // drop stale errors of other runtime users: the return code is about OUR launches
  hipStream_t stream = (hipStream_t)stream_;
  if (m < 0 || batch <= 0 || !sample_offsets || !n_vis || m >= (1ll << 31)) return TMAE_EARG;
  if (m > 0 && (!noise || !mask || !vis_index)) return TMAE_EARG;
  WsCarver ws(wsp, ws_bytes);
  int32_t* keepflag = ws.take<int32_t>((size_t)m);
  int32_t* pos = ws.take<int32_t>((size_t)m);
  size_t sb = tmae_scan_i32_workspace(m);
  char* scanws = ws.take<char>(sb);
  if (!ws.ok) return TMAE_EWS;
  if (m > 0) {
    (void)hipMemsetAsync(keepflag, 0, (size_t)m * 4, stream);
    hipLaunchKernelGGL(mask_select_kernel, dim3(batch), dim3(MASK_THREADS), 0, stream, noise, sample_offsets,
                       keep_frac, mask, keepflag);
  }
  int r = tmae_scan_i32(keepflag, pos, m, n_vis, scanws, sb, stream);
  if (r) return r;
  if (m > 0)
    hipLaunchKernelGGL(mask_compact_kernel, dim3(tmae_cdiv(m, 256)), dim3(256), 0, stream, keepflag, pos, m,
                       vis_index);
  return tmae_launch_status();
}
