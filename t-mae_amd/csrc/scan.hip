// Device-wide exclusive prefix sum of int32: tile scan -> (recursive) scan of tile sums -> add.
// The arrays scanned on this path are small (<= a few million flags), so the three-launch form is
// bandwidth-trivial; it keeps the library free of temp-storage negotiation.
#include "common.h"

#define SCAN_THREADS 256
#define SCAN_ITEMS 8
#define SCAN_TILE (SCAN_THREADS * SCAN_ITEMS)

__global__ __launch_bounds__(SCAN_THREADS) void scan_tile_kernel(const int32_t* __restrict__ in,
                                                                 int32_t* __restrict__ out, int64_t n,
                                                                 int32_t* __restrict__ block_sums) {
  __shared__ int wsum[SCAN_THREADS / TMAE_WAVE];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  int v[SCAN_ITEMS];
  int run = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    int64_t i = base + k;
    int t = (i < n) ? in[i] : 0;
    v[k] = run;
    run += t;
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = run;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  int woff = 0;
  for (int i = 0; i < w; ++i) woff += wsum[i];
  const int excl = woff + inc - run;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    int64_t i = base + k;
    if (i < n) out[i] = excl + v[k];
  }
  if (threadIdx.x == SCAN_THREADS - 1) block_sums[blockIdx.x] = woff + inc;
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_add_kernel(int32_t* __restrict__ out, int64_t n,
                                                                const int32_t* __restrict__ block_off) {
  const int off = block_off[blockIdx.x];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    int64_t i = base + (int64_t)k * SCAN_THREADS + threadIdx.x;
    if (i < n) out[i] += off;
  }
}

__global__ void scan_total_kernel(const int32_t* in, const int32_t* out, int64_t n, int32_t* total) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *total = (n > 0) ? in[n - 1] + out[n - 1] : 0;
}

// n <= SCAN_SMALL_MAX (the window counts of every stage, ~2 k .. 29 k, are): ONE workgroup, one launch -- each thread
// owns a contiguous run of ceil(n / 1024) elements (sum, block scan of the sums, exclusive writes) and the total comes
// out of the same kernel; the three-launch form plus the total kernel cost four ~5 us dispatches per scan, ~25 scans a step.
#define SCAN_SMALL_THREADS 1024
#define SCAN_SMALL_MAX 65536
__global__ __launch_bounds__(SCAN_SMALL_THREADS) void scan_small_kernel(const int32_t* __restrict__ in,
                                                                       int32_t* __restrict__ out, int n,
                                                                       int32_t* __restrict__ total) {
  __shared__ int wsum[SCAN_SMALL_THREADS / TMAE_WAVE];
  const int per = (n + SCAN_SMALL_THREADS - 1) / SCAN_SMALL_THREADS;
  const int lo = min((int)threadIdx.x * per, n), hi = min(lo + per, n);
  int run = 0;
  for (int i = lo; i < hi; ++i) run += in[i];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = run;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  int woff = 0;
  for (int i = 0; i < w; ++i) woff += wsum[i];
  int acc = woff + inc - run;
  for (int i = lo; i < hi; ++i) {
    const int t = in[i];                               // (in and out may not alias: out[i] is written after in[i] is read)
    out[i] = acc;
    acc += t;
  }
  if (total && threadIdx.x == SCAN_SMALL_THREADS - 1) *total = woff + inc;
}

size_t tmae_scan_i32_workspace(int64_t n) {
  size_t bytes = 0;
  while (n > 1) {
    int64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    bytes += 2 * tmae_align((size_t)nb * sizeof(int32_t));
    if (nb <= 1) break;
    n = nb;
  }
  return bytes + 512;
}

static int scan_rec(const int32_t* in, int32_t* out, int64_t n, WsCarver& ws, hipStream_t stream) {
  const int64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
  int32_t* sums = ws.take<int32_t>((size_t)nb);
  int32_t* sums_scanned = ws.take<int32_t>((size_t)nb);
  if (!ws.ok) return TMAE_EWS;
  hipLaunchKernelGGL(scan_tile_kernel, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, stream, in, out, n, sums);
  if (nb > 1) {
    int r = scan_rec(sums, sums_scanned, nb, ws, stream);
    if (r != 0) return r;
    hipLaunchKernelGGL(scan_add_kernel, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, stream, out, n, sums_scanned);
  }
  return 0;
}

int tmae_scan_i32(const int32_t* in, int32_t* out, int64_t n, int32_t* total, void* wsp, size_t ws_bytes,
                  hipStream_t stream) {
  (void)hipGetLastError();   // drop stale errors of other runtime users: the return code is about OUR launches
  if (n < 0 || (n > 0 && (!in || !out))) return TMAE_EARG;
  if (n > 0 && n <= SCAN_SMALL_MAX && in != out) {
    hipLaunchKernelGGL(scan_small_kernel, dim3(1), dim3(SCAN_SMALL_THREADS), 0, stream, in, out, (int)n, total);
    return tmae_launch_status();
  }
  if (n > 0) {
    WsCarver ws(wsp, ws_bytes);
    int r = scan_rec(in, out, n, ws, stream);
    if (r != 0) return r;
  }
  if (total) hipLaunchKernelGGL(scan_total_kernel, dim3(1), dim3(64), 0, stream, in, out, n, total);
  return tmae_launch_status();
}
