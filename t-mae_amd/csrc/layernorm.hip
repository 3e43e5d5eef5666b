// Fused residual-add + LayerNorm (post-norm encoder layers: sst_basic_block.py:77-84, wca_block.py:93-102)
// forward and backward.  Pure HBM streaming: one wavefront per row, d/64 contiguous elements per lane, fp32
// statistics; the backward's gamma/beta column sums are accumulated per workgroup in registers/LDS and finished by
// a fixed-order second pass (deterministic, no atomics).
#include "common.h"

template <class T, int VEC>
__device__ __forceinline__ void load_vec(const T* p, float* v) {
#pragma unroll
  for (int i = 0; i < VEC; ++i) v[i] = ld_f<T>(p + i);
}
template <class T, int VEC>
__device__ __forceinline__ void store_vec(T* p, const float* v) {
#pragma unroll
  for (int i = 0; i < VEC; ++i) st_f<T>(p + i, v[i]);
}

// y = LN(a + b) * gamma + beta ; xsum (optional) = a + b in T ; mean/rstd per row (f32)
template <class T, int VEC>
__global__ __launch_bounds__(256) void add_ln_fwd_kernel(const T* __restrict__ a, const T* __restrict__ b, int64_t m,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps,
                                                        T* __restrict__ xsum, T* __restrict__ y,
                                                        float* __restrict__ mean, float* __restrict__ rstd) {
  constexpr int D = VEC * 64;
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
  float g[VEC], bt[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { g[i] = gamma[lane * VEC + i]; bt[i] = beta[lane * VEC + i]; }
  for (int64_t r = wave; r < m; r += nwaves) {
    float v[VEC], w[VEC];
    load_vec<T, VEC>(a + r * D + lane * VEC, v);
    if (b) {
      load_vec<T, VEC>(b + r * D + lane * VEC, w);
#pragma unroll
      for (int i = 0; i < VEC; ++i) v[i] += w[i];
    }
    if (xsum) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) { T t; st_f<T>(&t, v[i]); v[i] = ld_f<T>(&t); }   // statistics of the STORED (rounded) sum
      store_vec<T, VEC>(xsum + r * D + lane * VEC, v);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) s += v[i];
    const float mu = wave_sum(s) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) { const float dlt = v[i] - mu; q += dlt * dlt; }
    const float rs = rsqrtf(wave_sum(q) * (1.0f / D) + eps);
#pragma unroll
    for (int i = 0; i < VEC; ++i) v[i] = (v[i] - mu) * rs * g[i] + bt[i];
    store_vec<T, VEC>(y + r * D + lane * VEC, v);
    if (lane == 0) { mean[r] = mu; rstd[r] = rs; }
  }
}

// dx = rstd * (g*dy - mean_c(g*dy) - xhat * mean_c(g*dy*xhat)) ; partial dgamma/dbeta per workgroup
template <class T, int VEC>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, int64_t m,
                                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                                    const float* __restrict__ gamma, T* __restrict__ dx,
                                                    float* __restrict__ part /*[grid][2][D]*/) {
  constexpr int D = VEC * 64;
  __shared__ float red[4][2][D];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t wave = (int64_t)blockIdx.x * 4 + w, nwaves = (int64_t)gridDim.x * 4;
  float g[VEC], dg[VEC], db[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { g[i] = gamma[lane * VEC + i]; dg[i] = 0.f; db[i] = 0.f; }
  for (int64_t r = wave; r < m; r += nwaves) {
    float v[VEC], d[VEC];
    load_vec<T, VEC>(x + r * D + lane * VEC, v);
    load_vec<T, VEC>(dy + r * D + lane * VEC, d);
    const float mu = mean[r], rs = rstd[r];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      v[i] = (v[i] - mu) * rs;                 // xhat
      dg[i] += d[i] * v[i];
      db[i] += d[i];
      d[i] *= g[i];
      s1 += d[i];
      s2 += d[i] * v[i];
    }
    s1 = wave_sum(s1) * (1.0f / D);
    s2 = wave_sum(s2) * (1.0f / D);
#pragma unroll
    for (int i = 0; i < VEC; ++i) d[i] = rs * (d[i] - s1 - v[i] * s2);
    store_vec<T, VEC>(dx + r * D + lane * VEC, d);
  }
#pragma unroll
  for (int i = 0; i < VEC; ++i) { red[w][0][lane * VEC + i] = dg[i]; red[w][1][lane * VEC + i] = db[i]; }
  __syncthreads();
  for (int e = threadIdx.x; e < 2 * D; e += 256) {
    const int which = e / D, c = e % D;
    part[((int64_t)blockIdx.x * 2 + which) * D + c] = red[0][which][c] + red[1][which][c] + red[2][which][c] + red[3][which][c];
  }
}

// one workgroup per 16 columns: 16 column lanes x 16 row lanes walk the partial rows, then a fixed-order LDS tree
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float* __restrict__ part, int nblocks, int d,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float red[16][17];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + cl;                 // column in [0, 2d)
  float acc = 0.f;
  if (e < 2 * d)
    for (int b = rl; b < nblocks; b += 16) acc += part[(int64_t)b * 2 * d + e];
  red[rl][cl] = acc;
  __syncthreads();
  if (rl == 0 && e < 2 * d) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += red[r][cl];
    if (e < d) dgamma[e] = t; else dbeta[e - d] = t;
  }
}

static int ln_grid(int64_t m) {
  int64_t g = (m + 31) / 32;            // >= 8 rows per wave
  if (g > 512) g = 512;
  if (g < 1) g = 1;
  return (int)g;
}

int tmae_add_layernorm_fwd(const void* a, const void* b, int dtype, int64_t m, int d, const float* gamma,
                           const float* beta, float eps, void* xsum, void* y, float* mean, float* rstd,
                           void* stream_) {
  (void)hipGetLastError();
  hipStream_t stream = (hipStream_t)stream_;
  if (m < 0 || (d != 128 && d != 256)) return TMAE_EARG;
  if (m == 0) return TMAE_OK;
  if (!a || !gamma || !beta || !y || !mean || !rstd) return TMAE_EARG;
  dim3 grid(ln_grid(m)), block(256);
#define FWD(T, V)                                                                                              \
  hipLaunchKernelGGL((add_ln_fwd_kernel<T, V>), grid, block, 0, stream, (const T*)a, (const T*)b, m, gamma, beta, \
                     eps, (T*)xsum, (T*)y, mean, rstd)
  if (dtype == TMAE_F32) { if (d == 128) FWD(float, 2); else FWD(float, 4); }
  else if (dtype == TMAE_BF16) { if (d == 128) FWD(__hip_bfloat16, 2); else FWD(__hip_bfloat16, 4); }
  else return TMAE_EDTYPE;
#undef FWD
  return tmae_launch_status();
}

size_t tmae_layernorm_bwd_workspace(int64_t m, int d) { return (size_t)ln_grid(m) * 2 * d * 4 + 256; }

int tmae_layernorm_bwd(const void* dy, const void* x, int dtype, int64_t m, int d, const float* mean,
                       const float* rstd, const float* gamma, void* dx, float* dgamma, float* dbeta, void* wsp,
                       size_t ws_bytes, void* stream_) {
  (void)hipGetLastError();
  hipStream_t stream = (hipStream_t)stream_;
  if (m < 0 || (d != 128 && d != 256)) return TMAE_EARG;
  if (!dgamma || !dbeta || !gamma) return TMAE_EARG;
  if (m > 0 && (!dy || !x || !mean || !rstd || !dx)) return TMAE_EARG;
  const int nb = ln_grid(m);
  WsCarver ws(wsp, ws_bytes);
  float* part = ws.take<float>((size_t)nb * 2 * d);
  if (!ws.ok) return TMAE_EWS;
  dim3 grid(nb), block(256);
#define BWD(T, V)                                                                                                   \
  hipLaunchKernelGGL((ln_bwd_kernel<T, V>), grid, block, 0, stream, (const T*)dy, (const T*)x, m, mean, rstd, gamma, \
                     (T*)dx, part)
  if (dtype == TMAE_F32) { if (d == 128) BWD(float, 2); else BWD(float, 4); }
  else if (dtype == TMAE_BF16) { if (d == 128) BWD(__hip_bfloat16, 2); else BWD(__hip_bfloat16, 4); }
  else return TMAE_EDTYPE;
#undef BWD
  hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3(tmae_cdiv(2 * d, 16)), dim3(256), 0, stream, part, nb, d, dgamma,
                     dbeta);
  return tmae_launch_status();
}
