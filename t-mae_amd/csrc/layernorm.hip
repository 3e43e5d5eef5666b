// Fused residual-add + LayerNorm (post-norm encoder layers: sst_basic_block.py:77-84, wca_block.py:93-102)
// forward and backward.  Pure HBM streaming: 16-byte accesses, several rows per wavefront, fp32
// statistics; the backward's gamma/beta column sums are accumulated per workgroup in registers/LDS and finished by
// a fixed-order second pass (deterministic, no atomics).
#include "common.h"

// Row layout: 8 consecutive channels per lane (16-byte accesses in bf16), D/8 adjacent lanes per row, 512/D rows per
// wavefront, two row groups per loop iteration (all loads of both issued before any arithmetic).
// VEC = D / 64 (2 or 4) is kept as the template parameter of the dispatch.

// y = LN(a + b) * gamma + beta ; xsum (optional) = a + b in T ; mean/rstd per row (f32)
// Round 3: bmask (optional, [m] in T) scales row r of b -- the cross layers add the attention update only to the query
// rows whose window holds previous-frame tokens (wca_block.py:93-96) -- and post (optional, [m, D]) is added to the
// normalised output: the block residual `x + encoder(x)` of SSTBlockV1 / WCABlock (spt_backbone.py:342-353) rides on
// the last norm of the block instead of an elementwise pass of its own.
template <class T, int VEC, bool EXTRA>
__global__ __launch_bounds__(256) void add_ln_fwd_kernel(const T* __restrict__ a, const T* __restrict__ b, int64_t m,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps,
                                                        T* __restrict__ xsum, T* __restrict__ y,
                                                        float* __restrict__ mean, float* __restrict__ rstd,
                                                        const T* __restrict__ bmask, const T* __restrict__ post) {
  constexpr int D = VEC * 64, LPR = D / 8, RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, sub = lane / LPR, cl = lane % LPR;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
  float g[8], bt[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { g[i] = gamma[cl * 8 + i]; bt[i] = beta[cl * 8 + i]; }
  const T* __restrict__ bsrc = b ? b : a;           // no second summand: read (and ignore) a again, branch-free
  for (int64_t r0 = wave * (2 * RPW); r0 < m; r0 += nwaves * (2 * RPW)) {
    float v[2][8], w[2][8], pz[2][8], bm[2];
    bool ok[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      // unconditional loads of a clamped row (rows past m are computed and dropped): a load under `if (r < m)` is
      // waited for before the next one is issued, so the four loads of an iteration would never overlap
      const int64_t r = r0 + u * RPW + sub, rc = r < m ? r : m - 1;
      ok[u] = r < m;
      load8<T>(a + rc * D + cl * 8, v[u]);
      load8<T>(bsrc + rc * D + cl * 8, w[u]);
      if (EXTRA && post) load8<T>(post + rc * D + cl * 8, pz[u]);          // wave-uniform
      bm[u] = (EXTRA && bmask) ? ld_f<T>(bmask + rc) : 1.0f;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int64_t r = r0 + u * RPW + sub;
      if (b) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[u][i] += w[u][i] * bm[u];
      }
      if (xsum) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { T t; st_f<T>(&t, v[u][i]); v[u][i] = ld_f<T>(&t); }   // statistics of the STORED (rounded) sum
        if (ok[u]) store8_nt<T>(xsum + r * D + cl * 8, v[u]);       // saved for the backward pass only
      }
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) s += v[u][i];
      const float mu = group_sum<LPR>(s) * (1.0f / D);
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) { const float dlt = v[u][i] - mu; q += dlt * dlt; }
      const float rs = rsqrtf(group_sum<LPR>(q) * (1.0f / D) + eps);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[u][i] = (v[u][i] - mu) * rs * g[i] + bt[i];
      if (EXTRA && post) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[u][i] += pz[u][i];
      }
      if (ok[u]) {
        store8<T>(y + r * D + cl * 8, v[u]);
        if (cl == 0) { mean[r] = mu; rstd[r] = rs; }
      }
    }
  }
}

// dx = rstd * (g*dy - mean_c(g*dy) - xhat * mean_c(g*dy*xhat)) ; partial dgamma/dbeta per workgroup
// Round 3, all optional: dx_skip = dx + skip (the gradient that reaches the first summand through the block residual,
// folded here instead of an AccumulateGrad add), dx_b = dx * bmask[row] (gradient of the masked second summand).
template <class T, int VEC, bool EXTRA>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, int64_t m,
                                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                                    const float* __restrict__ gamma, T* __restrict__ dx,
                                                    float* __restrict__ part /*[grid][2][D]*/,
                                                    const T* __restrict__ skip, T* __restrict__ dx_skip,
                                                    const T* __restrict__ bmask, T* __restrict__ dx_b) {
  constexpr int D = VEC * 64, LPR = D / 8, RPW = 64 / LPR;
  __shared__ float red[4][2][D];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, sub = lane / LPR, cl = lane % LPR;
  const int64_t wave = (int64_t)blockIdx.x * 4 + w, nwaves = (int64_t)gridDim.x * 4;
  float g[8], dg[8], db[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { g[i] = gamma[cl * 8 + i]; dg[i] = 0.f; db[i] = 0.f; }
  for (int64_t r0 = wave * (2 * RPW); r0 < m; r0 += nwaves * (2 * RPW)) {
    float v[2][8], d[2][8], sk[2][8], mu[2], rs[2], bm[2];
    bool ok[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int64_t r = r0 + u * RPW + sub, rc = r < m ? r : m - 1;
      ok[u] = r < m;
      load8<T>(x + rc * D + cl * 8, v[u]);                        // unconditional (clamped), dy masked below
      load8<T>(dy + rc * D + cl * 8, d[u]);
      if (EXTRA && skip) load8<T>(skip + rc * D + cl * 8, sk[u]);          // wave-uniform
      bm[u] = (EXTRA && bmask) ? ld_f<T>(bmask + rc) : 1.0f;
      mu[u] = mean[rc];
      rs[u] = rstd[rc];
#pragma unroll
      for (int i = 0; i < 8; ++i) d[u][i] = ok[u] ? d[u][i] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        v[u][i] = (v[u][i] - mu[u]) * rs[u];                 // xhat
        dg[i] += d[u][i] * v[u][i];
        db[i] += d[u][i];
        d[u][i] *= g[i];
        s1 += d[u][i];
        s2 += d[u][i] * v[u][i];
      }
      s1 = group_sum<LPR>(s1) * (1.0f / D);
      s2 = group_sum<LPR>(s2) * (1.0f / D);
#pragma unroll
      for (int i = 0; i < 8; ++i) d[u][i] = rs[u] * (d[u][i] - s1 - v[u][i] * s2);
      const int64_t ro = (r0 + u * RPW + sub) * D + cl * 8;
      if (ok[u] && (!EXTRA || dx)) store8<T>(dx + ro, d[u]);
      if (EXTRA && dx_b) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[u][i] = d[u][i] * bm[u];
        if (ok[u]) store8<T>(dx_b + ro, v[u]);
      }
      if (EXTRA && dx_skip) {
#pragma unroll
        for (int i = 0; i < 8; ++i) d[u][i] += sk[u][i];
        if (ok[u]) store8<T>(dx_skip + ro, d[u]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) { dg[i] = cross_group_sum<LPR>(dg[i]); db[i] = cross_group_sum<LPR>(db[i]); }
  if (sub == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { red[w][0][cl * 8 + i] = dg[i]; red[w][1][cl * 8 + i] = db[i]; }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 2 * D; e += 256) {
    const int which = e / D, c = e % D;
    part[((int64_t)blockIdx.x * 2 + which) * D + c] = red[0][which][c] + red[1][which][c] + red[2][which][c] + red[3][which][c];
  }
}

// one workgroup per 8 columns: 8 column lanes x 32 row lanes walk the partial rows (4 independent accumulators per
// thread), then a fixed-order LDS tree
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float* __restrict__ part, int nblocks, int d,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float red[32][9];
  const int cl = threadIdx.x & 7, rl = threadIdx.x >> 3;
  const int e = blockIdx.x * 8 + cl;                  // column in [0, 2d)
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (e < 2 * d) {
    const int64_t ld = 2 * (int64_t)d;
    int b = rl;
    for (; b + 96 < nblocks; b += 128) {
      a0 += part[(int64_t)b * ld + e];
      a1 += part[(int64_t)(b + 32) * ld + e];
      a2 += part[(int64_t)(b + 64) * ld + e];
      a3 += part[(int64_t)(b + 96) * ld + e];
    }
    for (; b < nblocks; b += 32) a0 += part[(int64_t)b * ld + e];
  }
  red[rl][cl] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (rl == 0 && e < 2 * d) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 32; ++r) t += red[r][cl];
    if (e < d) dgamma[e] = t; else dbeta[e - d] = t;
  }
}

static int ln_grid(int64_t m) {
  int64_t g = (m + 127) / 128;          // >= 32 rows per wave
  if (g > 1024) g = 1024;
  if (g < 1) g = 1;
  return (int)g;
}

int tmae_add_layernorm_fwd(const void* a, const void* b, int dtype, int64_t m, int d, const float* gamma,
                           const float* beta, float eps, void* xsum, void* y, float* mean, float* rstd,
                           const void* bmask, const void* post, void* stream_) {
  (void)hipGetLastError();
  hipStream_t stream = (hipStream_t)stream_;
  if (m < 0 || (d != 128 && d != 256)) return TMAE_EARG;
  if (m == 0) return TMAE_OK;
  if (!a || !gamma || !beta || !y || !mean || !rstd) return TMAE_EARG;
  dim3 grid(ln_grid(m)), block(256);
  const bool extra = (b && bmask) || post;      // the plain form keeps its lean instantiation
#define FWD(T, V)                                                                                              \
  do {                                                                                                         \
    if (extra)                                                                                                 \
      hipLaunchKernelGGL((add_ln_fwd_kernel<T, V, true>), grid, block, 0, stream, (const T*)a, (const T*)b, m, gamma, beta, \
                         eps, (T*)xsum, (T*)y, mean, rstd, (const T*)(b ? bmask : nullptr), (const T*)post);    \
    else                                                                                                       \
      hipLaunchKernelGGL((add_ln_fwd_kernel<T, V, false>), grid, block, 0, stream, (const T*)a, (const T*)b, m, gamma, beta, \
                         eps, (T*)xsum, (T*)y, mean, rstd, (const T*)nullptr, (const T*)nullptr);              \
  } while (0)
  if (dtype == TMAE_F32) { if (d == 128) FWD(float, 2); else FWD(float, 4); }
  else if (dtype == TMAE_BF16) { if (d == 128) FWD(__hip_bfloat16, 2); else FWD(__hip_bfloat16, 4); }
  else return TMAE_EDTYPE;
#undef FWD
  return tmae_launch_status();
}

size_t tmae_layernorm_bwd_workspace(int64_t m, int d) { return (size_t)ln_grid(m) * 2 * d * 4 + 256; }

int tmae_layernorm_bwd(const void* dy, const void* x, int dtype, int64_t m, int d, const float* mean,
                       const float* rstd, const float* gamma, void* dx, float* dgamma, float* dbeta,
                       const void* skip, void* dx_skip, const void* bmask, void* dx_b, void* wsp, size_t ws_bytes,
                       void* stream_) {
  (void)hipGetLastError();
  hipStream_t stream = (hipStream_t)stream_;
  if (m < 0 || (d != 128 && d != 256)) return TMAE_EARG;
  if (!dgamma || !dbeta || !gamma) return TMAE_EARG;
  if (m > 0 && (!dy || !x || !mean || !rstd || (!dx && !dx_skip && !dx_b))) return TMAE_EARG;
  if ((skip == nullptr) != (dx_skip == nullptr) || (bmask == nullptr) != (dx_b == nullptr)) return TMAE_EARG;
  const int nb = ln_grid(m);
  WsCarver ws(wsp, ws_bytes);
  float* part = ws.take<float>((size_t)nb * 2 * d);
  if (!ws.ok) return TMAE_EWS;
  dim3 grid(nb), block(256);
  const bool extra = skip || bmask;
#define BWD(T, V)                                                                                                   \
  do {                                                                                                              \
    if (extra)                                                                                                      \
      hipLaunchKernelGGL((ln_bwd_kernel<T, V, true>), grid, block, 0, stream, (const T*)dy, (const T*)x, m, mean, rstd, gamma, \
                         (T*)dx, part, (const T*)skip, (T*)dx_skip, (const T*)bmask, (T*)dx_b);                     \
    else                                                                                                            \
      hipLaunchKernelGGL((ln_bwd_kernel<T, V, false>), grid, block, 0, stream, (const T*)dy, (const T*)x, m, mean, rstd, gamma, \
                         (T*)dx, part, (const T*)nullptr, (T*)nullptr, (const T*)nullptr, (T*)nullptr);           \
  } while (0)
  if (dtype == TMAE_F32) { if (d == 128) BWD(float, 2); else BWD(float, 4); }
  else if (dtype == TMAE_BF16) { if (d == 128) BWD(__hip_bfloat16, 2); else BWD(__hip_bfloat16, 4); }
  else return TMAE_EDTYPE;
#undef BWD
  hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3(tmae_cdiv(2 * d, 8)), dim3(256), 0, stream, part, nb, d, dgamma,
                     dbeta);
  return tmae_launch_status();
}
