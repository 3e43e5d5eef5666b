// BatchNorm1d (training statistics) + ReLU over the rows of a token / point list [m, c], forward and backward
// (spconv_utils.post_act_block: conv + BatchNorm1d(eps 1e-3) + ReLU, pcdet/utils/spconv_utils.py:37-56; the VFE's
// Linear + BatchNorm1d + ReLU, model_utils/network_utils.py:25-40).  Pure HBM streaming with 16-byte accesses,
// several rows per wavefront; column sums are accumulated per lane in registers, per workgroup in LDS, and
// finished in a fixed order (deterministic, no atomics); the final combination runs in double.
#include "common.h"

// Row layout: 8 consecutive channels per lane (16-byte accesses in bf16), C/8 adjacent lanes per row, 512/C rows per
// wavefront, two row groups per loop iteration.  VEC = C / 64 (1, 2 or 4) stays the template parameter of the dispatch.
#define BN_LAYOUT                                                                   \
  constexpr int C = VEC * 64, LPR = C / 8, RPW = 64 / LPR;                          \
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, sub = lane / LPR, cl = lane % LPR; \
  const int64_t wave = (int64_t)blockIdx.x * 4 + w, nwaves = (int64_t)gridDim.x * 4; \
  (void)w

// workgroup partial of two per-channel sums held as s1[8], s2[8] per lane -> part[block][2][C]
template <int C, int LPR>
__device__ __forceinline__ void bn_block_partial(float* s1, float* s2, float (*red)[2][C], int w, int sub, int cl,
                                                 float* __restrict__ part) {
#pragma unroll
  for (int i = 0; i < 8; ++i) { s1[i] = cross_group_sum<LPR>(s1[i]); s2[i] = cross_group_sum<LPR>(s2[i]); }
  if (sub == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { red[w][0][cl * 8 + i] = s1[i]; red[w][1][cl * 8 + i] = s2[i]; }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 2 * C; e += 256) {
    const int which = e / C, c = e % C;
    part[(int64_t)blockIdx.x * 2 * C + e] = red[0][which][c] + red[1][which][c] + red[2][which][c] + red[3][which][c];
  }
}

// partial column sums of (x - p) and (x - p)^2, p = a per-channel pivot (row 0 of x when `use_pivot`, else 0).
// Shifting by a sample of the data keeps E[(x-p)^2] - E[x-p]^2 free of the cancellation that the raw moments suffer
// when |mean| >> std (dense BEV maps whose inactive region is one constant per channel); torch uses Welford there.
// Block 0 also writes the pivot behind the partials: part[grid*2*C + c].
template <class T, int VEC>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ x, int64_t m, int use_pivot,
                                                      float* __restrict__ part /*[grid][2][C] + [C]*/) {
  BN_LAYOUT;
  __shared__ float red[4][2][C];
  float s1[8], s2[8], pv[8];
  load8<T>(x + cl * 8, pv);
#pragma unroll
  for (int i = 0; i < 8; ++i) { s1[i] = 0.f; s2[i] = 0.f; pv[i] = use_pivot ? pv[i] : 0.f; }
  if (blockIdx.x == 0 && w == 0 && sub == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) part[(int64_t)gridDim.x * 2 * C + cl * 8 + i] = pv[i];
  }
  for (int64_t r0 = wave * (2 * RPW); r0 < m; r0 += nwaves * (2 * RPW)) {
    float v[2][8];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      // unconditional load of a clamped row, masked afterwards: a load under `if (r < m)` is waited for
      // (s_waitcnt vmcnt(0)) before the next one is issued, i.e. the two rows of an iteration never overlap
      const int64_t r = r0 + u * RPW + sub;
      load8<T>(x + (r < m ? r : m - 1) * C + cl * 8, v[u]);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[u][i] = r < m ? v[u][i] - pv[i] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) { s1[i] += v[u][i]; s2[i] += v[u][i] * v[u][i]; }
  }
  bn_block_partial<C, LPR>(s1, s2, red, w, sub, cl, part);
}

// mean / biased variance / rstd per channel from the partials (double accumulation, fixed order)
// (1024 threads = 16 channels x 64 partial rows each, four independent loads in flight per thread: with 256 threads the
// 64 dependent round trips of a thread made this tiny kernel 16 us, 42 times per step)
#define BN_FIN_RL 64
__device__ __forceinline__ void bn_partial_sums(const float* __restrict__ part, int nblocks, int c, int ch, int rl,
                                                double& a, double& b) {
  a = 0.0; b = 0.0;
  if (ch >= c) return;
  int k = rl;
  for (; k + 3 * BN_FIN_RL < nblocks; k += 4 * BN_FIN_RL) {
    float u[4], v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      u[q] = part[(int64_t)(k + q * BN_FIN_RL) * 2 * c + ch];
      v[q] = part[(int64_t)(k + q * BN_FIN_RL) * 2 * c + c + ch];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { a += u[q]; b += v[q]; }
  }
  for (; k < nblocks; k += BN_FIN_RL) { a += part[(int64_t)k * 2 * c + ch]; b += part[(int64_t)k * 2 * c + c + ch]; }
}

__global__ __launch_bounds__(16 * BN_FIN_RL) void bn_finalize_kernel(const float* __restrict__ part, int nblocks, int c,
                                                                    double m, float eps, float* __restrict__ mean,
                                                                    float* __restrict__ var, float* __restrict__ rstd) {
  __shared__ double red[BN_FIN_RL][17][2];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int ch = blockIdx.x * 16 + cl;
  double a, b;
  bn_partial_sums(part, nblocks, c, ch, rl, a, b);
  red[rl][cl][0] = a;
  red[rl][cl][1] = b;
  __syncthreads();
  if (rl == 0 && ch < c) {
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int r = 0; r < BN_FIN_RL; ++r) { s1 += red[r][cl][0]; s2 += red[r][cl][1]; }
    const double d = s1 / m;                                   // mean of (x - pivot)
    const double mu = (double)part[(int64_t)nblocks * 2 * c + ch] + d;
    double vr = s2 / m - d * d;
    if (vr < 0.0) vr = 0.0;
    mean[ch] = (float)mu;
    var[ch] = (float)vr;
    rstd[ch] = (float)(1.0 / sqrt(vr + (double)eps));
  }
}

template <class T, int VEC>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, int64_t m,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      int relu, T* __restrict__ y, const T* __restrict__ post = nullptr) {
  // post (optional, [m, C]): y = relu?(norm(x)) + post -- the residual shortcut of SSTBEVBackbone (sst_bev_backbone.py:35-41)
  // added where the normalised row is in registers anyway, one rounding instead of two
  BN_LAYOUT;
  float sc[8], sh[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int ch = cl * 8 + i;
    sc[i] = rstd[ch] * gamma[ch];
    sh[i] = beta[ch] - mean[ch] * sc[i];
  }
  for (int64_t r0 = wave * (2 * RPW); r0 < m; r0 += nwaves * (2 * RPW)) {
    float v[2][8];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int64_t r = r0 + u * RPW + sub;
      load8<T>(x + (r < m ? r : m - 1) * C + cl * 8, v[u]);      // unconditional (clamped): both rows in flight
    }
    float pv[2][8];
    if (post) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int64_t r = r0 + u * RPW + sub;
        load8<T>(post + (r < m ? r : m - 1) * C + cl * 8, pv[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int64_t r = r0 + u * RPW + sub;
      if (r < m) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float z = v[u][i] * sc[i] + sh[i];
          v[u][i] = relu ? fmaxf(z, 0.f) : z;
          if (post) v[u][i] += pv[u][i];
        }
        store8<T>(y + r * C + cl * 8, v[u]);
      }
    }
  }
}

// partial sums of dz and dz * xhat, dz = dy * (z > 0) with z recomputed from x
// dy2 (optional, same rows): the gradient is dy + dy2, summed in fp32 where the rows are in registers anyway -- the output of a
// stage's last BatchNorm feeds the next stage AND (split by frame) the cross-attention block (SiamWCA_MAE.py:262-291); autograd
// used to concatenate the two frame halves of the second gradient and add the result to the first: two elementwise passes and a
// bf16 rounding per stage.
template <class T, int VEC, bool TWO = false>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ dy, const T* __restrict__ dy2,
                                                           const T* __restrict__ x,
                                                           int64_t m, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, int relu,
                                                           float* __restrict__ part) {
  BN_LAYOUT;
  __shared__ float red[4][2][C];
  float mu[8], rs[8], g[8], bt[8], s1[8], s2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int ch = cl * 8 + i;
    mu[i] = mean[ch]; rs[i] = rstd[ch]; g[i] = gamma[ch]; bt[i] = beta[ch];
    s1[i] = 0.f; s2[i] = 0.f;
  }
  for (int64_t r0 = wave * (2 * RPW); r0 < m; r0 += nwaves * (2 * RPW)) {
    float v[2][8], d[2][8];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int64_t r = r0 + u * RPW + sub, rc = r < m ? r : m - 1;
      load8<T>(x + rc * C + cl * 8, v[u]);                        // unconditional (clamped), masked below
      load8<T>(dy + rc * C + cl * 8, d[u]);
      if constexpr (TWO) {
        float e[8];
        load8<T>(dy2 + rc * C + cl * 8, e);
#pragma unroll
        for (int i = 0; i < 8; ++i) d[u][i] += e[i];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) d[u][i] = r < m ? d[u][i] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float xh = (v[u][i] - mu[i]) * rs[i];
        const float dz = (relu && !(xh * g[i] + bt[i] > 0.f)) ? 0.f : d[u][i];
        s1[i] += dz;
        s2[i] += dz * xh;
      }
  }
  bn_block_partial<C, LPR>(s1, s2, red, w, sub, cl, part);
}

__global__ __launch_bounds__(16 * BN_FIN_RL) void bn_bwd_finalize_kernel(const float* __restrict__ part, int nblocks,
                                                                        int c, float* __restrict__ dbeta,
                                                                        float* __restrict__ dgamma) {
  __shared__ double red[BN_FIN_RL][17][2];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int ch = blockIdx.x * 16 + cl;
  double a, b;
  bn_partial_sums(part, nblocks, c, ch, rl, a, b);
  red[rl][cl][0] = a;
  red[rl][cl][1] = b;
  __syncthreads();
  if (rl == 0 && ch < c) {
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int r = 0; r < BN_FIN_RL; ++r) { s1 += red[r][cl][0]; s2 += red[r][cl][1]; }
    dbeta[ch] = (float)s1;
    dgamma[ch] = (float)s2;
  }
}

// The same two sums plus the UNMASKED column sum of dy (the fused decoder head's BatchNorm backward, csrc/deblock.hip, needs it to
// price the inactive cells: tmae_deblock_bn_tail) in one pass over dy and x -- it used to be a separate tmae_column_sums pass
// over dy (three launches of 82 us per step).  part [grid][3][C] (+ unused pivot row).
template <class T, int VEC>
__global__ __launch_bounds__(256) void bn_bwd_reduce3_kernel(const T* __restrict__ dy, const T* __restrict__ x, int64_t m,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            int relu, float* __restrict__ part) {
  BN_LAYOUT;
  __shared__ float red[4][3][C];
  float mu[8], rs[8], g[8], bt[8], s0[8], s1[8], s2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int ch = cl * 8 + i;
    mu[i] = mean[ch]; rs[i] = rstd[ch]; g[i] = gamma[ch]; bt[i] = beta[ch];
    s0[i] = 0.f; s1[i] = 0.f; s2[i] = 0.f;
  }
  for (int64_t r0 = wave * (2 * RPW); r0 < m; r0 += nwaves * (2 * RPW)) {
    float v[2][8], d[2][8];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int64_t r = r0 + u * RPW + sub, rc = r < m ? r : m - 1;
      load8<T>(x + rc * C + cl * 8, v[u]);                        // unconditional (clamped), masked below
      load8<T>(dy + rc * C + cl * 8, d[u]);
#pragma unroll
      for (int i = 0; i < 8; ++i) d[u][i] = r < m ? d[u][i] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float xh = (v[u][i] - mu[i]) * rs[i];
        const float dz = (relu && !(xh * g[i] + bt[i] > 0.f)) ? 0.f : d[u][i];
        s0[i] += d[u][i];
        s1[i] += dz;
        s2[i] += dz * xh;
      }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) { s0[i] = cross_group_sum<LPR>(s0[i]); s1[i] = cross_group_sum<LPR>(s1[i]); s2[i] = cross_group_sum<LPR>(s2[i]); }
  if (sub == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { red[w][0][cl * 8 + i] = s0[i]; red[w][1][cl * 8 + i] = s1[i]; red[w][2][cl * 8 + i] = s2[i]; }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 3 * C; e += 256) {
    const int which = e / C, c = e % C;
    part[(int64_t)blockIdx.x * 3 * C + e] = red[0][which][c] + red[1][which][c] + red[2][which][c] + red[3][which][c];
  }
}

// the three totals from part [nblocks][3][c]: double accumulation in a fixed order, 16 channels x 64 partial rows per workgroup
__global__ __launch_bounds__(16 * BN_FIN_RL) void bn_bwd_finalize3_kernel(const float* __restrict__ part, int nblocks, int c,
                                                                         float* __restrict__ sum_dy, float* __restrict__ dbeta,
                                                                         float* __restrict__ dgamma) {
  __shared__ double red[BN_FIN_RL][17][3];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int ch = blockIdx.x * 16 + cl;
  double a[3] = {0.0, 0.0, 0.0};
  if (ch < c) {
    int k = rl;
    for (; k + 3 * BN_FIN_RL < nblocks; k += 4 * BN_FIN_RL) {
      float u[4][3];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int t = 0; t < 3; ++t) u[q][t] = part[((int64_t)(k + q * BN_FIN_RL) * 3 + t) * c + ch];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int t = 0; t < 3; ++t) a[t] += u[q][t];
    }
    for (; k < nblocks; k += BN_FIN_RL)
#pragma unroll
      for (int t = 0; t < 3; ++t) a[t] += part[((int64_t)k * 3 + t) * c + ch];
  }
#pragma unroll
  for (int t = 0; t < 3; ++t) red[rl][cl][t] = a[t];
  __syncthreads();
  if (rl == 0 && ch < c) {
    double s[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < BN_FIN_RL; ++r)
#pragma unroll
      for (int t = 0; t < 3; ++t) s[t] += red[r][cl][t];
    sum_dy[ch] = (float)s[0];
    dbeta[ch] = (float)s[1];
    dgamma[ch] = (float)s[2];
  }
}

// dx = gamma * rstd * (dz - dbeta/m - xhat * dgamma/m)
template <class T, int VEC, bool TWO = false>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ dy2,
                                                          const T* __restrict__ x, int64_t m,
                                                          const float* __restrict__ mean,
                                                          const float* __restrict__ rstd,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, int relu,
                                                          const float* __restrict__ dbeta,
                                                          const float* __restrict__ dgamma, float invm,
                                                          T* __restrict__ dx) {
  BN_LAYOUT;
  float mu[8], rs[8], g[8], bt[8], a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int ch = cl * 8 + i;
    mu[i] = mean[ch]; rs[i] = rstd[ch]; g[i] = gamma[ch]; bt[i] = beta[ch];
    a[i] = dbeta[ch] * invm; b[i] = dgamma[ch] * invm;
  }
  for (int64_t r0 = wave * (2 * RPW); r0 < m; r0 += nwaves * (2 * RPW)) {
    float v[2][8], d[2][8];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int64_t r = r0 + u * RPW + sub, rc = r < m ? r : m - 1;
      load8<T>(x + rc * C + cl * 8, v[u]);                        // unconditional (clamped): four loads in flight
      load8<T>(dy + rc * C + cl * 8, d[u]);
      if constexpr (TWO) {
        float e[8];
        load8<T>(dy2 + rc * C + cl * 8, e);
#pragma unroll
        for (int i = 0; i < 8; ++i) d[u][i] += e[i];
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int64_t r = r0 + u * RPW + sub;
      if (r < m) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float xh = (v[u][i] - mu[i]) * rs[i];
          const float dz = (relu && !(xh * g[i] + bt[i] > 0.f)) ? 0.f : d[u][i];
          d[u][i] = g[i] * rs[i] * (dz - a[i] - xh * b[i]);
        }
        store8<T>(dx + r * C + cl * 8, d[u]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Backward of BatchNorm(+ReLU) over the rows of a DENSE map whose output is read at m << cells sites only -- the decoder's last
// norm, whose [B, 468, 468, 128] output reaches the loss through the gather at the current frame's voxels (SiamWCA_MAE.py:303-312:
// 21 % of the cells).  The gradient that arrives is the compact [m, c] one; scattering it into a dense, mostly zero dy first
// (tmae_sparse_to_dense) cost a 448 MB write, and both passes below then read those zeros back.  Here:
//   * the two sums run over the m gathered rows only (x read at their cells);
//   * the apply pass walks all cells and takes dz from the compact rows through the cell -> row map (-1: no gradient).
// ------------------------------------------------------------------------------------------------
template <class T, int VEC>
__global__ __launch_bounds__(256) void bn_bwd_reduce_rows_kernel(const T* __restrict__ dyc, const int32_t* __restrict__ ind,
                                                                const T* __restrict__ x, int64_t m, int ny, int nx,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                int relu, float* __restrict__ part) {
  BN_LAYOUT;
  __shared__ float red[4][2][C];
  float mu[8], rs[8], g[8], bt[8], s1[8], s2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int ch = cl * 8 + i;
    mu[i] = mean[ch]; rs[i] = rstd[ch]; g[i] = gamma[ch]; bt[i] = beta[ch];
    s1[i] = 0.f; s2[i] = 0.f;
  }
  for (int64_t r0 = wave * (2 * RPW); r0 < m; r0 += nwaves * (2 * RPW)) {
    float v[2][8], d[2][8];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int64_t r = r0 + u * RPW + sub, rc = r < m ? r : m - 1;
      const int64_t cell = ((int64_t)ind[rc * 3] * ny + ind[rc * 3 + 1]) * nx + ind[rc * 3 + 2];
      load8<T>(x + cell * C + cl * 8, v[u]);                      // unconditional (clamped row), masked below
      load8<T>(dyc + rc * C + cl * 8, d[u]);
#pragma unroll
      for (int i = 0; i < 8; ++i) d[u][i] = r < m ? d[u][i] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float xh = (v[u][i] - mu[i]) * rs[i];
        const float dz = (relu && !(xh * g[i] + bt[i] > 0.f)) ? 0.f : d[u][i];
        s1[i] += dz;
        s2[i] += dz * xh;
      }
  }
  bn_block_partial<C, LPR>(s1, s2, red, w, sub, cl, part);
}

template <class T, int VEC>
__global__ __launch_bounds__(256) void bn_bwd_apply_map_kernel(const T* __restrict__ dyc, const int32_t* __restrict__ rowmap,
                                                              const T* __restrict__ x, int64_t cells,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              int relu, const float* __restrict__ dbeta,
                                                              const float* __restrict__ dgamma, float invm, T* __restrict__ dx) {
  BN_LAYOUT;
  float mu[8], rs[8], g[8], bt[8], a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int ch = cl * 8 + i;
    mu[i] = mean[ch]; rs[i] = rstd[ch]; g[i] = gamma[ch]; bt[i] = beta[ch];
    a[i] = dbeta[ch] * invm; b[i] = dgamma[ch] * invm;
  }
  for (int64_t r0 = wave * (2 * RPW); r0 < cells; r0 += nwaves * (2 * RPW)) {
    float v[2][8], d[2][8];
    int src[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int64_t r = r0 + u * RPW + sub, rc = r < cells ? r : cells - 1;
      src[u] = rowmap[rc];
      load8<T>(x + rc * C + cl * 8, v[u]);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      load8<T>(dyc + (int64_t)(src[u] < 0 ? 0 : src[u]) * C + cl * 8, d[u]);      // unconditional (row 0 for "no gradient"), masked below
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int64_t r = r0 + u * RPW + sub;
      if (r < cells) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float xh = (v[u][i] - mu[i]) * rs[i];
          const float dz = (src[u] < 0 || (relu && !(xh * g[i] + bt[i] > 0.f))) ? 0.f : d[u][i];
          d[u][i] = g[i] * rs[i] * (dz - a[i] - xh * b[i]);
        }
        store8<T>(dx + r * C + cl * 8, d[u]);
      }
    }
  }
}

static int bn_grid(int64_t m) {
  int64_t g = (m + 255) / 256;          // >= 64 rows per wave
  if (g > 1024) g = 1024;
  if (g < 1) g = 1;
  return (int)g;
}

size_t tmae_bn_workspace(int64_t m, int c) { return ((size_t)bn_grid(m) * 2 * c + c) * 4 + 256; }

#define BN_DISPATCH(T, KERNEL, ...)                                                              \
  do {                                                                                            \
    if (c == 64) hipLaunchKernelGGL((KERNEL<T, 1>), grid, block, 0, stream, __VA_ARGS__);         \
    else if (c == 128) hipLaunchKernelGGL((KERNEL<T, 2>), grid, block, 0, stream, __VA_ARGS__);   \
    else hipLaunchKernelGGL((KERNEL<T, 4>), grid, block, 0, stream, __VA_ARGS__);                 \
  } while (0)
// the backward kernels: with a second gradient operand (dy2 != nullptr) the TWO instantiation
#define BN_DISPATCH2(T, KERNEL, two, ...)                                                                  \
  do {                                                                                                      \
    if (two) {                                                                                              \
      if (c == 64) hipLaunchKernelGGL((KERNEL<T, 1, true>), grid, block, 0, stream, __VA_ARGS__);           \
      else if (c == 128) hipLaunchKernelGGL((KERNEL<T, 2, true>), grid, block, 0, stream, __VA_ARGS__);     \
      else hipLaunchKernelGGL((KERNEL<T, 4, true>), grid, block, 0, stream, __VA_ARGS__);                   \
    } else {                                                                                                \
      if (c == 64) hipLaunchKernelGGL((KERNEL<T, 1, false>), grid, block, 0, stream, __VA_ARGS__);          \
      else if (c == 128) hipLaunchKernelGGL((KERNEL<T, 2, false>), grid, block, 0, stream, __VA_ARGS__);    \
      else hipLaunchKernelGGL((KERNEL<T, 4, false>), grid, block, 0, stream, __VA_ARGS__);                  \
    }                                                                                                       \
  } while (0)

static int bn_relu_fwd(const void* x_, int dtype, int64_t m, int c, const float* gamma, const float* beta, float eps,
                       int relu, const void* post_, void* y_, float* mean, float* var, float* rstd, void* wsp, size_t ws_bytes,
                       void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (m <= 0 || (c != 64 && c != 128 && c != 256) || !x_ || !gamma || !beta || !y_ || !mean || !var || !rstd)
    return TMAE_EARG;
  if (dtype != TMAE_F32 && dtype != TMAE_BF16) return TMAE_EDTYPE;
  const int nb = bn_grid(m);
  WsCarver ws(wsp, ws_bytes);
  float* part = ws.take<float>((size_t)nb * 2 * c + c);
  if (!ws.ok) return TMAE_EWS;
  dim3 grid(nb), block(256);
  if (dtype == TMAE_F32) {
    const float* x = (const float*)x_;
    float* y = (float*)y_;
    BN_DISPATCH(float, bn_stats_kernel, x, m, 1, part);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(tmae_cdiv(c, 16)), dim3(16 * BN_FIN_RL), 0, stream, part, nb, c, (double)m, eps, mean, var, rstd);
    BN_DISPATCH(float, bn_apply_kernel, x, m, mean, rstd, gamma, beta, relu, y, (const float*)post_);
  } else {
    const __hip_bfloat16* x = (const __hip_bfloat16*)x_;
    __hip_bfloat16* y = (__hip_bfloat16*)y_;
    BN_DISPATCH(__hip_bfloat16, bn_stats_kernel, x, m, 1, part);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(tmae_cdiv(c, 16)), dim3(16 * BN_FIN_RL), 0, stream, part, nb, c, (double)m, eps, mean, var, rstd);
    BN_DISPATCH(__hip_bfloat16, bn_apply_kernel, x, m, mean, rstd, gamma, beta, relu, y, (const __hip_bfloat16*)post_);
  }
  return tmae_launch_status();
}

int tmae_bn_relu_fwd(const void* x, int dtype, int64_t m, int c, const float* gamma, const float* beta, float eps, int relu,
                     void* y, float* mean, float* var, float* rstd, void* ws, size_t ws_bytes, void* stream) {
  (void)hipGetLastError();
  return bn_relu_fwd(x, dtype, m, c, gamma, beta, eps, relu, nullptr, y, mean, var, rstd, ws, ws_bytes, stream);
}

int tmae_bn_relu_add_fwd(const void* x, int dtype, int64_t m, int c, const float* gamma, const float* beta, float eps, int relu,
                         const void* post, void* y, float* mean, float* var, float* rstd, void* ws, size_t ws_bytes,
                         void* stream) {
  (void)hipGetLastError();
  if (!post) return TMAE_EARG;
  return bn_relu_fwd(x, dtype, m, c, gamma, beta, eps, relu, post, y, mean, var, rstd, ws, ws_bytes, stream);
}

static int bn_relu_bwd(const void* dy_, const void* dy2_, const void* x_, int dtype, int64_t m, int c, const float* mean,
                       const float* rstd, const float* gamma, const float* beta, int relu, void* dx_, float* dgamma, float* dbeta,
                       void* wsp, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (m <= 0 || (c != 64 && c != 128 && c != 256) || !dy_ || !x_ || !mean || !rstd || !gamma || !beta || !dx_ ||
      !dgamma || !dbeta)
    return TMAE_EARG;
  if (dtype != TMAE_F32 && dtype != TMAE_BF16) return TMAE_EDTYPE;
  if (dy2_ && ((uintptr_t)dy2_ & 15)) return TMAE_EARG;
  const int nb = bn_grid(m);
  WsCarver ws(wsp, ws_bytes);
  float* part = ws.take<float>((size_t)nb * 2 * c + c);
  if (!ws.ok) return TMAE_EWS;
  dim3 grid(nb), block(256);
  const bool two = dy2_ != nullptr;
  if (dtype == TMAE_F32) {
    const float *dy = (const float*)dy_, *dy2 = (const float*)dy2_, *x = (const float*)x_;
    float* dx = (float*)dx_;
    BN_DISPATCH2(float, bn_bwd_reduce_kernel, two, dy, dy2, x, m, mean, rstd, gamma, beta, relu, part);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(tmae_cdiv(c, 16)), dim3(16 * BN_FIN_RL), 0, stream, part, nb, c, dbeta, dgamma);
    BN_DISPATCH2(float, bn_bwd_apply_kernel, two, dy, dy2, x, m, mean, rstd, gamma, beta, relu, dbeta, dgamma, 1.0f / (float)m, dx);
  } else {
    const __hip_bfloat16 *dy = (const __hip_bfloat16*)dy_, *dy2 = (const __hip_bfloat16*)dy2_, *x = (const __hip_bfloat16*)x_;
    __hip_bfloat16* dx = (__hip_bfloat16*)dx_;
    BN_DISPATCH2(__hip_bfloat16, bn_bwd_reduce_kernel, two, dy, dy2, x, m, mean, rstd, gamma, beta, relu, part);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(tmae_cdiv(c, 16)), dim3(16 * BN_FIN_RL), 0, stream, part, nb, c, dbeta, dgamma);
    BN_DISPATCH2(__hip_bfloat16, bn_bwd_apply_kernel, two, dy, dy2, x, m, mean, rstd, gamma, beta, relu, dbeta, dgamma, 1.0f / (float)m, dx);
  }
  return tmae_launch_status();
}

int tmae_bn_relu_bwd(const void* dy, const void* x, int dtype, int64_t m, int c, const float* mean, const float* rstd,
                     const float* gamma, const float* beta, int relu, void* dx, float* dgamma, float* dbeta, void* ws,
                     size_t ws_bytes, void* stream) {
  (void)hipGetLastError();
  return bn_relu_bwd(dy, nullptr, x, dtype, m, c, mean, rstd, gamma, beta, relu, dx, dgamma, dbeta, ws, ws_bytes, stream);
}

// the gradient arrives in two pieces (dy + dy2, both [m, c] of x's dtype): see bn_bwd_reduce_kernel
int tmae_bn_relu_bwd2(const void* dy, const void* dy2, const void* x, int dtype, int64_t m, int c, const float* mean,
                      const float* rstd, const float* gamma, const float* beta, int relu, void* dx, float* dgamma, float* dbeta,
                      void* ws, size_t ws_bytes, void* stream) {
  (void)hipGetLastError();
  if (!dy2) return TMAE_EARG;
  return bn_relu_bwd(dy, dy2, x, dtype, m, c, mean, rstd, gamma, beta, relu, dx, dgamma, dbeta, ws, ws_bytes, stream);
}

// ------------------------------------------------------------------------------------------------
// split entry points (used by the fused sparse decoder head, deblock.hip): statistics with an explicit element
// count (BatchNorm2d over a dense grid whose inactive cells are exact zeros: sum over the active rows / all cells),
// partial backward sums, and the backward apply with given totals.
// ------------------------------------------------------------------------------------------------
static bool bn_args_ok(int64_t m, int c, int dtype) {
  return m > 0 && (c == 64 || c == 128 || c == 256) && (dtype == TMAE_F32 || dtype == TMAE_BF16);
}

int tmae_bn_stats(const void* x_, int dtype, int64_t m, int c, double count, float eps, float* mean, float* var,
                  float* rstd, void* wsp, size_t ws_bytes, void* stream_) {
  (void)hipGetLastError();
  hipStream_t stream = (hipStream_t)stream_;
  if (!bn_args_ok(m, c, dtype) || !x_ || !mean || !var || !rstd || count < 1.0) return TMAE_EARG;
  const int nb = bn_grid(m);
  WsCarver ws(wsp, ws_bytes);
  float* part = ws.take<float>((size_t)nb * 2 * c + c);
  if (!ws.ok) return TMAE_EWS;
  dim3 grid(nb), block(256);
  // the pivot only helps when every counted element is a row of x (count == m); with implicit zero cells the raw
  // moments are the right ones (the zeros ARE the bulk of the data)
  const int piv = count == (double)m ? 1 : 0;
  if (dtype == TMAE_F32) { const float* x = (const float*)x_; BN_DISPATCH(float, bn_stats_kernel, x, m, piv, part); }
  else { const __hip_bfloat16* x = (const __hip_bfloat16*)x_; BN_DISPATCH(__hip_bfloat16, bn_stats_kernel, x, m, piv, part); }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(tmae_cdiv(c, 16)), dim3(16 * BN_FIN_RL), 0, stream, part, nb, c, count, eps, mean, var, rstd);
  return tmae_launch_status();
}

// y = relu?((x - mean) * rstd * gamma + beta) with CALLER-provided statistics: the apply half of tmae_bn_relu_fwd, for
// statistics that were merged over ranks first (SyncBatchNorm, tools/train.py --sync_bn)
int tmae_bn_apply(const void* x_, int dtype, int64_t m, int c, const float* mean, const float* rstd, const float* gamma,
                  const float* beta, int relu, void* y_, void* stream_) {
  (void)hipGetLastError();
  hipStream_t stream = (hipStream_t)stream_;
  if (!bn_args_ok(m, c, dtype) || !x_ || !mean || !rstd || !gamma || !beta || !y_) return TMAE_EARG;
  dim3 grid(bn_grid(m)), block(256);
  if (dtype == TMAE_F32) {
    const float* x = (const float*)x_;
    float* y = (float*)y_;
    BN_DISPATCH(float, bn_apply_kernel, x, m, mean, rstd, gamma, beta, relu, y);
  } else {
    const __hip_bfloat16* x = (const __hip_bfloat16*)x_;
    __hip_bfloat16* y = (__hip_bfloat16*)y_;
    BN_DISPATCH(__hip_bfloat16, bn_apply_kernel, x, m, mean, rstd, gamma, beta, relu, y);
  }
  return tmae_launch_status();
}

int tmae_bn_bwd_sums(const void* dy_, const void* x_, int dtype, int64_t m, int c, const float* mean, const float* rstd,
                     const float* gamma, const float* beta, int relu, float* sum_dz, float* sum_dz_xhat, void* wsp,
                     size_t ws_bytes, void* stream_) {
  (void)hipGetLastError();
  hipStream_t stream = (hipStream_t)stream_;
  if (!bn_args_ok(m, c, dtype) || !dy_ || !x_ || !mean || !rstd || !gamma || !beta || !sum_dz || !sum_dz_xhat)
    return TMAE_EARG;
  const int nb = bn_grid(m);
  WsCarver ws(wsp, ws_bytes);
  float* part = ws.take<float>((size_t)nb * 2 * c + c);
  if (!ws.ok) return TMAE_EWS;
  dim3 grid(nb), block(256);
  if (dtype == TMAE_F32) {
    const float *dy = (const float*)dy_, *x = (const float*)x_;
    BN_DISPATCH(float, bn_bwd_reduce_kernel, dy, (const float*)nullptr, x, m, mean, rstd, gamma, beta, relu, part);
  } else {
    const __hip_bfloat16 *dy = (const __hip_bfloat16*)dy_, *x = (const __hip_bfloat16*)x_;
    BN_DISPATCH(__hip_bfloat16, bn_bwd_reduce_kernel, dy, (const __hip_bfloat16*)nullptr, x, m, mean, rstd, gamma, beta, relu, part);
  }
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(tmae_cdiv(c, 16)), dim3(16 * BN_FIN_RL), 0, stream, part, nb, c, sum_dz, sum_dz_xhat);
  return tmae_launch_status();
}

// tmae_bn_bwd_sums plus sum_dy [c] = the unmasked column sums of dy, in the same pass (workspace: 3/2 of tmae_bn_workspace: the
// caller passes tmae_bn_workspace(m, c) * 2)
int tmae_bn_bwd_sums3(const void* dy_, const void* x_, int dtype, int64_t m, int c, const float* mean, const float* rstd,
                      const float* gamma, const float* beta, int relu, float* sum_dy, float* sum_dz, float* sum_dz_xhat, void* wsp,
                      size_t ws_bytes, void* stream_) {
  (void)hipGetLastError();
  hipStream_t stream = (hipStream_t)stream_;
  if (!bn_args_ok(m, c, dtype) || !dy_ || !x_ || !mean || !rstd || !gamma || !beta || !sum_dy || !sum_dz || !sum_dz_xhat)
    return TMAE_EARG;
  const int nb = bn_grid(m);
  WsCarver ws(wsp, ws_bytes);
  float* part = ws.take<float>((size_t)nb * 3 * c + c);
  if (!ws.ok) return TMAE_EWS;
  dim3 grid(nb), block(256);
  if (dtype == TMAE_F32) {
    const float *dy = (const float*)dy_, *x = (const float*)x_;
    BN_DISPATCH(float, bn_bwd_reduce3_kernel, dy, x, m, mean, rstd, gamma, beta, relu, part);
  } else {
    const __hip_bfloat16 *dy = (const __hip_bfloat16*)dy_, *x = (const __hip_bfloat16*)x_;
    BN_DISPATCH(__hip_bfloat16, bn_bwd_reduce3_kernel, dy, x, m, mean, rstd, gamma, beta, relu, part);
  }
  hipLaunchKernelGGL(bn_bwd_finalize3_kernel, dim3(tmae_cdiv(c, 16)), dim3(16 * BN_FIN_RL), 0, stream, part, nb, c, sum_dy, sum_dz,
                     sum_dz_xhat);
  return tmae_launch_status();
}

int tmae_bn_bwd_apply(const void* dy_, const void* x_, int dtype, int64_t m, int c, const float* mean,
                      const float* rstd, const float* gamma, const float* beta, int relu, const float* dbeta,
                      const float* dgamma, double count, void* dx_, void* stream_) {
  (void)hipGetLastError();
  hipStream_t stream = (hipStream_t)stream_;
  if (!bn_args_ok(m, c, dtype) || !dy_ || !x_ || !mean || !rstd || !gamma || !beta || !dbeta || !dgamma || !dx_ ||
      count < 1.0)
    return TMAE_EARG;
  dim3 grid(bn_grid(m)), block(256);
  const float invm = (float)(1.0 / count);
  if (dtype == TMAE_F32) {
    const float *dy = (const float*)dy_, *x = (const float*)x_;
    float* dx = (float*)dx_;
    BN_DISPATCH(float, bn_bwd_apply_kernel, dy, (const float*)nullptr, x, m, mean, rstd, gamma, beta, relu, dbeta, dgamma, invm, dx);
  } else {
    const __hip_bfloat16 *dy = (const __hip_bfloat16*)dy_, *x = (const __hip_bfloat16*)x_;
    __hip_bfloat16* dx = (__hip_bfloat16*)dx_;
    BN_DISPATCH(__hip_bfloat16, bn_bwd_apply_kernel, dy, (const __hip_bfloat16*)nullptr, x, m, mean, rstd, gamma, beta, relu, dbeta, dgamma, invm, dx);
  }
  return tmae_launch_status();
}

// dx [cells, c], dgamma, dbeta of y = relu?(BatchNorm(x)) over the rows of a dense map x [batch, ny, nx, c] whose gradient is given at
// m sites only: dyc [m, c] (row j = site indices[j] = (b, y, x), int32 [m, 3]), rowmap [batch * ny * nx] int32 = the site's row in dyc
// or -1 (tmae_index_grid of `indices`); every other cell has zero gradient.  Statistics over all `cells` rows, as the forward's.
int tmae_bn_relu_bwd_gathered(const void* dyc_, const int32_t* indices, const int32_t* rowmap, int64_t m, const void* x_, int dtype,
                              int batch, int ny, int nx, int c, const float* mean, const float* rstd, const float* gamma,
                              const float* beta, int relu, void* dx_, float* dgamma, float* dbeta, void* wsp, size_t ws_bytes,
                              void* stream_) {
  (void)hipGetLastError();
  hipStream_t stream = (hipStream_t)stream_;
  const int64_t cells = (int64_t)batch * ny * nx;
  if (!bn_args_ok(cells, c, dtype) || batch <= 0 || ny <= 0 || nx <= 0 || m <= 0 || !dyc_ || !indices || !rowmap || !x_ || !mean ||
      !rstd || !gamma || !beta || !dx_ || !dgamma || !dbeta || ((uintptr_t)dyc_ & 15) || ((uintptr_t)x_ & 15) || ((uintptr_t)dx_ & 15))
    return TMAE_EARG;
  const int nb = bn_grid(m);
  WsCarver ws(wsp, ws_bytes);
  float* part = ws.take<float>((size_t)nb * 2 * c + c);
  if (!ws.ok) return TMAE_EWS;
  const float invm = (float)(1.0 / (double)cells);
  if (dtype == TMAE_F32) {
    const float *dyc = (const float*)dyc_, *x = (const float*)x_;
    float* dx = (float*)dx_;
    { dim3 grid(nb), block(256); BN_DISPATCH(float, bn_bwd_reduce_rows_kernel, dyc, indices, x, m, ny, nx, mean, rstd, gamma, beta, relu, part); }
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(tmae_cdiv(c, 16)), dim3(16 * BN_FIN_RL), 0, stream, part, nb, c, dbeta, dgamma);
    { dim3 grid(bn_grid(cells)), block(256); BN_DISPATCH(float, bn_bwd_apply_map_kernel, dyc, rowmap, x, cells, mean, rstd, gamma, beta, relu, dbeta, dgamma, invm, dx); }
  } else {
    const __hip_bfloat16 *dyc = (const __hip_bfloat16*)dyc_, *x = (const __hip_bfloat16*)x_;
    __hip_bfloat16* dx = (__hip_bfloat16*)dx_;
    { dim3 grid(nb), block(256); BN_DISPATCH(__hip_bfloat16, bn_bwd_reduce_rows_kernel, dyc, indices, x, m, ny, nx, mean, rstd, gamma, beta, relu, part); }
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(tmae_cdiv(c, 16)), dim3(16 * BN_FIN_RL), 0, stream, part, nb, c, dbeta, dgamma);
    { dim3 grid(bn_grid(cells)), block(256); BN_DISPATCH(__hip_bfloat16, bn_bwd_apply_map_kernel, dyc, rowmap, x, cells, mean, rstd, gamma, beta, relu, dbeta, dgamma, invm, dx); }
  }
  return tmae_launch_status();
}

// ------------------------------------------------------------------------------------------------
// BatchNorm backward of one deconvolution of the fused decoder head (csrc/deblock.hip) with its gradient rows read IN PLACE from
// the dense concat gradient: row R = (active voxel j, sub-cell q of its s x s block) of v [m s^2, c] corresponds to the c channels
// at offset coff of cell ((b ys + y) s + q / s, x s + q % s) of dcat [B, ys s, xs s, ldc].  Rounds 1-5 gathered those rows into a
// tensor g first (tmae_deblock_gather: 353 MB written for the stride-4 source, then read by two passes).
// ------------------------------------------------------------------------------------------------
struct DbRows {
  const int32_t* ind;       // [m, 3] (b, y, x) of the active voxels
  int ys, xs, s, ls2;       // source grid, stride, log2(s * s)
  int64_t ldc;              // row pitch of dcat in elements
  __device__ __forceinline__ int64_t off(int64_t row) const {        // element offset of row's first channel (without coff)
    const int64_t j = row >> ls2;
    const int q = (int)(row & ((1 << ls2) - 1));
    const int b = ind[j * 3], y = ind[j * 3 + 1], x = ind[j * 3 + 2];
    const int64_t cell = ((int64_t)b * ys * s + (int64_t)y * s + q / s) * ((int64_t)xs * s) + (int64_t)x * s + q % s;
    return cell * ldc;
  }
};

template <class T, int VEC>
__global__ __launch_bounds__(256) void db_bwd_reduce3_kernel(const T* __restrict__ dcat, DbRows map, const T* __restrict__ x, int64_t m,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ part) {
  BN_LAYOUT;
  __shared__ float red[4][3][C];
  float mu[8], rs[8], g[8], bt[8], s0[8], s1[8], s2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int ch = cl * 8 + i;
    mu[i] = mean[ch]; rs[i] = rstd[ch]; g[i] = gamma[ch]; bt[i] = beta[ch];
    s0[i] = 0.f; s1[i] = 0.f; s2[i] = 0.f;
  }
  for (int64_t r0 = wave * (2 * RPW); r0 < m; r0 += nwaves * (2 * RPW)) {
    float v[2][8], d[2][8];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int64_t r = r0 + u * RPW + sub, rc = r < m ? r : m - 1;
      load8<T>(x + rc * C + cl * 8, v[u]);                        // unconditional (clamped), masked below
      load8<T>(dcat + map.off(rc) + cl * 8, d[u]);
#pragma unroll
      for (int i = 0; i < 8; ++i) d[u][i] = r < m ? d[u][i] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float xh = (v[u][i] - mu[i]) * rs[i];
        const float dz = !(xh * g[i] + bt[i] > 0.f) ? 0.f : d[u][i];
        s0[i] += d[u][i];
        s1[i] += dz;
        s2[i] += dz * xh;
      }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) { s0[i] = cross_group_sum<LPR>(s0[i]); s1[i] = cross_group_sum<LPR>(s1[i]); s2[i] = cross_group_sum<LPR>(s2[i]); }
  if (sub == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { red[w][0][cl * 8 + i] = s0[i]; red[w][1][cl * 8 + i] = s1[i]; red[w][2][cl * 8 + i] = s2[i]; }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 3 * C; e += 256) {
    const int which = e / C, c = e % C;
    part[(int64_t)blockIdx.x * 3 * C + e] = red[0][which][c] + red[1][which][c] + red[2][which][c] + red[3][which][c];
  }
}

template <class T, int VEC>
__global__ __launch_bounds__(256) void db_bwd_apply_kernel(const T* __restrict__ dcat, DbRows map, const T* __restrict__ x, int64_t m,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ dbeta, const float* __restrict__ dgamma,
                                                          float invm, T* __restrict__ dx) {
  BN_LAYOUT;
  float mu[8], rs[8], g[8], bt[8], a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int ch = cl * 8 + i;
    mu[i] = mean[ch]; rs[i] = rstd[ch]; g[i] = gamma[ch]; bt[i] = beta[ch];
    a[i] = dbeta[ch] * invm; b[i] = dgamma[ch] * invm;
  }
  for (int64_t r0 = wave * (2 * RPW); r0 < m; r0 += nwaves * (2 * RPW)) {
    float v[2][8], d[2][8];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int64_t r = r0 + u * RPW + sub, rc = r < m ? r : m - 1;
      load8<T>(x + rc * C + cl * 8, v[u]);                        // unconditional (clamped): four loads in flight
      load8<T>(dcat + map.off(rc) + cl * 8, d[u]);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int64_t r = r0 + u * RPW + sub;
      if (r < m) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float xh = (v[u][i] - mu[i]) * rs[i];
          const float dz = !(xh * g[i] + bt[i] > 0.f) ? 0.f : d[u][i];
          d[u][i] = g[i] * rs[i] * (dz - a[i] - xh * b[i]);
        }
        store8<T>(dx + r * C + cl * 8, d[u]);
      }
    }
  }
}

extern "C" int tmae_deblock_bn_tail(const float* mean, const float* rstd, const float* gamma, const float* beta, const float* s_all,
                                    const float* s_act, const float* sum_dz, const float* sum_dzx, int c, float* dbeta, float* dgamma,
                                    void* stream_);

size_t tmae_deblock_bn_bwd_workspace(int64_t m, int s, int cout) {
  return ((size_t)bn_grid(m * s * s) * 3 * cout + 4 * (size_t)cout) * 4 + 512;
}

// dv [m s^2, cout], dgamma, dbeta of relu(BatchNorm2d(deconv)) for one source of the fused decoder head (tmae_deblock_scatter's
// backward): v [m s^2, cout] = the deconvolution's rows at the m active voxels (indices [m, 3] int32), dcat [batch, ys s, xs s, ldc]
// the gradient of the concat buffer (this source's channels at coff), s_all [cout] its column sums over ALL cells (tmae_column_sums,
// or the producing conv's epilogue), count = batch * ys s * xs s.  One pass for the three sums, the inactive cells' share
// (tmae_deblock_bn_tail), one pass for dv; s in {1, 2, 4}.
int tmae_deblock_bn_bwd(const void* dcat_, int dtype, int64_t ldc, int coff, const int32_t* indices, int64_t m, int ys, int xs, int s,
                        int cout, const void* v_, const float* mean, const float* rstd, const float* gamma, const float* beta,
                        const float* s_all, double count, void* dv_, float* dgamma, float* dbeta, void* wsp, size_t ws_bytes,
                        void* stream_) {
  (void)hipGetLastError();
  hipStream_t stream = (hipStream_t)stream_;
  const int c = cout;
  const int64_t rows = m * s * s;
  if (!bn_args_ok(rows, c, dtype) || (s != 1 && s != 2 && s != 4) || ys <= 0 || xs <= 0 || ldc < coff + cout || (ldc % 8) || (coff % 8) ||
      !dcat_ || !indices || !v_ || !mean || !rstd || !gamma || !beta || !s_all || !dv_ || !dgamma || !dbeta || count < 1.0 ||
      ((uintptr_t)dcat_ & 15) || ((uintptr_t)v_ & 15) || ((uintptr_t)dv_ & 15))
    return TMAE_EARG;
  const int nb = bn_grid(rows);
  WsCarver ws(wsp, ws_bytes);
  float* part = ws.take<float>((size_t)nb * 3 * c);
  float* sums = ws.take<float>((size_t)3 * c);          // s_act | sum_dz | sum_dz_xhat
  if (!ws.ok) return TMAE_EWS;
  DbRows map{indices, ys, xs, s, s == 1 ? 0 : (s == 2 ? 2 : 4), ldc};
  const float invm = (float)(1.0 / count);
  dim3 grid(nb), block(256);
  if (dtype == TMAE_F32) {
    const float *dcat = (const float*)dcat_ + coff, *x = (const float*)v_;
    BN_DISPATCH(float, db_bwd_reduce3_kernel, dcat, map, x, rows, mean, rstd, gamma, beta, part);
  } else {
    const __hip_bfloat16 *dcat = (const __hip_bfloat16*)dcat_ + coff, *x = (const __hip_bfloat16*)v_;
    BN_DISPATCH(__hip_bfloat16, db_bwd_reduce3_kernel, dcat, map, x, rows, mean, rstd, gamma, beta, part);
  }
  hipLaunchKernelGGL(bn_bwd_finalize3_kernel, dim3(tmae_cdiv(c, 16)), dim3(16 * BN_FIN_RL), 0, stream, part, nb, c, sums, sums + c,
                     sums + 2 * c);
  if (int e = tmae_deblock_bn_tail(mean, rstd, gamma, beta, s_all, sums, sums + c, sums + 2 * c, c, dbeta, dgamma, stream_)) return e;
  if (dtype == TMAE_F32) {
    const float *dcat = (const float*)dcat_ + coff, *x = (const float*)v_;
    BN_DISPATCH(float, db_bwd_apply_kernel, dcat, map, x, rows, mean, rstd, gamma, beta, dbeta, dgamma, invm, (float*)dv_);
  } else {
    const __hip_bfloat16 *dcat = (const __hip_bfloat16*)dcat_ + coff, *x = (const __hip_bfloat16*)v_;
    BN_DISPATCH(__hip_bfloat16, db_bwd_apply_kernel, dcat, map, x, rows, mean, rstd, gamma, beta, dbeta, dgamma, invm, (__hip_bfloat16*)dv_);
  }
  return tmae_launch_status();
}
