// A11, decoder head: ConvTranspose2d(k = s, stride s) + BatchNorm2d + ReLU of a SPARSE BEV tensor, written straight
// into a channel slice of the dense channels-last concat buffer (SiamWCA_MAE.dense_conv, SiamWCA_MAE.py:231-253 with
// the modules of :79-98).
//
// A k = s transposed conv maps every active input cell to its own s x s block of output cells and every other
// output cell to exact zero, so the dense result is: rows v = feat @ W [m, s*s*cout] on the active cells, zero
// elsewhere.  BatchNorm2d's batch statistics over ALL B*Y*X cells follow from the sums over the active rows (count =
// all cells), every inactive cell becomes the per-channel constant relu(beta - mean*rstd*gamma), and the backward
// needs the dense gradient only as (a) its rows at the active cells and (b) its per-channel column sums.  The dense
// tensors are therefore touched once (one write forward, one read backward) instead of ~15 times.
#include "common.h"

// out[cell, coff + c] = relu(v[row(cell), sub(cell), c] * sc[c] + sh[c])   or   relu(sh[c]) on inactive cells
// A thread owns ONE 16-byte channel chunk for the whole kernel (its scale / shift pairs are formed once, not four
// parameter loads per channel and cell) and walks over DS_CPT cells; cell coordinates from 32-bit divisions.  The first
// version (one thread per (cell, chunk), 64-bit index arithmetic, parameters re-read per element) wrote its 450 MB at
// 2.2 TB/s.
#define DS_CPT 8
template <class T>
__global__ __launch_bounds__(256) void deblock_scatter_kernel(const T* __restrict__ v, const int32_t* __restrict__ grid,
                                                             int batch, int ys, int xs, int s, int cout,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ rstd,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, T* __restrict__ out,
                                                             int ldc, int coff) {
  constexpr int VEC = 16 / sizeof(T);
  const unsigned chunks = (unsigned)(cout / VEC);               // divides 256 (checked by the launcher)
  const unsigned cpp = 256u / chunks;                           // cells per pass of the block
  const unsigned Y = (unsigned)(ys * s), X = (unsigned)(xs * s);
  const unsigned ncell = (unsigned)batch * Y * X;               // < 2^31 (launcher)
  const unsigned ch = threadIdx.x % chunks, sub = threadIdx.x / chunks;
  const int c0 = (int)ch * VEC;
  float sc[VEC], sh[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    sc[k] = rstd[c0 + k] * gamma[c0 + k];
    sh[k] = beta[c0 + k] - mean[c0 + k] * sc[k];
  }
  const unsigned cell0 = blockIdx.x * (cpp * DS_CPT) + sub;
#pragma unroll
  for (int it = 0; it < DS_CPT; ++it) {
    const unsigned cell = cell0 + (unsigned)it * cpp;
    if (cell >= ncell) break;
    const unsigned x = cell % X, yb = cell / X, y = yb % Y, b = yb / Y;
    const int idx = grid[(b * (unsigned)ys + y / (unsigned)s) * (unsigned)xs + x / (unsigned)s];
    T tmp[VEC];
    if (idx >= 0)
      *reinterpret_cast<uint4*>(tmp) = *reinterpret_cast<const uint4*>(
          v + ((int64_t)idx * s * s + (y % (unsigned)s) * s + (x % (unsigned)s)) * cout + c0);
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const float z = (idx >= 0 ? ld_f<T>(&tmp[k]) * sc[k] : 0.f) + sh[k];
      st_f<T>(&tmp[k], fmaxf(z, 0.f));
    }
    *reinterpret_cast<uint4*>(out + (int64_t)cell * ldc + coff + c0) = *reinterpret_cast<uint4*>(tmp);
  }
}

// The same for ALL sources of the concat buffer in one launch (round 6): a workgroup's threads cover whole ldc-wide rows -- thread t
// owns chunk t % (ldc / VEC) of the row, i.e. one 16-byte chunk of ONE source for the whole kernel -- so every cell's row leaves as
// one contiguous 768-byte store burst.  One launch per source wrote its 256-byte slice of each row: three passes of partial lines
// over the 1.35 GB buffer (3 x 126 us at 3.5 TB/s).
#define DS_MAXSRC 4
struct DbSrc {
  const void* v;
  const int32_t* grid;
  const float *mean, *rstd, *gamma, *beta;
  int ys, xs, s, cout, coff;
};
struct DbSrcs { DbSrc src[DS_MAXSRC]; int n; };

template <class T>
__global__ __launch_bounds__(256) void deblock_scatter_multi_kernel(DbSrcs P, int batch, int Yi, int Xi, T* __restrict__ out, int ldc) {
  constexpr int VEC = 16 / sizeof(T);
  const unsigned chunks = (unsigned)(ldc / VEC);                // chunks per row (<= 256: launcher)
  const unsigned cpp = 256u / chunks;                           // cells per pass of the block
  const unsigned Y = (unsigned)Yi, X = (unsigned)Xi;
  const unsigned ncell = (unsigned)batch * Y * X;
  const unsigned ch = threadIdx.x % chunks, sub = threadIdx.x / chunks;
  if (sub >= cpp) return;                                       // 256 % chunks spare threads
  const int c0g = (int)ch * VEC;                                // channel of the chunk in the concat row
  int si = 0;
#pragma unroll
  for (int k = 1; k < DS_MAXSRC; ++k)
    if (k < P.n && c0g >= P.src[k].coff) si = k;
  const DbSrc S = P.src[si];
  const int c0 = c0g - S.coff;
  const T* __restrict__ v = (const T*)S.v;
  float sc[VEC], sh[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    sc[k] = S.rstd[c0 + k] * S.gamma[c0 + k];
    sh[k] = S.beta[c0 + k] - S.mean[c0 + k] * sc[k];
  }
  // the stride differs from lane group to lane group here (one source per chunk range): shifts and masks, not per-lane integer
  // divisions (the launcher admits powers of two only) -- the first version divided and ran 471 us against 3 x 126 us
  const unsigned ls = (unsigned)S.s, ys = (unsigned)S.ys, xs = (unsigned)S.xs, sm = (1u << ls) - 1u;   // S.s holds log2(stride)
  const unsigned cell0 = blockIdx.x * (cpp * DS_CPT) + sub;
#pragma unroll
  for (int it = 0; it < DS_CPT; ++it) {
    const unsigned cell = cell0 + (unsigned)it * cpp;
    if (cell >= ncell) break;
    const unsigned x = cell % X, yb = cell / X, y = yb % Y, b = yb / Y;
    const int idx = S.grid[(b * ys + (y >> ls)) * xs + (x >> ls)];
    T tmp[VEC];
    if (idx >= 0)
      *reinterpret_cast<uint4*>(tmp) = *reinterpret_cast<const uint4*>(
          v + ((((int64_t)idx << (2 * ls)) + (((y & sm) << ls) | (x & sm))) * S.cout + c0));
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const float z = (idx >= 0 ? ld_f<T>(&tmp[k]) * sc[k] : 0.f) + sh[k];
      st_f<T>(&tmp[k], fmaxf(z, 0.f));
    }
    *reinterpret_cast<uint4*>(out + (int64_t)cell * ldc + c0g) = *reinterpret_cast<uint4*>(tmp);
  }
}

// g[row, sub, c] = dcat[cell(row, sub), coff + c]
template <class T>
__global__ __launch_bounds__(256) void deblock_gather_kernel(const T* __restrict__ dcat, int ldc, int coff,
                                                            const int32_t* __restrict__ indices, int64_t m, int ys,
                                                            int xs, int s, int cout, T* __restrict__ g) {
  constexpr int VEC = 16 / sizeof(T);
  const int chunks = cout / VEC;
  int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= m * s * s * chunks) return;
  const int ch = (int)(e % chunks);
  const int64_t rs = e / chunks;
  const int sub = (int)(rs % (s * s));
  const int64_t row = rs / (s * s);
  const int b = indices[row * 3], yi = indices[row * 3 + 1], xi = indices[row * 3 + 2];
  const int64_t cell = ((int64_t)b * ys * s + yi * s + sub / s) * (xs * s) + xi * s + sub % s;
  *reinterpret_cast<uint4*>(g + rs * cout + ch * VEC) =
      *reinterpret_cast<const uint4*>(dcat + cell * ldc + coff + ch * VEC);
}

// per-channel sums over all rows of a [rows, c] matrix (c <= 512, multiple of 64): partials + fixed-order finish
template <class T, int VEC>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, int64_t rows, float* __restrict__ part) {
  constexpr int C = VEC * 64;
  __shared__ float red[4][C];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t wave = (int64_t)blockIdx.x * 4 + w, nwaves = (int64_t)gridDim.x * 4;
  float s[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) s[i] = 0.f;
  for (int64_t r = wave; r < rows; r += nwaves) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) s[i] += ld_f<T>(x + r * C + lane * VEC + i);
  }
#pragma unroll
  for (int i = 0; i < VEC; ++i) red[w][lane * VEC + i] = s[i];
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) part[(int64_t)blockIdx.x * C + c] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
}

__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ part, int nblocks, int c,
                                                           float* __restrict__ out) {
  __shared__ double red[16][17];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int ch = blockIdx.x * 16 + cl;
  double a = 0.0;
  if (ch < c)
    for (int k = rl; k < nblocks; k += 16) a += part[(int64_t)k * c + ch];
  red[rl][cl] = a;
  __syncthreads();
  if (rl == 0 && ch < c) {
    double t = 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += red[r][cl];
    out[ch] = (float)t;
  }
}

static int esz(int dtype) { return dtype == TMAE_F32 ? 4 : (dtype == TMAE_BF16 ? 2 : 0); }

int tmae_deblock_scatter(const void* v, int dtype, const int32_t* grid, int batch, int ys, int xs, int s, int cout,
                         const float* mean, const float* rstd, const float* gamma, const float* beta, void* out,
                         int ldc, int coff, void* stream_) {
  (void)hipGetLastError();
  hipStream_t stream = (hipStream_t)stream_;
  const int es = esz(dtype);
  if (!es) return TMAE_EDTYPE;
  if (!grid || !mean || !rstd || !gamma || !beta || !out || batch <= 0 || ys <= 0 || xs <= 0 || s <= 0 || cout <= 0 ||
      (cout * es) % 16 || (ldc * es) % 16 || (coff * es) % 16)
    return TMAE_EARG;
  const int chunks = cout * es / 16;
  const int64_t ncell = (int64_t)batch * ys * s * xs * s;
  if (chunks > 256 || 256 % chunks || ncell >= ((int64_t)1 << 31)) return TMAE_EARG;   // one chunk per thread, 32-bit cell ids
  const int64_t per_block = (int64_t)(256 / chunks) * DS_CPT;
  if (dtype == TMAE_F32)
    hipLaunchKernelGGL(deblock_scatter_kernel<float>, dim3(tmae_cdiv(ncell, per_block)), dim3(256), 0, stream,
                       (const float*)v, grid, batch, ys, xs, s, cout, mean, rstd, gamma, beta, (float*)out, ldc, coff);
  else
    hipLaunchKernelGGL(deblock_scatter_kernel<__hip_bfloat16>, dim3(tmae_cdiv(ncell, per_block)), dim3(256), 0, stream,
                       (const __hip_bfloat16*)v, grid, batch, ys, xs, s, cout, mean, rstd, gamma, beta,
                       (__hip_bfloat16*)out, ldc, coff);
  return tmae_launch_status();
}

// n_src <= 4 sources written in one launch: host arrays v[i], grid[i], mean[i] ... and ys[i], xs[i], s[i], cout[i] (the channel
// slices are laid side by side in source order: coff[i] = cout[0] + ... + cout[i-1], their sum = ldc); every source must cover the
// same dense grid (ys[i] * s[i] = Y, xs[i] * s[i] = X).
int tmae_deblock_scatter_multi(int n_src, const void* const* v, int dtype, const int32_t* const* grid, int batch, const int* ys,
                               const int* xs, const int* s, const int* cout, const float* const* mean, const float* const* rstd,
                               const float* const* gamma, const float* const* beta, void* out, int ldc, void* stream_) {
  (void)hipGetLastError();
  hipStream_t stream = (hipStream_t)stream_;
  const int es = esz(dtype);
  if (!es) return TMAE_EDTYPE;
  if (n_src < 1 || n_src > DS_MAXSRC || !v || !grid || !ys || !xs || !s || !cout || !mean || !rstd || !gamma || !beta || !out || batch <= 0 ||
      ldc <= 0 || (ldc * es) % 16)
    return TMAE_EARG;
  DbSrcs P;
  P.n = n_src;
  int coff = 0;
  const int Y = ys[0] * s[0], X = xs[0] * s[0];
  for (int i = 0; i < n_src; ++i) {
    if (!v[i] || !grid[i] || !mean[i] || !rstd[i] || !gamma[i] || !beta[i] || ys[i] <= 0 || xs[i] <= 0 || s[i] <= 0 || cout[i] <= 0 ||
        (cout[i] * es) % 16 || ys[i] * s[i] != Y || xs[i] * s[i] != X || (s[i] & (s[i] - 1)))
      return TMAE_EARG;                                       // (strides: powers of two -- the kernel shifts)
    int ls = 0;
    while ((1 << ls) < s[i]) ++ls;
    P.src[i] = DbSrc{v[i], grid[i], mean[i], rstd[i], gamma[i], beta[i], ys[i], xs[i], ls, cout[i], coff};
    coff += cout[i];
  }
  for (int i = n_src; i < DS_MAXSRC; ++i) P.src[i] = P.src[0];
  if (coff != ldc) return TMAE_EARG;
  const int chunks = ldc * es / 16;
  const int64_t ncell = (int64_t)batch * Y * X;
  if (chunks > 256 || ncell >= ((int64_t)1 << 31)) return TMAE_EARG;
  const int64_t per_block = (int64_t)(256 / chunks) * DS_CPT;
  if (dtype == TMAE_F32)
    hipLaunchKernelGGL(deblock_scatter_multi_kernel<float>, dim3(tmae_cdiv(ncell, per_block)), dim3(256), 0, stream, P, batch, Y, X,
                       (float*)out, ldc);
  else
    hipLaunchKernelGGL(deblock_scatter_multi_kernel<__hip_bfloat16>, dim3(tmae_cdiv(ncell, per_block)), dim3(256), 0, stream, P, batch,
                       Y, X, (__hip_bfloat16*)out, ldc);
  return tmae_launch_status();
}

int tmae_deblock_gather(const void* dcat, int dtype, int ldc, int coff, const int32_t* indices, int64_t m, int ys,
                        int xs, int s, int cout, void* g, void* stream_) {
  (void)hipGetLastError();
  hipStream_t stream = (hipStream_t)stream_;
  const int es = esz(dtype);
  if (!es) return TMAE_EDTYPE;
  if (m < 0 || ys <= 0 || xs <= 0 || s <= 0 || cout <= 0 || (cout * es) % 16 || (ldc * es) % 16 || (coff * es) % 16)
    return TMAE_EARG;
  if (m == 0) return TMAE_OK;
  if (!dcat || !indices || !g) return TMAE_EARG;
  const int64_t total = m * s * s * (cout * es / 16);
  if (dtype == TMAE_F32)
    hipLaunchKernelGGL(deblock_gather_kernel<float>, dim3(tmae_cdiv(total, 256)), dim3(256), 0, stream,
                       (const float*)dcat, ldc, coff, indices, m, ys, xs, s, cout, (float*)g);
  else
    hipLaunchKernelGGL(deblock_gather_kernel<__hip_bfloat16>, dim3(tmae_cdiv(total, 256)), dim3(256), 0, stream,
                       (const __hip_bfloat16*)dcat, ldc, coff, indices, m, ys, xs, s, cout, (__hip_bfloat16*)g);
  return tmae_launch_status();
}

static int colsum_grid(int64_t rows) {
  int64_t g = (rows + 63) / 64;
  if (g > 1024) g = 1024;
  if (g < 1) g = 1;
  return (int)g;
}

size_t tmae_column_sums_workspace(int64_t rows, int c) { return (size_t)colsum_grid(rows) * c * 4 + 256; }

int tmae_column_sums(const void* x, int dtype, int64_t rows, int c, float* out, void* wsp, size_t ws_bytes,
                     void* stream_) {
  (void)hipGetLastError();
  hipStream_t stream = (hipStream_t)stream_;
  if (rows <= 0 || c <= 0 || c % 64 || c > 512 || !x || !out) return TMAE_EARG;
  if (dtype != TMAE_F32 && dtype != TMAE_BF16) return TMAE_EDTYPE;
  const int nb = colsum_grid(rows);
  WsCarver ws(wsp, ws_bytes);
  float* part = ws.take<float>((size_t)nb * c);
  if (!ws.ok) return TMAE_EWS;
  dim3 grid(nb), block(256);
#define CS(T, V) hipLaunchKernelGGL((colsum_kernel<T, V>), grid, block, 0, stream, (const T*)x, rows, part)
#define CSD(T)                                                                                                    \
  switch (c / 64) { case 1: CS(T, 1); break; case 2: CS(T, 2); break; case 3: CS(T, 3); break; case 4: CS(T, 4); break; \
                    case 5: CS(T, 5); break; case 6: CS(T, 6); break; case 7: CS(T, 7); break; default: CS(T, 8); }
  if (dtype == TMAE_F32) { CSD(float) } else { CSD(__hip_bfloat16) }
#undef CSD
#undef CS
  hipLaunchKernelGGL(colsum_finish_kernel, dim3(tmae_cdiv(c, 16)), dim3(256), 0, stream, part, nb, c, out);
  return tmae_launch_status();
}

// The inactive cells' share of a deblock norm's parameter gradients.  Every output cell that no active input cell feeds holds
// the same value per channel, z0 = beta - mean rstd gamma with xhat0 = -mean rstd, so its part of the sums over ALL cells is
// (sum of dy over the inactive cells) x [z0 > 0]: dbeta = sum_dz + rest, dgamma = sum_dzx + rest xhat0 with
// rest = (s_all - s_act) [z0 > 0].  One launch instead of eleven [c]-sized elementwise launches per deblock.
__global__ __launch_bounds__(256) void deblock_bn_tail_kernel(const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ s_all, const float* __restrict__ s_act,
                                                             const float* __restrict__ sum_dz,
                                                             const float* __restrict__ sum_dzx, int c,
                                                             float* __restrict__ dbeta, float* __restrict__ dgamma) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= c) return;
  const float xhat0 = -mean[j] * rstd[j];
  const float live = (beta[j] + xhat0 * gamma[j]) > 0.f ? 1.f : 0.f;
  const float rest = (s_all[j] - s_act[j]) * live;
  dbeta[j] = sum_dz[j] + rest;
  dgamma[j] = sum_dzx[j] + rest * xhat0;
}

int tmae_deblock_bn_tail(const float* mean, const float* rstd, const float* gamma, const float* beta, const float* s_all,
                         const float* s_act, const float* sum_dz, const float* sum_dzx, int c, float* dbeta, float* dgamma,
                         void* stream_) {
  (void)hipGetLastError();
  if (c <= 0 || !mean || !rstd || !gamma || !beta || !s_all || !s_act || !sum_dz || !sum_dzx || !dbeta || !dgamma) return TMAE_EARG;
  hipLaunchKernelGGL(deblock_bn_tail_kernel, dim3(tmae_cdiv(c, 256)), dim3(256), 0, (hipStream_t)stream_, mean, rstd, gamma,
                     beta, s_all, s_act, sum_dz, sum_dzx, c, dbeta, dgamma);
  return tmae_launch_status();
}
