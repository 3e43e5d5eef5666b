// A9/A11: 2-D sparse convolution rulebook + gather forms, sparse<->dense (channels-last).
//
// The rulebook is not a hash table: with a dense row-index grid per sparse tensor a tap lookup is one
// L2-resident int32 load.  Output order is lexicographic (b,y,x) (prefix sum over the dense output
// grid), which is the canonical order SURVEY A-9 fixes.  The contraction itself runs as ONE GEMM per
// conv over the gathered [m_out, 9*cin] matrix (output-stationary: no atomics, deterministic), and the
// data gradient uses the transposed neighbour table, again without atomics.
#include "common.h"
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void down_flag_kernel(const int32_t* __restrict__ grid_in, int batch, int ny,
                                                       int nx, int oy, int ox, int32_t* __restrict__ flag) {
  int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= (int64_t)batch * oy * ox) return;
  const int x = (int)(c % ox), y = (int)((c / ox) % oy), b = (int)(c / ((int64_t)ox * oy));
  int any = 0;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = 2 * y - 1 + ky;
    if (iy < 0 || iy >= ny) continue;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = 2 * x - 1 + kx;
      if (ix < 0 || ix >= nx) continue;
      any |= (grid_in[((int64_t)b * ny + iy) * nx + ix] >= 0) ? 1 : 0;
    }
  }
  flag[c] = any;
}

__global__ __launch_bounds__(256) void down_emit_kernel(const int32_t* __restrict__ flag,
                                                       const int32_t* __restrict__ rank, int batch, int oy, int ox,
                                                       int32_t* __restrict__ out_grid,
                                                       int32_t* __restrict__ out_indices) {
  int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= (int64_t)batch * oy * ox) return;
  if (flag[c]) {
    const int r = rank[c];
    out_grid[c] = r;
    out_indices[(int64_t)r * 3] = (int)(c / ((int64_t)ox * oy));
    out_indices[(int64_t)r * 3 + 1] = (int)((c / ox) % oy);
    out_indices[(int64_t)r * 3 + 2] = (int)(c % ox);
  } else {
    out_grid[c] = -1;
  }
}

size_t tmae_spconv_down_outputs_workspace(int batch, int oy, int ox) {
  size_t cells = (size_t)batch * oy * ox;
  return 2 * tmae_align(cells * 4) + tmae_scan_i32_workspace((int64_t)cells) + 1024;
}

int tmae_spconv_down_outputs(const int32_t* grid_in, int batch, int ny, int nx, int oy, int ox, int32_t* out_grid,
                             int32_t* out_indices, int32_t* n_out, void* wsp, size_t ws_bytes, void* stream_) {
  (void)hipGetLastError();   // drop stale errors of other runtime users: the return code is about OUR launches
  hipStream_t stream = (hipStream_t)stream_;
  if (!grid_in || !out_grid || !out_indices || !n_out || batch <= 0 || ny <= 0 || nx <= 0) return TMAE_EARG;
  if (oy != (ny + 2 - 3) / 2 + 1 || ox != (nx + 2 - 3) / 2 + 1) return TMAE_EARG;
  const int64_t cells = (int64_t)batch * oy * ox;
  WsCarver ws(wsp, ws_bytes);
  int32_t* flag = ws.take<int32_t>((size_t)cells);
  int32_t* rank = ws.take<int32_t>((size_t)cells);
  size_t sb = tmae_scan_i32_workspace(cells);
  char* scanws = ws.take<char>(sb);
  if (!ws.ok) return TMAE_EWS;
  hipLaunchKernelGGL(down_flag_kernel, dim3(tmae_cdiv(cells, 256)), dim3(256), 0, stream, grid_in, batch, ny, nx, oy,
                     ox, flag);
  int r = tmae_scan_i32(flag, rank, cells, n_out, scanws, sb, stream);
  if (r) return r;
  hipLaunchKernelGGL(down_emit_kernel, dim3(tmae_cdiv(cells, 256)), dim3(256), 0, stream, flag, rank, batch, oy, ox,
                     out_grid, out_indices);
  return tmae_launch_status();
}

__global__ __launch_bounds__(256) void nbr_kernel(const int32_t* __restrict__ out_ind, int64_t m_out,
                                                 const int32_t* __restrict__ grid_in, int batch, int ny, int nx,
                                                 int stride, int32_t* __restrict__ nbr) {
  int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= m_out * 9) return;
  const int64_t o = e / 9;
  const int t = (int)(e - o * 9), ky = t / 3, kx = t % 3;
  const int b = out_ind[o * 3], iy = out_ind[o * 3 + 1] * stride - 1 + ky, ix = out_ind[o * 3 + 2] * stride - 1 + kx;
  int r = -1;
  if (b >= 0 && b < batch && iy >= 0 && iy < ny && ix >= 0 && ix < nx) r = grid_in[((int64_t)b * ny + iy) * nx + ix];
  nbr[e] = r;
}

__global__ __launch_bounds__(256) void nbr_t_kernel(const int32_t* __restrict__ in_ind, int64_t m_in,
                                                   const int32_t* __restrict__ grid_out, int batch, int oy, int ox,
                                                   int stride, int32_t* __restrict__ nbr_t) {
  int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= m_in * 9) return;
  const int64_t i = e / 9;
  const int t = (int)(e - i * 9), ky = t / 3, kx = t % 3;
  const int b = in_ind[i * 3];
  const int ty = in_ind[i * 3 + 1] + 1 - ky, tx = in_ind[i * 3 + 2] + 1 - kx;   // = out * stride
  int r = -1;
  if (b >= 0 && b < batch && ty >= 0 && tx >= 0 && (ty % stride) == 0 && (tx % stride) == 0) {
    const int y = ty / stride, x = tx / stride;
    if (y < oy && x < ox) r = grid_out[((int64_t)b * oy + y) * ox + x];
  }
  nbr_t[e] = r;
}

int tmae_spconv_neighbors(const int32_t* out_indices, int64_t m_out, const int32_t* grid_in, int batch, int ny,
                          int nx, int stride, int32_t* nbr, void* stream_) {
  (void)hipGetLastError();   // drop stale errors of other runtime users: the return code is about OUR launches
  hipStream_t stream = (hipStream_t)stream_;
  if (m_out < 0 || batch <= 0 || ny <= 0 || nx <= 0 || (stride != 1 && stride != 2)) return TMAE_EARG;
  if (m_out == 0) return TMAE_OK;
  if (!out_indices || !grid_in || !nbr) return TMAE_EARG;
  hipLaunchKernelGGL(nbr_kernel, dim3(tmae_cdiv(m_out * 9, 256)), dim3(256), 0, stream, out_indices, m_out, grid_in,
                     batch, ny, nx, stride, nbr);
  return tmae_launch_status();
}

int tmae_spconv_neighbors_t(const int32_t* in_indices, int64_t m_in, const int32_t* grid_out, int batch, int oy,
                            int ox, int stride, int32_t* nbr_t, void* stream_) {
  (void)hipGetLastError();   // drop stale errors of other runtime users: the return code is about OUR launches
  hipStream_t stream = (hipStream_t)stream_;
  if (m_in < 0 || batch <= 0 || oy <= 0 || ox <= 0 || (stride != 1 && stride != 2)) return TMAE_EARG;
  if (m_in == 0) return TMAE_OK;
  if (!in_indices || !grid_out || !nbr_t) return TMAE_EARG;
  hipLaunchKernelGGL(nbr_t_kernel, dim3(tmae_cdiv(m_in * 9, 256)), dim3(256), 0, stream, in_indices, m_in, grid_out,
                     batch, oy, ox, stride, nbr_t);
  return tmae_launch_status();
}

// cols[o, t*c + ch] = feat[nbr[o,t], ch]  -- 16-byte chunks, chunk index fastest => fully coalesced stores
template <int VEC_BYTES>
__global__ __launch_bounds__(256) void gather9_kernel(const char* __restrict__ feat, int row_bytes,
                                                     const int32_t* __restrict__ nbr, int64_t m_out,
                                                     char* __restrict__ cols) {
  const int chunks = row_bytes / VEC_BYTES;
  int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= m_out * 9 * chunks) return;
  const int64_t ot = e / chunks;
  const int ch = (int)(e - ot * chunks);
  const int r = nbr[ot];
  uint4 val = make_uint4(0, 0, 0, 0);
  if (r >= 0) val = *reinterpret_cast<const uint4*>(feat + (int64_t)r * row_bytes + (int64_t)ch * VEC_BYTES);
  // the im2col matrix (~1 GB) is written once and read once by the GEMM: streamed past the caches (nt)
  __builtin_nontemporal_store(__builtin_bit_cast(u32x4_t, val), reinterpret_cast<u32x4_t*>(cols + e * VEC_BYTES));
}

template <class T, int VEC>
__global__ __launch_bounds__(256) void gather9_t_kernel(const T* __restrict__ dcols, int c,
                                                       const int32_t* __restrict__ nbr_t, int64_t m_in,
                                                       T* __restrict__ din) {
  const int chunks = c / VEC;
  int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= m_in * chunks) return;
  const int64_t i = e / chunks;
  const int c0 = (int)(e - i * chunks) * VEC;
  float acc[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
  // all nine rulebook entries, then all nine rows, unconditionally (an absent tap reads row 0 and is masked): a load
  // under `if (o >= 0)` is waited for before the next one is issued -- nine dependent round trips per thread
  int o[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) o[t] = nbr_t[i * 9 + t];
  uint4 raw[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
    raw[t] = *reinterpret_cast<const uint4*>(dcols + ((int64_t)(o[t] < 0 ? 0 : o[t]) * 9 + t) * c + c0);
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    T tmp[VEC];
    *reinterpret_cast<uint4*>(tmp) = raw[t];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] += o[t] < 0 ? 0.f : ld_f<T>(&tmp[k]);
  }
  T outv[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) st_f<T>(&outv[k], acc[k]);
  *reinterpret_cast<uint4*>(din + i * c + c0) = *reinterpret_cast<uint4*>(outv);
}

static int esize(int dtype) { return dtype == TMAE_F32 ? 4 : (dtype == TMAE_BF16 ? 2 : 0); }

int tmae_spconv_gather(const void* feat, int dtype, int64_t m_in, int c, const int32_t* nbr, int64_t m_out,
                       void* cols, void* stream_) {
  (void)hipGetLastError();   // drop stale errors of other runtime users: the return code is about OUR launches
  hipStream_t stream = (hipStream_t)stream_;
  const int es = esize(dtype);
  if (!es) return TMAE_EDTYPE;
  if (m_in < 0 || m_out < 0 || c <= 0 || (c * es) % 16) return TMAE_EARG;
  if (m_out == 0) return TMAE_OK;
  if (!nbr || !cols || (m_in > 0 && !feat)) return TMAE_EARG;
  const int64_t total = m_out * 9 * (c * es / 16);
  hipLaunchKernelGGL(gather9_kernel<16>, dim3(tmae_cdiv(total, 256)), dim3(256), 0, stream, (const char*)feat, c * es,
                     nbr, m_out, (char*)cols);
  return tmae_launch_status();
}

int tmae_spconv_gather_t(const void* dcols, int dtype, int64_t m_out, int c, const int32_t* nbr_t, int64_t m_in,
                         void* din, void* stream_) {
  (void)hipGetLastError();   // drop stale errors of other runtime users: the return code is about OUR launches
  hipStream_t stream = (hipStream_t)stream_;
  const int es = esize(dtype);
  if (!es) return TMAE_EDTYPE;
  if (m_in < 0 || m_out < 0 || c <= 0 || (c * es) % 16) return TMAE_EARG;
  if (m_in == 0) return TMAE_OK;
  if (!nbr_t || !din || (m_out > 0 && !dcols)) return TMAE_EARG;
  if (dtype == TMAE_F32)
    hipLaunchKernelGGL((gather9_t_kernel<float, 4>), dim3(tmae_cdiv(m_in * (c / 4), 256)), dim3(256), 0, stream,
                       (const float*)dcols, c, nbr_t, m_in, (float*)din);
  else
    hipLaunchKernelGGL((gather9_t_kernel<__hip_bfloat16, 8>), dim3(tmae_cdiv(m_in * (c / 8), 256)), dim3(256), 0,
                       stream, (const __hip_bfloat16*)dcols, c, nbr_t, m_in, (__hip_bfloat16*)din);
  return tmae_launch_status();
}

// ------------------------------------------------------------------------------------------------
// sparse -> dense NHWC (one pass: zero fill fused with the row copy) and dense -> rows
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void to_dense_kernel(const char* __restrict__ feat, int row_bytes,
                                                      const int32_t* __restrict__ grid, int64_t cells,
                                                      char* __restrict__ out) {
  const int chunks = row_bytes / 16;
  int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= cells * chunks) return;
  const int64_t cell = e / chunks;
  const int ch = (int)(e - cell * chunks);
  const int r = grid[cell];
  uint4 val = make_uint4(0, 0, 0, 0);
  if (r >= 0) val = *reinterpret_cast<const uint4*>(feat + (int64_t)r * row_bytes + (int64_t)ch * 16);
  *reinterpret_cast<uint4*>(out + e * 16) = val;
}

__global__ __launch_bounds__(256) void dense_gather_kernel(const char* __restrict__ dense, int row_bytes, int batch,
                                                          int ny, int nx, const int32_t* __restrict__ ind, int64_t m,
                                                          char* __restrict__ rows) {
  const int chunks = row_bytes / 16;
  int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= m * chunks) return;
  const int64_t r = e / chunks;
  const int ch = (int)(e - r * chunks);
  const int b = ind[r * 3], y = ind[r * 3 + 1], x = ind[r * 3 + 2];
  uint4 val = make_uint4(0, 0, 0, 0);
  if (b >= 0 && b < batch && y >= 0 && y < ny && x >= 0 && x < nx)
    val = *reinterpret_cast<const uint4*>(dense + (((int64_t)b * ny + y) * nx + x) * row_bytes + (int64_t)ch * 16);
  *reinterpret_cast<uint4*>(rows + e * 16) = val;
}

int tmae_sparse_to_dense(const void* feat, int dtype, int64_t m, int c, const int32_t* grid, int batch, int ny,
                         int nx, void* out, void* stream_) {
  (void)hipGetLastError();   // drop stale errors of other runtime users: the return code is about OUR launches
  hipStream_t stream = (hipStream_t)stream_;
  const int es = esize(dtype);
  if (!es) return TMAE_EDTYPE;
  if (m < 0 || c <= 0 || (c * es) % 16 || batch <= 0 || ny <= 0 || nx <= 0 || !grid || !out || (m > 0 && !feat))
    return TMAE_EARG;
  const int64_t cells = (int64_t)batch * ny * nx;
  hipLaunchKernelGGL(to_dense_kernel, dim3(tmae_cdiv(cells * (c * es / 16), 256)), dim3(256), 0, stream,
                     (const char*)feat, c * es, grid, cells, (char*)out);
  return tmae_launch_status();
}

int tmae_dense_gather(const void* dense, int dtype, int batch, int ny, int nx, int c, const int32_t* indices,
                      int64_t m, void* rows, void* stream_) {
  (void)hipGetLastError();   // drop stale errors of other runtime users: the return code is about OUR launches
  hipStream_t stream = (hipStream_t)stream_;
  const int es = esize(dtype);
  if (!es) return TMAE_EDTYPE;
  if (m < 0 || c <= 0 || (c * es) % 16 || batch <= 0 || ny <= 0 || nx <= 0) return TMAE_EARG;
  if (m == 0) return TMAE_OK;
  if (!dense || !indices || !rows) return TMAE_EARG;
  hipLaunchKernelGGL(dense_gather_kernel, dim3(tmae_cdiv(m * (c * es / 16), 256)), dim3(256), 0, stream,
                     (const char*)dense, c * es, batch, ny, nx, indices, m, (char*)rows);
  return tmae_launch_status();
}
