// A4/A5/A10: dense index grid, window partition + region batching, positional-embedding add.
//
// Design: the BEV grid is tiny (batch*468*468 int32 = 7 MB at batch 8) so every sparse tensor gets a
// dense row-index grid.  A wavefront owns one 8x8 window: lane l reads cell (l/8, l%8); one ballot gives
// the window's token count and, by popcount of the lower lanes, each token's stable in-window rank --
// exactly the quantity the reference builds with atomics + unique + sort (sst_ops_gpu.cu:14-20,
// sst_utils.py:61-107).  No sort, no atomics, no host sync.
#include "common.h"

#define MAX_LEVELS 8
struct LevelTable {
  int n;
  int max_tokens[MAX_LEVELS], lower[MAX_LEVELS], upper[MAX_LEVELS];
};

__global__ __launch_bounds__(256) void grid_scatter_kernel(const int32_t* __restrict__ ind, int64_t m, int batch,
                                                          int ny, int nx, int32_t* __restrict__ grid) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  const int b = ind[i * 3], y = ind[i * 3 + 1], x = ind[i * 3 + 2];
  if (b < 0 || b >= batch || y < 0 || y >= ny || x < 0 || x >= nx) return;   // never index out of the grid
  grid[((int64_t)b * ny + y) * nx + x] = (int32_t)i;
}

int tmae_index_grid(const int32_t* indices, int64_t m, int batch, int ny, int nx, int32_t* grid, void* stream_) {
  (void)hipGetLastError();   // drop stale errors of other runtime users: the return code is about OUR launches
  hipStream_t stream = (hipStream_t)stream_;
  if (m < 0 || batch <= 0 || ny <= 0 || nx <= 0 || !grid || (m > 0 && !indices)) return TMAE_EARG;
  (void)hipMemsetAsync(grid, 0xFF, (size_t)batch * ny * nx * 4, stream);
  if (m > 0)
    hipLaunchKernelGGL(grid_scatter_kernel, dim3(tmae_cdiv(m, 256)), dim3(256), 0, stream, indices, m, batch, ny, nx,
                       grid);
  return tmae_launch_status();
}

// one wavefront per dense window (b, wcx, wcy); dense window id dw = (b*Wx + wcx)*Wy + wcy, which orders
// windows exactly like the reference's batch_win_inds (sst_utils.py:48-51: x-major, then y).
__global__ __launch_bounds__(256) void win_count_kernel(const int32_t* __restrict__ grid,
                                                       const int32_t* __restrict__ grid_other, int batch, int ny,
                                                       int nx, int wy, int wx, int Wy, int Wx, int sy, int sx,
                                                       int32_t* __restrict__ wcount,
                                                       int32_t* __restrict__ wcount_other,
                                                       int32_t* __restrict__ inner) {
  const int lane = threadIdx.x & 63;
  const int64_t dw = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (dw >= (int64_t)batch * Wx * Wy) return;
  const int wcy = (int)(dw % Wy);
  const int wcx = (int)((dw / Wy) % Wx);
  const int b = (int)(dw / ((int64_t)Wy * Wx));
  const int ly = lane / wx, lx = lane % wx;
  const int y = wcy * wy - sy + ly, x = wcx * wx - sx + lx;
  const bool in = (ly < wy) && y >= 0 && y < ny && x >= 0 && x < nx;
  const int64_t cell = ((int64_t)b * ny + y) * nx + x;
  const int v = in ? grid[cell] : -1;
  const unsigned long long mask = __ballot(v >= 0);
  if (lane == 0) wcount[dw] = __popcll(mask);
  if (v >= 0) inner[v] = __popcll(mask & ((1ull << lane) - 1ull));
  if (grid_other) {
    const int vo = in ? grid_other[cell] : -1;
    const unsigned long long mo = __ballot(vo >= 0);
    if (lane == 0) wcount_other[dw] = __popcll(mo);
  }
}

__device__ __forceinline__ int level_of(int cnt, const LevelTable& lt) {
  int lvl = -1;
  for (int l = 0; l < lt.n; ++l)
    if (cnt >= lt.lower[l] && cnt < lt.upper[l]) lvl = l;     // later levels override, as the reference loop does
  return lvl;
}

__global__ __launch_bounds__(256) void win_level_kernel(const int32_t* __restrict__ wcount,
                                                       const int32_t* __restrict__ wcount_other, int64_t nwin,
                                                       LevelTable lt, int32_t* __restrict__ wlevel,
                                                       int32_t* __restrict__ flags /*[n_levels][nwin]*/) {
  int64_t dw = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (dw >= nwin) return;
  const int c = wcount[dw];
  int eff = c;
  bool dropped = false;
  if (wcount_other) {
    const int co = wcount_other[dw];
    dropped = (c == 0) || (co == 0);      // SiamWCA.py:86-91
    eff = max(c, co);                      // SiamWCA.py:94-118
  }
  const int lvl = level_of(eff, lt);
  const bool kept = (c > 0) && !dropped && lvl >= 0;
  wlevel[dw] = dropped ? (-2 - max(lvl, 0)) : lvl;   // <= -2 encodes "window dropped", level = -2 - value
  if (flags)
    for (int l = 0; l < lt.n; ++l) flags[(int64_t)l * nwin + dw] = (kept && lvl == l) ? 1 : 0;
}

__global__ __launch_bounds__(256) void win_emit_kernel(const int32_t* __restrict__ ind, int64_t m, int wy, int wx,
                                                      int Wy, int Wx, int sy, int sx,
                                                      const int32_t* __restrict__ wlevel,
                                                      const int32_t* __restrict__ ranks /*[n_levels][nwin]*/,
                                                      int64_t nwin, const int32_t* __restrict__ inner, LevelTable lt,
                                                      int64_t* __restrict__ bwi, int64_t* __restrict__ ciw,
                                                      int32_t* __restrict__ level, uint8_t* __restrict__ keep,
                                                      int64_t* __restrict__ f2w) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  const int b = ind[i * 3], y = ind[i * 3 + 1], x = ind[i * 3 + 2];
  const int scy = y + sy, scx = x + sx;
  const int wcy = scy / wy, wcx = scx / wx;
  // reference id: b*(Wx*Wy*Wz) + wcx*(Wy*Wz) + wcy*Wz + wcz with Wz = 2, wcz = 0 (sst_utils.py:23-51)
  if (bwi) bwi[i] = ((int64_t)b * Wx * Wy + (int64_t)wcx * Wy + wcy) * 2;
  if (ciw) { ciw[i * 3] = 0; ciw[i * 3 + 1] = scy % wy; ciw[i * 3 + 2] = scx % wx; }
  const int64_t dw = ((int64_t)b * Wx + wcx) * Wy + wcy;
  int lv = wlevel[dw];
  const bool dropped = lv <= -2;
  if (dropped) lv = -2 - lv;
  const int T = (lv >= 0) ? lt.max_tokens[lv] : 0;
  const int in = inner[i];
  const bool k = !dropped && lv >= 0 && in < T;
  if (level) level[i] = lv;
  if (keep) keep[i] = k ? 1 : 0;
  if (f2w) f2w[i] = k ? ((int64_t)ranks[(int64_t)lv * nwin + dw] * T + in) : -1;
}

static inline void win_dims(int ny, int nx, int wy, int wx, int& Wy, int& Wx) {
  Wy = (ny + wy - 1) / wy + 1;   // ceil(grid / win) + 1 (sst_utils.py:23-25)
  Wx = (nx + wx - 1) / wx + 1;
}

size_t tmae_window_bucket_workspace(int batch, int ny, int nx, int wy, int wx, int n_levels) {
  int Wy, Wx;
  win_dims(ny, nx, wy, wx, Wy, Wx);
  size_t nwin = (size_t)batch * Wy * Wx;
  return (3 + 2 * (size_t)n_levels) * tmae_align(nwin * 4) + (size_t)n_levels * tmae_scan_i32_workspace(nwin) + 4096;
}

int tmae_window_bucket(const int32_t* indices, int64_t m, const int32_t* grid, const int32_t* grid_other, int batch,
                       int ny, int nx, int wy, int wx, int do_shift, const int32_t* levels_host, int n_levels,
                       int64_t* bwi, int64_t* ciw, int32_t* inner, int32_t* level, uint8_t* keep, int64_t* f2w,
                       int32_t* win_per_level, void* wsp, size_t ws_bytes, void* stream_) {
  (void)hipGetLastError();   // drop stale errors of other runtime users: the return code is about OUR launches
  hipStream_t stream = (hipStream_t)stream_;
  if (m < 0 || batch <= 0 || ny <= 0 || nx <= 0 || wy <= 0 || wx <= 0 || wy * wx > 64 || !grid || !levels_host ||
      n_levels <= 0 || n_levels > MAX_LEVELS || !inner || (m > 0 && !indices))
    return TMAE_EARG;
  // the per-level window ranks (one scan per level) only serve flat2win and the per-level window counts: a caller that wants
  // neither -- the cross-attention blocks need `keep` alone -- passes both as NULL and gets three launches instead of 3 + n_levels
  const bool need_ranks = f2w != nullptr || win_per_level != nullptr;
  if (need_ranks && !win_per_level) return TMAE_EARG;
  LevelTable lt;
  lt.n = n_levels;
  for (int l = 0; l < n_levels; ++l) {
    lt.max_tokens[l] = levels_host[l * 3];
    lt.lower[l] = levels_host[l * 3 + 1];
    lt.upper[l] = levels_host[l * 3 + 2];
  }
  int Wy, Wx;
  win_dims(ny, nx, wy, wx, Wy, Wx);
  const int64_t nwin = (int64_t)batch * Wy * Wx;
  const int sy = do_shift ? wy / 2 : wy, sx = do_shift ? wx / 2 : wx;
  WsCarver ws(wsp, ws_bytes);
  int32_t* wcount = ws.take<int32_t>((size_t)nwin);
  int32_t* wcount_o = ws.take<int32_t>((size_t)nwin);
  int32_t* wlevel = ws.take<int32_t>((size_t)nwin);
  int32_t* flags = ws.take<int32_t>((size_t)nwin * n_levels);
  int32_t* ranks = ws.take<int32_t>((size_t)nwin * n_levels);
  size_t sb = tmae_scan_i32_workspace(nwin);
  char* scanws = ws.take<char>(sb * n_levels);
  if (!ws.ok) return TMAE_EWS;
  hipLaunchKernelGGL(win_count_kernel, dim3(tmae_cdiv(nwin, 4)), dim3(256), 0, stream, grid, grid_other, batch, ny,
                     nx, wy, wx, Wy, Wx, sy, sx, wcount, wcount_o, inner);
  hipLaunchKernelGGL(win_level_kernel, dim3(tmae_cdiv(nwin, 256)), dim3(256), 0, stream, wcount,
                     grid_other ? wcount_o : (const int32_t*)nullptr, nwin, lt, wlevel, need_ranks ? flags : (int32_t*)nullptr);
  for (int l = 0; need_ranks && l < n_levels; ++l) {
    int r = tmae_scan_i32(flags + (int64_t)l * nwin, ranks + (int64_t)l * nwin, nwin, win_per_level + l,
                          scanws + sb * l, sb, stream);
    if (r) return r;
  }
  if (m > 0)
    hipLaunchKernelGGL(win_emit_kernel, dim3(tmae_cdiv(m, 256)), dim3(256), 0, stream, indices, m, wy, wx, Wy, Wx, sy,
                       sx, wlevel, ranks, nwin, inner, lt, bwi, ciw, level, keep, f2w);
  return tmae_launch_status();
}

// ------------------------------------------------------------------------------------------------
// out = x + pos_table[cell]  (16-byte vectorised rows)
// ------------------------------------------------------------------------------------------------
template <class T, int VEC>
__global__ __launch_bounds__(256) void add_pos_kernel(const T* __restrict__ x, int64_t m, int d,
                                                     const int32_t* __restrict__ ind, int wy, int wx, int sy, int sx,
                                                     const float* __restrict__ table, T* __restrict__ out) {
  const int chunks = d / VEC;
  int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= m * chunks) return;
  const int64_t r = e / chunks;
  const int c0 = (int)(e - r * chunks) * VEC;
  const int y = ind[r * 3 + 1], xx = ind[r * 3 + 2];
  const int cell = ((y + sy) % wy) * wx + (xx + sx) % wx;
  const T* xi = x + r * d + c0;
  const float* ti = table + (int64_t)cell * d + c0;
  T* oi = out + r * d + c0;
  T tmp[VEC];
  *reinterpret_cast<uint4*>(tmp) = *reinterpret_cast<const uint4*>(xi);
#pragma unroll
  for (int k = 0; k < VEC; ++k) st_f<T>(&tmp[k], ld_f<T>(&tmp[k]) + ti[k]);
  *reinterpret_cast<uint4*>(oi) = *reinterpret_cast<uint4*>(tmp);
}

int tmae_add_pos_embed(const void* x, int dtype, int64_t m, int d, const int32_t* indices, int wy, int wx,
                       int do_shift, const float* pos_table, void* out, void* stream_) {
  (void)hipGetLastError();   // drop stale errors of other runtime users: the return code is about OUR launches
  hipStream_t stream = (hipStream_t)stream_;
  if (m < 0 || d <= 0 || wy <= 0 || wx <= 0) return TMAE_EARG;
  if (m == 0) return TMAE_OK;
  if (!x || !indices || !pos_table || !out) return TMAE_EARG;
  const int sy = do_shift ? wy / 2 : wy, sx = do_shift ? wx / 2 : wx;
  if (dtype == TMAE_F32) {
    if (d % 4) return TMAE_EARG;
    hipLaunchKernelGGL((add_pos_kernel<float, 4>), dim3(tmae_cdiv(m * (d / 4), 256)), dim3(256), 0, stream,
                       (const float*)x, m, d, indices, wy, wx, sy, sx, pos_table, (float*)out);
  } else if (dtype == TMAE_BF16) {
    if (d % 8) return TMAE_EARG;
    hipLaunchKernelGGL((add_pos_kernel<__hip_bfloat16, 8>), dim3(tmae_cdiv(m * (d / 8), 256)), dim3(256), 0, stream,
                       (const __hip_bfloat16*)x, m, d, indices, wy, wx, sy, sx, pos_table, (__hip_bfloat16*)out);
  } else {
    return TMAE_EDTYPE;
  }
  return tmae_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Cell of every token inside its (shifted) window, in the two forms the position-folded in-projection uses
// (token_gemm.hip POS variants): cells [m] = xc | yc << 3 for the forward's one-hot k-step, and onehot [m,16] bf16
// (columns 0..7 = one-hot xc, 8..15 = one-hot yc), the operand whose product with dY gives the position part of the
// in-projection weight gradient: dW[:, :d/2] += (dY^T onehot[:, :8]) ex, dW[:, d/2:] += (dY^T onehot[:, 8:]) ey.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void window_cells_kernel(const int32_t* __restrict__ ind, int64_t m, int64_t m_pad,
                                                          int wy, int wx, int sy, int sx, uint8_t* __restrict__ cells,
                                                          uint4* __restrict__ onehot) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= m) {                       // the padding tmae_linear_wgrad_cells may read: zeros, written here (no fill launch)
    if (r < m_pad) cells[r] = 0;
    return;
  }
  const unsigned yc = (unsigned)((ind[r * 3 + 1] + sy) % wy), xc = (unsigned)((ind[r * 3 + 2] + sx) % wx);
  cells[r] = (uint8_t)(xc | (yc << 3));
  const unsigned vx = (xc & 1u) ? 0x3F800000u : 0x00003F80u, ix = xc >> 1;
  const unsigned vy = (yc & 1u) ? 0x3F800000u : 0x00003F80u, iy = yc >> 1;
  if (!onehot) return;
  onehot[2 * r] = make_uint4(ix == 0u ? vx : 0u, ix == 1u ? vx : 0u, ix == 2u ? vx : 0u, ix == 3u ? vx : 0u);
  onehot[2 * r + 1] = make_uint4(iy == 0u ? vy : 0u, iy == 1u ? vy : 0u, iy == 2u ? vy : 0u, iy == 3u ? vy : 0u);
}

int tmae_window_cells(const int32_t* indices, int64_t m, int64_t cells_len, int wy, int wx, int do_shift, uint8_t* cells,
                      void* onehot, void* stream_) {
  (void)hipGetLastError();
  hipStream_t stream = (hipStream_t)stream_;
  if (m < 0 || cells_len < m || wy <= 0 || wx <= 0 || wy > 8 || wx > 8) return TMAE_EARG;
  if (cells_len == 0) return TMAE_OK;
  if (!indices || !cells || ((uintptr_t)onehot & 15)) return TMAE_EARG;
  const int sy = do_shift ? wy / 2 : wy, sx = do_shift ? wx / 2 : wx;
  hipLaunchKernelGGL(window_cells_kernel, dim3(tmae_cdiv(cells_len, 256)), dim3(256), 0, stream, indices, m, cells_len, wy,
                     wx, sy, sx, cells, (uint4*)onehot);
  return tmae_launch_status();
}
