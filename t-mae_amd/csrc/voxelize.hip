// A1/A2/A5/A12: dynamic voxelisation, point->voxel CSR, VFE point features, segment max,
// per-voxel point grouping.  All of these are HBM/latency bound index kernels: one pass over the
// points with coalesced row reads; uniqueness comes from a dense occupancy grid (the pillar grid is
// only batch*468*468 cells, L2-resident) + prefix sums instead of a 4-column int64 row sort.
#include "common.h"

// ------------------------------------------------------------------------------------------------
// voxelize
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vox_key_kernel(const float* __restrict__ pts, int row, int64_t n, int batch,
                                                     float rx, float ry, float rz, float vx, float vy, float vz,
                                                     int gx, int gy, int gz, int32_t* __restrict__ flag,
                                                     int32_t* __restrict__ key) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float* p = pts + i * row;
  // IEEE fp32 subtract + divide, truncation toward zero (common_utils.py:74); no reciprocal, no fma.
  float qx = div_rn(sub_rn(p[1], rx), vx);
  float qy = div_rn(sub_rn(p[2], ry), vy);
  float qz = div_rn(sub_rn(p[3], rz), vz);
  float fb = p[0];
  bool ok = (qx > -1.0f) && (qy > -1.0f) && (qz > -1.0f) && (qx < (float)gx) && (qy < (float)gy) &&
            (qz < (float)gz) && (fb > -1.0f) && (fb < (float)batch);
  int k = 0;
  if (ok) {
    int cx = (int)qx, cy = (int)qy, cz = (int)qz, b = (int)fb;   // trunc; (-1,0) -> 0 is kept (A-1)
    k = ((b * gz + cz) * gy + cy) * gx + cx;
  }
  flag[i] = ok ? 1 : 0;
  key[i] = k;
}

__global__ __launch_bounds__(256) void vox_compact_kernel(const float* __restrict__ pts, int row, int64_t n,
                                                         const int32_t* __restrict__ flag,
                                                         const int32_t* __restrict__ pos,
                                                         const int32_t* __restrict__ key, int gx, int gy, int gz,
                                                         float* __restrict__ pts_out, int64_t* __restrict__ pc,
                                                         int32_t* __restrict__ keyc, int32_t* __restrict__ occ) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n || !flag[i]) return;
  const int64_t j = pos[i];
  const float* p = pts + i * row;
  float* o = pts_out + j * row;
  for (int c = 0; c < row; ++c) o[c] = p[c];
  const int k = key[i];
  int cx = k % gx, t = k / gx;
  int cy = t % gy; t /= gy;
  int cz = t % gz, b = t / gz;
  int64_t* c = pc + j * 4;
  c[0] = b; c[1] = cz; c[2] = cy; c[3] = cx;
  keyc[j] = k;
  occ[k] = 1;
}

__global__ __launch_bounds__(256) void vox_emit_kernel(const int32_t* __restrict__ occ,
                                                      const int32_t* __restrict__ rank, int64_t cells, int gx,
                                                      int gy, int gz, int64_t* __restrict__ vc) {
  int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= cells || !occ[k]) return;
  int64_t* c = vc + (int64_t)rank[k] * 4;
  int kk = (int)k;
  int cx = kk % gx, t = kk / gx;
  int cy = t % gy; t /= gy;
  c[0] = t / gz; c[1] = t % gz; c[2] = cy; c[3] = cx;
}

__global__ __launch_bounds__(256) void vox_inverse_kernel(const int32_t* __restrict__ keyc,
                                                         const int32_t* __restrict__ rank,
                                                         const int32_t* __restrict__ counts, int64_t n,
                                                         int64_t* __restrict__ inverse) {
  int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n || j >= counts[0]) return;
  inverse[j] = rank[keyc[j]];
}

__global__ void vox_counts_kernel(const int32_t* occ, const int32_t* rank, int64_t cells, int batch,
                                  int cells_per_sample, int32_t* counts) {
  int b = threadIdx.x;
  if (b < batch) {
    int64_t lo = (int64_t)b * cells_per_sample, hi = lo + cells_per_sample;
    int end = (hi < cells) ? rank[hi] : (rank[cells - 1] + occ[cells - 1]);
    counts[2 + b] = end - rank[lo];
  }
}

size_t tmae_voxelize_workspace(int64_t n, int batch, int gx, int gy, int gz) {
  int64_t cells = (int64_t)batch * gx * gy * gz;
  return 4 * tmae_align((size_t)n * 4) + 2 * tmae_align((size_t)cells * 4) + tmae_scan_i32_workspace(n) +
         tmae_scan_i32_workspace(cells) + 4096;
}

int tmae_voxelize(const float* points, int row, int64_t n, int batch, float rx, float ry, float rz, float vx, float vy,
                  float vz, int gx, int gy, int gz, float* points_out, int64_t* point_coords, int64_t* inverse,
                  int64_t* voxel_coords, int32_t* counts, void* wsp, size_t ws_bytes, void* stream_) {
  (void)hipGetLastError();   // drop stale errors of other runtime users: the return code is about OUR launches
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || batch <= 0 || gx <= 0 || gy <= 0 || gz <= 0 || !counts || row < 4 || row > 16) return TMAE_EARG;
  int64_t cells = (int64_t)batch * gx * gy * gz;
  if (cells >= (1ll << 31) || n >= (1ll << 31)) return TMAE_EARG;
  if (n > 0 && (!points || !points_out || !point_coords || !inverse || !voxel_coords)) return TMAE_EARG;
  WsCarver ws(wsp, ws_bytes);
  int32_t* flag = ws.take<int32_t>((size_t)n);
  int32_t* pos = ws.take<int32_t>((size_t)n);
  int32_t* key = ws.take<int32_t>((size_t)n);
  int32_t* keyc = ws.take<int32_t>((size_t)n);
  int32_t* occ = ws.take<int32_t>((size_t)cells);
  int32_t* rank = ws.take<int32_t>((size_t)cells);
  size_t s1 = tmae_scan_i32_workspace(n), s2 = tmae_scan_i32_workspace(cells);
  char* scan1 = ws.take<char>(s1);
  char* scan2 = ws.take<char>(s2);
  if (!ws.ok) return TMAE_EWS;
  (void)hipMemsetAsync(occ, 0, (size_t)cells * 4, stream);
  if (n > 0)
    hipLaunchKernelGGL(vox_key_kernel, dim3(tmae_cdiv(n, 256)), dim3(256), 0, stream, points, row, n, batch, rx, ry, rz,
                       vx, vy, vz, gx, gy, gz, flag, key);
  int r = tmae_scan_i32(flag, pos, n, counts + 0, scan1, s1, stream);
  if (r) return r;
  if (n > 0)
    hipLaunchKernelGGL(vox_compact_kernel, dim3(tmae_cdiv(n, 256)), dim3(256), 0, stream, points, row, n, flag, pos, key,
                       gx, gy, gz, points_out, point_coords, keyc, occ);
  r = tmae_scan_i32(occ, rank, cells, counts + 1, scan2, s2, stream);
  if (r) return r;
  hipLaunchKernelGGL(vox_emit_kernel, dim3(tmae_cdiv(cells, 256)), dim3(256), 0, stream, occ, rank, cells, gx, gy, gz,
                     voxel_coords);
  if (n > 0)
    hipLaunchKernelGGL(vox_inverse_kernel, dim3(tmae_cdiv(n, 256)), dim3(256), 0, stream, keyc, rank, counts, n,
                       inverse);
  hipLaunchKernelGGL(vox_counts_kernel, dim3(1), dim3(tmae_align(batch, 64)), 0, stream, occ, rank, cells, batch,
                     gx * gy * gz, counts);
  return tmae_launch_status();
}

// ------------------------------------------------------------------------------------------------
// point -> group CSR (stable) and in-group rank
// ------------------------------------------------------------------------------------------------
// Counting sort instead of a radix sort of (group, point) pairs: groups are voxels / windows with a handful of
// members, so (1) integer counts per group + exclusive scan give the segment offsets, (2) every element takes a slot of
// its segment through an atomic cursor (arbitrary order inside the segment), (3) every element counts the members of
// its segment with a SMALLER element id -- its stable rank -- and moves there.  Deterministic and stable like the
// sort it replaces (42 library launches per step), with sum(len^2) reads that stay in L2 (len ~ 2-3, <= a few hundred).
__global__ __launch_bounds__(256) void csr_count_kernel(const int64_t* __restrict__ g, int64_t n,
                                                       int32_t* __restrict__ cnt) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) atomicAdd(cnt + g[i], 1);   // integer count: deterministic
}

__global__ __launch_bounds__(256) void csr_scatter_kernel(const int64_t* __restrict__ g, int64_t n,
                                                         const int32_t* __restrict__ offsets,
                                                         int32_t* __restrict__ cursor, int32_t* __restrict__ tmp) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t k = g[i];
  tmp[offsets[k] + atomicAdd(cursor + k, 1)] = (int32_t)i;
}

// slot j of the unordered segment image -> perm[offsets + stable rank]; rank_out (optional) [element] = that rank
__global__ __launch_bounds__(256) void csr_rank_kernel(const int64_t* __restrict__ g, int64_t n,
                                                      const int32_t* __restrict__ offsets,
                                                      const int32_t* __restrict__ tmp, int32_t* __restrict__ perm,
                                                      int64_t* __restrict__ rank_out) {
  int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const int32_t e = tmp[j];
  const int64_t k = g[e];
  const int lo = offsets[k], hi = offsets[k + 1];
  int r = 0;
  for (int t = lo; t < hi; ++t) r += tmp[t] < e;
  if (perm) perm[lo + r] = e;
  if (rank_out) rank_out[e] = r;
}

size_t tmae_segment_csr_workspace(int64_t n, int64_t m) {
  return tmae_align((size_t)n * 4) + 2 * tmae_align((size_t)(m + 1) * 4) + tmae_scan_i32_workspace(m + 1) + 4096;
}

static int csr_build(const int64_t* g, int64_t n, int64_t m, int32_t* perm, int32_t* offsets, int64_t* rank_out,
                     void* wsp, size_t ws_bytes, hipStream_t stream) {
  if (n < 0 || m < 0 || n >= (1ll << 31) || m >= (1ll << 31) || !offsets) return TMAE_EARG;
  if (n > 0 && (!g || (!perm && !rank_out))) return TMAE_EARG;
  WsCarver ws(wsp, ws_bytes);
  int32_t* tmp = ws.take<int32_t>((size_t)n);
  int32_t* cnt = ws.take<int32_t>((size_t)m + 1);
  int32_t* cursor = ws.take<int32_t>((size_t)m + 1);
  size_t sb = tmae_scan_i32_workspace(m + 1);
  char* scanws = ws.take<char>(sb);
  if (!ws.ok) return TMAE_EWS;
  (void)hipMemsetAsync(cnt, 0, (size_t)(m + 1) * 4, stream);
  (void)hipMemsetAsync(cursor, 0, (size_t)(m + 1) * 4, stream);
  if (n > 0) hipLaunchKernelGGL(csr_count_kernel, dim3(tmae_cdiv(n, 256)), dim3(256), 0, stream, g, n, cnt);
  int r = tmae_scan_i32(cnt, offsets, m + 1, nullptr, scanws, sb, stream);   // offsets[m] = n
  if (r) return r;
  if (n > 0) {
    hipLaunchKernelGGL(csr_scatter_kernel, dim3(tmae_cdiv(n, 256)), dim3(256), 0, stream, g, n, offsets, cursor, tmp);
    hipLaunchKernelGGL(csr_rank_kernel, dim3(tmae_cdiv(n, 256)), dim3(256), 0, stream, g, n, offsets, tmp, perm,
                       rank_out);
  }
  return tmae_launch_status();
}

int tmae_segment_csr(const int64_t* inverse, int64_t n, int64_t m, int32_t* perm, int32_t* offsets, void* ws,
                     size_t ws_bytes, void* stream) {
  (void)hipGetLastError();   // drop stale errors of other runtime users: the return code is about OUR launches
  return csr_build(inverse, n, m, perm, offsets, nullptr, ws, ws_bytes, (hipStream_t)stream);
}

size_t tmae_ingroup_rank_workspace(int64_t n, int64_t num_groups) {
  return tmae_segment_csr_workspace(n, num_groups) + tmae_align((size_t)(num_groups + 1) * 4);
}

int tmae_ingroup_rank(const int64_t* group, int64_t n, int64_t num_groups, int64_t* out, void* wsp, size_t ws_bytes,
                      void* stream_) {
  (void)hipGetLastError();   // drop stale errors of other runtime users: the return code is about OUR launches
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || num_groups < 0 || (n > 0 && (!group || !out))) return TMAE_EARG;
  if (n == 0) return TMAE_OK;
  WsCarver ws(wsp, ws_bytes);
  int32_t* offsets = ws.take<int32_t>((size_t)num_groups + 1);
  if (!ws.ok) return TMAE_EWS;
  return csr_build(group, n, num_groups, nullptr, offsets, out, ws.base + ws.used, ws.size - ws.used, stream);
}

// ------------------------------------------------------------------------------------------------
// VFE point features
// ------------------------------------------------------------------------------------------------
#define VFE_MAXF 8
__global__ __launch_bounds__(256) void voxel_mean_kernel(const float* __restrict__ pts, int row,
                                                        const int32_t* __restrict__ perm,
                                                        const int32_t* __restrict__ offsets, int64_t m,
                                                        float* __restrict__ mean) {
  int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (v >= m) return;
  const int F = row - 1;
  const int lo = offsets[v], hi = offsets[v + 1];
  float s[VFE_MAXF];
#pragma unroll
  for (int c = 0; c < VFE_MAXF; ++c) s[c] = 0.f;
  // ascending point id: deterministic sum.  Four points per round: their ids, then their rows, are loaded side by side and added
  // in order -- the rolled loop was two dependent round trips per POINT, and the launch lasted as long as its fullest voxel
  // (~100 points next to the sensor: 127 us for 18 MB).
  int j = lo;
  for (; j + 3 < hi; j += 4) {
    int id[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) id[q] = perm[j + q];
    float t[4][VFE_MAXF];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float* p = pts + (int64_t)id[q] * row;
#pragma unroll
      for (int c = 0; c < VFE_MAXF; ++c) t[q][c] = c < F ? p[1 + c] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int c = 0; c < VFE_MAXF; ++c)
        if (c < F) s[c] = add_rn(s[c], t[q][c]);
  }
  for (; j < hi; ++j) {
    const float* p = pts + (int64_t)perm[j] * row;
#pragma unroll
    for (int c = 0; c < VFE_MAXF; ++c)
      if (c < F) s[c] = add_rn(s[c], p[1 + c]);
  }
  const float cnt = (float)max(hi - lo, 1);
  float* o = mean + v * F;
#pragma unroll
  for (int c = 0; c < VFE_MAXF; ++c)
    if (c < F) o[c] = __fdiv_rn(s[c], cnt);
}

// feats = [f_center(3) | point features (F) | f_cluster(3)]  (temporal_dyn_vfe.py:88-109)
__global__ __launch_bounds__(256) void point_feat_kernel(const float* __restrict__ pts, int row,
                                                        const int64_t* __restrict__ pc,
                                                        const int64_t* __restrict__ inv,
                                                        const float* __restrict__ mean, int64_t n, float rx, float ry,
                                                        float rz, float vx, float vy, float vz,
                                                        float* __restrict__ feats) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int F = row - 1;
  const float* p = pts + i * row;
  const int64_t* c = pc + i * 4;
  const float* mu = mean + inv[i] * F;
  const float x = p[1], y = p[2], z = p[3];
  // f_center = p - ((coord + 0.5) * vs + rmin): separate fp32 roundings as in temporal_dyn_vfe.py:93-96
  const float ccx = add_rn(mul_rn(add_rn((float)c[3], 0.5f), vx), rx);
  const float ccy = add_rn(mul_rn(add_rn((float)c[2], 0.5f), vy), ry);
  const float ccz = add_rn(mul_rn(add_rn((float)c[1], 0.5f), vz), rz);
  float* f = feats + i * (F + 6);
  f[0] = sub_rn(x, ccx); f[1] = sub_rn(y, ccy); f[2] = sub_rn(z, ccz);
  for (int k = 0; k < F; ++k) f[3 + k] = p[1 + k];
  f[3 + F] = sub_rn(x, mu[0]); f[4 + F] = sub_rn(y, mu[1]); f[5 + F] = sub_rn(z, mu[2]);
}

// The same features as ONE bf16 matrix [n, 32] = [hi(16) | lo(16)], hi = bf16(f), lo = bf16(f - hi), columns past
// F+6 zero: x = hi + lo to ~2^-17 relative, so a bf16 MFMA product over the 32 columns against [W | W] is the
// fp32-input Linear of the VFE (absolute coordinates reach 75 m: a plain bf16 cast would round them to 0.25-0.5 m,
// coarser than the 0.32 m pillar).
__global__ __launch_bounds__(256) void point_feat_split_kernel(const float* __restrict__ pts, int row,
                                                              const int64_t* __restrict__ pc,
                                                              const int64_t* __restrict__ inv,
                                                              const float* __restrict__ mean, int64_t n, float rx,
                                                              float ry, float rz, float vx, float vy, float vz,
                                                              __hip_bfloat16* __restrict__ out,
                                                              const int32_t* __restrict__ perm,
                                                              int64_t* __restrict__ inv_csr) {
  // perm / inv_csr (both or neither): row j of `out` is point perm[j] -- the rows come out SORTED BY VOXEL, inv_csr[j] = that
  // point's voxel -- so that everything downstream that walks a voxel's points (the segment max and its backward) reads
  // consecutive rows instead of gathering 256-byte rows all over the point list
  const int64_t j_ = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j_ >= n) return;
  const int64_t i = perm ? (int64_t)perm[j_] : j_;
  const int F = row - 1;
  const float* p = pts + i * row;
  const int64_t* c = pc + i * 4;
  const float* mu = mean + inv[i] * F;
  const float x = p[1], y = p[2], z = p[3];
  const float ccx = add_rn(mul_rn(add_rn((float)c[3], 0.5f), vx), rx);
  const float ccy = add_rn(mul_rn(add_rn((float)c[2], 0.5f), vy), ry);
  const float ccz = add_rn(mul_rn(add_rn((float)c[1], 0.5f), vz), rz);
  float f[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) f[k] = 0.f;
  f[0] = sub_rn(x, ccx); f[1] = sub_rn(y, ccy); f[2] = sub_rn(z, ccz);
#pragma unroll
  for (int k = 0; k < VFE_MAXF; ++k)
    if (k < F) f[3 + k] = p[1 + k];
  const float dcl[3] = {sub_rn(x, mu[0]), sub_rn(y, mu[1]), sub_rn(z, mu[2])};
#pragma unroll
  for (int q = 3; q < 16; ++q) {                       // f[3 + F + k] = dcl[k] with compile-time register indices
    const int k = q - 3 - F;
    if (k >= 0 && k < 3) f[q] = k == 0 ? dcl[0] : (k == 1 ? dcl[1] : dcl[2]);
  }
  __hip_bfloat16 hi[16], lo[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    hi[k] = __float2bfloat16(f[k]);
    lo[k] = __float2bfloat16(f[k] - __bfloat162float(hi[k]));
  }
  if (inv_csr) inv_csr[j_] = inv[i];
  uint4* o = reinterpret_cast<uint4*>(out + j_ * 32);
  o[0] = reinterpret_cast<const uint4*>(hi)[0];
  o[1] = reinterpret_cast<const uint4*>(hi)[1];
  o[2] = reinterpret_cast<const uint4*>(lo)[0];
  o[3] = reinterpret_cast<const uint4*>(lo)[1];
}

int tmae_vfe_point_features_bf16x2(const float* points, int row, const int64_t* pc, const int64_t* inverse,
                                   const int32_t* perm, const int32_t* offsets, int64_t n, int64_t m, float rx,
                                   float ry, float rz, float vx, float vy, float vz, float* voxel_mean, void* feats_hl,
                                   int64_t* inverse_csr, void* stream_) {
  (void)hipGetLastError();
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || m < 0 || row < 4 || row - 1 > VFE_MAXF) return TMAE_EARG;
  if (n == 0 || m == 0) return TMAE_OK;
  if (!points || !pc || !inverse || !perm || !offsets || !voxel_mean || !feats_hl) return TMAE_EARG;
  hipLaunchKernelGGL(voxel_mean_kernel, dim3(tmae_cdiv(m, 256)), dim3(256), 0, stream, points, row, perm, offsets, m,
                     voxel_mean);
  hipLaunchKernelGGL(point_feat_split_kernel, dim3(tmae_cdiv(n, 256)), dim3(256), 0, stream, points, row, pc, inverse,
                     voxel_mean, n, rx, ry, rz, vx, vy, vz, (__hip_bfloat16*)feats_hl,
                     inverse_csr ? perm : (const int32_t*)nullptr, inverse_csr);
  return tmae_launch_status();
}

int tmae_vfe_point_features(const float* points, int row, const int64_t* pc, const int64_t* inverse,
                            const int32_t* perm, const int32_t* offsets, int64_t n, int64_t m, float rx, float ry, float rz, float vx,
                            float vy, float vz, float* voxel_mean, float* feats, void* stream_) {
  (void)hipGetLastError();   // drop stale errors of other runtime users: the return code is about OUR launches
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || m < 0 || row < 4 || row - 1 > VFE_MAXF) return TMAE_EARG;
  if (n == 0 || m == 0) return TMAE_OK;
  if (!points || !pc || !inverse || !perm || !offsets || !voxel_mean || !feats) return TMAE_EARG;
  hipLaunchKernelGGL(voxel_mean_kernel, dim3(tmae_cdiv(m, 256)), dim3(256), 0, stream, points, row, perm, offsets, m,
                     voxel_mean);
  hipLaunchKernelGGL(point_feat_kernel, dim3(tmae_cdiv(n, 256)), dim3(256), 0, stream, points, row, pc, inverse,
                     voxel_mean, n, rx, ry, rz, vx, vy, vz, feats);
  return tmae_launch_status();
}

// ------------------------------------------------------------------------------------------------
// segment max (one wavefront per voxel, channels across lanes -> coalesced row reads)
// ------------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void segmax_fwd_kernel(const T* __restrict__ x, int64_t m, int c,
                                                        const int32_t* __restrict__ perm,
                                                        const int32_t* __restrict__ offsets, T* __restrict__ out,
                                                        int32_t* __restrict__ argmax) {
  const int lane = threadIdx.x & 63;
  int64_t v = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (v >= m) return;
  const int lo = offsets[v], hi = offsets[v + 1];
  for (int ch = lane; ch < c; ch += 64) {
    float best = -INFINITY;
    int arg = -1;
    for (int j = lo; j < hi; ++j) {
      const int row = perm ? perm[j] : j;
      const float val = ld_f<T>(x + (int64_t)row * c + ch);
      if (val > best || arg < 0) { best = val; arg = row; }
    }
    if (arg < 0) best = 0.f;
    st_f<T>(out + v * c + ch, best);
    argmax[v * c + ch] = arg;
  }
}

template <class T>
__global__ __launch_bounds__(256) void segmax_bwd_kernel(const T* __restrict__ dout, int64_t n, int c,
                                                        const int64_t* __restrict__ inv,
                                                        const int32_t* __restrict__ argmax, T* __restrict__ dx) {
  int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n * c) return;
  const int64_t p = e / c;
  const int ch = (int)(e - p * c);
  const int64_t v = inv[p];
  const bool hit = argmax[v * c + ch] == (int)p;
  if (hit) dx[e] = dout[v * c + ch];
  else st_f<T>(dx + e, 0.f);
}

// 16-byte variants (c a multiple of 8 with c/8 a power of two <= 64): c/8 adjacent lanes per voxel / point row, 8
// channels per lane.  A voxel holds ~2-3 points, so one wavefront per voxel left most of the wave's width idle.
template <class T> __device__ __forceinline__ float round_to(float v);
template <> __device__ __forceinline__ float round_to<float>(float v) { return v; }
template <> __device__ __forceinline__ float round_to<__hip_bfloat16>(float v) { return __bfloat162float(__float2bfloat16(v)); }

// AFF (round 6): the rows are normalised on the way in -- val = relu?(x * sc + sh) with sc = rstd gamma, sh = beta - mean sc, rounded
// to T like the tensor a separate BatchNorm + ReLU pass would have written -- so the VFE's last norm needs no apply pass of its
// own over every point of the frame (temporal_dyn_vfe.py:110-113: Linear, BatchNorm1d, ReLU, scatter_max).
template <class T, int LPV, bool AFF = false>
__global__ __launch_bounds__(256) void segmax_fwd8_kernel(const T* __restrict__ x, int64_t m,
                                                         const int32_t* __restrict__ perm,
                                                         const int32_t* __restrict__ offsets, T* __restrict__ out,
                                                         int32_t* __restrict__ argmax, const float* __restrict__ mean = nullptr,
                                                         const float* __restrict__ rstd = nullptr,
                                                         const float* __restrict__ gamma = nullptr,
                                                         const float* __restrict__ beta = nullptr, int relu = 0) {
  constexpr int C = LPV * 8, VPW = 64 / LPV;
  const int lane = threadIdx.x & 63, sub = lane / LPV, cl = lane % LPV;
  const int64_t v = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * VPW + sub;
  if (v >= m) return;
  const int lo = offsets[v], hi = offsets[v + 1];
  float best[8];
  int arg[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { best[i] = -INFINITY; arg[i] = -1; }
  float sc[8], sh[8];
  if constexpr (AFF) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int ch = cl * 8 + i;
      sc[i] = rstd[ch] * gamma[ch];
      sh[i] = beta[ch] - mean[ch] * sc[i];
    }
  }
  // four rows per round, all four loads in flight before the first comparison (a voxel holds 2-3 points: one dependent load
  // per point left this kernel waiting for memory latency, not bandwidth); rows past the segment re-read its last row
  for (int j0 = lo; j0 < hi; j0 += 4) {
    int rows[4];
    float val[4][8];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = min(j0 + u, hi - 1);
      rows[u] = perm ? perm[j] : j;                 // perm == NULL: the rows of x are already sorted by voxel
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) load8<T>(x + (int64_t)rows[u] * C + cl * 8, val[u]);
    if constexpr (AFF) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float z = val[u][i] * sc[i] + sh[i];            // the expression of bn_apply_kernel, then its store's rounding
          val[u][i] = round_to<T>(relu ? fmaxf(z, 0.f) : z);
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (j0 + u < hi) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (val[u][i] > best[i] || arg[i] < 0) { best[i] = val[u][i]; arg[i] = rows[u]; }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (arg[i] < 0) best[i] = 0.f;
  store8<T>(out + v * C + cl * 8, best);
  int4* ap = reinterpret_cast<int4*>(argmax + v * C + cl * 8);
  ap[0] = make_int4(arg[0], arg[1], arg[2], arg[3]);
  ap[1] = make_int4(arg[4], arg[5], arg[6], arg[7]);
}

template <class T, int LPV>
__global__ __launch_bounds__(256) void segmax_bwd8_kernel(const T* __restrict__ dout, int64_t n,
                                                         const int64_t* __restrict__ inv,
                                                         const int32_t* __restrict__ argmax, T* __restrict__ dx) {
  constexpr int C = LPV * 8;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t p = e / LPV;
  if (p >= n) return;
  const int cl = (int)(e % LPV);
  const int64_t v = inv[p];
  const int4* ap = reinterpret_cast<const int4*>(argmax + v * C + cl * 8);
  const int4 a0 = ap[0], a1 = ap[1];
  const int arg[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
  float d[8];
  load8<T>(dout + v * C + cl * 8, d);
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (arg[i] != (int)p) d[i] = 0.f;
  store8<T>(dx + p * C + cl * 8, d);
}

static bool segmax_vec_ok(int c, const void* a, const void* b) {
  const int l = c / 8;
  return c % 8 == 0 && (l == 8 || l == 16 || l == 32 || l == 64) && !(((uintptr_t)a | (uintptr_t)b) & 15);
}

#define SEGMAX8(KERNEL, T, GRID, ...)                                                                   \
  do {                                                                                                  \
    switch (c / 8) {                                                                                    \
      case 8: hipLaunchKernelGGL((KERNEL<T, 8>), GRID, dim3(256), 0, stream, __VA_ARGS__); break;       \
      case 16: hipLaunchKernelGGL((KERNEL<T, 16>), GRID, dim3(256), 0, stream, __VA_ARGS__); break;     \
      case 32: hipLaunchKernelGGL((KERNEL<T, 32>), GRID, dim3(256), 0, stream, __VA_ARGS__); break;     \
      default: hipLaunchKernelGGL((KERNEL<T, 64>), GRID, dim3(256), 0, stream, __VA_ARGS__); break;     \
    }                                                                                                   \
  } while (0)

int tmae_segment_max_fwd(const void* x, int dtype, int64_t n, int64_t m, int c, const int32_t* perm,
                         const int32_t* offsets, void* out, int32_t* argmax, void* stream_) {
  (void)hipGetLastError();   // drop stale errors of other runtime users: the return code is about OUR launches
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || m < 0 || c <= 0) return TMAE_EARG;
  if (m == 0) return TMAE_OK;
  if (!x || !offsets || !out || !argmax) return TMAE_EARG;       // perm may be NULL: rows already in CSR order
  if (segmax_vec_ok(c, x, out) && !((uintptr_t)argmax & 15) && (dtype == TMAE_F32 || dtype == TMAE_BF16)) {
    const dim3 g8(tmae_cdiv(m, 4 * (512 / c)));
    if (dtype == TMAE_F32)
      SEGMAX8(segmax_fwd8_kernel, float, g8, (const float*)x, m, perm, offsets, (float*)out, argmax);
    else
      SEGMAX8(segmax_fwd8_kernel, __hip_bfloat16, g8, (const __hip_bfloat16*)x, m, perm, offsets,
              (__hip_bfloat16*)out, argmax);
    return tmae_launch_status();
  }
  dim3 grid(tmae_cdiv(m, 4)), block(256);
  if (dtype == TMAE_F32)
    hipLaunchKernelGGL(segmax_fwd_kernel<float>, grid, block, 0, stream, (const float*)x, m, c, perm, offsets,
                       (float*)out, argmax);
  else if (dtype == TMAE_BF16)
    hipLaunchKernelGGL(segmax_fwd_kernel<__hip_bfloat16>, grid, block, 0, stream, (const __hip_bfloat16*)x, m, c,
                       perm, offsets, (__hip_bfloat16*)out, argmax);
  else
    return TMAE_EDTYPE;
  return tmae_launch_status();
}

// tmae_segment_max_fwd over relu?(BatchNorm(x)) with the norm's statistics given: out / argmax as if the normalised rows had been
// written (in x's dtype) and then reduced.  c in {64, 128, 256} with 16-byte aligned rows; anything else TMAE_EARG (normalise first).
int tmae_segment_max_bn_fwd(const void* x, int dtype, int64_t n, int64_t m, int c, const int32_t* perm, const int32_t* offsets,
                            const float* mean, const float* rstd, const float* gamma, const float* beta, int relu, void* out,
                            int32_t* argmax, void* stream_) {
  (void)hipGetLastError();
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || m < 0 || (c != 64 && c != 128 && c != 256)) return TMAE_EARG;
  if (m == 0) return TMAE_OK;
  if (!x || !offsets || !out || !argmax || !mean || !rstd || !gamma || !beta) return TMAE_EARG;
  if (!segmax_vec_ok(c, x, out) || ((uintptr_t)argmax & 15) || (dtype != TMAE_F32 && dtype != TMAE_BF16)) return TMAE_EARG;
  const dim3 g8(tmae_cdiv(m, 4 * (512 / c))), blk(256);
#define SEGMAX8_AFF(T)                                                                                                  \
  do {                                                                                                                  \
    if (c == 64) hipLaunchKernelGGL((segmax_fwd8_kernel<T, 8, true>), g8, blk, 0, stream, (const T*)x, m, perm, offsets, (T*)out, argmax, mean, rstd, gamma, beta, relu); \
    else if (c == 128) hipLaunchKernelGGL((segmax_fwd8_kernel<T, 16, true>), g8, blk, 0, stream, (const T*)x, m, perm, offsets, (T*)out, argmax, mean, rstd, gamma, beta, relu); \
    else hipLaunchKernelGGL((segmax_fwd8_kernel<T, 32, true>), g8, blk, 0, stream, (const T*)x, m, perm, offsets, (T*)out, argmax, mean, rstd, gamma, beta, relu); \
  } while (0)
  if (dtype == TMAE_F32) SEGMAX8_AFF(float); else SEGMAX8_AFF(__hip_bfloat16);
#undef SEGMAX8_AFF
  return tmae_launch_status();
}

int tmae_segment_max_bwd(const void* dout, int dtype, int64_t n, int64_t m, int c, const int64_t* inverse,
                         const int32_t* argmax, void* dx, void* stream_) {
  (void)hipGetLastError();   // drop stale errors of other runtime users: the return code is about OUR launches
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || m < 0 || c <= 0) return TMAE_EARG;
  if (n == 0) return TMAE_OK;
  if (!dout || !inverse || !argmax || !dx) return TMAE_EARG;
  if (segmax_vec_ok(c, dout, dx) && !((uintptr_t)argmax & 15) && (dtype == TMAE_F32 || dtype == TMAE_BF16)) {
    const dim3 g8(tmae_cdiv(n * (c / 8), 256));
    if (dtype == TMAE_F32)
      SEGMAX8(segmax_bwd8_kernel, float, g8, (const float*)dout, n, inverse, argmax, (float*)dx);
    else
      SEGMAX8(segmax_bwd8_kernel, __hip_bfloat16, g8, (const __hip_bfloat16*)dout, n, inverse, argmax,
              (__hip_bfloat16*)dx);
    return tmae_launch_status();
  }
  dim3 grid(tmae_cdiv(n * c, 256)), block(256);
  if (dtype == TMAE_F32)
    hipLaunchKernelGGL(segmax_bwd_kernel<float>, grid, block, 0, stream, (const float*)dout, n, c, inverse, argmax,
                       (float*)dx);
  else if (dtype == TMAE_BF16)
    hipLaunchKernelGGL(segmax_bwd_kernel<__hip_bfloat16>, grid, block, 0, stream, (const __hip_bfloat16*)dout, n, c,
                       inverse, argmax, (__hip_bfloat16*)dx);
  else
    return TMAE_EDTYPE;
  return tmae_launch_status();
}

// ------------------------------------------------------------------------------------------------
// group_inner_inds + normalised gt points
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void group_points_kernel(const float* __restrict__ pts, int row,
                                                          const int64_t* __restrict__ vc,
                                                          const int32_t* __restrict__ perm,
                                                          const int32_t* __restrict__ offsets, int64_t m, int k,
                                                          float rx, float ry, float rz, float vx, float vy, float vz,
                                                          int64_t* __restrict__ ginds, float* __restrict__ gt) {
  int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= m * k) return;
  const int64_t v = e / k;
  const int t = (int)(e - v * k);
  const int lo = offsets[v], cnt = offsets[v + 1] - lo;
  int64_t idx = -1;
  float gx_ = 0.f, gy_ = 0.f, gz_ = 0.f;
  if (cnt > 0) {
    idx = perm[lo + (t < cnt ? t : t % cnt)];          // cyclic repeat, sst_ops_gpu.cu:30-39
    const float* p = pts + idx * row;
    const int64_t* c = vc + v * 4;
    const float cx = add_rn(mul_rn(add_rn((float)c[3], 0.5f), vx), rx);
    const float cy = add_rn(mul_rn(add_rn((float)c[2], 0.5f), vy), ry);
    const float cz = add_rn(mul_rn(add_rn((float)c[1], 0.5f), vz), rz);
    gx_ = sub_rn(p[1], cx); gy_ = sub_rn(p[2], cy); gz_ = sub_rn(p[3], cz);
  }
  if (ginds) ginds[e] = idx;
  float* g = gt + e * 3;
  g[0] = gx_; g[1] = gy_; g[2] = gz_;
}

int tmae_group_points(const float* points, int row, const int64_t* voxel_coords, const int32_t* perm, const int32_t* offsets,
                      int64_t m, int k, float rx, float ry, float rz, float vx, float vy, float vz,
                      int64_t* group_inds, float* gt, void* stream_) {
  (void)hipGetLastError();   // drop stale errors of other runtime users: the return code is about OUR launches
  hipStream_t stream = (hipStream_t)stream_;
  if (m < 0 || k <= 0 || row < 4) return TMAE_EARG;
  if (m == 0) return TMAE_OK;
  if (!points || !voxel_coords || !perm || !offsets || !gt) return TMAE_EARG;
  hipLaunchKernelGGL(group_points_kernel, dim3(tmae_cdiv(m * k, 256)), dim3(256), 0, stream, points, row, voxel_coords,
                     perm, offsets, m, k, rx, ry, rz, vx, vy, vz, group_inds, gt);
  return tmae_launch_status();
}
