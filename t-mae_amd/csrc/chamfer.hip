// A13: Chamfer distance between the predicted (<=64) and real (<=64) points of every voxel.
// One wavefront per voxel: lane j holds gt point j; the 16 predictions are broadcast one at a time, the
// pred->gt minimum is a wave reduction, the gt->pred minimum a per-lane running minimum.  Brute force,
// everything in registers; HBM traffic is the algorithmic minimum (each point read once).
#include "common.h"
#include <cstdlib>

// minimum over the wavefront, result in every lane: four DPP row rotations (VALU) + two cross-row exchanges
__device__ __forceinline__ float wave_min_all(float v) {
#define TMAE_ROR(n) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + (n), 0xF, 0xF, false))
  v = fminf(v, TMAE_ROR(8));
  v = fminf(v, TMAE_ROR(4));
  v = fminf(v, TMAE_ROR(2));
  v = fminf(v, TMAE_ROR(1));
#undef TMAE_ROR
  v = fminf(v, __shfl_xor(v, 16, 64));
  v = fminf(v, __shfl_xor(v, 32, 64));
  return v;
}

__global__ __launch_bounds__(256) void chamfer_fwd_kernel(const float* __restrict__ pred,
                                                         const float* __restrict__ gt,
                                                         const float* __restrict__ weights, int64_t m, int np,
                                                         int ng, float* __restrict__ per_voxel,
                                                         int8_t* __restrict__ idx_x, int8_t* __restrict__ idx_y) {
  const int lane = threadIdx.x & 63;
  const int64_t v = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (v >= m) return;
  const float w = weights[v];
  if (w == 0.f) {                       // unmasked voxel: contributes nothing (weights = mask, SiamWCA_MAE.py:163)
    if (lane == 0) per_voxel[v] = 0.f;
    return;
  }
  float gx = 0.f, gy = 0.f, gz = 0.f, px = 0.f, py = 0.f, pz = 0.f;
  if (lane < ng) { const float* g = gt + (v * ng + lane) * 3; gx = g[0]; gy = g[1]; gz = g[2]; }
  if (lane < np) { const float* p = pred + (v * np + lane) * 3; px = p[0]; py = p[1]; pz = p[2]; }
  float best = INFINITY;
  int besti = 0;
  float cx_sum = 0.f;
  for (int i = 0; i < np; ++i) {                         // i is wave-uniform: the broadcasts are v_readlane
    const float ax = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(px), i));
    const float ay = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(py), i));
    const float az = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pz), i));
    const float dx = ax - gx, dy = ay - gy, dz = az - gz;
    const float d = (lane < ng) ? (dx * dx + dy * dy + dz * dz) : INFINITY;
    if (d < best) { best = d; besti = i; }
    const float dmin = wave_min_all(d);                  // exact (min of floats); ties -> the lowest gt index
    const int j = __ffsll((unsigned long long)__ballot(d == dmin)) - 1;
    cx_sum += dmin;
    if (lane == i) idx_x[v * np + i] = (int8_t)j;
  }
  if (lane < ng) idx_y[v * ng + lane] = (int8_t)besti;
  const float cy_sum = wave_sum(lane < ng ? best : 0.f);
  if (lane == 0) per_voxel[v] = w * (cx_sum / (float)np + cy_sum / (float)ng);
}

__global__ __launch_bounds__(256) void chamfer_bwd_kernel(const float* __restrict__ pred,
                                                         const float* __restrict__ gt,
                                                         const float* __restrict__ weights,
                                                         const int8_t* __restrict__ idx_x,
                                                         const int8_t* __restrict__ idx_y,
                                                         const float* __restrict__ scale, int64_t m, int np, int ng,
                                                         float* __restrict__ dpred) {
  __shared__ float accs[4][192];
  const int lane = threadIdx.x & 63;
  const int64_t v = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (v >= m) return;
  const float w = weights[v];
  if (w == 0.f) {
    if (lane < np) { float* d = dpred + (v * np + lane) * 3; d[0] = 0.f; d[1] = 0.f; d[2] = 0.f; }
    return;
  }
  float gx = 0.f, gy = 0.f, gz = 0.f, px = 0.f, py = 0.f, pz = 0.f;
  int iy = -1, ix = 0;
  if (lane < ng) { const float* g = gt + (v * ng + lane) * 3; gx = g[0]; gy = g[1]; gz = g[2]; iy = idx_y[v * ng + lane]; }
  if (lane < np) { const float* p = pred + (v * np + lane) * 3; px = p[0]; py = p[1]; pz = p[2]; ix = idx_x[v * np + lane]; }
  // pred -> gt term: 2/np (p_i - g_nn(i))
  const float nx_ = __shfl(gx, ix, 64), ny_ = __shfl(gy, ix, 64), nz_ = __shfl(gz, ix, 64);
  const float a = 2.0f / (float)np, b = 2.0f / (float)ng;
  float ax = a * (px - nx_), ay = a * (py - ny_), az = a * (pz - nz_);
  // gt -> pred term: 2/ng sum_{j: nn(j) = i} (p_i - g_j): every gt lane adds (p_nn - g_j) into its nearest
  // prediction's slot of a wave-private LDS accumulator (fixed tree order is not needed: <= 64 adds of similar
  // magnitude per slot; the adds commute up to fp32 rounding) -- 3 LDS atomics per lane instead of 64 x 4 shuffles
  {
    float* acc = &accs[threadIdx.x >> 6][0];
    for (int e = lane; e < np * 3; e += 64) acc[e] = 0.f;
    __builtin_amdgcn_wave_barrier();
    const int src = iy >= 0 ? iy : 0;                      // shuffles run with every lane active
    const float qx = __shfl(px, src, 64), qy = __shfl(py, src, 64), qz = __shfl(pz, src, 64);
    if (lane < ng && iy >= 0) {
      atomicAdd(&acc[iy * 3 + 0], qx - gx);
      atomicAdd(&acc[iy * 3 + 1], qy - gy);
      atomicAdd(&acc[iy * 3 + 2], qz - gz);
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < np) { ax += b * acc[lane * 3]; ay += b * acc[lane * 3 + 1]; az += b * acc[lane * 3 + 2]; }
  }
  if (lane < np) {
    const float s = scale[0] * w;
    float* d = dpred + (v * np + lane) * 3;
    d[0] = s * ax; d[1] = s * ay; d[2] = s * az;
  }
}

// ------------------------------------------------------------------------------------------------
// The shipped shape -- 16 predictions against 64 real points per voxel (t_mae_ssl.yaml:96-97) -- without the per-
// prediction wave reductions, LDS shuffles, predicated stores and LDS float atomics of the generic kernels above (they
// ran at 1.0 TB/s of their ~360 MB).  Lane l holds gt point l and prediction l & 15 (each 16-lane DPP row holds all 16
// predictions next to its 16 gt points).  Sixteen row rotations bring every gt point of a row past every prediction:
//   pred -> gt: a per-lane running minimum over the row's 16 gt points, the 4 rows combined once at the end;
//   gt -> pred: the distance is rotated back to the gt point's home lane and minimised there.
// An index register travels with every rotation, so nothing depends on the rotation direction.  Exact ties (they
// come from the cyclic repeats that pad gt clouds below 64 points, sst_ops_gpu.cu:30-39) resolve to the first
// candidate met instead of the lowest index: the repeats are bit-identical points, so the loss and every gradient
// are the same numbers either way.
// ------------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ float rot16f(float v) {
  if constexpr (N == 0) return v;
  else return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + N, 0xF, 0xF, false));
}
template <int N>
__device__ __forceinline__ int rot16i(int v) {
  if constexpr (N == 0) return v;
  else return __builtin_amdgcn_update_dpp(0, v, 0x120 + N, 0xF, 0xF, false);
}
// min / sum over the 4 lanes {i, i+16, i+32, i+48}, result in all of them (v_permlane16/32_swap, see attention_mfma.hip)
__device__ __forceinline__ float quad4_min(float v) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fminf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fminf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ int quad4_min_i(int v) {
  auto a = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
  v = min((int)a[0], (int)a[1]);
  auto b = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
  return min((int)b[0], (int)b[1]);
}
__device__ __forceinline__ float quad4_sum(float v) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float row16_sum_f(float v) {
  v += rot16f<8>(v); v += rot16f<4>(v); v += rot16f<2>(v); v += rot16f<1>(v);
  return v;
}

__global__ __launch_bounds__(256) void chamfer_fwd_16x64_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                               const float* __restrict__ weights, int64_t m,
                                                               float* __restrict__ per_voxel, int8_t* __restrict__ idx_x,
                                                               int8_t* __restrict__ idx_y) {
  const int lane = threadIdx.x & 63, i = lane & 15;
  const int64_t v = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (v >= m) return;
  const float w = weights[v];
  if (w == 0.f) {                       // unmasked voxel: contributes nothing (weights = mask, SiamWCA_MAE.py:163)
    if (lane == 0) per_voxel[v] = 0.f;
    return;
  }
  const float* g = gt + (v * 64 + lane) * 3;
  const float* p = pred + (v * 16 + i) * 3;
  const float gx = g[0], gy = g[1], gz = g[2], px = p[0], py = p[1], pz = p[2];
  float bx = INFINITY, by = INFINITY;                // pred -> gt minimum over this row's gts / gt -> pred minimum
  int bxi = 0, byi = 0;
#define CH_STEP(R)                                                                                    \
  {                                                                                                   \
    const float dx = rot16f<R>(gx) - px, dy = rot16f<R>(gy) - py, dz = rot16f<R>(gz) - pz;            \
    const int gi = rot16i<R>(lane);                       /* the gt point this lane sees in step R */  \
    const float d = dx * dx + dy * dy + dz * dz;                                                      \
    if (d < bx) { bx = d; bxi = gi; }                                                                 \
    const float db = rot16f<(16 - R) & 15>(d);            /* back to the gt point's home lane */       \
    const int pb = rot16i<(16 - R) & 15>(i);              /* ... with the prediction it was measured against */ \
    if (db < by) { by = db; byi = pb; }                                                               \
  }
  CH_STEP(0) CH_STEP(1) CH_STEP(2) CH_STEP(3) CH_STEP(4) CH_STEP(5) CH_STEP(6) CH_STEP(7)
  CH_STEP(8) CH_STEP(9) CH_STEP(10) CH_STEP(11) CH_STEP(12) CH_STEP(13) CH_STEP(14) CH_STEP(15)
#undef CH_STEP
  // pred -> gt: combine the four rows (disjoint gt ranges); among equal minima the lowest gt index
  const float cmin = quad4_min(bx);
  const int cidx = quad4_min_i(bx == cmin ? bxi : 127);
  const float cx_sum = row16_sum_f(cmin);            // over the 16 predictions (every row holds the same values)
  if (lane < 16) idx_x[v * 16 + lane] = (int8_t)cidx;
  idx_y[v * 64 + lane] = (int8_t)byi;
  const float cy_sum = wave_sum(by);
  if (lane == 0) per_voxel[v] = w * (cx_sum / 16.0f + cy_sum / 64.0f);
}

__global__ __launch_bounds__(256) void chamfer_bwd_16x64_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                               const float* __restrict__ weights,
                                                               const int8_t* __restrict__ idx_x,
                                                               const int8_t* __restrict__ idx_y,
                                                               const float* __restrict__ scale, int64_t m,
                                                               float* __restrict__ dpred) {
  const int lane = threadIdx.x & 63, i = lane & 15;
  const int64_t v = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (v >= m) return;
  const float w = weights[v];
  if (w == 0.f) {
    if (lane < 16) { float* d = dpred + (v * 16 + lane) * 3; d[0] = 0.f; d[1] = 0.f; d[2] = 0.f; }
    return;
  }
  const float* g = gt + (v * 64 + lane) * 3;
  const float* p = pred + (v * 16 + i) * 3;
  const float gx = g[0], gy = g[1], gz = g[2], px = p[0], py = p[1], pz = p[2];
  const int iy = idx_y[v * 64 + lane], ix = idx_x[v * 16 + i];
  // gt -> pred term: 2/64 sum_{j: nn(j) = i} (p_i - g_j): every gt point of the row passes by with its nearest-prediction
  // index; the four rows' partial sums meet at the end (fixed order: no atomics)
  float ax = 0.f, ay = 0.f, az = 0.f;
#define CB_STEP(R)                                                                                    \
  {                                                                                                   \
    const float m_ = rot16i<R>(iy) == i ? 1.0f : 0.0f;                                                \
    ax += m_ * (px - rot16f<R>(gx)); ay += m_ * (py - rot16f<R>(gy)); az += m_ * (pz - rot16f<R>(gz)); \
  }
  CB_STEP(0) CB_STEP(1) CB_STEP(2) CB_STEP(3) CB_STEP(4) CB_STEP(5) CB_STEP(6) CB_STEP(7)
  CB_STEP(8) CB_STEP(9) CB_STEP(10) CB_STEP(11) CB_STEP(12) CB_STEP(13) CB_STEP(14) CB_STEP(15)
#undef CB_STEP
  ax = quad4_sum(ax); ay = quad4_sum(ay); az = quad4_sum(az);
  // pred -> gt term: 2/16 (p_i - g_nn(i))
  const float nx_ = __shfl(gx, ix, 64), ny_ = __shfl(gy, ix, 64), nz_ = __shfl(gz, ix, 64);
  const float a = 2.0f / 16.0f, b = 2.0f / 64.0f;
  if (lane < 16) {
    const float s = scale[0] * w;
    float* d = dpred + (v * 16 + lane) * 3;
    d[0] = s * (a * (px - nx_) + b * ax);
    d[1] = s * (a * (py - ny_) + b * ay);
    d[2] = s * (a * (pz - nz_) + b * az);
  }
}

int tmae_chamfer_fwd(const float* pred, const float* gt, const float* weights, int64_t m, int np, int ng,
                     float* per_voxel, int8_t* idx_x, int8_t* idx_y, void* stream_) {
  (void)hipGetLastError();   // drop stale errors of other runtime users: the return code is about OUR launches
  hipStream_t stream = (hipStream_t)stream_;
  if (m < 0 || np <= 0 || np > 64 || ng <= 0 || ng > 64) return TMAE_EARG;
  if (m == 0) return TMAE_OK;
  if (!pred || !gt || !weights || !per_voxel || !idx_x || !idx_y) return TMAE_EARG;
  static const bool generic = TMAE_AB_INT("TMAE_CHAMFER_GENERIC", 0) != 0;
  if (np == 16 && ng == 64 && !generic)
    hipLaunchKernelGGL(chamfer_fwd_16x64_kernel, dim3(tmae_cdiv(m, 4)), dim3(256), 0, stream, pred, gt, weights, m,
                       per_voxel, idx_x, idx_y);
  else
    hipLaunchKernelGGL(chamfer_fwd_kernel, dim3(tmae_cdiv(m, 4)), dim3(256), 0, stream, pred, gt, weights, m, np, ng,
                       per_voxel, idx_x, idx_y);
  return tmae_launch_status();
}

int tmae_chamfer_bwd(const float* pred, const float* gt, const float* weights, const int8_t* idx_x,
                     const int8_t* idx_y, const float* scale, int64_t m, int np, int ng, float* dpred,
                     void* stream_) {
  (void)hipGetLastError();   // drop stale errors of other runtime users: the return code is about OUR launches
  hipStream_t stream = (hipStream_t)stream_;
  if (m < 0 || np <= 0 || np > 64 || ng <= 0 || ng > 64) return TMAE_EARG;
  if (m == 0) return TMAE_OK;
  if (!pred || !gt || !weights || !idx_x || !idx_y || !scale || !dpred) return TMAE_EARG;
  static const bool generic = TMAE_AB_INT("TMAE_CHAMFER_GENERIC", 0) != 0;
  if (np == 16 && ng == 64 && !generic)
    hipLaunchKernelGGL(chamfer_bwd_16x64_kernel, dim3(tmae_cdiv(m, 4)), dim3(256), 0, stream, pred, gt, weights, idx_x,
                       idx_y, scale, m, dpred);
  else
    hipLaunchKernelGGL(chamfer_bwd_kernel, dim3(tmae_cdiv(m, 4)), dim3(256), 0, stream, pred, gt, weights, idx_x, idx_y,
                       scale, m, np, ng, dpred);
  return tmae_launch_status();
}
