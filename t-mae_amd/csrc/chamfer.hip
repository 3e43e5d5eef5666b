// A13: Chamfer distance between the predicted (<=64) and real (<=64) points of every voxel.
// One wavefront per voxel: lane j holds gt point j; the 16 predictions are broadcast one at a time, the
// pred->gt minimum is a wave reduction, the gt->pred minimum a per-lane running minimum.  Brute force,
// everything in registers; HBM traffic is the algorithmic minimum (each point read once).
#include "common.h"

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

int tmae_chamfer_fwd(const float* pred, const float* gt, const float* weights, int64_t m, int np, int ng,
                     float* per_voxel, int8_t* idx_x, int8_t* idx_y, void* stream_) {
  (void)hipGetLastError();   // drop stale errors of other runtime users: the return code is about OUR launches
  hipStream_t stream = (hipStream_t)stream_;
  if (m < 0 || np <= 0 || np > 64 || ng <= 0 || ng > 64) return TMAE_EARG;
  if (m == 0) return TMAE_OK;
  if (!pred || !gt || !weights || !per_voxel || !idx_x || !idx_y) return TMAE_EARG;
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
  hipLaunchKernelGGL(chamfer_bwd_kernel, dim3(tmae_cdiv(m, 4)), dim3(256), 0, stream, pred, gt, weights, idx_x, idx_y,
                     scale, m, np, ng, dpred);
  return tmae_launch_status();
}

int tmae_abi_version(void) { return 4; }
