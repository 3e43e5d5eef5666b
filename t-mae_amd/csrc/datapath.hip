// Two-frame data path in front of the step, one launch group per frame of a sample:
// ego-point removal (once_temporal_dataset.remove_ego_points, once_temporal_dataset.py:43-45), alignment of the
// previous frame into the current one (once_utils.convert_prv_frame_to_cur, once_utils.py:4-29: float64), the
// joint world augmentation (DataAugmentor.random_world_flip / rotation / scaling, data_augmentor.py:55-142; the
// rotation is an fp32 product with (cos, sin, -sin, cos) as in common_utils.rotate_points_along_z), the range crop on
// x / y (common_utils.mask_points_by_range) and the collate's sample-index column (dataset.py:203-207), with an
// order-preserving compaction.  The reference does all of it in numpy on the dataloader workers.
#include "common.h"

struct FrameXform {
  double r1[9], t1[3];     // step 1: p_global = R1 p + t1            (has1)
  double m2[12];           // step 2: p_cur = M2[:3,:3] p_global + M2[:3,3] (has2; M2 = inverse of the current pose)
  int has1, has2;
  float ego_radius, cosa, sina, scale, xmin, ymin, xmax, ymax;
  int flip_x, flip_y;
};

// Removal of the scene points inside the (enlarged) boxes that gt_sampling pastes (database_sampler.py:201-205 ->
// box_utils.remove_points_in_boxes3d -> roiaware_pool3d.cpp:119-140 check_pt_in_box3d_cpu): the test runs on the ALIGNED,
// not yet augmented point in fp32 -- |z - cz| > dz / 2 rejects; (x - cx, y - cy) rotated by -heading in fp32 with
// (float)cos / (float)sin of the double angle; inside iff |lx| < dx / 2 + 1e-2f and |ly| < dy / 2 + 1e-2f, the thresholds in double.
// rb [nb, 8] doubles prepared by the host: cx, cy, cz, cosa, sina (float values), dz / 2, dx / 2 + margin, dy / 2 + margin.
#define FRAME_MAX_RB 64
__global__ __launch_bounds__(256) void frame_xform_kernel(const float* __restrict__ in, int row, int64_t n, FrameXform f,
                                                         float* __restrict__ xyz /* [n,3] */, int32_t* __restrict__ flag,
                                                         const double* __restrict__ rb, int nb) {
  __shared__ double srb[FRAME_MAX_RB * 8];
  for (int t = threadIdx.x; t < nb * 8; t += 256) srb[t] = rb[t];
  if (nb > 0) __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float* p = in + i * row;
  const float x0 = p[0], y0 = p[1], z0 = p[2];
  const bool ego = fabsf(x0) < f.ego_radius && fabsf(y0) < f.ego_radius;
  double gx = x0, gy = y0, gz = z0;
  if (f.has1) {
    const double ax = gx, ay = gy, az = gz;
    gx = ax * f.r1[0] + ay * f.r1[1] + az * f.r1[2] + f.t1[0];
    gy = ax * f.r1[3] + ay * f.r1[4] + az * f.r1[5] + f.t1[1];
    gz = ax * f.r1[6] + ay * f.r1[7] + az * f.r1[8] + f.t1[2];
  }
  if (f.has2) {
    const double ax = gx, ay = gy, az = gz;
    gx = ax * f.m2[0] + ay * f.m2[1] + az * f.m2[2] + f.m2[3];
    gy = ax * f.m2[4] + ay * f.m2[5] + az * f.m2[6] + f.m2[7];
    gz = ax * f.m2[8] + ay * f.m2[9] + az * f.m2[10] + f.m2[11];
  }
  bool boxed = false;
  {
    const float px = (float)gx, py = (float)gy, pz = (float)gz;
    for (int b = 0; b < nb; ++b) {
      const double* q = srb + b * 8;
      if ((double)fabsf(sub_rn(pz, (float)q[2])) > q[5]) continue;
      const float sx = sub_rn(px, (float)q[0]), sy = sub_rn(py, (float)q[1]), ca = (float)q[3], sa = (float)q[4];
      const float lx = add_rn(mul_rn(sx, ca), mul_rn(sy, -sa)), ly = add_rn(mul_rn(sx, sa), mul_rn(sy, ca));
      boxed |= ((double)fabsf(lx) < q[6]) & ((double)fabsf(ly) < q[7]);
    }
  }
  if (f.flip_x) gy = -gy;                       // 'x': mirror about the x axis, data_augmentor.py:69-70
  if (f.flip_y) gx = -gx;
  const float x = (float)gx, y = (float)gy, z = (float)gz;
  // [x y z] . [[c, s, 0], [-s, c, 0], [0, 0, 1]] in fp32
  float xr = add_rn(mul_rn(x, f.cosa), mul_rn(y, -f.sina));
  float yr = add_rn(mul_rn(x, f.sina), mul_rn(y, f.cosa));
  xr = mul_rn(xr, f.scale);
  yr = mul_rn(yr, f.scale);
  const float zr = mul_rn(z, f.scale);
  xyz[i * 3] = xr; xyz[i * 3 + 1] = yr; xyz[i * 3 + 2] = zr;
  flag[i] = (!ego && !boxed && xr >= f.xmin && xr <= f.xmax && yr >= f.ymin && yr <= f.ymax) ? 1 : 0;
}

__global__ __launch_bounds__(256) void frame_emit_kernel(const float* __restrict__ in, int row, int64_t n,
                                                        const float* __restrict__ xyz, const int32_t* __restrict__ flag,
                                                        const int32_t* __restrict__ pos, float batch_idx,
                                                        float* __restrict__ out /* [*, row + 1] */) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n || !flag[i]) return;
  float* o = out + (int64_t)pos[i] * (row + 1);
  o[0] = batch_idx;
  o[1] = xyz[i * 3]; o[2] = xyz[i * 3 + 1]; o[3] = xyz[i * 3 + 2];
  for (int c = 3; c < row; ++c) o[c + 1] = in[i * row + c];
}

size_t tmae_frame_prepare_workspace(int64_t n) {
  return tmae_align((size_t)n * 12) + 2 * tmae_align((size_t)n * 4) + tmae_scan_i32_workspace(n) + 1024;
}

int tmae_frame_prepare(const float* points, int row, int64_t n, const double* r1t1 /* 12 or null */,
                       const double* m2 /* 12 or null */, float ego_radius, int flip_x, int flip_y, float cosa, float sina,
                       float scale, float xmin, float ymin, float xmax, float ymax, int batch_idx, float* out,
                       int32_t* count, void* wsp, size_t ws_bytes, void* stream_) {
  return tmae_frame_prepare_boxes(points, row, n, r1t1, m2, ego_radius, flip_x, flip_y, cosa, sina, scale, xmin, ymin, xmax,
                                  ymax, batch_idx, nullptr, 0, out, count, wsp, ws_bytes, stream_);
}

int tmae_frame_prepare_boxes(const float* points, int row, int64_t n, const double* r1t1, const double* m2, float ego_radius,
                             int flip_x, int flip_y, float cosa, float sina, float scale, float xmin, float ymin, float xmax,
                             float ymax, int batch_idx, const double* remove_boxes /* device [n_boxes, 8] or null */,
                             int n_boxes, float* out, int32_t* count, void* wsp, size_t ws_bytes, void* stream_) {
  (void)hipGetLastError();
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || row < 4 || !count || n_boxes < 0 || n_boxes > FRAME_MAX_RB || (n_boxes > 0 && !remove_boxes)) return TMAE_EARG;
  if (n == 0) return (int)hipMemsetAsync(count, 0, 4, stream);
  if (!points || !out) return TMAE_EARG;
  FrameXform f;
  f.has1 = r1t1 != nullptr;
  f.has2 = m2 != nullptr;
  for (int i = 0; i < 9; ++i) f.r1[i] = r1t1 ? r1t1[i] : 0.0;
  for (int i = 0; i < 3; ++i) f.t1[i] = r1t1 ? r1t1[9 + i] : 0.0;
  for (int i = 0; i < 12; ++i) f.m2[i] = m2 ? m2[i] : 0.0;
  f.ego_radius = ego_radius; f.cosa = cosa; f.sina = sina; f.scale = scale;
  f.xmin = xmin; f.ymin = ymin; f.xmax = xmax; f.ymax = ymax; f.flip_x = flip_x; f.flip_y = flip_y;
  WsCarver ws(wsp, ws_bytes);
  float* xyz = ws.take<float>((size_t)n * 3);
  int32_t* flag = ws.take<int32_t>((size_t)n);
  int32_t* pos = ws.take<int32_t>((size_t)n);
  const size_t sb = tmae_scan_i32_workspace(n);
  void* scanws = ws.take<char>(sb);
  if (!ws.ok) return TMAE_EWS;
  const dim3 grid(tmae_cdiv(n, 256)), block(256);
  hipLaunchKernelGGL(frame_xform_kernel, grid, block, 0, stream, points, row, n, f, xyz, flag, remove_boxes, n_boxes);
  const int r = tmae_scan_i32(flag, pos, n, count, scanws, sb, stream);
  if (r) return r;
  hipLaunchKernelGGL(frame_emit_kernel, grid, block, 0, stream, points, row, n, xyz, flag, pos, (float)batch_idx, out);
  return tmae_launch_status();
}
