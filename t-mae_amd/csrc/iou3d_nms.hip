// Rotated-box BEV overlap / IoU and NMS for the detection head (pcdet/ops/iou3d_nms: iou3d_nms_kernel.cu:113-330,
// iou3d_nms.cpp:104-150, used by model_nms_utils.class_agnostic_nms and generate_recall_record).
// Boxes are rows (x, y, z, dx, dy, dz, heading).  The intersection of two rotated rectangles is obtained by clipping
// rectangle A against the four edges of B (convex clipping, <= 8 vertices, all in registers); the reference collects
// corner-in-box points and edge intersections and sorts them around their centroid -- the same polygon.
// NMS: a 64 x 64 tile of the upper triangle per workgroup writes one 64-bit suppression word per (row, column block),
// exactly the reference's mask layout; the sequential greedy pass over the rows runs on the DEVICE in one wavefront
// (the reference copies the mask to the host and loops there), so no synchronisation is needed.
#include "common.h"

struct P2 { float x, y; };

__device__ __forceinline__ void rect_corners(const float* b, P2* c) {
  const float cs = cosf(b[6]), sn = sinf(b[6]);
  const float hx = b[3] * 0.5f, hy = b[4] * 0.5f;
  const float px[4] = {-hx, hx, hx, -hx}, py[4] = {-hy, -hy, hy, hy};          // counter-clockwise
#pragma unroll
  for (int i = 0; i < 4; ++i) { c[i].x = b[0] + px[i] * cs - py[i] * sn; c[i].y = b[1] + px[i] * sn + py[i] * cs; }
}

__device__ float box_overlap_bev(const float* a, const float* b) {
  P2 poly[10], tmp[10], cb[4];
  rect_corners(a, poly);
  rect_corners(b, cb);
  int n = 4;
#pragma unroll 1
  for (int e = 0; e < 4 && n > 0; ++e) {
    const P2 ea = cb[e], eb = cb[(e + 1) & 3];
    const float ex = eb.x - ea.x, ey = eb.y - ea.y;
    int m = 0;
    for (int i = 0; i < n; ++i) {
      const P2 p = poly[i], q = poly[i + 1 == n ? 0 : i + 1];
      const float sp = ex * (p.y - ea.y) - ey * (p.x - ea.x);
      const float sq = ex * (q.y - ea.y) - ey * (q.x - ea.x);
      if (sp >= 0.f) tmp[m++] = p;
      if ((sp >= 0.f) != (sq >= 0.f)) {
        const float t = sp / (sp - sq);
        tmp[m].x = p.x + t * (q.x - p.x);
        tmp[m].y = p.y + t * (q.y - p.y);
        ++m;
      }
    }
    n = m;
    for (int i = 0; i < n; ++i) poly[i] = tmp[i];
  }
  if (n < 3) return 0.f;
  float area = 0.f;
  for (int i = 0; i < n; ++i) {
    const P2 p = poly[i], q = poly[i + 1 == n ? 0 : i + 1];
    area += (p.x - poly[0].x) * (q.y - poly[0].y) - (q.x - poly[0].x) * (p.y - poly[0].y);   // relative to a vertex: less cancellation
  }
  return fabsf(area) * 0.5f;
}

__device__ __forceinline__ float iou_bev_dev(const float* a, const float* b) {
  const float sa = a[3] * a[4], sb = b[3] * b[4];
  const float o = box_overlap_bev(a, b);
  return o / fmaxf(sa + sb - o, 1e-8f);
}

// mode 0: BEV overlap area, 1: BEV IoU, 2: 3-D IoU (overlap x height overlap / union volume, iou3d_nms_utils.py:48-81)
__global__ __launch_bounds__(256) void boxes_pairwise_kernel(const float* __restrict__ A, int na, const float* __restrict__ B,
                                                            int nb, int mode, float* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool paired = (mode & 4) != 0;             // mode | 4: out[i] = f(A[i], B[i]) (na == nb), e.g. the IoU head's targets
  mode &= 3;
  if (e >= (paired ? (int64_t)na : (int64_t)na * nb)) return;
  const float* a = A + (paired ? e : e / nb) * 7;
  const float* b = B + (paired ? e : e % nb) * 7;
  float v;
  if (mode == 0) v = box_overlap_bev(a, b);
  else if (mode == 1) v = iou_bev_dev(a, b);
  else {
    const float h = fmaxf(fminf(a[2] + a[5] * 0.5f, b[2] + b[5] * 0.5f) - fmaxf(a[2] - a[5] * 0.5f, b[2] - b[5] * 0.5f), 0.f);
    const float o = box_overlap_bev(a, b) * h;
    v = o / fmaxf(a[3] * a[4] * a[5] + b[3] * b[4] * b[5] - o, 1e-6f);
  }
  out[e] = v;
}

// mask[row, cb] bit i = IoU(box row, box cb*64 + i) > thresh, only for columns after the row (score order)
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ boxes, int n, float thresh,
                                                     unsigned long long* __restrict__ mask, int col_blocks) {
  __shared__ float cbox[64 * 7];
  const int rb = blockIdx.y, cb = blockIdx.x;
  if (cb < rb) {                                    // lower triangle: never read with a set bit, but keep it defined
    const int row = rb * 64 + threadIdx.x;
    if (row < n) mask[(int64_t)row * col_blocks + cb] = 0ull;
    return;
  }
  const int ncol = min(n - cb * 64, 64);
  if ((int)threadIdx.x < ncol)
    for (int j = 0; j < 7; ++j) cbox[threadIdx.x * 7 + j] = boxes[(int64_t)(cb * 64 + threadIdx.x) * 7 + j];
  __syncthreads();
  const int row = rb * 64 + threadIdx.x;
  if (row >= n) return;
  float rbx[7];
  for (int j = 0; j < 7; ++j) rbx[j] = boxes[(int64_t)row * 7 + j];
  unsigned long long t = 0ull;
  const int start = (rb == cb) ? threadIdx.x + 1 : 0;
  for (int i = start; i < ncol; ++i)
    if (iou_bev_dev(rbx, cbox + i * 7) > thresh) t |= 1ull << i;
  mask[(int64_t)row * col_blocks + cb] = t;
}

// greedy pass in score order, one wavefront: lane l owns the removal words l, l+64, ...
__global__ __launch_bounds__(64) void nms_select_kernel(const unsigned long long* __restrict__ mask, int n, int col_blocks,
                                                       int64_t* __restrict__ keep, int32_t* __restrict__ num_keep) {
  extern __shared__ unsigned long long removed[];
  const int lane = threadIdx.x;
  for (int w = lane; w < col_blocks; w += 64) removed[w] = 0ull;
  __syncthreads();
  int cnt = 0;
  for (int i = 0; i < n; ++i) {
    const bool dead = (removed[i >> 6] >> (i & 63)) & 1ull;         // same word for every lane: LDS broadcast
    if (!dead) {
      if (lane == 0) keep[cnt] = i;
      ++cnt;
      for (int w = lane; w < col_blocks; w += 64) removed[w] |= mask[(int64_t)i * col_blocks + w];
    }
    __syncthreads();
  }
  if (lane == 0) *num_keep = cnt;
}

int tmae_boxes_pairwise(const float* boxes_a, int na, const float* boxes_b, int nb, int mode, float* out, void* stream_) {
  (void)hipGetLastError();
  if (na < 0 || nb < 0 || mode < 0 || (mode & 3) > 2 || mode > 6 || ((mode & 4) && na != nb)) return TMAE_EARG;
  if (na == 0 || nb == 0) return TMAE_OK;
  if (!boxes_a || !boxes_b || !out) return TMAE_EARG;
  hipLaunchKernelGGL(boxes_pairwise_kernel, dim3(tmae_cdiv((int64_t)na * nb, 256)), dim3(256), 0, (hipStream_t)stream_,
                     boxes_a, na, boxes_b, nb, mode, out);
  return tmae_launch_status();
}

size_t tmae_nms_bev_workspace(int n) { return (size_t)n * ((n + 63) / 64) * 8 + 256; }

int tmae_nms_bev(const float* boxes_sorted, int n, float thresh, int64_t* keep, int32_t* num_keep, void* wsp,
                 size_t ws_bytes, void* stream_) {
  (void)hipGetLastError();
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || !num_keep) return TMAE_EARG;
  if (n == 0) return (int)hipMemsetAsync(num_keep, 0, 4, stream);
  if (!boxes_sorted || !keep || n > 65536) return TMAE_EARG;
  const int cbk = (n + 63) / 64;
  WsCarver ws(wsp, ws_bytes);
  unsigned long long* mask = ws.take<unsigned long long>((size_t)n * cbk);
  if (!ws.ok) return TMAE_EWS;
  hipLaunchKernelGGL(nms_mask_kernel, dim3(cbk, cbk), dim3(64), 0, stream, boxes_sorted, n, thresh, mask, cbk);
  hipLaunchKernelGGL(nms_select_kernel, dim3(1), dim3(64), (size_t)cbk * 8, stream, mask, n, cbk, keep, num_keep);
  return tmae_launch_status();
}
