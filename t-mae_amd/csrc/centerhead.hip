// Fine-tune path (BASELINE configs[4]): CenterHead target assignment and the CenterNet focal loss.
//
// Targets (CenterHead.assign_targets / assign_target_of_single_head, center_head.py:107-231 with
// centernet_utils.gaussian_radius / draw_gaussian_to_heatmap, centernet_utils.py:9-74): the reference loops over
// samples and boxes in Python on the CPU and draws every Gaussian with a masked torch.max.  Here one workgroup per
// sample compacts the boxes of the head's classes (stable order = the reference's slot index k), one thread per box
// writes the regression targets and splats its Gaussian with an integer atomicMax (heat values are >= 0, so the
// float order is the int order; max is order independent, hence deterministic).  All index arithmetic follows the
// reference's fp32 operation order without fused multiply-adds, the truncations are bit-exact.
//
// Focal loss (loss_utils.neg_loss_cornernet, loss_utils.py:273-309, on sigmoid(x) clamped to [1e-4, 1-1e-4],
// center_head.py:233-244): one fused pass for the loss sums and one for the gradient w.r.t. the logits, instead of
// ~20 elementwise launches over the [B, C, H, W] map.
#include "common.h"

__device__ __forceinline__ float gaussian_radius_f32(float h, float w, double ov) {
  // centernet_utils.gaussian_radius in torch fp32 arithmetic: python-float factors are applied as fp32 scalars
  const float one_m = (float)(1.0 - ov), one_p = (float)(1.0 + ov);
  const float b1 = add_rn(h, w);
  const float c1 = div_rn(mul_rn(mul_rn(w, h), one_m), one_p);
  const float sq1 = sqrtf(sub_rn(mul_rn(b1, b1), mul_rn(4.0f, c1)));
  const float r1 = div_rn(add_rn(b1, sq1), 2.0f);
  const float b2 = mul_rn(2.0f, add_rn(h, w));
  const float c2 = mul_rn(mul_rn(one_m, w), h);
  const float sq2 = sqrtf(sub_rn(mul_rn(b2, b2), mul_rn(16.0f, c2)));
  const float r2 = div_rn(add_rn(b2, sq2), 2.0f);
  const float four_a3 = (float)(4.0 * (4.0 * ov));              // python: 4 * a3 with a3 = 4 * min_overlap, in double
  const float b3 = mul_rn((float)(-2.0 * ov), add_rn(h, w));
  const float c3 = mul_rn(mul_rn((float)(ov - 1.0), w), h);
  const float sq3 = sqrtf(sub_rn(mul_rn(b3, b3), mul_rn(four_a3, c3)));
  const float r3 = div_rn(add_rn(b3, sq3), 2.0f);
  return fminf(fminf(r1, r2), r3);
}

// one workgroup per sample
__global__ __launch_bounds__(256) void centerhead_targets_kernel(
    const float* __restrict__ gt, int nbox, int ncode /* 8 */, const int32_t* __restrict__ cls_map /* [num_class+1] */,
    int num_class_all, int H, int W, float pcr_x, float pcr_y, float vs_x, float vs_y, float stride, int nmax,
    double overlap, int min_radius, float* __restrict__ heat /* [B, C, H, W], zeroed */, int C,
    float* __restrict__ ret_boxes /* [B, nmax, ncode] zeroed */, int64_t* __restrict__ inds, int64_t* __restrict__ mask) {
  __shared__ int wave_cnt[4];
  __shared__ int base_s;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const float* g = gt + (int64_t)b * nbox * ncode;
  if (tid == 0) base_s = 0;
  __syncthreads();
  for (int i0 = 0; i0 < nbox; i0 += 256) {
    const int i = i0 + tid;
    int local = -1;
    if (i < nbox) {
      const int c = (int)g[(int64_t)i * ncode + ncode - 1];
      if (c >= 0 && c <= num_class_all) local = cls_map[c];          // -1: not a class of this head (0 = padding row)
    }
    const bool keep = local >= 0;
    const unsigned long long bal = __ballot(keep);
    if (lane == 0) wave_cnt[w] = __popcll(bal);
    __syncthreads();
    int k = base_s + __popcll(bal & ((1ull << lane) - 1ull));
    for (int ww = 0; ww < w; ++ww) k += wave_cnt[ww];
    __syncthreads();
    if (tid == 0) base_s += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    if (keep && k < nmax) {
      const float* bx = g + (int64_t)i * ncode;
      const float x = bx[0], y = bx[1], z = bx[2];
      float cx = div_rn(div_rn(sub_rn(x, pcr_x), vs_x), stride);
      float cy = div_rn(div_rn(sub_rn(y, pcr_y), vs_y), stride);
      cx = fminf(fmaxf(cx, 0.f), (float)W - 0.5f);
      cy = fminf(fmaxf(cy, 0.f), (float)H - 0.5f);
      const int cxi = (int)cx, cyi = (int)cy;
      const float dx = div_rn(div_rn(bx[3], vs_x), stride), dy = div_rn(div_rn(bx[4], vs_y), stride);
      if (dx > 0.f && dy > 0.f && cxi >= 0 && cxi <= W && cyi >= 0 && cyi <= H) {
        int r = (int)gaussian_radius_f32(dx, dy, overlap);
        if (r < min_radius) r = min_radius;
        // draw_gaussian_to_heatmap: sigma = (2r+1)/6, value exp(-(dx^2+dy^2)/(2 sigma^2)) in float64, cast to f32
        const double sigma = (double)(2 * r + 1) / 6.0, den = 2.0 * sigma * sigma;
        const int left = min(cxi, r), right = min(W - cxi, r + 1), top = min(cyi, r), bottom = min(H - cyi, r + 1);
        float* hm = heat + ((int64_t)b * C + local) * H * W;
        for (int yy = -top; yy < bottom; ++yy)
          for (int xx = -left; xx < right; ++xx) {
            const float v = (float)exp(-(double)(xx * xx + yy * yy) / den);
            atomicMax(reinterpret_cast<int*>(hm + (int64_t)(cyi + yy) * W + (cxi + xx)), __float_as_int(v));
          }
        const int64_t o = (int64_t)b * nmax + k;
        inds[o] = (int64_t)cyi * W + cxi;
        mask[o] = 1;
        float* rb = ret_boxes + o * ncode;
        rb[0] = sub_rn(cx, (float)cxi);
        rb[1] = sub_rn(cy, (float)cyi);
        rb[2] = z;
        rb[3] = logf(bx[3]); rb[4] = logf(bx[4]); rb[5] = logf(bx[5]);
        rb[6] = cosf(bx[6]);
        rb[7] = sinf(bx[6]);
        for (int e = 8; e < ncode; ++e) rb[e] = bx[e - 1];
      }
    }
    __syncthreads();
  }
}

int tmae_centerhead_targets(const float* gt_boxes, int batch, int nbox, int ncode, const int32_t* cls_map,
                            int num_class_all, int num_class_head, int H, int W, float pcr_x, float pcr_y, float vs_x,
                            float vs_y, float stride, int nmax, double overlap, int min_radius, float* heatmap,
                            float* target_boxes, int64_t* inds, int64_t* mask, void* stream_) {
  (void)hipGetLastError();
  hipStream_t stream = (hipStream_t)stream_;
  if (batch <= 0 || nbox < 0 || ncode < 8 || num_class_head <= 0 || H <= 0 || W <= 0 || nmax <= 0 || !cls_map || !heatmap ||
      !target_boxes || !inds || !mask || vs_x <= 0.f || vs_y <= 0.f || stride <= 0.f)
    return TMAE_EARG;
  if (nbox == 0) return TMAE_OK;
  if (!gt_boxes) return TMAE_EARG;
  hipLaunchKernelGGL(centerhead_targets_kernel, dim3(batch), dim3(256), 0, stream, gt_boxes, nbox, ncode, cls_map,
                     num_class_all, H, W, pcr_x, pcr_y, vs_x, vs_y, stride, nmax, overlap, min_radius, heatmap,
                     num_class_head, target_boxes, inds, mask);
  return tmae_launch_status();
}

// ------------------------------------------------------------------------------------------------
// focal loss: sums of pos_loss, neg_loss and the number of positives; gradient w.r.t. the logits
// ------------------------------------------------------------------------------------------------
#define FL_LO 1e-4f
#define FL_HI (1.0f - 1e-4f)

template <class T>
__global__ __launch_bounds__(256) void focal_fwd_kernel(const T* __restrict__ x, const float* __restrict__ t, int64_t n,
                                                       double* __restrict__ part /* [grid][3] */) {
  __shared__ double red[4][3];
  double pos = 0.0, neg = 0.0, cnt = 0.0;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const float xv = ld_f<T>(x + e), tv = t[e];
    float p = 1.0f / (1.0f + expf(-xv));
    p = fminf(fmaxf(p, FL_LO), FL_HI);
    if (tv == 1.0f) {
      const float q = 1.0f - p;
      pos += (double)(logf(p) * q * q);
      cnt += 1.0;
    } else if (tv < 1.0f) {
      const float w1 = 1.0f - tv, w2 = w1 * w1;
      neg += (double)(logf(1.0f - p) * p * p * (w2 * w2));
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    pos += __shfl_xor(pos, o, 64);
    neg += __shfl_xor(neg, o, 64);
    cnt += __shfl_xor(cnt, o, 64);
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { red[w][0] = pos; red[w][1] = neg; red[w][2] = cnt; }
  __syncthreads();
  if (threadIdx.x < 3)
    part[(int64_t)blockIdx.x * 3 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// out[0] = loss, out[1] = num_pos, out[2] = pos sum, out[3] = neg sum
__global__ __launch_bounds__(256) void focal_finish_kernel(const double* __restrict__ part, int nblocks,
                                                          float* __restrict__ out) {
  __shared__ double red[256][3];
  double a[3] = {0.0, 0.0, 0.0};
  for (int b = threadIdx.x; b < nblocks; b += 256)
    for (int j = 0; j < 3; ++j) a[j] += part[(int64_t)b * 3 + j];
  for (int j = 0; j < 3; ++j) red[threadIdx.x][j] = a[j];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s)
      for (int j = 0; j < 3; ++j) red[threadIdx.x][j] += red[threadIdx.x + s][j];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double pos = red[0][0], neg = red[0][1], cnt = red[0][2];
    out[0] = (float)(cnt == 0.0 ? -neg : -(pos + neg) / cnt);
    out[1] = (float)cnt;
    out[2] = (float)pos;
    out[3] = (float)neg;
  }
}

template <class T>
__global__ __launch_bounds__(256) void focal_bwd_kernel(const T* __restrict__ x, const float* __restrict__ t, int64_t n,
                                                       const float* __restrict__ stats /* out of the forward */,
                                                       const float* __restrict__ gout /* scalar */, T* __restrict__ dx) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const float cnt = stats[1];
  const float scale = -gout[0] / (cnt == 0.f ? 1.0f : cnt);
  const float xv = ld_f<T>(x + e), tv = t[e];
  const float s = 1.0f / (1.0f + expf(-xv));
  float d = 0.f;
  if (s >= FL_LO && s <= FL_HI) {            // torch.clamp passes the gradient inside [min, max] only
    const float p = s, q = 1.0f - p;
    float dLdp = 0.f;
    if (tv == 1.0f) dLdp = q * q / p - 2.0f * logf(p) * q;
    else if (tv < 1.0f) {
      const float w1 = 1.0f - tv, w2 = w1 * w1;
      dLdp = (w2 * w2) * (2.0f * p * logf(q) - p * p / q);
    }
    d = scale * dLdp * p * q;
  }
  st_f<T>(dx + e, d);
}

static int focal_grid(int64_t n) {
  int64_t g = (n + 256 * 16 - 1) / (256 * 16);
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

size_t tmae_focal_loss_workspace(int64_t n) { return (size_t)focal_grid(n) * 3 * 8 + 256; }

int tmae_focal_loss_fwd(const void* logits, int dtype, const float* target, int64_t n, float* out4, void* wsp,
                        size_t ws_bytes, void* stream_) {
  (void)hipGetLastError();
  hipStream_t stream = (hipStream_t)stream_;
  if (n <= 0 || !logits || !target || !out4) return TMAE_EARG;
  if (dtype != TMAE_F32 && dtype != TMAE_BF16) return TMAE_EDTYPE;
  const int nb = focal_grid(n);
  WsCarver ws(wsp, ws_bytes);
  double* part = ws.take<double>((size_t)nb * 3);
  if (!ws.ok) return TMAE_EWS;
  if (dtype == TMAE_F32)
    hipLaunchKernelGGL(focal_fwd_kernel<float>, dim3(nb), dim3(256), 0, stream, (const float*)logits, target, n, part);
  else
    hipLaunchKernelGGL(focal_fwd_kernel<__hip_bfloat16>, dim3(nb), dim3(256), 0, stream, (const __hip_bfloat16*)logits,
                       target, n, part);
  hipLaunchKernelGGL(focal_finish_kernel, dim3(1), dim3(256), 0, stream, part, nb, out4);
  return tmae_launch_status();
}

int tmae_focal_loss_bwd(const void* logits, int dtype, const float* target, int64_t n, const float* stats4,
                        const float* grad_out, void* dlogits, void* stream_) {
  (void)hipGetLastError();
  hipStream_t stream = (hipStream_t)stream_;
  if (n <= 0 || !logits || !target || !stats4 || !grad_out || !dlogits) return TMAE_EARG;
  if (dtype != TMAE_F32 && dtype != TMAE_BF16) return TMAE_EDTYPE;
  const dim3 grid(tmae_cdiv(n, 256)), block(256);
  if (dtype == TMAE_F32)
    hipLaunchKernelGGL(focal_bwd_kernel<float>, grid, block, 0, stream, (const float*)logits, target, n, stats4, grad_out,
                       (float*)dlogits);
  else
    hipLaunchKernelGGL(focal_bwd_kernel<__hip_bfloat16>, grid, block, 0, stream, (const __hip_bfloat16*)logits, target, n,
                       stats4, grad_out, (__hip_bfloat16*)dlogits);
  return tmae_launch_status();
}
