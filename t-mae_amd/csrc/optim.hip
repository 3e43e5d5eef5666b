// A15: the optimizer step of the T-MAE recipe -- decoupled weight decay p *= 1 - wd*lr (fastai_optim.py:139-150) followed
// by Adam (fastai_optim.py:135-152 -> torch.optim.Adam.step: exp_avg.lerp_(g, 1 - b1), exp_avg_sq = b2 * exp_avg_sq +
// (1 - b2) g*g, p -= lr / bc1 * exp_avg / (sqrt(exp_avg_sq) / sqrt(bc2) + eps)) -- for ALL parameter tensors of the
// optimizer in ONE launch.  torch's fused Adam plus the foreach multiply take ~36 multi-tensor launches per step for the
// 266 tensors / 9.1 M parameters of this model (0.85 ms, latency-bound); the arithmetic is 28 bytes per parameter.
// A block owns one 4096-element chunk of one tensor (host-built tables: the tensor of every chunk); 16-byte accesses
// when all four pointers of the tensor are 16-byte aligned (gradients may be views into a DDP bucket).
#include "common.h"

struct AdamEntry {           // 8 x int64 in the host-built table
  float* p;                  // parameter (fp32), updated in place
  const float* g;            // gradient (fp32) or NULL: decay only (a parameter that received no gradient)
  float* m;                  // exp_avg
  float* v;                  // exp_avg_sq
  float* step;               // torch.optim.Adam's per-parameter step tensor (device, 1 float) or NULL
  int64_t numel_chunk0;      // numel | first chunk << 40
  __hip_bfloat16* copy;      // bf16 copy of the updated parameter (what the autocast path reads) or NULL
  int64_t step_no;           // THIS tensor's step number after the increment (>= 1 where g != NULL): tensors that missed a
                             // gradient in some earlier step lag behind the others and keep their own bias corrections
};

#define ADAM_CHUNK 4096

__global__ __launch_bounds__(256) void adam_step_kernel(const AdamEntry* __restrict__ tab, const int* __restrict__ chunk_tensor,
                                                       float decay, float w1, float b2, float w2, double lr,
                                                       double beta1, double beta2, float eps) {
  const AdamEntry e = tab[chunk_tensor[blockIdx.x]];
  // the scalar factors in double, as torch.optim.Adam computes them on the host (python floats); per ENTRY: one thread,
  // through LDS (a double pow is a few hundred instructions)
  __shared__ float s_fac[2];
  if (threadIdx.x == 0 && e.g) {
    const double bc1 = 1.0 - pow(beta1, (double)e.step_no), bc2 = 1.0 - pow(beta2, (double)e.step_no);
    s_fac[0] = (float)(lr / bc1);
    s_fac[1] = (float)(1.0 / sqrt(bc2));
  }
  __syncthreads();
  const float step_size = s_fac[0], inv_bc2_sqrt = s_fac[1], step_value = (float)e.step_no;
  const int64_t numel = e.numel_chunk0 & (((int64_t)1 << 40) - 1), chunk0 = e.numel_chunk0 >> 40;
  const int64_t base = ((int64_t)blockIdx.x - chunk0) * ADAM_CHUNK;
  const int64_t left = numel - base;
  const int n = left < ADAM_CHUNK ? (int)left : ADAM_CHUNK;
  if (base == 0 && threadIdx.x == 0 && e.step && e.g) *e.step = step_value;
  float* p = e.p + base;
  __hip_bfloat16* cp = e.copy ? e.copy + base : nullptr;
  if (!e.g) {
    for (int k = threadIdx.x; k < n; k += 256) {
      const float pp = p[k] * decay;
      p[k] = pp;
      if (cp) cp[k] = __float2bfloat16(pp);
    }
    return;
  }
  const float* g = e.g + base;
  float* m = e.m + base;
  float* v = e.v + base;
  auto upd = [&](float& pp, float gg, float& mm, float& vv) {
    pp *= decay;
    mm = mm + (gg - mm) * w1;                 // lerp(exp_avg, grad, 1 - beta1)
    vv = b2 * vv + w2 * gg * gg;
    pp -= step_size * mm / (sqrtf(vv) * inv_bc2_sqrt + eps);
  };
  const bool al = !(((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) && !((uintptr_t)cp & 7);
  if (al && n == ADAM_CHUNK) {
#pragma unroll
    for (int r = 0; r < ADAM_CHUNK / 1024; ++r) {
      const int k = (r * 256 + threadIdx.x) * 4;
      float4 pp = *reinterpret_cast<float4*>(p + k), mm = *reinterpret_cast<float4*>(m + k), vv = *reinterpret_cast<float4*>(v + k);
      const float4 gg = *reinterpret_cast<const float4*>(g + k);
      upd(pp.x, gg.x, mm.x, vv.x); upd(pp.y, gg.y, mm.y, vv.y); upd(pp.z, gg.z, mm.z, vv.z); upd(pp.w, gg.w, mm.w, vv.w);
      *reinterpret_cast<float4*>(p + k) = pp; *reinterpret_cast<float4*>(m + k) = mm; *reinterpret_cast<float4*>(v + k) = vv;
      if (cp) {
        __hip_bfloat16 t4[4] = {__float2bfloat16(pp.x), __float2bfloat16(pp.y), __float2bfloat16(pp.z), __float2bfloat16(pp.w)};
        *reinterpret_cast<uint2*>(cp + k) = *reinterpret_cast<const uint2*>(t4);
      }
    }
  } else {
    for (int k = threadIdx.x; k < n; k += 256) {
      float pp = p[k], mm = m[k], vv = v[k];
      upd(pp, g[k], mm, vv);
      p[k] = pp; m[k] = mm; v[k] = vv;
      if (cp) cp[k] = __float2bfloat16(pp);
    }
  }
}

int tmae_adam_step(const void* table, const int32_t* chunk_tensor, int64_t total_chunks, float lr, float beta1, float beta2,
                   float eps, float weight_decay, void* stream_) {
  (void)hipGetLastError();
  if (total_chunks < 0 || total_chunks >= ((int64_t)1 << 31) || !(beta1 >= 0.f && beta1 < 1.f) ||
      !(beta2 >= 0.f && beta2 < 1.f))
    return TMAE_EARG;
  if (total_chunks == 0) return TMAE_OK;
  if (!table || !chunk_tensor || ((uintptr_t)table & 7)) return TMAE_EARG;
  const float decay = (float)(1.0 - (double)weight_decay * (double)lr);
  hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)total_chunks), dim3(256), 0, (hipStream_t)stream_,
                     (const AdamEntry*)table, (const int*)chunk_tensor, decay, (float)(1.0 - (double)beta1), beta2,
                     (float)(1.0 - (double)beta2), (double)lr, (double)beta1, (double)beta2, eps);
  return tmae_launch_status();
}

// ------------------------------------------------------------------------------------------------
// BatchNorm running statistics of ALL norm layers of a forward pass in one launch (torch.nn.BatchNorm's update in
// training mode, torch/nn/modules/batchnorm.py: running = (1 - momentum) * running + momentum * batch statistic, unbiased
// variance; num_batches_tracked += 1).  The step has ~26 such updates (network_utils.py:31, spconv_utils.py:50-54,
// SiamWCA_MAE.py:91-115); as torch._foreach calls they were ~14 multi-tensor launches.  One workgroup per BUFFER applies
// that buffer's updates in order (the VFE norm runs once per frame: two updates of the same buffer).
struct BnBuf {               // 4 x int64
  float* running;            // running_mean or running_var
  int64_t numel;
  int64_t first, count;      // its updates in the update table
};
struct BnUpd {               // 2 x int64
  const float* stat;         // batch mean / biased batch variance [numel]
  int64_t keep_scale;        // float bits: keep (1 - momentum) | scale (momentum [* n / (n - 1)]) << 32
};

__global__ __launch_bounds__(128) void bn_running_update_kernel(const BnBuf* __restrict__ bufs, const BnUpd* __restrict__ upd,
                                                              int64_t* const* __restrict__ counters, int ncounters) {
  if (blockIdx.x == 0)
    for (int c = threadIdx.x; c < ncounters; c += 128)                           // num_batches_tracked: one entry per use
      atomicAdd(reinterpret_cast<unsigned long long*>(counters[c]), 1ull);      // (a module used twice appears twice)
  const BnBuf b = bufs[blockIdx.x];
  for (int64_t e = threadIdx.x; e < b.numel; e += 128) {
    float r = b.running[e];
    for (int64_t u = 0; u < b.count; ++u) {
      const BnUpd q = upd[b.first + u];
      const float keep = __uint_as_float((unsigned)(q.keep_scale & 0xFFFFFFFF)), scale = __uint_as_float((unsigned)(q.keep_scale >> 32));
      r = r * keep + q.stat[e] * scale;
    }
    b.running[e] = r;
  }
}

int tmae_bn_running_update(const void* bufs, int nbufs, const void* updates, const void* counters, int ncounters,
                           void* stream_) {
  (void)hipGetLastError();
  if (nbufs < 0 || ncounters < 0) return TMAE_EARG;
  if (nbufs == 0) return TMAE_OK;
  if (!bufs || !updates || (ncounters > 0 && !counters) || ((uintptr_t)bufs & 7) || ((uintptr_t)updates & 7)) return TMAE_EARG;
  hipLaunchKernelGGL(bn_running_update_kernel, dim3(nbufs), dim3(128), 0, (hipStream_t)stream_, (const BnBuf*)bufs,
                     (const BnUpd*)updates, (int64_t* const*)counters, ncounters);
  return tmae_launch_status();
}
