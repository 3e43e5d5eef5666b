// A6/A7/A10: ragged window cosine attention, forward and backward (fp32 accumulate).
//
// One workgroup per (8x8 window, head group); wave w of the block owns head hg*HG + w, lane l owns the
// window cell (l/8, l%8).  The window's query / key tokens are read straight from the dense index
// grids, so the reference's flat2window / window2flat padding (sst_utils.py:118-192), its per-level
// loop and its key-padding masks disappear: a window with 5 tokens does 5x5 work, not 16x16.
// K-hat / V rows of the window are staged once per head in LDS (padded rows, 16-byte reads that
// broadcast to the whole wave); every query lane then streams over the keys with an online softmax.
// Shapes on this path are tiny (T <= 64, dh 16/32), so the kernel is gather/latency bound, not FLOP
// bound; see DESIGN.md for the roofline.
#include "common.h"
#include <stdlib.h>

#define WIN 8          // window edge (cells); 8x8 = 64 cells = one wavefront
#define ROWPAD 4       // LDS row padding (floats): keeps 16-byte alignment, breaks the power-of-two stride

template <class T, int DH>
__device__ __forceinline__ void load_row(const T* p, float* r) {
#pragma unroll
  for (int c = 0; c < DH; ++c) r[c] = ld_f<T>(p + c);
}

struct WinGeom {
  int b, wcy, wcx;
};

__device__ __forceinline__ void window_tokens(const int32_t* __restrict__ grid_q, const int32_t* __restrict__ grid_k,
                                              int64_t dw, int ny, int nx, int Wy, int Wx, int sy, int sx, int lane,
                                              int& tq, int& tk) {
  const int wcy = (int)(dw % Wy);
  const int wcx = (int)((dw / Wy) % Wx);
  const int b = (int)(dw / ((int64_t)Wy * Wx));
  const int y = wcy * WIN - sy + (lane >> 3), x = wcx * WIN - sx + (lane & 7);
  const bool in = y >= 0 && y < ny && x >= 0 && x < nx;
  const int64_t cell = ((int64_t)b * ny + y) * nx + x;
  tq = in ? grid_q[cell] : -1;
  tk = in ? grid_k[cell] : -1;
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <class T, int DH>
__global__ __launch_bounds__(64 * (64 / DH)) void win_attn_fwd_kernel(
    const T* __restrict__ q, int64_t ldq, const T* __restrict__ k, int64_t ldk, const T* __restrict__ v, int64_t ldv,
    int nhead, const int32_t* __restrict__ grid_q, const int32_t* __restrict__ grid_k, int ny, int nx, int Wy, int Wx,
    int sy, int sx, const float* __restrict__ tau, float tau_min, T* __restrict__ out, int64_t ldo,
    float* __restrict__ lse, int tau_stride) {
  constexpr int HG = 64 / DH;
  constexpr int LD = DH + ROWPAD;
  __shared__ __attribute__((aligned(16))) float Ks[HG][64][LD];
  __shared__ __attribute__((aligned(16))) float Vs[HG][64][LD];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int head = blockIdx.y * HG + w;
  int tq, tk;
  window_tokens(grid_q, grid_k, blockIdx.x, ny, nx, Wy, Wx, sy, sx, lane, tq, tk);
  const unsigned long long mq = __ballot(tq >= 0), mk = __ballot(tk >= 0);
  if (mq == 0ull) return;                                   // block-uniform: every wave sees the same window
  const int nk = __popcll(mk);
  const int hoff = head * DH;
  if (nk == 0) {                                            // cross-attention window with no key: zero rows
    if (tq >= 0) {
      T* o = out + (int64_t)tq * ldo + hoff;
#pragma unroll
      for (int c = 0; c < DH; ++c) st_f<T>(o + c, 0.f);
      lse[(int64_t)tq * nhead + head] = 0.f;
    }
    return;
  }
  if (tk >= 0) {                                            // stage this head's normalised key and value rows
    const int slot = __popcll(mk & ((1ull << lane) - 1ull));
    float r[DH];
    load_row<T, DH>(k + (int64_t)tk * ldk + hoff, r);
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < DH; ++c) ss += r[c] * r[c];
    const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);      // F.normalize eps (cosine_msa.py:150-151)
#pragma unroll
    for (int c = 0; c < DH; ++c) Ks[w][slot][c] = r[c] * inv;
    load_row<T, DH>(v + (int64_t)tk * ldv + hoff, r);
#pragma unroll
    for (int c = 0; c < DH; ++c) Vs[w][slot][c] = r[c];
  }
  __syncthreads();
  if (tq < 0) return;
  // attn / tau.clamp(min=tau_min) (cosine_msa.py:155-161); tau_stride 1: one temperature per head (non_shared_tau, :453-454)
  const float inv_tau = 1.0f / fmaxf(tau[head * tau_stride], tau_min);
  float qh[DH], o[DH];
  load_row<T, DH>(q + (int64_t)tq * ldq + hoff, qh);
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < DH; ++c) ss += qh[c] * qh[c];
  const float qs = inv_tau / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
  for (int c = 0; c < DH; ++c) { qh[c] *= qs; o[c] = 0.f; }
  float mrun = -INFINITY, l = 0.f;
  for (int j = 0; j < nk; ++j) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < DH; c += 4) {
      const float4 kk = *reinterpret_cast<const float4*>(&Ks[w][j][c]);
      s += qh[c] * kk.x + qh[c + 1] * kk.y + qh[c + 2] * kk.z + qh[c + 3] * kk.w;
    }
    const float mnew = fmaxf(mrun, s);
    const float corr = expf(mrun - mnew);                   // exp(-inf) = 0 on the first key
    const float p = expf(s - mnew);
    l = l * corr + p;
#pragma unroll
    for (int c = 0; c < DH; c += 4) {
      const float4 vv = *reinterpret_cast<const float4*>(&Vs[w][j][c]);
      o[c] = o[c] * corr + p * vv.x;
      o[c + 1] = o[c + 1] * corr + p * vv.y;
      o[c + 2] = o[c + 2] * corr + p * vv.z;
      o[c + 3] = o[c + 3] * corr + p * vv.w;
    }
    mrun = mnew;
  }
  const float invl = 1.0f / l;
  T* op = out + (int64_t)tq * ldo + hoff;
#pragma unroll
  for (int c = 0; c < DH; ++c) st_f<T>(op + c, o[c] * invl);
  lse[(int64_t)tq * nhead + head] = mrun + logf(l);
}

// ------------------------------------------------------------------------------------------------
// backward: phase A (lane = query) -> dq, d tau;  phase B (lane = key) -> dk, dv.  P is recomputed
// from the saved log-sum-exp; nothing of size T x T is stored.
// ------------------------------------------------------------------------------------------------
template <class T, int DH>
__global__ __launch_bounds__(64 * (64 / DH)) void win_attn_bwd_kernel(
    const T* __restrict__ q, int64_t ldq, const T* __restrict__ k, int64_t ldk, const T* __restrict__ v, int64_t ldv,
    const T* __restrict__ outp, int64_t ldo, const T* __restrict__ dout, int64_t lddo, const float* __restrict__ lse,
    int nhead, const int32_t* __restrict__ grid_q, const int32_t* __restrict__ grid_k, int ny, int nx, int Wy, int Wx,
    int sy, int sx, const float* __restrict__ tau, float tau_min, T* __restrict__ dq, int64_t lddq,
    T* __restrict__ dk, int64_t lddk, T* __restrict__ dv, int64_t lddv, float* __restrict__ dtau_partial, int tau_stride) {
  constexpr int HG = 64 / DH;
  constexpr int LD = DH + ROWPAD;
  __shared__ __attribute__((aligned(16))) float Ks[HG][64][LD];
  __shared__ __attribute__((aligned(16))) float Vs[HG][64][LD];
  __shared__ __attribute__((aligned(16))) float Qs[HG][64][LD];
  __shared__ __attribute__((aligned(16))) float Gs[HG][64][LD];   // dO rows
  __shared__ float Ls[HG][64], Ds[HG][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int head = blockIdx.y * HG + w;
  const int hoff = head * DH;
  int tq, tk;
  window_tokens(grid_q, grid_k, blockIdx.x, ny, nx, Wy, Wx, sy, sx, lane, tq, tk);
  const unsigned long long mq = __ballot(tq >= 0), mk = __ballot(tk >= 0);
  const int nq = __popcll(mq), nk = __popcll(mk);
  float* dtp = dtau_partial + (int64_t)blockIdx.x * nhead + head;
  if (nq == 0 || nk == 0) {                                 // nothing attended here: zero gradients
    if (lane == 0) *dtp = 0.f;
    if (nq > 0 && tq >= 0) {
      T* g = dq + (int64_t)tq * lddq + hoff;
#pragma unroll
      for (int c = 0; c < DH; ++c) st_f<T>(g + c, 0.f);
    }
    if (nk > 0 && tk >= 0 && grid_q != grid_k) {
      T* g1 = dk + (int64_t)tk * lddk + hoff;
      T* g2 = dv + (int64_t)tk * lddv + hoff;
#pragma unroll
      for (int c = 0; c < DH; ++c) { st_f<T>(g1 + c, 0.f); st_f<T>(g2 + c, 0.f); }
    }
    return;
  }
  const float inv_tau = 1.0f / fmaxf(tau[head * tau_stride], tau_min);
  float kh[DH], vj[DH];
  float knorm = 1.f;
  if (tk >= 0) {
    const int slot = __popcll(mk & ((1ull << lane) - 1ull));
    load_row<T, DH>(k + (int64_t)tk * ldk + hoff, kh);
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < DH; ++c) ss += kh[c] * kh[c];
    knorm = fmaxf(sqrtf(ss), 1e-12f);
    const float inv = 1.0f / knorm;
#pragma unroll
    for (int c = 0; c < DH; ++c) { kh[c] *= inv; Ks[w][slot][c] = kh[c]; }
    load_row<T, DH>(v + (int64_t)tk * ldv + hoff, vj);
#pragma unroll
    for (int c = 0; c < DH; ++c) Vs[w][slot][c] = vj[c];
  }
  float qh[DH], go[DH];
  float qnorm = 1.f, lse_i = 0.f, d_i = 0.f;
  if (tq >= 0) {
    const int slot = __popcll(mq & ((1ull << lane) - 1ull));
    load_row<T, DH>(q + (int64_t)tq * ldq + hoff, qh);
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < DH; ++c) ss += qh[c] * qh[c];
    qnorm = fmaxf(sqrtf(ss), 1e-12f);
    const float inv = 1.0f / qnorm;
    load_row<T, DH>(dout + (int64_t)tq * lddo + hoff, go);
    float orow[DH];
    load_row<T, DH>(outp + (int64_t)tq * ldo + hoff, orow);
#pragma unroll
    for (int c = 0; c < DH; ++c) {
      qh[c] *= inv;
      Qs[w][slot][c] = qh[c];
      Gs[w][slot][c] = go[c];
      d_i += go[c] * orow[c];                               // D_i = dO_i . O_i = sum_j P_ij dP_ij
    }
    lse_i = lse[(int64_t)tq * nhead + head];
    Ls[w][slot] = lse_i;
    Ds[w][slot] = d_i;
  }
  __syncthreads();
  // ---- phase A: query lanes
  float dtau_acc = 0.f;
  if (tq >= 0) {
    float dqh[DH];
#pragma unroll
    for (int c = 0; c < DH; ++c) dqh[c] = 0.f;
    for (int j = 0; j < nk; ++j) {
      float cs = 0.f, dp = 0.f;
#pragma unroll
      for (int c = 0; c < DH; c += 4) {
        const float4 kk = *reinterpret_cast<const float4*>(&Ks[w][j][c]);
        const float4 vv = *reinterpret_cast<const float4*>(&Vs[w][j][c]);
        cs += qh[c] * kk.x + qh[c + 1] * kk.y + qh[c + 2] * kk.z + qh[c + 3] * kk.w;
        dp += go[c] * vv.x + go[c + 1] * vv.y + go[c + 2] * vv.z + go[c + 3] * vv.w;
      }
      const float s = cs * inv_tau;
      const float p = expf(s - lse_i);
      const float ds = p * (dp - d_i);
      dtau_acc += ds * s;
      const float g = ds * inv_tau;
#pragma unroll
      for (int c = 0; c < DH; c += 4) {
        const float4 kk = *reinterpret_cast<const float4*>(&Ks[w][j][c]);
        dqh[c] += g * kk.x; dqh[c + 1] += g * kk.y; dqh[c + 2] += g * kk.z; dqh[c + 3] += g * kk.w;
      }
    }
    // through q_hat = q / max(|q|, eps): dq = (dq_hat - q_hat (q_hat . dq_hat)) / |q|
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < DH; ++c) dot += dqh[c] * qh[c];
    if (qnorm <= 1e-12f) dot = 0.f;                          // clamped branch: q_hat = q / eps, no projection
    const float inv = 1.0f / qnorm;
    T* g = dq + (int64_t)tq * lddq + hoff;
#pragma unroll
    for (int c = 0; c < DH; ++c) st_f<T>(g + c, (dqh[c] - qh[c] * dot) * inv);
  }
  dtau_acc = wave_sum(dtau_acc);
  if (lane == 0) *dtp = dtau_acc;                            // sum_ij dS_ij * s_ij ; caller applies -1/tau_c
  // ---- phase B: key lanes
  if (tk >= 0) {
    float dkh[DH], dvj[DH];
#pragma unroll
    for (int c = 0; c < DH; ++c) { dkh[c] = 0.f; dvj[c] = 0.f; }
    for (int i = 0; i < nq; ++i) {
      float cs = 0.f, dp = 0.f;
#pragma unroll
      for (int c = 0; c < DH; c += 4) {
        const float4 qq = *reinterpret_cast<const float4*>(&Qs[w][i][c]);
        const float4 gg = *reinterpret_cast<const float4*>(&Gs[w][i][c]);
        cs += qq.x * kh[c] + qq.y * kh[c + 1] + qq.z * kh[c + 2] + qq.w * kh[c + 3];
        dp += gg.x * vj[c] + gg.y * vj[c + 1] + gg.z * vj[c + 2] + gg.w * vj[c + 3];
      }
      const float s = cs * inv_tau;
      const float p = expf(s - Ls[w][i]);
      const float g = p * (dp - Ds[w][i]) * inv_tau;
#pragma unroll
      for (int c = 0; c < DH; c += 4) {
        const float4 qq = *reinterpret_cast<const float4*>(&Qs[w][i][c]);
        const float4 gg = *reinterpret_cast<const float4*>(&Gs[w][i][c]);
        dkh[c] += g * qq.x; dkh[c + 1] += g * qq.y; dkh[c + 2] += g * qq.z; dkh[c + 3] += g * qq.w;
        dvj[c] += p * gg.x; dvj[c + 1] += p * gg.y; dvj[c + 2] += p * gg.z; dvj[c + 3] += p * gg.w;
      }
    }
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < DH; ++c) dot += dkh[c] * kh[c];
    if (knorm <= 1e-12f) dot = 0.f;
    const float inv = 1.0f / knorm;
    T* g1 = dk + (int64_t)tk * lddk + hoff;
    T* g2 = dv + (int64_t)tk * lddv + hoff;
#pragma unroll
    for (int c = 0; c < DH; ++c) {
      st_f<T>(g1 + c, (dkh[c] - kh[c] * dot) * inv);
      st_f<T>(g2 + c, dvj[c]);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
// bf16 runs on the MFMA kernels of attention_mfma.hip; TMAE_ATTN_IMPL=valu forces the fp32-VALU kernels (A/B runs)
int tmae_win_attn_fwd_mfma(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                           int64_t mq, int64_t mk, int nhead, int dh, const int32_t* grid_q, const int32_t* grid_k,
                           int batch, int ny, int nx, int do_shift, const float* tau, float tau_min, void* out,
                           int64_t ldo, float* lse, const int32_t* worklist, int tau_stride, hipStream_t stream);
int tmae_win_attn_bwd_mfma(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                           const void* out, int64_t ldo, const void* dout, int64_t lddo, const float* lse, int64_t mq,
                           int64_t mk, int nhead, int dh, const int32_t* grid_q, const int32_t* grid_k, int batch,
                           int ny, int nx, int do_shift, const float* tau, float tau_min, void* dq, int64_t lddq,
                           void* dk, int64_t lddk, void* dv, int64_t lddv, float* dtau_partial,
                           const int32_t* worklist, int tau_stride, hipStream_t stream);
// bf16 inputs take the MFMA kernels; TMAE_ATTN_VALU=1 (debug build only, common.h) sends them to the fp32-style VALU kernels
static bool use_mfma() {
  static const int valu = TMAE_AB_INT("TMAE_ATTN_VALU", 0);
  return valu == 0;
}

static inline void attn_dims(int ny, int nx, int& Wy, int& Wx) {
  Wy = (ny + WIN - 1) / WIN + 1;
  Wx = (nx + WIN - 1) / WIN + 1;
}

// sum of the per-(window, head) tau partials in a fixed order, ONE launch for any n: block b sums its contiguous strip
// (thread t the elements t, t+1024, ... of it; LDS tree) into strip_sum[b]; the block that takes the last ticket adds the
// strip sums in the order b = 0, 1, ... and applies the clamp rule.  (The caller used to reduce the ~1e5..5e5 partials with
// a library reduction first: two more launches per attention module and step.)  The ticket counter and the strip sums
// are module-scope device variables: calls must be ordered on one stream (they are: the training stream).
#define DTAU_MAX_BLOCKS 64
__device__ unsigned dtau_ticket = 0;
__device__ float dtau_strip_sum[DTAU_MAX_BLOCKS];

// listed (may be NULL) / per: the class byte of window e / per from the work list (attention_mfma.hip: win_class_kernel) -- a
// window that is in no list wrote no partial and counts as zero, so the caller need not pre-zero the partials.
// off / stride: the partials e * stride + off, e < n (one temperature per head: stride = heads, off = the head, n = windows; one
// launch per head, ordered on the stream like any other two calls); shared temperature: off 0, stride 1, n = windows x heads
__global__ __launch_bounds__(1024) void dtau_finish_kernel(const float* __restrict__ part_raw, int64_t n,
                                                          const float* __restrict__ tau, float tau_min,
                                                          float* __restrict__ dtau, const int8_t* __restrict__ listed, int heads,
                                                          int off, int stride) {
  struct Part {
    const float* p; const int8_t* l; int per, off, stride;
    __device__ __forceinline__ float operator[](int64_t e) const {
      const int64_t f = e * stride + off;
      return (l == nullptr || l[f / per] >= 0) ? p[f] : 0.f;
    }
  } part{part_raw, listed, heads, off, stride};
  __shared__ float red[1024];
  __shared__ bool last;
  const int nb = gridDim.x;
  const int64_t per = (n + nb - 1) / nb;
  const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int64_t e = lo + threadIdx.x;
  for (; e + 3072 < hi; e += 4096) { a0 += part[e]; a1 += part[e + 1024]; a2 += part[e + 2048]; a3 += part[e + 3072]; }
  for (; e < hi; e += 1024) a0 += part[e];
  red[threadIdx.x] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    dtau_strip_sum[blockIdx.x] = red[0];
    __threadfence();                                           // the strip sum is visible before the ticket is taken
    last = atomicAdd(&dtau_ticket, 1u) == (unsigned)(nb - 1);
  }
  __syncthreads();
  if (last && threadIdx.x == 0) {
    __threadfence();
    float tot = 0.f;
    for (int b = 0; b < nb; ++b) tot += __hip_atomic_load(&dtau_strip_sum[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int ti = stride > 1 ? off : 0;
    const float t = tau[ti];
    dtau[ti] = t >= tau_min ? -tot / t : 0.f;
    dtau_ticket = 0;                                           // ready for the next call on the stream
  }
}

int tmae_win_attn_dtau(const float* dtau_partial, int64_t n, const float* tau, float tau_min, float* dtau,
                       const int32_t* worklist, int nhead, int tau_per_head, void* stream_) {
  (void)hipGetLastError();
  if (n < 0 || !tau || !dtau || (n > 0 && !dtau_partial) || (worklist && (nhead <= 0 || n % nhead))) return TMAE_EARG;
  if (tau_per_head != 0 && tau_per_head != 1) return TMAE_EARG;
  if (tau_per_head && (nhead <= 0 || n % nhead)) return TMAE_EARG;
  // the class bytes behind the four lists (tmae_window_worklist_size: counts | lists | class bytes); n = windows x heads
  const int8_t* listed = worklist ? reinterpret_cast<const int8_t*>(worklist + 8 + 4 * (n / nhead)) : nullptr;
  const int launches = tau_per_head ? nhead : 1, stride = tau_per_head ? nhead : 1;
  const int64_t cnt = n / stride;
  int64_t nb = (cnt + 8191) / 8192;                            // >= 8 elements per thread and block
  if (nb < 1) nb = 1;
  if (nb > DTAU_MAX_BLOCKS) nb = DTAU_MAX_BLOCKS;
  for (int h = 0; h < launches; ++h)
    hipLaunchKernelGGL(dtau_finish_kernel, dim3((unsigned)nb), dim3(1024), 0, (hipStream_t)stream_, dtau_partial, cnt, tau,
                       tau_min, dtau, listed, nhead > 0 ? nhead : 1, h, stride);
  return tmae_launch_status();
}

int64_t tmae_win_attn_num_blocks(int batch, int ny, int nx, int nhead, int dh) {
  (void)dh;
  int Wy, Wx;
  attn_dims(ny, nx, Wy, Wx);
  return (int64_t)batch * Wy * Wx * nhead;
}

static int attn_check(int64_t mq, int64_t mk, int nhead, int dh, int batch, int ny, int nx) {
  if (mq < 0 || mk < 0 || nhead <= 0 || batch <= 0 || ny <= 0 || nx <= 0) return TMAE_EARG;
  if (dh != 16 && dh != 32) return TMAE_EARG;
  if (nhead % (64 / dh)) return TMAE_EARG;
  return 0;
}

int tmae_win_attn_fwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, int dtype,
                      int64_t mq, int64_t mk, int nhead, int dh, const int32_t* grid_q, const int32_t* grid_k,
                      int batch, int ny, int nx, int do_shift, const float* tau, float tau_min, void* out, int64_t ldo,
                      float* lse, const int32_t* worklist, int tau_per_head, void* stream_) {
  (void)hipGetLastError();   // drop stale errors of other runtime users: the return code is about OUR launches
  hipStream_t stream = (hipStream_t)stream_;
  int r = attn_check(mq, mk, nhead, dh, batch, ny, nx);
  if (r) return r;
  if (mq == 0) return TMAE_OK;
  if (!q || !grid_q || !grid_k || !tau || !out || !lse || (mk > 0 && (!k || !v))) return TMAE_EARG;
  if (tau_per_head != 0 && tau_per_head != 1) return TMAE_EARG;
  int Wy, Wx;
  attn_dims(ny, nx, Wy, Wx);
  const int s = do_shift ? WIN / 2 : WIN;
  dim3 grid((unsigned)((int64_t)batch * Wy * Wx), (unsigned)(nhead / (64 / dh)));
  dim3 block(64 * (64 / dh));
#define FWD(T, DH)                                                                                                   \
  hipLaunchKernelGGL((win_attn_fwd_kernel<T, DH>), grid, block, 0, stream, (const T*)q, ldq, (const T*)k, ldk,       \
                     (const T*)v, ldv, nhead, grid_q, grid_k, ny, nx, Wy, Wx, s, s, tau, tau_min, (T*)out, ldo, lse, tau_per_head)
  if (dtype == TMAE_F32) { if (dh == 16) FWD(float, 16); else FWD(float, 32); }
  else if (dtype == TMAE_BF16) {
    if (use_mfma() && nhead % 4 == 0 && ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && !((uintptr_t)q & 15) &&
        !((uintptr_t)k & 15) && !((uintptr_t)v & 15))
      return tmae_win_attn_fwd_mfma(q, ldq, k, ldk, v, ldv, mq, mk, nhead, dh, grid_q, grid_k, batch, ny, nx, do_shift,
                                    tau, tau_min, out, ldo, lse, worklist, tau_per_head, stream);
    if (dh == 16) FWD(__hip_bfloat16, 16); else FWD(__hip_bfloat16, 32);
  }
  else return TMAE_EDTYPE;
#undef FWD
  return tmae_launch_status();
}

int tmae_win_attn_bwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                      const void* out, int64_t ldo, const void* dout, int64_t lddo, const float* lse, int dtype,
                      int64_t mq, int64_t mk, int nhead, int dh, const int32_t* grid_q, const int32_t* grid_k,
                      int batch, int ny, int nx, int do_shift, const float* tau, float tau_min, void* dq, int64_t lddq,
                      void* dk, int64_t lddk, void* dv, int64_t lddv, float* dtau_partial, const int32_t* worklist,
                      int tau_per_head, void* stream_) {
  (void)hipGetLastError();   // drop stale errors of other runtime users: the return code is about OUR launches
  hipStream_t stream = (hipStream_t)stream_;
  int r = attn_check(mq, mk, nhead, dh, batch, ny, nx);
  if (r) return r;
  if (!grid_q || !grid_k || !tau || !dtau_partial || (tau_per_head != 0 && tau_per_head != 1)) return TMAE_EARG;
  if (mq > 0 && (!q || !out || !dout || !lse || !dq)) return TMAE_EARG;
  if (mk > 0 && (!k || !v || !dk || !dv)) return TMAE_EARG;
  int Wy, Wx;
  attn_dims(ny, nx, Wy, Wx);
  const int s = do_shift ? WIN / 2 : WIN;
  dim3 grid((unsigned)((int64_t)batch * Wy * Wx), (unsigned)(nhead / (64 / dh)));
  dim3 block(64 * (64 / dh));
#define BWD(T, DH)                                                                                                   \
  hipLaunchKernelGGL((win_attn_bwd_kernel<T, DH>), grid, block, 0, stream, (const T*)q, ldq, (const T*)k, ldk,       \
                     (const T*)v, ldv, (const T*)out, ldo, (const T*)dout, lddo, lse, nhead, grid_q, grid_k, ny, nx, \
                     Wy, Wx, s, s, tau, tau_min, (T*)dq, lddq, (T*)dk, lddk, (T*)dv, lddv, dtau_partial, tau_per_head)
  if (dtype == TMAE_F32) { if (dh == 16) BWD(float, 16); else BWD(float, 32); }
  else if (dtype == TMAE_BF16) {
    const bool al = !(ldq % 8) && !(ldk % 8) && !(ldv % 8) && !(ldo % 8) && !(lddo % 8) && !((uintptr_t)q & 15) &&
                    !((uintptr_t)k & 15) && !((uintptr_t)v & 15) && !((uintptr_t)out & 15) && !((uintptr_t)dout & 15);
    if (use_mfma() && nhead % 4 == 0 && al && mq > 0 && mk > 0)
      return tmae_win_attn_bwd_mfma(q, ldq, k, ldk, v, ldv, out, ldo, dout, lddo, lse, mq, mk, nhead, dh, grid_q, grid_k,
                                    batch, ny, nx, do_shift, tau, tau_min, dq, lddq, dk, lddk, dv, lddv, dtau_partial,
                                    worklist, tau_per_head, stream);
    if (dh == 16) BWD(__hip_bfloat16, 16); else BWD(__hip_bfloat16, 32);
  }
  else return TMAE_EDTYPE;
#undef BWD
  return tmae_launch_status();
}
