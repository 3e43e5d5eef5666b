// A6/A7 (bf16 path): ragged window cosine attention on the matrix cores.
//
// One wavefront per (8x8 window, head); 4 heads per workgroup.  The window's tokens come from the dense index
// grids (see attention.hip); its <=64 queries / keys are cut into 16-row tiles.
//   S^T = K-hat . Q-hat^T   16x16x32 (dh 32) / 16x16x16 (dh 16) MFMA, operands loaded straight from the token rows
//                           (the contraction index is the contiguous channel axis: plain 16-/8-byte row reads),
//                           L2-normalised and scaled by 1/max(tau,tau_min) in registers before the bf16 cast.
//   softmax                  in registers: the "swapped" product leaves one query per lane column, its keys in the
//                            4 accumulator rows x 4 lane groups, so max / sum are 2 cross-lane steps (fp32).
//   O = P . V                16x16x16 MFMA; P is taken from the accumulators as-is (their layout IS the A-operand
//                            layout), V comes from a row-major LDS image through ds_read_b64_tr_b16.
// No padding to 16/32/64-token levels, no key masks in memory, nothing of size T x T leaves the registers.
#include "common.h"

#define WIN 8
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ s16x4 tr_read4(const char* lds_ptr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)lds_ptr);
}
__device__ __forceinline__ short f2bf(float v) {
  __hip_bfloat16 b = __float2bfloat16(v);
  return *reinterpret_cast<short*>(&b);
}
__device__ __forceinline__ float bf2f(short s) { return __uint_as_float(((unsigned)(unsigned short)s) << 16); }

// Row fragment of a token: channels [FR*g, FR*g+FR) of head `hoff`, FR = 8 (dh 32) or 4 (dh 16), as floats.
template <int FR>
__device__ __forceinline__ void load_row_frag(const __hip_bfloat16* base, int64_t ld, int tok, int hoff, int g,
                                              float* f) {
  if (tok < 0) {
#pragma unroll
    for (int j = 0; j < FR; ++j) f[j] = 0.f;
    return;
  }
  const __hip_bfloat16* p = base + (int64_t)tok * ld + hoff + FR * g;
  if constexpr (FR == 8) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) { f[2 * j] = __uint_as_float(w[j] << 16); f[2 * j + 1] = __uint_as_float(w[j] & 0xFFFF0000u); }
  } else {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xFFFF0000u);
    f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xFFFF0000u);
  }
}

// L2-normalise a token row that is spread over the 4 lane groups (same lane&15), times `scale`.
template <int FR>
__device__ __forceinline__ float normalize_frag(float* f, float scale) {
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < FR; ++j) ss += f[j] * f[j];
  ss += __shfl_xor(ss, 16, 64);
  ss += __shfl_xor(ss, 32, 64);
  const float nrm = fmaxf(sqrtf(ss), 1e-12f);
  const float inv = scale / nrm;
#pragma unroll
  for (int j = 0; j < FR; ++j) f[j] *= inv;
  return nrm;                                   // max(|x|, eps) of the whole row
}

template <int DH> struct Frag;
template <> struct Frag<32> { typedef s16x8 T; };
template <> struct Frag<16> { typedef s16x4 T; };

template <int FR>
__device__ __forceinline__ typename Frag<FR * 4>::T pack_frag(const float* f) {
  typename Frag<FR * 4>::T r;
#pragma unroll
  for (int j = 0; j < FR; ++j) r[j] = f2bf(f[j]);
  return r;
}

// hi/lo split of an fp32 fragment into two bf16 fragments (f ~= hi + lo): three MFMAs (hi.hi + hi.lo + lo.hi) give
// the cosine logits to ~2^-16 relative instead of bf16's 2^-8 -- they are divided by tau >= 0.01 before the exp.
template <int FR>
__device__ __forceinline__ void split_frag(const float* f, typename Frag<FR * 4>::T& hi, typename Frag<FR * 4>::T& lo) {
#pragma unroll
  for (int j = 0; j < FR; ++j) {
    const short h = f2bf(f[j]);
    hi[j] = h;
    lo[j] = f2bf(f[j] - bf2f(h));
  }
}

__device__ __forceinline__ f32x4 mfma_s(const s16x8& a, const s16x8& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a), *reinterpret_cast<const bf16x8*>(&b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma_s(const s16x4& a, const s16x4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}

template <int DH>
__global__ __launch_bounds__(256) void win_attn_fwd_mfma_kernel(
    const __hip_bfloat16* __restrict__ q, int64_t ldq, const __hip_bfloat16* __restrict__ k, int64_t ldk,
    const __hip_bfloat16* __restrict__ v, int64_t ldv, int nhead, const int32_t* __restrict__ grid_q,
    const int32_t* __restrict__ grid_k, int ny, int nx, int Wy, int Wx, int sy, int sx,
    const float* __restrict__ tau, float tau_min, __hip_bfloat16* __restrict__ out, int64_t ldo,
    float* __restrict__ lse) {
  constexpr int FR = DH / 4;                 // channels per lane in a row fragment
  constexpr int CT = DH / 16;                // 16-channel output tiles
  constexpr int RB = DH * 2 + 16;            // V image row pitch (bytes): 16-byte aligned, off the power of two
  typedef typename Frag<DH>::T frag_t;
  __shared__ int toks[2][64];
  __shared__ __attribute__((aligned(16))) char vimg[4][64 * RB];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, g = lane >> 4, i = lane & 15;
  const int head = blockIdx.y * 4 + w, hoff = head * DH;
  // window tokens (every wave computes the same; wave 0 publishes the compacted lists)
  const int64_t dw = blockIdx.x;
  const int wcy = (int)(dw % Wy), wcx = (int)((dw / Wy) % Wx), b = (int)(dw / ((int64_t)Wy * Wx));
  const int y = wcy * WIN - sy + (lane >> 3), x = wcx * WIN - sx + (lane & 7);
  const bool in = y >= 0 && y < ny && x >= 0 && x < nx;
  const int64_t cell = ((int64_t)b * ny + y) * nx + x;
  const int tq = in ? grid_q[cell] : -1, tk = in ? grid_k[cell] : -1;
  const unsigned long long mq = __ballot(tq >= 0), mk = __ballot(tk >= 0);
  if (mq == 0ull) return;
  const int Tq = __popcll(mq), Tk = __popcll(mk);
  if (Tk == 0) {                              // cross-attention window without keys: zero rows (not "kept")
    if (tq >= 0) {
      __hip_bfloat16* o = out + (int64_t)tq * ldo + hoff;
#pragma unroll
      for (int c = 0; c < DH; ++c) o[c] = __float2bfloat16(0.f);
      lse[(int64_t)tq * nhead + head] = 0.f;
    }
    return;
  }
  if (w == 0) {
    if (tq >= 0) toks[0][__popcll(mq & ((1ull << lane) - 1ull))] = tq;
    if (tk >= 0) toks[1][__popcll(mk & ((1ull << lane) - 1ull))] = tk;
  }
  // stage this head's V rows (row = key slot) for the transposed reads
  if (tk >= 0) {
    const int slot = __popcll(mk & ((1ull << lane) - 1ull));
    const uint4* src = reinterpret_cast<const uint4*>(v + (int64_t)tk * ldv + hoff);
    uint4* dst = reinterpret_cast<uint4*>(&vimg[w][slot * RB]);
#pragma unroll
    for (int c = 0; c < DH / 8; ++c) dst[c] = src[c];
  }
  if (lane >= Tk) {                          // rows no key writes: zero them (P is 0 there, but 0 * garbage = NaN)
    uint4* dst = reinterpret_cast<uint4*>(&vimg[w][lane * RB]);
#pragma unroll
    for (int c = 0; c < DH / 8; ++c) dst[c] = make_uint4(0, 0, 0, 0);
  }
  __syncthreads();
  const int nq = (Tq + 15) >> 4, nk = (Tk + 15) >> 4;
  const float inv_tau = 1.0f / fmaxf(tau[0], tau_min);
  // K-hat fragments of every key tile stay in registers
  frag_t kf[4], kl[4];
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    const int slot = kt * 16 + i;
    if (kt < nk) {                                   // wave-uniform
      float f[FR];
      load_row_frag<FR>(k, ldk, slot < Tk ? toks[1][slot] : -1, hoff, g, f);
      normalize_frag<FR>(f, 1.0f);
      split_frag<FR>(f, kf[kt], kl[kt]);
    } else {
#pragma unroll
      for (int j = 0; j < FR; ++j) { kf[kt][j] = 0; kl[kt][j] = 0; }
    }
  }
  // V fragments (B operand of P.V): [key tile][channel tile], 4 keys x 1 channel per lane
  s16x4 vf[4][CT];
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
      vf[kt][ct] = tr_read4(&vimg[w][(kt * 16 + 4 * g + (i >> 2)) * RB + ct * 32 + 8 * (i & 3)]);

  for (int qt = 0; qt < nq; ++qt) {
    float f[FR];
    const int qslot = qt * 16 + i;
    load_row_frag<FR>(q, ldq, qslot < Tq ? toks[0][qslot] : -1, hoff, g, f);
    normalize_frag<FR>(f, inv_tau);
    frag_t qf, ql;
    split_frag<FR>(f, qf, ql);
    f32x4 st[4];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      st[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (kt < nk) {
        st[kt] = mfma_s(kl[kt], qf, st[kt]);                       // S^T tile: rows = keys 4g+r, col = query i
        st[kt] = mfma_s(kf[kt], ql, st[kt]);
        st[kt] = mfma_s(kf[kt], qf, st[kt]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (kt * 16 + 4 * g + r >= Tk) st[kt][r] = -INFINITY;
        mx = fmaxf(mx, st[kt][r]);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float l = 0.f;
    s16x4 pf[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __expf(st[kt][r] - mx);
        l += p;
        pf[kt][r] = f2bf(p);
      }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float invl = 1.0f / l;
    f32x4 o[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      o[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
        if (kt < nk) o[ct] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pf[kt], vf[kt][ct], o[ct], 0, 0, 0);
    }
    // O tile: rows = queries 4g+r, col = channel i
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qs = qt * 16 + 4 * g + r;
      const float il = __shfl(invl, 4 * g + r, 64);
      if (qs < Tq) {
        __hip_bfloat16* op = out + (int64_t)toks[0][qs] * ldo + hoff + i;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) op[ct * 16] = __float2bfloat16(o[ct][r] * il);
      }
    }
    if (g == 0 && qslot < Tq) lse[(int64_t)toks[0][qslot] * nhead + head] = mx + __logf(l);
  }
}

int tmae_win_attn_fwd_mfma(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                           int64_t mq, int64_t mk, int nhead, int dh, const int32_t* grid_q, const int32_t* grid_k,
                           int batch, int ny, int nx, int do_shift, const float* tau, float tau_min, void* out,
                           int64_t ldo, float* lse, hipStream_t stream) {
  if (nhead % 4 || (dh != 16 && dh != 32)) return TMAE_EARG;
  // 16-byte row fragments / V rows: bases and pitches must keep every head slice 16-byte aligned
  if ((ldq % 8) || (ldk % 8) || (ldv % 8) || ((uintptr_t)q & 15) || ((uintptr_t)k & 15) || ((uintptr_t)v & 15))
    return TMAE_EARG;
  const int Wy = (ny + WIN - 1) / WIN + 1, Wx = (nx + WIN - 1) / WIN + 1;
  const int s = do_shift ? WIN / 2 : WIN;
  dim3 grid((unsigned)((int64_t)batch * Wy * Wx), (unsigned)(nhead / 4));
  if (dh == 32)
    hipLaunchKernelGGL(win_attn_fwd_mfma_kernel<32>, grid, dim3(256), 0, stream, (const __hip_bfloat16*)q, ldq,
                       (const __hip_bfloat16*)k, ldk, (const __hip_bfloat16*)v, ldv, nhead, grid_q, grid_k, ny, nx, Wy,
                       Wx, s, s, tau, tau_min, (__hip_bfloat16*)out, ldo, lse);
  else
    hipLaunchKernelGGL(win_attn_fwd_mfma_kernel<16>, grid, dim3(256), 0, stream, (const __hip_bfloat16*)q, ldq,
                       (const __hip_bfloat16*)k, ldk, (const __hip_bfloat16*)v, ldv, nhead, grid_q, grid_k, ny, nx, Wy,
                       Wx, s, s, tau, tau_min, (__hip_bfloat16*)out, ldo, lse);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// backward (bf16): P is recomputed from the saved log-sum-exp in BOTH orientations, because the five products
// need the T x T factor once with the query on the lane column (dQ-hat = dS.K-hat) and once with the key on it
// (dV = P^T.dO, dK-hat = dS^T.Q-hat) -- recomputing two tiny MFMAs is cheaper than any cross-lane transpose.
// The right-hand factors (K-hat, Q-hat/tau, dO) are row-major LDS images read with ds_read_b64_tr_b16.
// ------------------------------------------------------------------------------------------------
template <int DH>
__global__ __launch_bounds__(256) void win_attn_bwd_mfma_kernel(
    const __hip_bfloat16* __restrict__ q, int64_t ldq, const __hip_bfloat16* __restrict__ k, int64_t ldk,
    const __hip_bfloat16* __restrict__ v, int64_t ldv, const __hip_bfloat16* __restrict__ outp, int64_t ldo,
    const __hip_bfloat16* __restrict__ dout, int64_t lddo, const float* __restrict__ lse, int nhead,
    const int32_t* __restrict__ grid_q, const int32_t* __restrict__ grid_k, int ny, int nx, int Wy, int Wx, int sy,
    int sx, const float* __restrict__ tau, float tau_min, __hip_bfloat16* __restrict__ dq, int64_t lddq,
    __hip_bfloat16* __restrict__ dk, int64_t lddk, __hip_bfloat16* __restrict__ dv, int64_t lddv,
    float* __restrict__ dtau_partial) {
  constexpr int FR = DH / 4;
  constexpr int CT = DH / 16;
  constexpr int RB = DH * 2 + 16;
  typedef typename Frag<DH>::T frag_t;
  __shared__ int toks[2][64];
  __shared__ __attribute__((aligned(16))) char kimg[4][64 * RB];   // K-hat
  __shared__ __attribute__((aligned(16))) char qimg[4][64 * RB];   // Q-hat / tau_c
  __shared__ __attribute__((aligned(16))) char gimg[4][64 * RB];   // dO
  __shared__ float qnorm[4][64], knorm[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, g = lane >> 4, i = lane & 15;
  const int head = blockIdx.y * 4 + w, hoff = head * DH;
  const int64_t dw = blockIdx.x;
  const int wcy = (int)(dw % Wy), wcx = (int)((dw / Wy) % Wx), b = (int)(dw / ((int64_t)Wy * Wx));
  const int y = wcy * WIN - sy + (lane >> 3), x = wcx * WIN - sx + (lane & 7);
  const bool in = y >= 0 && y < ny && x >= 0 && x < nx;
  const int64_t cell = ((int64_t)b * ny + y) * nx + x;
  const int tq = in ? grid_q[cell] : -1, tk = in ? grid_k[cell] : -1;
  const unsigned long long mq = __ballot(tq >= 0), mk = __ballot(tk >= 0);
  const int Tq = __popcll(mq), Tk = __popcll(mk);
  float* dtp = dtau_partial + dw * nhead + head;
  if (Tq == 0 || Tk == 0) {
    if (lane == 0) *dtp = 0.f;
    if (Tq > 0 && tq >= 0) {
      __hip_bfloat16* p = dq + (int64_t)tq * lddq + hoff;
#pragma unroll
      for (int c = 0; c < DH; ++c) p[c] = __float2bfloat16(0.f);
    }
    if (Tk > 0 && tk >= 0 && grid_q != grid_k) {
      __hip_bfloat16* p1 = dk + (int64_t)tk * lddk + hoff;
      __hip_bfloat16* p2 = dv + (int64_t)tk * lddv + hoff;
#pragma unroll
      for (int c = 0; c < DH; ++c) { p1[c] = __float2bfloat16(0.f); p2[c] = __float2bfloat16(0.f); }
    }
    return;
  }
  const float inv_tau = 1.0f / fmaxf(tau[0], tau_min);
  if (w == 0) {
    if (tq >= 0) toks[0][__popcll(mq & ((1ull << lane) - 1ull))] = tq;
    if (tk >= 0) toks[1][__popcll(mk & ((1ull << lane) - 1ull))] = tk;
  }
  __syncthreads();                                   // token lists visible
  const int nq = (Tq + 15) >> 4, nk = (Tk + 15) >> 4;
  // ---- stage the row-major LDS images from 16-byte row fragments: lane (g,i) owns chunk g of row tile*16+i.
  //      Slots beyond T get zero fragments, so every row of every image is defined.
  frag_t kf[4], kl[4], vr[4];
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    const int slot = kt * 16 + i;
    if (kt < nk) {                                   // wave-uniform: most windows hold a single 16-token tile
      const int tokk = slot < Tk ? toks[1][slot] : -1;
      float f[FR];
      load_row_frag<FR>(k, ldk, tokk, hoff, g, f);
      const float nrm = normalize_frag<FR>(f, 1.0f);
      split_frag<FR>(f, kf[kt], kl[kt]);
      if (g == 0) knorm[w][slot] = nrm;
      load_row_frag<FR>(v, ldv, tokk, hoff, g, f);
      vr[kt] = pack_frag<FR>(f);
    } else {
#pragma unroll
      for (int j = 0; j < FR; ++j) { kf[kt][j] = 0; kl[kt][j] = 0; vr[kt][j] = 0; }
    }
    *reinterpret_cast<frag_t*>(&kimg[w][slot * RB + FR * g * 2]) = kf[kt];
  }
#pragma unroll
  for (int qt = 0; qt < 4; ++qt) {
    const int slot = qt * 16 + i;
    if (qt < nq) {
      const int tokq = slot < Tq ? toks[0][slot] : -1;
      float f[FR];
      load_row_frag<FR>(q, ldq, tokq, hoff, g, f);
      const float nrm = normalize_frag<FR>(f, inv_tau);
      *reinterpret_cast<frag_t*>(&qimg[w][slot * RB + FR * g * 2]) = pack_frag<FR>(f);
      if (g == 0) qnorm[w][slot] = nrm;
      load_row_frag<FR>(dout, lddo, tokq, hoff, g, f);
      *reinterpret_cast<frag_t*>(&gimg[w][slot * RB + FR * g * 2]) = pack_frag<FR>(f);   // exact: bf16 -> f32 -> bf16
    }
    // tiles >= nq are never read: the q loop and its transposed reads stop at nq
  }
  __syncthreads();
  f32x4 dKa[4][CT], dVa[4][CT];
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) { dKa[kt][ct] = f32x4{0.f, 0.f, 0.f, 0.f}; dVa[kt][ct] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  float dtau_acc = 0.f;

  for (int qt = 0; qt < nq; ++qt) {
    const int qslot = qt * 16 + i;
    const bool qok = qslot < Tq;
    const int qtok = qok ? toks[0][qslot] : -1;
    frag_t qf, ql;
    {
      float f[FR];
      load_row_frag<FR>(q, ldq, qtok, hoff, g, f);
      normalize_frag<FR>(f, inv_tau);
      split_frag<FR>(f, qf, ql);
    }
    const frag_t gf = *reinterpret_cast<const frag_t*>(&gimg[w][qslot * RB + FR * g * 2]);
    // D = dO . O and LSE of query i (swapped orientation) -> shuffled copies for queries 4g+r (unswapped)
    float of[FR], gfl[FR];
    load_row_frag<FR>(outp, ldo, qtok, hoff, g, of);
#pragma unroll
    for (int j = 0; j < FR; ++j) gfl[j] = bf2f(gf[j]);
    float dsum = 0.f;
#pragma unroll
    for (int j = 0; j < FR; ++j) dsum += of[j] * gfl[j];
    dsum += __shfl_xor(dsum, 16, 64);
    dsum += __shfl_xor(dsum, 32, 64);
    const float lse_i = qok ? lse[(int64_t)qtok * nhead + head] : 0.f;
    float lse_r[4], d_r[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { lse_r[r] = __shfl(lse_i, 4 * g + r, 64); d_r[r] = __shfl(dsum, 4 * g + r, 64); }
    // transposed right-hand fragments of this query tile
    s16x4 trQ[CT], trG[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int off = (qt * 16 + 4 * g + (i >> 2)) * RB + ct * 32 + 8 * (i & 3);
      trQ[ct] = tr_read4(&qimg[w][off]);
      trG[ct] = tr_read4(&gimg[w][off]);
    }
    f32x4 dQa[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) dQa[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      if (kt < nk) {
        const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
        // swapped: rows = keys 4g+r of this tile, column = query i
        f32x4 sT = mfma_s(kl[kt], qf, z);
        sT = mfma_s(kf[kt], ql, sT);
        sT = mfma_s(kf[kt], qf, sT);
        const f32x4 dPT = mfma_s(vr[kt], gf, z);
        s16x4 dsT;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool ok = qok && (kt * 16 + 4 * g + r < Tk);
          const float p = ok ? __expf(sT[r] - lse_i) : 0.f;
          const float ds = p * (dPT[r] - dsum);
          dtau_acc += ok ? ds * sT[r] : 0.f;
          dsT[r] = f2bf(ds);
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const s16x4 trK = tr_read4(&kimg[w][(kt * 16 + 4 * g + (i >> 2)) * RB + ct * 32 + 8 * (i & 3)]);
          dQa[ct] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(dsT, trK, dQa[ct], 0, 0, 0);
        }
        // unswapped: rows = queries 4g+r, column = key i
        f32x4 sU = mfma_s(ql, kf[kt], z);
        sU = mfma_s(qf, kl[kt], sU);
        sU = mfma_s(qf, kf[kt], sU);
        const f32x4 dPU = mfma_s(gf, vr[kt], z);
        s16x4 pU, dsU;
        const bool kok = kt * 16 + i < Tk;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool ok = kok && (qt * 16 + 4 * g + r < Tq);
          const float p = ok ? __expf(sU[r] - lse_r[r]) : 0.f;
          pU[r] = f2bf(p);
          dsU[r] = f2bf(p * (dPU[r] - d_r[r]));
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          dVa[kt][ct] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pU, trG[ct], dVa[kt][ct], 0, 0, 0);
          dKa[kt][ct] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(dsU, trQ[ct], dKa[kt][ct], 0, 0, 0);
        }
      }
    }
    // dq = (dq-hat - q-hat (q-hat . dq-hat)) / |q| ; accumulators: rows = queries 4g+r, column = channel ct*16+i
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qs = qt * 16 + 4 * g + r;
      float qh[CT], dqh[CT], dot = 0.f;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        qh[ct] = bf2f(*reinterpret_cast<const short*>(&qimg[w][qs * RB + (ct * 16 + i) * 2])) * (1.0f / inv_tau);
        dqh[ct] = dQa[ct][r] * inv_tau;
        dot += qh[ct] * dqh[ct];
      }
      dot += __shfl_xor(dot, 1, 64);
      dot += __shfl_xor(dot, 2, 64);
      dot += __shfl_xor(dot, 4, 64);
      dot += __shfl_xor(dot, 8, 64);
      if (qs < Tq) {
        const float nrm = qnorm[w][qs];
        if (nrm <= 1e-12f) dot = 0.f;
        const float inv = 1.0f / nrm;
        __hip_bfloat16* p = dq + (int64_t)toks[0][qs] * lddq + hoff + i;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) p[ct * 16] = __float2bfloat16((dqh[ct] - qh[ct] * dot) * inv);
      }
    }
  }
  dtau_acc = wave_sum(dtau_acc);
  if (lane == 0) *dtp = dtau_acc;
  // ---- dk, dv: rows = keys 4g+r of tile kt, column = channel ct*16+i
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    if (kt < nk) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ks = kt * 16 + 4 * g + r;
        float kh[CT], dot = 0.f;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          kh[ct] = bf2f(*reinterpret_cast<const short*>(&kimg[w][ks * RB + (ct * 16 + i) * 2]));
          dot += kh[ct] * dKa[kt][ct][r];
        }
        dot += __shfl_xor(dot, 1, 64);
        dot += __shfl_xor(dot, 2, 64);
        dot += __shfl_xor(dot, 4, 64);
        dot += __shfl_xor(dot, 8, 64);
        if (ks < Tk) {
          const float nrm = knorm[w][ks];
          if (nrm <= 1e-12f) dot = 0.f;
          const float inv = 1.0f / nrm;
          const int tokk = toks[1][ks];
          __hip_bfloat16* p1 = dk + (int64_t)tokk * lddk + hoff + i;
          __hip_bfloat16* p2 = dv + (int64_t)tokk * lddv + hoff + i;
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) {
            p1[ct * 16] = __float2bfloat16((dKa[kt][ct][r] - kh[ct] * dot) * inv);
            p2[ct * 16] = __float2bfloat16(dVa[kt][ct][r]);
          }
        }
      }
    }
  }
}

int tmae_win_attn_bwd_mfma(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                           const void* out, int64_t ldo, const void* dout, int64_t lddo, const float* lse, int nhead,
                           int dh, const int32_t* grid_q, const int32_t* grid_k, int batch, int ny, int nx,
                           int do_shift, const float* tau, float tau_min, void* dq, int64_t lddq, void* dk,
                           int64_t lddk, void* dv, int64_t lddv, float* dtau_partial, hipStream_t stream) {
  if (nhead % 4 || (dh != 16 && dh != 32)) return TMAE_EARG;
  const int Wy = (ny + WIN - 1) / WIN + 1, Wx = (nx + WIN - 1) / WIN + 1;
  const int s = do_shift ? WIN / 2 : WIN;
  dim3 grid((unsigned)((int64_t)batch * Wy * Wx), (unsigned)(nhead / 4));
#define BWDM(DH)                                                                                                      \
  hipLaunchKernelGGL(win_attn_bwd_mfma_kernel<DH>, grid, dim3(256), 0, stream, (const __hip_bfloat16*)q, ldq,         \
                     (const __hip_bfloat16*)k, ldk, (const __hip_bfloat16*)v, ldv, (const __hip_bfloat16*)out, ldo,   \
                     (const __hip_bfloat16*)dout, lddo, lse, nhead, grid_q, grid_k, ny, nx, Wy, Wx, s, s, tau, tau_min, \
                     (__hip_bfloat16*)dq, lddq, (__hip_bfloat16*)dk, lddk, (__hip_bfloat16*)dv, lddv, dtau_partial)
  if (dh == 32) BWDM(32); else BWDM(16);
#undef BWDM
  return (int)hipGetLastError();
}

