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
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ s16x4 tr_read4(const char* lds_ptr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)lds_ptr);
}
__device__ __forceinline__ short f2bf(float v) {
  __hip_bfloat16 b = __float2bfloat16(v);
  return *reinterpret_cast<short*>(&b);
}
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
  return (unsigned)(unsigned short)f2bf(a) | ((unsigned)(unsigned short)f2bf(b) << 16);
}
__device__ __forceinline__ float bf2f(short s) { return __uint_as_float(((unsigned)(unsigned short)s) << 16); }
// 1/x in one instruction (1 ulp); an IEEE division costs ~10 VALU instructions and these kernels are VALU-bound
#define MASKED_LOGIT (-30000.0f)   // finite: 0 * MASKED_LOGIT stays 0 in the tau gradient
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// sum over the 16 lanes of a DPP row (lanes 16k .. 16k+15), result in every lane: four rotate-and-add steps on the
// VALU (a __shfl_xor is a ds_bpermute through the LDS crossbar)
__device__ __forceinline__ float row16_sum(float v) {
#define TMAE_ROR(n) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + (n), 0xF, 0xF, false))
  v += TMAE_ROR(8);
  v += TMAE_ROR(4);
  v += TMAE_ROR(2);
  v += TMAE_ROR(1);
#undef TMAE_ROR
  return v;
}

// Row fragment of a token: channels [FR*g, FR*g+FR) of head `hoff`, FR = 8 (dh 32) or 4 (dh 16).
// The load is UNCONDITIONAL (an absent token, tok < 0, reads row 0) and the zeroing happens at unpack time: a load under
// `if (tok >= 0)` makes the compiler wait for it (s_waitcnt vmcnt(0)) before the next one can be issued, which turned the
// K, V, Q, dO rows of a window into four back-to-back memory round trips.  All raw loads of a workgroup are issued
// first, then unpacked.
template <int FR> struct RawFrag;
template <> struct RawFrag<8> { typedef u32x4 T; };
template <> struct RawFrag<4> { typedef u32x2 T; };

template <int FR>
__device__ __forceinline__ typename RawFrag<FR>::T load_row_raw(const __hip_bfloat16* base, int64_t ld, int tok, int hoff,
                                                                int g) {
  const __hip_bfloat16* p = base + (int64_t)(tok < 0 ? 0 : tok) * ld + hoff + FR * g;
  return *reinterpret_cast<const typename RawFrag<FR>::T*>(p);
}

template <int FR>
__device__ __forceinline__ void unpack_row(const typename RawFrag<FR>::T& u, int tok, float* f) {
#pragma unroll
  for (int j = 0; j < FR / 2; ++j) {
    const unsigned w = tok < 0 ? 0u : u[j];
    f[2 * j] = __uint_as_float(w << 16);
    f[2 * j + 1] = __uint_as_float(w & 0xFFFF0000u);
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
  const float inv = scale * fast_rcp(nrm);
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


// Row images whose transposed reads feed the SWAPPED products (rows = channels, columns = tokens): tile ct, position
// 4a + b of an image row holds channel (DH/4) a + 4 ct + b, so that the lane group a = lane>>4 ends up with DH/4
// CONSECUTIVE channels of its token (one 16-byte store for dh 32, 8 bytes for dh 16) instead of one 2-byte store per
// (row, tile).  A lane owns the channels (DH/4) g .. of a row: for dh 32 they go to two 8-byte places, dh 16: identity.
template <int DH>
__device__ __forceinline__ void store_img_frag(char* row_base, int g, const typename Frag<DH>::T& f) {
  if constexpr (DH == 32) {
    s16x4 lo = {f[0], f[1], f[2], f[3]}, hi = {f[4], f[5], f[6], f[7]};
    *reinterpret_cast<s16x4*>(row_base + 8 * g) = lo;            // tile 0, positions 4g .. 4g+3
    *reinterpret_cast<s16x4*>(row_base + 32 + 8 * g) = hi;       // tile 1, positions 4g .. 4g+3
  } else {
    *reinterpret_cast<s16x4*>(row_base + 8 * g) = f;
  }
}

// FR consecutive channels of one token row as one 16-byte (FR 8) / 8-byte (FR 4) store
template <int FR>
__device__ __forceinline__ void store_row_frag(__hip_bfloat16* p, const float* f) {
  if constexpr (FR == 8) {
    uint4 u;
    u.x = pack_bf16x2(f[0], f[1]); u.y = pack_bf16x2(f[2], f[3]); u.z = pack_bf16x2(f[4], f[5]); u.w = pack_bf16x2(f[6], f[7]);
    *reinterpret_cast<uint4*>(p) = u;
  } else {
    uint2 u;
    u.x = pack_bf16x2(f[0], f[1]); u.y = pack_bf16x2(f[2], f[3]);
    *reinterpret_cast<uint2*>(p) = u;
  }
}

__device__ __forceinline__ f32x4 mfma_s(const s16x8& a, const s16x8& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a), *reinterpret_cast<const bf16x8*>(&b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma_s(const s16x4& a, const s16x4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------
// window work lists: the windows that hold both queries and keys, binned by the number of 16-token tiles they
// need (class 0: <=16 tokens, 1: <=32, 2: <=64).  This is the reference's "region batching" (drop levels
// 16/32/64, spt_backbone.py:47-71) reduced to three index lists -- no padded tensors.  The class selects a kernel
// instantiation whose LDS images / register tiles are sized for it, so the ~85 % of windows with <=16 tokens run
// at 4x the occupancy of a worst-case (64-token) workgroup.  layout: wl[0..2] = counts, wl[4 + c*nwin + j] = ids.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void win_class_kernel(const int32_t* __restrict__ grid_q,
                                                       const int32_t* __restrict__ grid_k, int batch, int ny, int nx,
                                                       int Wy, int Wx, int sy, int sx, int8_t* __restrict__ cls) {
  const int lane = threadIdx.x & 63;
  const int64_t nwin = (int64_t)batch * Wy * Wx;
  const int64_t dw = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (dw >= nwin) return;
  const int wcy = (int)(dw % Wy), wcx = (int)((dw / Wy) % Wx), b = (int)(dw / ((int64_t)Wy * Wx));
  const int y = wcy * WIN - sy + (lane >> 3), x = wcx * WIN - sx + (lane & 7);
  const bool in = y >= 0 && y < ny && x >= 0 && x < nx;
  const int64_t cell = ((int64_t)b * ny + y) * nx + x;
  const int tq = in ? grid_q[cell] : -1, tk = in ? grid_k[cell] : -1;
  const int Tq = __popcll(__ballot(tq >= 0)), Tk = __popcll(__ballot(tk >= 0));
  if (lane == 0) {
    const int t = max(Tq, Tk);
    cls[dw] = (Tq > 0 && Tk > 0) ? (int8_t)(t <= 16 ? 0 : (t <= 32 ? 1 : 2)) : (int8_t)-1;
  }
}

// compaction: one thread per window; a wave reserves its range with ONE atomic per class (list order only affects
// scheduling, never results)
__global__ __launch_bounds__(256) void win_worklist_kernel(const int8_t* __restrict__ cls, int64_t nwin,
                                                          int32_t* __restrict__ wl) {
  const int lane = threadIdx.x & 63;
  const int64_t dw = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int c = dw < nwin ? (int)cls[dw] : -1;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const unsigned long long m = __ballot(c == k);
    if (m == 0ull) continue;
    int base = 0;
    if (lane == 0) base = atomicAdd(wl + k, __popcll(m));
    base = __shfl(base, 0, 64);
    if (c == k) wl[4 + (int64_t)k * nwin + base + __popcll(m & ((1ull << lane) - 1ull))] = (int32_t)dw;
  }
}

size_t tmae_window_worklist_size(int batch, int ny, int nx) {
  const int Wy = (ny + WIN - 1) / WIN + 1, Wx = (nx + WIN - 1) / WIN + 1;
  const size_t nwin = (size_t)batch * Wy * Wx;
  return 4 + 3 * nwin + (nwin + 3) / 4;                 // counts | three lists | class bytes (scratch)
}

int tmae_window_worklist(const int32_t* grid_q, const int32_t* grid_k, int batch, int ny, int nx, int do_shift,
                         int32_t* worklist, void* stream_) {
  (void)hipGetLastError();
  hipStream_t stream = (hipStream_t)stream_;
  if (!grid_q || !grid_k || !worklist || batch <= 0 || ny <= 0 || nx <= 0) return TMAE_EARG;
  const int Wy = (ny + WIN - 1) / WIN + 1, Wx = (nx + WIN - 1) / WIN + 1;
  const int s = do_shift ? WIN / 2 : WIN;
  const int64_t nwin = (int64_t)batch * Wy * Wx;
  int8_t* cls = reinterpret_cast<int8_t*>(worklist + 4 + 3 * nwin);
  hipMemsetAsync(worklist, 0, 16, stream);
  hipLaunchKernelGGL(win_class_kernel, dim3(tmae_cdiv(nwin, 4)), dim3(256), 0, stream, grid_q, grid_k, batch, ny, nx,
                     Wy, Wx, s, s, cls);
  hipLaunchKernelGGL(win_worklist_kernel, dim3(tmae_cdiv(nwin, 256)), dim3(256), 0, stream, cls, nwin, worklist);
  return (int)hipGetLastError();
}

// resolves the window of this workgroup; returns false when the block has nothing to do
__device__ __forceinline__ bool pick_window(const int32_t* __restrict__ wl, int cls, int64_t nwin, int64_t& dw) {
  if (wl) {
    if ((int)blockIdx.x >= wl[cls]) return false;
    dw = wl[4 + (int64_t)cls * nwin + blockIdx.x];
  } else {
    dw = blockIdx.x;
  }
  return true;
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <int DH, int NT>
__global__ __launch_bounds__(256) void win_attn_fwd_mfma_kernel(
    const __hip_bfloat16* __restrict__ q, int64_t ldq, const __hip_bfloat16* __restrict__ k, int64_t ldk,
    const __hip_bfloat16* __restrict__ v, int64_t ldv, int nhead, const int32_t* __restrict__ grid_q,
    const int32_t* __restrict__ grid_k, int ny, int nx, int Wy, int Wx, int sy, int sx,
    const float* __restrict__ tau, float tau_min, __hip_bfloat16* __restrict__ out, int64_t ldo,
    float* __restrict__ lse, const int32_t* __restrict__ wl, int cls, int64_t nwin) {
  constexpr int FR = DH / 4;                 // channels per lane in a row fragment
  constexpr int CT = DH / 16;                // 16-channel output tiles
  constexpr int RB = DH * 2 + 16;            // V image row pitch (bytes): 16-byte aligned, off the power of two
  constexpr int ROWS = NT * 16;
  typedef typename Frag<DH>::T frag_t;
  __shared__ int toks[2][64];
  __shared__ __attribute__((aligned(16))) char vimg[4][ROWS * RB];
  int64_t dw;
  if (!pick_window(wl, cls, nwin, dw)) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, g = lane >> 4, i = lane & 15;
  const int head = blockIdx.y * 4 + w, hoff = head * DH;
  const int wcy = (int)(dw % Wy), wcx = (int)((dw / Wy) % Wx), b = (int)(dw / ((int64_t)Wy * Wx));
  const int y = wcy * WIN - sy + (lane >> 3), x = wcx * WIN - sx + (lane & 7);
  const bool in = y >= 0 && y < ny && x >= 0 && x < nx;
  const int64_t cell = ((int64_t)b * ny + y) * nx + x;
  const int tq = in ? grid_q[cell] : -1, tk = in ? grid_k[cell] : -1;
  const unsigned long long mq = __ballot(tq >= 0), mk = __ballot(tk >= 0);
  if (mq == 0ull) return;
  const int Tq = __popcll(mq), Tk = __popcll(mk);
  if (Tk == 0) {                              // (dense launch only) cross-attention window without keys: zero rows
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
  __syncthreads();
  const int nq = (Tq + 15) >> 4, nk = (Tk + 15) >> 4;
  const float inv_tau = 1.0f / fmaxf(tau[0], tau_min);
  // ---- all global row loads are issued here, one dependent round after the token ids
  frag_t kf[NT], kl[NT], qf[NT], ql[NT];
  typedef typename RawFrag<FR>::T raw_t;
  raw_t rk[NT], rv[NT], rq[NT];
  int tokk_[NT], tokq_[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {                    // no branches here: every load of the wave is in flight at once
    const int slot = t * 16 + i;
    tokk_[t] = slot < Tk ? toks[1][slot] : -1;
    tokq_[t] = slot < Tq ? toks[0][slot] : -1;
    rk[t] = load_row_raw<FR>(k, ldk, tokk_[t], hoff, g);
    rv[t] = load_row_raw<FR>(v, ldv, tokk_[t], hoff, g);
    rq[t] = load_row_raw<FR>(q, ldq, tokq_[t], hoff, g);
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int slot = t * 16 + i;
    float f[FR];
    if (t < nk) {                                   // wave-uniform
      unpack_row<FR>(rk[t], tokk_[t], f);
      normalize_frag<FR>(f, 1.0f);
      split_frag<FR>(f, kf[t], kl[t]);
      unpack_row<FR>(rv[t], tokk_[t], f);
      store_img_frag<DH>(&vimg[w][slot * RB], g, pack_frag<FR>(f));
    } else {
#pragma unroll
      for (int j = 0; j < FR; ++j) { kf[t][j] = 0; kl[t][j] = 0; }
      frag_t z;
#pragma unroll
      for (int j = 0; j < FR; ++j) z[j] = 0;
      store_img_frag<DH>(&vimg[w][slot * RB], g, z);                        // P is 0 there, but 0 * garbage = NaN
    }
    if (t < nq) {
      unpack_row<FR>(rq[t], tokq_[t], f);
      normalize_frag<FR>(f, inv_tau);
      split_frag<FR>(f, qf[t], ql[t]);
    }
  }
  __syncthreads();
  // V fragments (A operand of O^T = V^T.P^T): [key tile][channel tile], 4 keys x 1 (permuted) channel per lane
  s16x4 vf[NT][CT];
#pragma unroll
  for (int kt = 0; kt < NT; ++kt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
      vf[kt][ct] = tr_read4(&vimg[w][(kt * 16 + 4 * g + (i >> 2)) * RB + ct * 32 + 8 * (i & 3)]);

  // key rows past Tk get a large negative logit through the MFMA's C operand (exp underflows to exactly 0): no
  // per-element masking in the loops, and tiles kt >= nk need no special case
  // (only where the 4*NT registers are free: the 64-token class would lose a resident wave and masks per element)
  constexpr bool KB = NT <= 2;
  f32x4 kbias[KB ? NT : 1];
  if constexpr (KB) {
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) kbias[kt][r] = (kt * 16 + 4 * g + r < Tk) ? 0.f : MASKED_LOGIT;
  }
#pragma unroll
  for (int qt = 0; qt < NT; ++qt) {
    if (qt < nq) {
      const int qslot = qt * 16 + i;
      f32x4 st[NT];
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
        if constexpr (KB) st[kt] = kbias[kt]; else st[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (kt < nk) {
          st[kt] = mfma_s(kl[kt], qf[qt], st[kt]);                   // S^T tile: rows = keys 4g+r, col = query i
          st[kt] = mfma_s(kf[kt], ql[qt], st[kt]);
          st[kt] = mfma_s(kf[kt], qf[qt], st[kt]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if constexpr (!KB) { if (kt * 16 + 4 * g + r >= Tk) st[kt][r] = MASKED_LOGIT; }
          mx = fmaxf(mx, st[kt][r]);
        }
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float l = 0.f;
      s16x4 pf[NT];
#pragma unroll
      for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __expf(st[kt][r] - mx);
          l += p;
          pf[kt][r] = f2bf(p);
        }
      l += __shfl_xor(l, 16, 64);
      l += __shfl_xor(l, 32, 64);
      const float invl = fast_rcp(l);
      // O^T = V^T . P^T (swapped: rows = channels, column = query i): the V fragments serve as the A operand, the
      // probabilities stay where the S^T accumulators left them
      f32x4 o[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        o[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
          if (kt < nk) o[ct] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(vf[kt][ct], pf[kt], o[ct], 0, 0, 0);
      }
      // lane (g, i): query i, channels (DH/4) g + 4 ct + r -- DH/4 consecutive channels: one wide store
      if (qslot < Tq) {
        float of[FR];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) of[4 * ct + r] = o[ct][r] * invl;
        store_row_frag<FR>(out + (int64_t)toks[0][qslot] * ldo + hoff + FR * g, of);
      }
      if (g == 0 && qslot < Tq) lse[(int64_t)toks[0][qslot] * nhead + head] = mx + __logf(l);
    }
  }
}

static int64_t class_grid(int cls, int64_t nwin, int64_t mq, int64_t mk) {
  // class c windows hold > 16*c tokens (c = 1: >= 17, c = 2: >= 33) in one of the two frames
  const int64_t bound = cls == 0 ? nwin : (mq + mk) / (cls == 1 ? 17 : 33);
  return bound < nwin ? bound : nwin;
}

int tmae_win_attn_fwd_mfma(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                           int64_t mq, int64_t mk, int nhead, int dh, const int32_t* grid_q, const int32_t* grid_k,
                           int batch, int ny, int nx, int do_shift, const float* tau, float tau_min, void* out,
                           int64_t ldo, float* lse, const int32_t* worklist, hipStream_t stream) {
  if (nhead % 4 || (dh != 16 && dh != 32)) return TMAE_EARG;
  // 16-byte row fragments / V rows: bases and pitches must keep every head slice 16-byte aligned
  if ((ldq % 8) || (ldk % 8) || (ldv % 8) || ((uintptr_t)q & 15) || ((uintptr_t)k & 15) || ((uintptr_t)v & 15))
    return TMAE_EARG;
  const int Wy = (ny + WIN - 1) / WIN + 1, Wx = (nx + WIN - 1) / WIN + 1;
  const int s = do_shift ? WIN / 2 : WIN;
  const int64_t nwin = (int64_t)batch * Wy * Wx;
#define FWDM(DH, NT, CLS, GX)                                                                                        \
  hipLaunchKernelGGL((win_attn_fwd_mfma_kernel<DH, NT>), dim3((unsigned)(GX), (unsigned)(nhead / 4)), dim3(256), 0,  \
                     stream, (const __hip_bfloat16*)q, ldq, (const __hip_bfloat16*)k, ldk, (const __hip_bfloat16*)v,  \
                     ldv, nhead, grid_q, grid_k, ny, nx, Wy, Wx, s, s, tau, tau_min, (__hip_bfloat16*)out, ldo, lse,  \
                     worklist, CLS, nwin)
  if (!worklist) {
    if (dh == 32) FWDM(32, 4, 0, nwin); else FWDM(16, 4, 0, nwin);
  } else {
    const int64_t g0 = class_grid(0, nwin, mq, mk), g1 = class_grid(1, nwin, mq, mk), g2 = class_grid(2, nwin, mq, mk);
    if (dh == 32) {
      if (g0 > 0) FWDM(32, 1, 0, g0);
      if (g1 > 0) FWDM(32, 2, 1, g1);
      if (g2 > 0) FWDM(32, 4, 2, g2);
    } else {
      if (g0 > 0) FWDM(16, 1, 0, g0);
      if (g1 > 0) FWDM(16, 2, 1, g1);
      if (g2 > 0) FWDM(16, 4, 2, g2);
    }
  }
#undef FWDM
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// backward (bf16): P is recomputed from the saved log-sum-exp with the query on the lane column; dV = P^T.dO and
// dK-hat = dS^T.Q-hat need the key on the lane column, which a 16x16 bf16 tile written with one 8-byte store per lane
// and read back with ds_read_b64_tr_b16 provides.  All three gradient products are taken TRANSPOSED (rows = channels,
// column = token) with the row-major LDS images (K-hat, Q-hat/tau, dO; channel-permuted, see store_img_frag) as the A
// operand: a lane ends up with DH/4 consecutive channels of ONE token -- the same channels as its own row fragment --
// so the normalisation Jacobian reads q-hat / k-hat from registers and every gradient row leaves in 16-byte
// (dh 32) / 8-byte (dh 16) stores.  (With rows = tokens each lane wrote 2 bytes per (row, tile): the L2 request
// rate of those stores was 16 % of the stage-1 kernel.)
// NT = 4 is compiled for two waves per SIMD (the LDS images allow no more): 233 registers, no scratch.
// ------------------------------------------------------------------------------------------------
template <int DH, int NT>
__global__ __launch_bounds__(256, NT >= 4 ? 2 : 1) void win_attn_bwd_mfma_kernel(
    const __hip_bfloat16* __restrict__ q, int64_t ldq, const __hip_bfloat16* __restrict__ k, int64_t ldk,
    const __hip_bfloat16* __restrict__ v, int64_t ldv, const __hip_bfloat16* __restrict__ outp, int64_t ldo,
    const __hip_bfloat16* __restrict__ dout, int64_t lddo, const float* __restrict__ lse, int nhead,
    const int32_t* __restrict__ grid_q, const int32_t* __restrict__ grid_k, int ny, int nx, int Wy, int Wx, int sy,
    int sx, const float* __restrict__ tau, float tau_min, __hip_bfloat16* __restrict__ dq, int64_t lddq,
    __hip_bfloat16* __restrict__ dk, int64_t lddk, __hip_bfloat16* __restrict__ dv, int64_t lddv,
    float* __restrict__ dtau_partial, const int32_t* __restrict__ wl, int cls, int64_t nwin) {
  constexpr int FR = DH / 4;
  constexpr int CT = DH / 16;
  constexpr int RB = DH * 2 + 16;
  constexpr int ROWS = NT * 16;
  typedef typename Frag<DH>::T frag_t;
  __shared__ int toks[2][64];
  __shared__ __attribute__((aligned(16))) char kimg[4][ROWS * RB];   // K-hat
  __shared__ __attribute__((aligned(16))) char qimg[4][ROWS * RB];   // Q-hat / tau_c
  __shared__ __attribute__((aligned(16))) char gimg[4][ROWS * RB];   // dO
  __shared__ float qnorm[4][ROWS], knorm[4][ROWS];
  constexpr int TP = 40;                                              // pitch (bytes) of the 16x16 bf16 transpose tiles
  __shared__ __attribute__((aligned(16))) char ptile[4][2][16 * TP];
  int64_t dw;
  if (!pick_window(wl, cls, nwin, dw)) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, g = lane >> 4, i = lane & 15;
  const int head = blockIdx.y * 4 + w, hoff = head * DH;
  const int wcy = (int)(dw % Wy), wcx = (int)((dw / Wy) % Wx), b = (int)(dw / ((int64_t)Wy * Wx));
  const int y = wcy * WIN - sy + (lane >> 3), x = wcx * WIN - sx + (lane & 7);
  const bool in = y >= 0 && y < ny && x >= 0 && x < nx;
  const int64_t cell = ((int64_t)b * ny + y) * nx + x;
  const int tq = in ? grid_q[cell] : -1, tk = in ? grid_k[cell] : -1;
  const unsigned long long mq = __ballot(tq >= 0), mk = __ballot(tk >= 0);
  const int Tq = __popcll(mq), Tk = __popcll(mk);
  float* dtp = dtau_partial + dw * nhead + head;
  if (Tq == 0 || Tk == 0) {                   // (dense launch only) nothing attended here: zero gradients
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
  const float tau_c = fmaxf(tau[0], tau_min), inv_tau = 1.0f / tau_c;
  if (w == 0) {
    if (tq >= 0) toks[0][__popcll(mq & ((1ull << lane) - 1ull))] = tq;
    if (tk >= 0) toks[1][__popcll(mk & ((1ull << lane) - 1ull))] = tk;
  }
  __syncthreads();                                   // token lists visible
  const int nq = (Tq + 15) >> 4, nk = (Tk + 15) >> 4;
  // ---- every global row load of the workgroup is issued here (one dependent round after the token ids); the
  //      row-major LDS images are written from the same 16-byte fragments: lane (g,i) owns chunk g of row tile*16+i.
  frag_t kf[NT], kl[NT], vr[NT], qf[NT], ql[NT], gf[NT];
  float lse_i[NT];
  typedef typename RawFrag<FR>::T raw_t;
  raw_t rk[NT], rv[NT], rq[NT], rg[NT];
  int tokk_[NT], tokq_[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {                     // no branches here: every load of the wave is in flight at once
    const int slot = t * 16 + i;
    tokk_[t] = slot < Tk ? toks[1][slot] : -1;
    tokq_[t] = slot < Tq ? toks[0][slot] : -1;
    rk[t] = load_row_raw<FR>(k, ldk, tokk_[t], hoff, g);
    rv[t] = load_row_raw<FR>(v, ldv, tokk_[t], hoff, g);
    rq[t] = load_row_raw<FR>(q, ldq, tokq_[t], hoff, g);
    rg[t] = load_row_raw<FR>(dout, lddo, tokq_[t], hoff, g);
    lse_i[t] = lse[(int64_t)(tokq_[t] < 0 ? 0 : tokq_[t]) * nhead + head];
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int slot = t * 16 + i;
    float f[FR];
    if (t < nk) {                                    // wave-uniform
      unpack_row<FR>(rk[t], tokk_[t], f);
      const float nrm = normalize_frag<FR>(f, 1.0f);
      split_frag<FR>(f, kf[t], kl[t]);
      if (g == 0) knorm[w][slot] = nrm;
      unpack_row<FR>(rv[t], tokk_[t], f);
      vr[t] = pack_frag<FR>(f);
    } else {
#pragma unroll
      for (int j = 0; j < FR; ++j) { kf[t][j] = 0; kl[t][j] = 0; vr[t][j] = 0; }
    }
    store_img_frag<DH>(&kimg[w][slot * RB], g, kf[t]);
    if (tokq_[t] < 0 || t >= nq) lse_i[t] = INFINITY;                      // no query: p = exp(s - inf) = 0
    if (t < nq) {
      unpack_row<FR>(rq[t], tokq_[t], f);
      const float nrm = normalize_frag<FR>(f, inv_tau);
      split_frag<FR>(f, qf[t], ql[t]);
      store_img_frag<DH>(&qimg[w][slot * RB], g, qf[t]);
      if (g == 0) qnorm[w][slot] = nrm;
      float gfl[FR];
      unpack_row<FR>(rg[t], tokq_[t], gfl);
      gf[t] = pack_frag<FR>(gfl);                                        // exact: bf16 -> f32 -> bf16
      store_img_frag<DH>(&gimg[w][slot * RB], g, gf[t]);
    }
  }
  __syncthreads();
  f32x4 dKa[NT][CT], dVa[NT][CT];
#pragma unroll
  for (int kt = 0; kt < NT; ++kt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) { dKa[kt][ct] = f32x4{0.f, 0.f, 0.f, 0.f}; dVa[kt][ct] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  float dtau_acc = 0.f;
  constexpr bool KB = NT <= 2;             // MASKED_LOGIT on key rows past Tk through the C operand (see the forward)
  f32x4 kbias[KB ? NT : 1];
  if constexpr (KB) {
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) kbias[kt][r] = (kt * 16 + 4 * g + r < Tk) ? 0.f : MASKED_LOGIT;
  }

#pragma unroll
  for (int qt = 0; qt < NT; ++qt) {
    if (qt < nq) {
      const int qslot = qt * 16 + i;
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
      // pass 1 over the key tiles: logits, probabilities and dP; D_i = sum_j P_ij dP_ij is taken from THESE values
      // (not from dO . O with the bf16-rounded saved output), so that sum_j dS_ij = 0 holds to fp32 rounding --
      // the tau gradient sum_ij dS_ij s_ij is a difference of large terms and is biased otherwise
      // NT == 4: the tiles are recomputed in pass 2 (4 MFMAs + 4 exps per tile pair, far below the budget of this
      // latency-bound kernel) instead of being held in 48 registers, which keeps two waves per SIMD resident.
      constexpr bool RECOMP = NT >= 4;
      f32x4 sTk[RECOMP ? 1 : NT], pTk[RECOMP ? 1 : NT], dPk[RECOMP ? 1 : NT];
      float dacc = 0.f;
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
        if (kt < nk) {
          const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
          // swapped: rows = keys 4g+r of this tile, column = query i; masked keys / absent queries give p = 0
          f32x4 sT = mfma_s(kl[kt], qf[qt], KB ? kbias[KB ? kt : 0] : z);
          sT = mfma_s(kf[kt], ql[qt], sT);
          sT = mfma_s(kf[kt], qf[qt], sT);
          const f32x4 dP = mfma_s(vr[kt], gf[qt], z);
          if constexpr (!RECOMP) { sTk[kt] = sT; dPk[kt] = dP; }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float p = __expf(sT[r] - lse_i[qt]);
            if constexpr (!KB) { if (kt * 16 + 4 * g + r >= Tk) p = 0.f; }
            if constexpr (!RECOMP) pTk[kt][r] = p;
            dacc += p * dP[r];
          }
        }
      }
      dacc += __shfl_xor(dacc, 16, 64);
      dacc += __shfl_xor(dacc, 32, 64);
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
        if (kt < nk) {
          f32x4 sT, dP, pT;
          if constexpr (RECOMP) {
            const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
            sT = mfma_s(kl[kt], qf[qt], z);
            sT = mfma_s(kf[kt], ql[qt], sT);
            sT = mfma_s(kf[kt], qf[qt], sT);
            dP = mfma_s(vr[kt], gf[qt], z);
#pragma unroll
            for (int r = 0; r < 4; ++r) pT[r] = (kt * 16 + 4 * g + r < Tk) ? __expf(sT[r] - lse_i[qt]) : 0.f;
          } else {
            sT = sTk[kt]; dP = dPk[kt]; pT = pTk[kt];
          }
          s16x4 dsT, pTb;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = pT[r];
            const float ds = p * (dP[r] - dacc);
            dtau_acc += ds * sT[r];                        // p = 0 entries: 0 * finite
            dsT[r] = f2bf(ds);
            pTb[r] = f2bf(p);
          }
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) {
            const s16x4 trK = tr_read4(&kimg[w][(kt * 16 + 4 * g + (i >> 2)) * RB + ct * 32 + 8 * (i & 3)]);
            dQa[ct] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(trK, dsT, dQa[ct], 0, 0, 0);   // dQ-hat^T
          }
          // the key-on-lane factors for dV / dK-hat: P and dS of this tile transposed through a 16x16 LDS tile
          // (one 8-byte store + one hardware-transposing read each) instead of recomputing exp() per element
          *reinterpret_cast<s16x4*>(&ptile[w][0][i * TP + 8 * g]) = pTb;
          *reinterpret_cast<s16x4*>(&ptile[w][1][i * TP + 8 * g]) = dsT;
          __builtin_amdgcn_wave_barrier();               // same wave: LDS ops execute in order
          const int toff = (4 * g + (i >> 2)) * TP + 8 * (i & 3);
          const s16x4 pU = tr_read4(&ptile[w][0][toff]);   // P^T  [key = i][query 4g+j]
          const s16x4 dsU = tr_read4(&ptile[w][1][toff]);  // dS^T [key = i][query 4g+j]
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) {
            dVa[kt][ct] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(trG[ct], pU, dVa[kt][ct], 0, 0, 0);    // dV^T
            dKa[kt][ct] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(trQ[ct], dsU, dKa[kt][ct], 0, 0, 0);   // dK-hat^T
          }
        }
      }
      // dq = (dq-hat - q-hat (q-hat . dq-hat)) / |q|.  The products are taken transposed (rows = channels) over the
      // permuted images, so lane (g, i) holds channels FR g + 4 ct + r of query qt*16+i: the SAME channels as its own
      // row fragment qf[qt] -- q-hat comes from registers, the dot product needs two shuffles, one wide store per lane
      {
        float dqh[FR], qh[FR], dot = 0.f;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            qh[4 * ct + r] = bf2f(qf[qt][4 * ct + r]) * tau_c;
            dqh[4 * ct + r] = dQa[ct][r] * inv_tau;
            dot += qh[4 * ct + r] * dqh[4 * ct + r];
          }
        dot += __shfl_xor(dot, 16, 64);
        dot += __shfl_xor(dot, 32, 64);
        if (qslot < Tq) {
          const float nrm = qnorm[w][qslot];
          if (nrm <= 1e-12f) dot = 0.f;
          const float inv = fast_rcp(nrm);
          float o[FR];
#pragma unroll
          for (int j = 0; j < FR; ++j) o[j] = (dqh[j] - qh[j] * dot) * inv;
          store_row_frag<FR>(dq + (int64_t)toks[0][qslot] * lddq + hoff + FR * g, o);
        }
      }
    }
  }
  dtau_acc = wave_sum(dtau_acc);
  if (lane == 0) *dtp = dtau_acc;
  // ---- dk, dv (transposed accumulators): lane (g, i) holds channels FR g + 4 ct + r of key kt*16+i
#pragma unroll
  for (int kt = 0; kt < NT; ++kt) {
    if (kt < nk) {
      const int ks = kt * 16 + i;
      float kh[FR], dkh[FR], dvv[FR], dot = 0.f;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          kh[4 * ct + r] = bf2f(kf[kt][4 * ct + r]);
          dkh[4 * ct + r] = dKa[kt][ct][r];
          dvv[4 * ct + r] = dVa[kt][ct][r];
          dot += kh[4 * ct + r] * dkh[4 * ct + r];
        }
      dot += __shfl_xor(dot, 16, 64);
      dot += __shfl_xor(dot, 32, 64);
      if (ks < Tk) {
        const float nrm = knorm[w][ks];
        if (nrm <= 1e-12f) dot = 0.f;
        const float inv = fast_rcp(nrm);
        const int tokk = toks[1][ks];
        float o[FR];
#pragma unroll
        for (int j = 0; j < FR; ++j) o[j] = (dkh[j] - kh[j] * dot) * inv;
        store_row_frag<FR>(dk + (int64_t)tokk * lddk + hoff + FR * g, o);
        store_row_frag<FR>(dv + (int64_t)tokk * lddv + hoff + FR * g, dvv);
      }
    }
  }
}

int tmae_win_attn_bwd_mfma(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                           const void* out, int64_t ldo, const void* dout, int64_t lddo, const float* lse, int64_t mq,
                           int64_t mk, int nhead, int dh, const int32_t* grid_q, const int32_t* grid_k, int batch,
                           int ny, int nx, int do_shift, const float* tau, float tau_min, void* dq, int64_t lddq,
                           void* dk, int64_t lddk, void* dv, int64_t lddv, float* dtau_partial,
                           const int32_t* worklist, hipStream_t stream) {
  if (nhead % 4 || (dh != 16 && dh != 32)) return TMAE_EARG;
  const int Wy = (ny + WIN - 1) / WIN + 1, Wx = (nx + WIN - 1) / WIN + 1;
  const int s = do_shift ? WIN / 2 : WIN;
  const int64_t nwin = (int64_t)batch * Wy * Wx;
#define BWDM(DH, NT, CLS, GX)                                                                                        \
  hipLaunchKernelGGL((win_attn_bwd_mfma_kernel<DH, NT>), dim3((unsigned)(GX), (unsigned)(nhead / 4)), dim3(256), 0,  \
                     stream, (const __hip_bfloat16*)q, ldq, (const __hip_bfloat16*)k, ldk, (const __hip_bfloat16*)v,  \
                     ldv, (const __hip_bfloat16*)out, ldo, (const __hip_bfloat16*)dout, lddo, lse, nhead, grid_q,     \
                     grid_k, ny, nx, Wy, Wx, s, s, tau, tau_min, (__hip_bfloat16*)dq, lddq, (__hip_bfloat16*)dk, lddk, \
                     (__hip_bfloat16*)dv, lddv, dtau_partial, worklist, CLS, nwin)
  if (!worklist) {
    if (dh == 32) BWDM(32, 4, 0, nwin); else BWDM(16, 4, 0, nwin);
  } else {
    const int64_t g0 = class_grid(0, nwin, mq, mk), g1 = class_grid(1, nwin, mq, mk), g2 = class_grid(2, nwin, mq, mk);
    if (dh == 32) {
      if (g0 > 0) BWDM(32, 1, 0, g0);
      if (g1 > 0) BWDM(32, 2, 1, g1);
      if (g2 > 0) BWDM(32, 4, 2, g2);
    } else {
      if (g0 > 0) BWDM(16, 1, 0, g0);
      if (g1 > 0) BWDM(16, 2, 1, g1);
      if (g2 > 0) BWDM(16, 4, 2, g2);
    }
  }
#undef BWDM
  return (int)hipGetLastError();
}
