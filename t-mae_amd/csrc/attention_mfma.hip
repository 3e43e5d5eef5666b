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
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
// two floats -> one dword of two bf16 (round to nearest even): ONE v_cvt_pk_bf16_f32 (the scalar conversions compiled to a
// convert per element plus shifts and ors to pack them)
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}
__device__ __forceinline__ float bf2f(short s) { return __uint_as_float(((unsigned)(unsigned short)s) << 16); }
// exp(s - m) with m given as m * log2(e): one fused multiply-add and the bare v_exp_f32 -- __expf(s - m) compiles to subtract,
// multiply, v_exp_f32.  m = +inf (no query) gives exp2(-inf) = 0, MASKED_LOGIT underflows to exactly 0 as before.
// Scalar on purpose, and this file is compiled with -fno-slp-vectorize (build.py): v_pk_*_f32 beside MFMAs costs several times
// the two scalar instructions it replaces (MI355X_MICROARCH.md, per-instruction constants; same-box A/B of the packed form of
// the loops below: +0.2 ms per step).
#define LOG2E_F 1.4426950408889634f
__device__ __forceinline__ float exp_minus(const float s, const float m_log2e) {
  return __builtin_amdgcn_exp2f(__builtin_fmaf(s, LOG2E_F, -m_log2e));
}
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

// Row fragment of a token: channels [FR*g, FR*g+FR) of head `hoff`, FR = 8 (dh 32) or 4 (dh 16), fetched with a
// BUFFER load: the address is a wave-uniform descriptor + a 32-bit byte offset (one v_mad_u32_u24 instead of the
// 64-bit multiply-add chain of a flat address -- these kernels are VALU-bound, profiles/round2_attention_pmc.md), and
// an absent token (tok < 0) gets the offset 0xFFFFFFFF: the descriptor's range check returns zeros, so there is no
// select at unpack time and no branch around the load (a load under `if (tok >= 0)` is waited for before the next one
// is issued).  All raw loads of a workgroup are issued first, then unpacked.
template <int FR> struct RawFrag;
template <> struct RawFrag<8> { typedef u32x4 T; };
template <> struct RawFrag<4> { typedef u32x2 T; };

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
// byte offset of (token row, byte column); rows and pitches stay below 2^24 (checked by the launcher)
__device__ __forceinline__ unsigned row_off(int tok, unsigned ld_bytes, unsigned col_bytes) {
  return tok < 0 ? 0xFFFFFFFFu : __umul24((unsigned)tok, ld_bytes) + col_bytes;
}
template <int FR>
__device__ __forceinline__ typename RawFrag<FR>::T load_row_raw(__amdgpu_buffer_rsrc_t rs, unsigned off) {
  if constexpr (FR == 8) return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
  else return __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, 0, 0));
}

template <int FR>
__device__ __forceinline__ void unpack_row(const typename RawFrag<FR>::T& u, float* f) {
#pragma unroll
  for (int j = 0; j < FR / 2; ++j) {
    f[2 * j] = __uint_as_float(u[j] << 16);
    f[2 * j + 1] = __uint_as_float(u[j] & 0xFFFF0000u);
  }
}

// reductions over the 4 lanes {i, i+16, i+32, i+48} that share a token row: v_permlane16_swap / v_permlane32_swap
// (gfx950) exchange register rows inside the VALU -- a __shfl_xor is an address computation plus a ds_bpermute
// through the LDS crossbar and a wait.  swap(v, v) returns (even-row copy, odd-row copy): their sum / max is the
// xor-16 (xor-32) butterfly step for every lane.
__device__ __forceinline__ float quad_sum(float v) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float quad_max(float v) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// L2-normalise a token row that is spread over the 4 lane groups (same lane&15), times `scale`.
template <int FR>
__device__ __forceinline__ float normalize_frag(float* f, float scale) {
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < FR; ++j) ss += f[j] * f[j];
  ss = quad_sum(ss);
  const float nrm = fmaxf(__builtin_amdgcn_sqrtf(ss), 1e-12f);   // v_sqrt_f32 (1 ulp); sqrtf() expands to ~17 instructions
  const float inv = scale * fast_rcp(nrm);
#pragma unroll
  for (int j = 0; j < FR; ++j) f[j] *= inv;
  return nrm;                                   // max(|x|, eps) of the whole row
}

template <int DH> struct Frag;
template <> struct Frag<32> { typedef s16x8 T; };
template <> struct Frag<16> { typedef s16x4 T; };

// a raw row fragment IS the MFMA fragment of its bf16 values (same element order): V and dO go straight from the load to the
// matrix core -- rounds 1-4 unpacked them to fp32 and packed them again (exact, and 20 VALU instructions per fragment)
template <int FR>
__device__ __forceinline__ typename Frag<FR * 4>::T raw_as_frag(const typename RawFrag<FR>::T& u) {
  return __builtin_bit_cast(typename Frag<FR * 4>::T, u);
}


template <int FR>
__device__ __forceinline__ typename Frag<FR * 4>::T pack_frag(const float* f) {
  typename RawFrag<FR>::T u;
#pragma unroll
  for (int j = 0; j < FR / 2; ++j) u[j] = pack_bf16x2(f[2 * j], f[2 * j + 1]);
  return __builtin_bit_cast(typename Frag<FR * 4>::T, u);
}

// hi/lo split of an fp32 fragment into two bf16 fragments (f ~= hi + lo): three MFMAs (hi.hi + hi.lo + lo.hi) give
// the cosine logits to ~2^-16 relative instead of bf16's 2^-8 -- they are divided by tau >= 0.01 before the exp.
template <int FR>
__device__ __forceinline__ void split_frag(const float* f, typename Frag<FR * 4>::T& hi, typename Frag<FR * 4>::T& lo) {
  typename RawFrag<FR>::T uh, ul;
#pragma unroll
  for (int j = 0; j < FR / 2; ++j) {
    const unsigned h = pack_bf16x2(f[2 * j], f[2 * j + 1]);
    uh[j] = h;
    ul[j] = pack_bf16x2(f[2 * j] - __uint_as_float(h << 16), f[2 * j + 1] - __uint_as_float(h & 0xFFFF0000u));
  }
  hi = __builtin_bit_cast(typename Frag<FR * 4>::T, uh);
  lo = __builtin_bit_cast(typename Frag<FR * 4>::T, ul);
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

// FR consecutive channels of one token row as one 16-byte (FR 8) / 8-byte (FR 4) buffer store; an offset of
// 0xFFFFFFFF (absent token) is dropped by the range check: no branch around the store
template <int FR>
__device__ __forceinline__ void store_row_frag(__amdgpu_buffer_rsrc_t rs, unsigned off, const float* f) {
  if constexpr (FR == 8) {
    u32x4 u = {pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7])};
    __builtin_amdgcn_raw_buffer_store_b128(u, rs, (int)off, 0, 0);
  } else {
    u32x2 u = {pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3])};
    __builtin_amdgcn_raw_buffer_store_b64(u, rs, (int)off, 0, 0);
  }
}

__device__ __forceinline__ f32x4 mfma_s(const s16x8& a, const s16x8& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a), *reinterpret_cast<const bf16x8*>(&b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma_s(const s16x4& a, const s16x4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------
// window work lists: the windows that hold both queries and keys, binned by size -- class 0: <= 8 tokens in both
// frames (TWO such windows share one 16-row tile, see PAIR below), 1: <= 16, 2: <= 32, 3: <= 64.  This is the
// reference's "region batching" (drop levels 16/32/64, spt_backbone.py:47-71) reduced to index lists -- no padded
// tensors.  The class selects a kernel instantiation whose LDS images / register tiles are sized for it.
// layout: wl[0..3] = counts, wl[8 + c*nwin + j] = (b << 16) | (wcx << 8) | wcy  (packed: decoding the flat window id
// costs three integer divisions by run-time divisors, ~150 scalar instructions per wave).
// ------------------------------------------------------------------------------------------------
#define WL_HDR 8
#define WL_CLASSES 4
__global__ __launch_bounds__(256) void win_class_kernel(const int32_t* __restrict__ grid_q,
                                                       const int32_t* __restrict__ grid_k, int batch, int ny, int nx,
                                                       int Wy, int Wx, int sy, int sx, int8_t* __restrict__ cls,
                                                       int32_t* __restrict__ wl_hdr) {
  // the list counters are zeroed HERE (the compaction kernel that adds to them is the next launch on the stream): one
  // memset dispatch less per work list
  if (blockIdx.x == 0 && threadIdx.x < WL_HDR) wl_hdr[threadIdx.x] = 0;
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
    cls[dw] = (Tq > 0 && Tk > 0) ? (int8_t)(t <= 8 ? 0 : (t <= 16 ? 1 : (t <= 32 ? 2 : 3))) : (int8_t)-1;
  }
}

// The complement of the work lists: the rows nobody writes.  A window with queries but no keys (cross attention: the other frame
// is empty there) is in no list, and its query rows of out / dq must read as zeros; likewise the key rows of dk / dv of a
// window with keys but no queries.  Callers used to pre-zero those WHOLE tensors (20 fills of 60-240 MB per step); this launch
// (one wave per window, the two grids read once) zeroes only the orphan rows.  Valid when every token lies in a window of the
// two grids (no token dropping: a dropped token is in no window and nobody would zero it).
// q-side rows (Tk == 0): q0 [., wq0 elements] with pitch ldq0, qf [., nq_f] floats (lse);  k-side rows (Tq == 0): k0, k1.
__global__ __launch_bounds__(256) void win_orphan_zero_kernel(const int32_t* __restrict__ grid_q, const int32_t* __restrict__ grid_k,
                                                             int batch, int ny, int nx, int Wy, int Wx, int sy, int sx,
                                                             __hip_bfloat16* __restrict__ q0, int64_t ldq0, int wq0,
                                                             float* __restrict__ qf, int nqf, __hip_bfloat16* __restrict__ k0,
                                                             int64_t ldk0, __hip_bfloat16* __restrict__ k1, int64_t ldk1, int wk) {
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
  const uint4 z = make_uint4(0u, 0u, 0u, 0u);
  if (Tq > 0 && Tk == 0 && tq >= 0) {
    if (q0) {
      uint4* r = reinterpret_cast<uint4*>(q0 + (int64_t)tq * ldq0);
      for (int c = 0; c < wq0 / 8; ++c) r[c] = z;
    }
    if (qf)
      for (int c = 0; c < nqf; ++c) qf[(int64_t)tq * nqf + c] = 0.f;
  }
  if (Tk > 0 && Tq == 0 && tk >= 0) {
    if (k0) {
      uint4* r = reinterpret_cast<uint4*>(k0 + (int64_t)tk * ldk0);
      for (int c = 0; c < wk / 8; ++c) r[c] = z;
    }
    if (k1) {
      uint4* r = reinterpret_cast<uint4*>(k1 + (int64_t)tk * ldk1);
      for (int c = 0; c < wk / 8; ++c) r[c] = z;
    }
  }
}

int tmae_win_attn_zero_orphans(const int32_t* grid_q, const int32_t* grid_k, int batch, int ny, int nx, int do_shift, void* q0,
                               int64_t ldq0, int wq0, float* qf, int nqf, void* k0, int64_t ldk0, void* k1, int64_t ldk1, int wk,
                               void* stream_) {
  (void)hipGetLastError();
  if (!grid_q || !grid_k || batch <= 0 || ny <= 0 || nx <= 0 || (int64_t)batch * ny * nx >= ((int64_t)1 << 31)) return TMAE_EARG;
  if ((q0 && ((wq0 % 8) || (ldq0 % 8) || ldq0 < wq0 || ((uintptr_t)q0 & 15))) || (qf && nqf <= 0)) return TMAE_EARG;
  if ((k0 && ((wk % 8) || (ldk0 % 8) || ldk0 < wk || ((uintptr_t)k0 & 15))) || (k1 && ((wk % 8) || (ldk1 % 8) || ldk1 < wk || ((uintptr_t)k1 & 15))))
    return TMAE_EARG;
  const int Wy = (ny + WIN - 1) / WIN + 1, Wx = (nx + WIN - 1) / WIN + 1;
  const int s = do_shift ? WIN / 2 : WIN;
  const int64_t nwin = (int64_t)batch * Wy * Wx;
  hipLaunchKernelGGL(win_orphan_zero_kernel, dim3(tmae_cdiv(nwin, 4)), dim3(256), 0, (hipStream_t)stream_, grid_q, grid_k, batch, ny,
                     nx, Wy, Wx, s, s, (__hip_bfloat16*)q0, ldq0, wq0, qf, nqf, (__hip_bfloat16*)k0, ldk0, (__hip_bfloat16*)k1, ldk1, wk);
  return tmae_launch_status();
}

// compaction: one thread per window; a wave reserves its range with ONE atomic per class (list order only affects
// scheduling, never results)
__global__ __launch_bounds__(256) void win_worklist_kernel(const int8_t* __restrict__ cls, int64_t nwin, int Wy, int Wx,
                                                          int32_t* __restrict__ wl) {
  const int lane = threadIdx.x & 63;
  const int64_t dw = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int c = dw < nwin ? (int)cls[dw] : -1;
  const int wcy = (int)(dw % Wy), wcx = (int)((dw / Wy) % Wx), b = (int)(dw / ((int64_t)Wy * Wx));
  const int32_t packed = (b << 16) | (wcx << 8) | wcy;
#pragma unroll
  for (int k = 0; k < WL_CLASSES; ++k) {
    const unsigned long long m = __ballot(c == k);
    if (m == 0ull) continue;
    int base = 0;
    if (lane == 0) base = atomicAdd(wl + k, __popcll(m));
    base = __shfl(base, 0, 64);
    if (c == k) wl[WL_HDR + (int64_t)k * nwin + base + __popcll(m & ((1ull << lane) - 1ull))] = packed;
  }
}

size_t tmae_window_worklist_size(int batch, int ny, int nx) {
  const int Wy = (ny + WIN - 1) / WIN + 1, Wx = (nx + WIN - 1) / WIN + 1;
  const size_t nwin = (size_t)batch * Wy * Wx;
  return WL_HDR + WL_CLASSES * nwin + (nwin + 3) / 4;                 // counts | the lists | class bytes (scratch)
}

int tmae_window_worklist(const int32_t* grid_q, const int32_t* grid_k, int batch, int ny, int nx, int do_shift,
                         int32_t* worklist, void* stream_) {
  (void)hipGetLastError();
  hipStream_t stream = (hipStream_t)stream_;
  if (!grid_q || !grid_k || !worklist || batch <= 0 || ny <= 0 || nx <= 0) return TMAE_EARG;
  const int Wy = (ny + WIN - 1) / WIN + 1, Wx = (nx + WIN - 1) / WIN + 1;
  if (Wy > 255 || Wx > 255 || batch > 32767) return TMAE_EARG;        // packed window coordinates
  const int s = do_shift ? WIN / 2 : WIN;
  const int64_t nwin = (int64_t)batch * Wy * Wx;
  int8_t* cls = reinterpret_cast<int8_t*>(worklist + WL_HDR + WL_CLASSES * nwin);
  hipLaunchKernelGGL(win_class_kernel, dim3(tmae_cdiv(nwin, 4)), dim3(256), 0, stream, grid_q, grid_k, batch, ny, nx,
                     Wy, Wx, s, s, cls, worklist);
  hipLaunchKernelGGL(win_worklist_kernel, dim3(tmae_cdiv(nwin, 256)), dim3(256), 0, stream, cls, nwin, Wy, Wx, worklist);
  return (int)hipGetLastError();
}

// The tokens of this workgroup's window(s).  PAIR: TWO windows of <= 8 tokens share the 16 rows of the single tile --
// rows 0-7 belong to list entry 2j, rows 8-15 to entry 2j+1 -- and a block-diagonal mask on the logits keeps them
// apart (a masked probability is exactly 0, so every product that follows stays block-diagonal): the ~2/3 of all
// windows that hold <= 8 tokens (all of the masked current frame's) cost half a wavefront each.
// Wave 0 fills toks[side][slot] (-1 = no token); returns false when the block has nothing to do.
struct WinInfo {
  int Tq, Tk;            // tokens of the window (PAIR: of window A)
  int TqB, TkB;          // PAIR: of window B
  int dwin;              // flat id of window (A): the slot of its tau-gradient partial
  int dwinB;             // PAIR: flat id of window B, -1 if the unit holds one window
};

template <bool PAIR>
__device__ __forceinline__ bool find_tokens(const int32_t* __restrict__ wl, int cls, int64_t nwin,
                                            const int32_t* __restrict__ grid_q, const int32_t* __restrict__ grid_k,
                                            int ny, int nx, int Wy, int Wx, int sy, int sx, int (*toks)[64], int lane,
                                            int w, WinInfo& wi) {
  int b[2], wcy[2], wcx[2];
  int nw = 1;
  if (wl) {
    const int cnt = wl[cls];
    const int e0 = PAIR ? 2 * (int)blockIdx.x : (int)blockIdx.x;
    if (e0 >= cnt) return false;
    const int32_t* list = wl + WL_HDR + (int64_t)cls * nwin;
    const int p0 = list[e0];
    b[0] = p0 >> 16; wcx[0] = (p0 >> 8) & 255; wcy[0] = p0 & 255;
    if (PAIR && e0 + 1 < cnt) {
      const int p1 = list[e0 + 1];
      b[1] = p1 >> 16; wcx[1] = (p1 >> 8) & 255; wcy[1] = p1 & 255;
      nw = 2;
    }
  } else {
    const unsigned u = blockIdx.x;                       // dense launch (no work list): one window per block
    wcy[0] = (int)(u % (unsigned)Wy); wcx[0] = (int)((u / (unsigned)Wy) % (unsigned)Wx);
    b[0] = (int)(u / (unsigned)(Wy * Wx));
  }
  wi.dwin = (b[0] * Wx + wcx[0]) * Wy + wcy[0];
  wi.dwinB = (PAIR && nw == 2) ? (b[1] * Wx + wcx[1]) * Wy + wcy[1] : -1;
  wi.TqB = wi.TkB = 0;
  int tq[2] = {-1, -1}, tk[2] = {-1, -1};
#pragma unroll
  for (int n = 0; n < (PAIR ? 2 : 1); ++n) {
    if (n < nw) {
      const int y = wcy[n] * WIN - sy + (lane >> 3), x = wcx[n] * WIN - sx + (lane & 7);
      const bool in = y >= 0 && y < ny && x >= 0 && x < nx;
      const int cell = (b[n] * ny + y) * nx + x;            // < 2^31: the grids are int32-indexed tensors
      tq[n] = in ? grid_q[cell] : -1;
      tk[n] = in ? grid_k[cell] : -1;
    }
  }
  const unsigned long long mq0 = __ballot(tq[0] >= 0), mk0 = __ballot(tk[0] >= 0);
  wi.Tq = __popcll(mq0);
  wi.Tk = __popcll(mk0);
  const unsigned long long below = (1ull << lane) - 1ull;
  if (PAIR) {
    const unsigned long long mq1 = __ballot(tq[1] >= 0), mk1 = __ballot(tk[1] >= 0);
    wi.TqB = __popcll(mq1);
    wi.TkB = __popcll(mk1);
    if (w == 0) {
      if (lane < 16) { toks[0][lane] = -1; toks[1][lane] = -1; }
      if (tq[0] >= 0) toks[0][__popcll(mq0 & below)] = tq[0];
      if (tk[0] >= 0) toks[1][__popcll(mk0 & below)] = tk[0];
      if (tq[1] >= 0) toks[0][8 + __popcll(mq1 & below)] = tq[1];
      if (tk[1] >= 0) toks[1][8 + __popcll(mk1 & below)] = tk[1];
    }
  } else if (w == 0) {
    toks[0][lane] = -1;
    toks[1][lane] = -1;
    if (tq[0] >= 0) toks[0][__popcll(mq0 & below)] = tq[0];
    if (tk[0] >= 0) toks[1][__popcll(mk0 & below)] = tk[0];
  }
  return true;
}

// logit bias of key row `kr` (0..16*NT-1) for the lane's query column i: 0 where the key exists (PAIR: and belongs
// to the query's window), else MASKED_LOGIT -- it enters through the MFMA's C operand, exp underflows to exactly 0
template <bool PAIR>
__device__ __forceinline__ float key_bias(int kr, int i, const WinInfo& wi) {
  bool ok;
  if (PAIR) ok = ((kr >> 3) == (i >> 3)) && ((kr & 7) < ((kr >> 3) ? wi.TkB : wi.Tk));
  else ok = kr < wi.Tk;
  return ok ? 0.f : MASKED_LOGIT;
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
struct AttnBufs {          // byte sizes of the tensors for the buffer descriptors (launcher-checked < 2^31)
  unsigned q, k, v, o, g, lse, dq, dk, dv;
};

// __launch_bounds__(256, 2): at least two waves per SIMD, i.e. at most 256 registers per lane.  Not for occupancy (the kernels use
// 28-107) but for the MFMA FORM: with a budget above 256 the compiler assumes it may need the AGPR half of the register file and
// selects the AGPR-destination MFMAs -- every logit then travels to the VALU through a v_accvgpr_read (508 moves in the dh-32
// 64-token forward, 16 % of its instructions).  Below 257 it selects the VGPR form and the moves are gone.
// SPLIT (see TAU_SPLIT_BELOW): the logits from the hi/lo split of q-hat / k-hat (three MFMAs per tile) or from their bf16 values
// alone (one MFMA, no split arithmetic).  The kernels carry both bodies and take one per launch-uniform temperature.
//
// What the split buys is 2^-16 instead of 2^-9 relative accuracy per normalised channel, i.e. an absolute logit error of
// ~5e-4 / tau without it (32 products of magnitude ~1/32, rounding errors adding in quadrature, divided by tau).  The
// probabilities that leave the softmax are rounded to bf16 for the P.V product anyway (2^-9 relative = a logit error of 2e-3):
// at tau_c >= TAU_SPLIT_BELOW = 0.25 the un-split logit error stays below that rounding and the split is arithmetic nobody can see
// -- ~20 VALU instructions per row fragment and two of three logit MFMAs in kernels that are bound by vector issue (DESIGN.md
// section 6j).  The module starts at tau = 1 (cosine_msa.py:453-456); a temperature trained below 0.25 -- down to the clamp at
// 0.01, where the un-split error would be 5 % -- takes the split body.  The reference itself runs this bmm on fp16 operands
// under AMP (2^-11).
#define TAU_SPLIT_BELOW 0.25f
// own = max(tau of wave w's head, tau_min): scales the wave's logits; blk = the smallest clamped temperature of the workgroup's
// four heads (blockIdx.y * 4 ..): picks the body for all four waves, because both bodies hold workgroup barriers (the split body is
// the more exact one).  tau_stride 0 (the layer's one temperature, every shipped configuration) runs exactly the instructions it
// ran before the per-head form existed -- one scalar load of tau[0] behind the token search; issuing four loads at the top of the
// kernel, or one vector load of tau[head] here, cost the short classes 5-19 % (+0.25 ms per step, same-box A/B).  tau_stride 1
// (non_shared_tau): scalar loads as well, the head index made wave-uniform first.
struct TauPick { float own, blk; };
__device__ __forceinline__ TauPick pick_tau(const float* __restrict__ tau, float tau_min, int tau_stride, int w) {
  if (tau_stride == 0) {
    const float t = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(fmaxf(tau[0], tau_min))));
    return TauPick{t, t};
  }
  const int h0 = (int)blockIdx.y * 4, ws = __builtin_amdgcn_readfirstlane(w);
  const float own = fmaxf(tau[(h0 + ws) * tau_stride], tau_min);
  float blk = own;
#pragma unroll
  for (int j = 0; j < 4; ++j) blk = fminf(blk, fmaxf(tau[(h0 + j) * tau_stride], tau_min));
  return TauPick{__uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(own))),
                 __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(blk)))};
}

template <int DH, int NT, bool PAIR, bool SPLIT>
__device__ __forceinline__ void win_attn_fwd_body(
    const __hip_bfloat16* __restrict__ q, int64_t ldq, const __hip_bfloat16* __restrict__ k, int64_t ldk,
    const __hip_bfloat16* __restrict__ v, int64_t ldv, int nhead, float inv_tau, __hip_bfloat16* __restrict__ out, int64_t ldo,
    float* __restrict__ lse, AttnBufs nb, const WinInfo& wi, int (*toks)[64], char (*vimg)[NT * 16 * (DH * 2 + 16)]) {
  constexpr int FR = DH / 4;                 // channels per lane in a row fragment
  constexpr int CT = DH / 16;                // 16-channel output tiles
  constexpr int RB = DH * 2 + 16;            // V image row pitch (bytes): 16-byte aligned, off the power of two
  typedef typename Frag<DH>::T frag_t;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, g = lane >> 4, i = lane & 15;
  const int head = blockIdx.y * 4 + w, hoff = head * DH;
  const int Tq = wi.Tq, Tk = wi.Tk;
  const __amdgpu_buffer_rsrc_t rsq = make_rsrc(q, nb.q), rsk = make_rsrc(k, nb.k), rsv = make_rsrc(v, nb.v),
                               rso = make_rsrc(out, nb.o), rsl = make_rsrc(lse, nb.lse);
  const unsigned colb = (unsigned)(hoff + FR * g) * 2u;
  const int nq = PAIR ? 1 : (Tq + 15) >> 4, nk = PAIR ? 1 : (Tk + 15) >> 4;
  // ---- all global row loads are issued here, one dependent round after the token ids
  frag_t kf[NT], kl[NT], qf[NT], ql[NT];
  typedef typename RawFrag<FR>::T raw_t;
  raw_t rk[NT], rv[NT], rq[NT];
  int tokq_[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {                    // no branches here: every load of the wave is in flight at once
    const int slot = t * 16 + i;
    const int tokk = toks[1][slot];
    tokq_[t] = toks[0][slot];
    rk[t] = load_row_raw<FR>(rsk, row_off(tokk, (unsigned)ldk * 2u, colb));
    rv[t] = load_row_raw<FR>(rsv, row_off(tokk, (unsigned)ldv * 2u, colb));
    rq[t] = load_row_raw<FR>(rsq, row_off(tokq_[t], (unsigned)ldq * 2u, colb));
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int slot = t * 16 + i;
    float f[FR];
    if (t < nk) {                                   // wave-uniform
      unpack_row<FR>(rk[t], f);
      normalize_frag<FR>(f, 1.0f);
      if constexpr (SPLIT) split_frag<FR>(f, kf[t], kl[t]); else kf[t] = pack_frag<FR>(f);
      store_img_frag<DH>(&vimg[w][slot * RB], g, raw_as_frag<FR>(rv[t]));
    } else {
#pragma unroll
      for (int j = 0; j < FR; ++j) { kf[t][j] = 0; if constexpr (SPLIT) kl[t][j] = 0; }
      frag_t z;
#pragma unroll
      for (int j = 0; j < FR; ++j) z[j] = 0;
      store_img_frag<DH>(&vimg[w][slot * RB], g, z);                        // P is 0 there, but 0 * garbage = NaN
    }
    if (t < nq) {
      unpack_row<FR>(rq[t], f);
      normalize_frag<FR>(f, inv_tau);
      if constexpr (SPLIT) split_frag<FR>(f, qf[t], ql[t]); else qf[t] = pack_frag<FR>(f);
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

  // key rows that do not exist (PAIR: or belong to the other window) get a large negative logit through the MFMA's C
  // operand (exp underflows to exactly 0): no per-element masking in the loops, and tiles kt >= nk need no special
  // case (only where the 4*NT registers are free: the 64-token class would lose a resident wave and masks per element)
  constexpr bool KB = NT <= 2;
  f32x4 kbias[NT];
  if constexpr (KB) {
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) kbias[kt][r] = key_bias<PAIR>(kt * 16 + 4 * g + r, i, wi);
  } else {
    // (!KB, the 64-token class) one bias vector per key tile as well, made once per window: all zeros except in the tile that
    // holds the first absent row.  (Until round 5 a single vector was selected into the C operand by `kt == nk - 1`: a
    // wave-uniform condition, but the select compiled to four v_cndmask per tile and pass: 100 of the backward's 2 100 VALU
    // instructions.)  No branch between a logit MFMA and its first VALU reader anywhere in these kernels: the compiler leaves
    // the taken side of such a branch without the wait states an MFMA result needs (tools/check_mfma_hazards.py).
    const int first_absent = Tk - 4 * g;       // row kt * 16 + 4 g + r is absent when kt * 16 + r >= Tk - 4 g
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) kbias[kt][r] = (kt * 16 + r >= first_absent) ? MASKED_LOGIT : 0.f;
  }
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int qt = 0; qt < NT; ++qt) {
    if (qt < nq) {
      f32x4 st[NT];
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
        if (kt < nk) {
          if constexpr (SPLIT) {
            st[kt] = mfma_s(kl[kt], qf[qt], kbias[kt]);                    // S^T tile: rows = keys 4g+r, col = query i
            st[kt] = mfma_s(kf[kt], ql[qt], st[kt]);
            st[kt] = mfma_s(kf[kt], qf[qt], st[kt]);
          } else {
            st[kt] = mfma_s(kf[kt], qf[qt], kbias[kt]);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[kt][r]);
          // (two v_max3_f32 in inline asm would save the canonicalising v_max_f32 the compiler puts in front of fmaxf() on an MFMA
          //  result -- but the compiler does not pad the MFMA -> VALU hazard in front of inline asm: check_mfma_hazards.py flags it)
          asm volatile("" : "+v"(mx));                               // the first reader stays in the MFMAs' block
        } else if constexpr (KB) {
          st[kt] = kbias[kt];                                        // all rows absent: p = 0 below
        }
      }
      mx = quad_max(mx);
      const float mxl = mx * LOG2E_F;
      float l = 0.f;
      s16x4 pf[NT];
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
        if (KB || kt < nk) {
          float pe[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            pe[r] = exp_minus(st[kt][r], mxl);
            l += pe[r];
          }
          const u32x2 pu = {pack_bf16x2(pe[0], pe[1]), pack_bf16x2(pe[2], pe[3])};
          pf[kt] = __builtin_bit_cast(s16x4, pu);
        }
      }
      l = quad_sum(l);
      const float invl = fast_rcp(l);
      // O^T = V^T . P^T (swapped: rows = channels, column = query i): the V fragments serve as the A operand, the
      // probabilities stay where the S^T accumulators left them
      f32x4 o[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)              // key tile 0 is always present: it starts from the inline constant 0
          if (kt < nk) o[ct] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(vf[kt][ct], pf[kt], kt == 0 ? zero4 : o[ct], 0, 0, 0);
      }
      // lane (g, i): query i, channels (DH/4) g + 4 ct + r -- DH/4 consecutive channels: one wide store (dropped by the
      // descriptor's range check when the slot holds no query)
      float of[FR];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) of[4 * ct + r] = o[ct][r] * invl;
      store_row_frag<FR>(rso, row_off(tokq_[qt], (unsigned)ldo * 2u, colb), of);
      const unsigned loff = (g == 0 && tokq_[qt] >= 0) ? (unsigned)(tokq_[qt] * nhead + head) * 4u : 0xFFFFFFFFu;
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(mx + __logf(l)), rsl, (int)loff, 0, 0);
    }
  }
}

template <int DH, int NT, bool PAIR>
__global__ __launch_bounds__(256, 2) void win_attn_fwd_mfma_kernel(
    const __hip_bfloat16* __restrict__ q, int64_t ldq, const __hip_bfloat16* __restrict__ k, int64_t ldk,
    const __hip_bfloat16* __restrict__ v, int64_t ldv, int nhead, const int32_t* __restrict__ grid_q,
    const int32_t* __restrict__ grid_k, int ny, int nx, int Wy, int Wx, int sy, int sx,
    const float* __restrict__ tau, float tau_min, __hip_bfloat16* __restrict__ out, int64_t ldo,
    float* __restrict__ lse, const int32_t* __restrict__ wl, int cls, int64_t nwin, AttnBufs nb, int tau_stride) {
  static_assert(!PAIR || NT == 1, "two windows per tile: single-tile class only");
  constexpr int RB = DH * 2 + 16, ROWS = NT * 16;
  __shared__ int toks[2][64];
  __shared__ __attribute__((aligned(16))) char vimg[4][ROWS * RB];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int head = blockIdx.y * 4 + w, hoff = head * DH;
  WinInfo wi;
  if (!find_tokens<PAIR>(wl, cls, nwin, grid_q, grid_k, ny, nx, Wy, Wx, sy, sx, toks, lane, w, wi)) return;
  if (wi.Tq == 0) return;                     // (dense launch only) no queries here
  if (wi.Tk == 0) {                           // (dense launch only) cross-attention window without keys: zero rows
    __syncthreads();
    const int tq = toks[0][lane];
    if (tq >= 0) {
      __hip_bfloat16* o = out + (int64_t)tq * ldo + hoff;
#pragma unroll
      for (int c = 0; c < DH; ++c) o[c] = __float2bfloat16(0.f);
      lse[(int64_t)tq * nhead + head] = 0.f;
    }
    return;
  }
  __syncthreads();
  // tau is one number for the whole launch (cosine_msa.py:453-456; tau_stride 0): a scalar load, a scalar compare, one branch per
  // wave -- or one per head (non_shared_tau, tau_stride 1): the wave's own head scales its logits, the SMALLEST temperature of the
  // workgroup's four heads picks the body, because both bodies hold workgroup barriers (the split body is the more exact one)
  const TauPick tp = pick_tau(tau, tau_min, tau_stride, w);
  const float tau_c = tp.own;
  const float inv_tau = fast_rcp(tau_c);                       // v_rcp_f32 (1 ulp; the backward uses the same instruction): an IEEE
                                                                 // division is ~12 instructions of the ~370 a 16-token window costs
  if (tp.blk < TAU_SPLIT_BELOW)
    win_attn_fwd_body<DH, NT, PAIR, true>(q, ldq, k, ldk, v, ldv, nhead, inv_tau, out, ldo, lse, nb, wi, toks, vimg);
  else
    win_attn_fwd_body<DH, NT, PAIR, false>(q, ldq, k, ldk, v, ldv, nhead, inv_tau, out, ldo, lse, nb, wi, toks, vimg);
}

static int64_t class_grid(int cls, int64_t nwin, int64_t mq, int64_t mk) {
  // class 0: pairs of windows; class c > 0 windows hold >= 8 * 2^(c-1) + 1 tokens in one of the two frames
  if (cls == 0) return (nwin + 1) / 2;
  const int64_t bound = (mq + mk) / (cls == 1 ? 9 : (cls == 2 ? 17 : 33));
  return bound < nwin ? bound : nwin;
}

// buffer descriptors address with 32-bit byte offsets and row_off multiplies 24-bit factors
static bool attn_sizes_ok(int64_t rows, int64_t ld) { return rows < (1 << 24) && ld * 2 < (1 << 24) && rows * ld * 2 < ((int64_t)1 << 31); }
static unsigned attn_bytes(int64_t rows, int64_t ld, int width) { return rows > 0 ? (unsigned)(((rows - 1) * ld + width) * 2) : 0u; }

int tmae_win_attn_fwd_mfma(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                           int64_t mq, int64_t mk, int nhead, int dh, const int32_t* grid_q, const int32_t* grid_k,
                           int batch, int ny, int nx, int do_shift, const float* tau, float tau_min, void* out,
                           int64_t ldo, float* lse, const int32_t* worklist, int tau_stride, hipStream_t stream) {
  if (nhead % 4 || (dh != 16 && dh != 32)) return TMAE_EARG;
  // 16-byte row fragments / V rows: bases and pitches must keep every head slice 16-byte aligned
  if ((ldq % 8) || (ldk % 8) || (ldv % 8) || ((uintptr_t)q & 15) || ((uintptr_t)k & 15) || ((uintptr_t)v & 15))
    return TMAE_EARG;
  if (!attn_sizes_ok(mq, ldq) || !attn_sizes_ok(mk, ldk) || !attn_sizes_ok(mk, ldv) || !attn_sizes_ok(mq, ldo) ||
      (int64_t)batch * ny * nx >= ((int64_t)1 << 31))
    return TMAE_EARG;
  const int Wy = (ny + WIN - 1) / WIN + 1, Wx = (nx + WIN - 1) / WIN + 1;
  const int s = do_shift ? WIN / 2 : WIN;
  const int64_t nwin = (int64_t)batch * Wy * Wx;
  const int d = nhead * dh;
  AttnBufs nb = {};
  nb.q = attn_bytes(mq, ldq, d); nb.k = attn_bytes(mk, ldk, d); nb.v = attn_bytes(mk, ldv, d);
  nb.o = attn_bytes(mq, ldo, d); nb.lse = (unsigned)(mq * nhead * 4);
#define FWDM(DH, NT, PAIR, CLS, GX)                                                                                  \
  hipLaunchKernelGGL((win_attn_fwd_mfma_kernel<DH, NT, PAIR>), dim3((unsigned)(GX), (unsigned)(nhead / 4)), dim3(256), \
                     0, stream, (const __hip_bfloat16*)q, ldq, (const __hip_bfloat16*)k, ldk, (const __hip_bfloat16*)v, \
                     ldv, nhead, grid_q, grid_k, ny, nx, Wy, Wx, s, s, tau, tau_min, (__hip_bfloat16*)out, ldo, lse, \
                     worklist, CLS, nwin, nb, tau_stride)
  if (!worklist) {
    if (dh == 32) FWDM(32, 4, false, 0, nwin); else FWDM(16, 4, false, 0, nwin);
  } else {
    const int64_t g0 = class_grid(0, nwin, mq, mk), g1 = class_grid(1, nwin, mq, mk), g2 = class_grid(2, nwin, mq, mk),
                  g3 = class_grid(3, nwin, mq, mk);
    if (dh == 32) {
      if (g0 > 0) FWDM(32, 1, true, 0, g0);
      if (g1 > 0) FWDM(32, 1, false, 1, g1);
      if (g2 > 0) FWDM(32, 2, false, 2, g2);
      if (g3 > 0) FWDM(32, 4, false, 3, g3);
    } else {
      if (g0 > 0) FWDM(16, 1, true, 0, g0);
      if (g1 > 0) FWDM(16, 1, false, 1, g1);
      if (g2 > 0) FWDM(16, 2, false, 2, g2);
      if (g3 > 0) FWDM(16, 4, false, 3, g3);
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
// the LDS of one backward workgroup (declared by the kernel, shared by its two bodies)
template <int DH, int NT> struct BwdLds {
  static constexpr int RB = DH * 2 + 16, ROWS = NT * 16, TP = 40;     // TP: pitch (bytes) of the 16x16 bf16 transpose tiles
  int toks[2][64];
  __attribute__((aligned(16))) char kimg[4][ROWS * RB];   // K-hat
  // Q-hat / tau_c and dO: ONE 16-row tile each, rewritten by its wave at the top of every query tile (round 5).  Holding all NT
  // tiles cost 41 KB of the 69 KB a dh-32 64-token workgroup used: two workgroups per CU, two waves per SIMD, in a kernel that
  // waits on latency (its first 2 us are global loads, every logit tile is an MFMA -> VALU -> exp -> MFMA chain).  With 28 KB and
  // the 168-register build the launch bounds ask for, three workgroups fit.
  __attribute__((aligned(16))) char qimg[4][16 * RB];     // Q-hat / tau_c of the current query tile
  __attribute__((aligned(16))) char gimg[4][16 * RB];     // dO of the current query tile
  float qnorm[4][ROWS], knorm[4][ROWS];
  __attribute__((aligned(16))) char ptile[4][2][16 * TP];
};

template <int DH, int NT, bool PAIR, bool SPLIT>
__device__ __forceinline__ void win_attn_bwd_body(
    const __hip_bfloat16* __restrict__ q, int64_t ldq, const __hip_bfloat16* __restrict__ k, int64_t ldk,
    const __hip_bfloat16* __restrict__ v, int64_t ldv, const __hip_bfloat16* __restrict__ dout, int64_t lddo,
    const float* __restrict__ lse, int nhead, float tau_c, __hip_bfloat16* __restrict__ dq, int64_t lddq,
    __hip_bfloat16* __restrict__ dk, int64_t lddk, __hip_bfloat16* __restrict__ dv, int64_t lddv,
    float* __restrict__ dtau_partial, float* __restrict__ dtp, AttnBufs nb, const WinInfo& wi, BwdLds<DH, NT>& L) {
  constexpr int FR = DH / 4;
  constexpr int CT = DH / 16;
  constexpr int RB = DH * 2 + 16;
  constexpr int TP = BwdLds<DH, NT>::TP;
  typedef typename Frag<DH>::T frag_t;
  auto& toks = L.toks; auto& kimg = L.kimg; auto& qimg = L.qimg; auto& gimg = L.gimg; auto& qnorm = L.qnorm; auto& knorm = L.knorm;
  auto& ptile = L.ptile;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, g = lane >> 4, i = lane & 15;
  const int head = blockIdx.y * 4 + w, hoff = head * DH;
  const int Tq = wi.Tq, Tk = wi.Tk;
  const float inv_tau = fast_rcp(tau_c);
  const __amdgpu_buffer_rsrc_t rsq = make_rsrc(q, nb.q), rsk = make_rsrc(k, nb.k), rsv = make_rsrc(v, nb.v),
                               rsg = make_rsrc(dout, nb.g), rsl = make_rsrc(lse, nb.lse), rsdq = make_rsrc(dq, nb.dq),
                               rsdk = make_rsrc(dk, nb.dk), rsdv = make_rsrc(dv, nb.dv);
  const unsigned colb = (unsigned)(hoff + FR * g) * 2u;
  const int nq = PAIR ? 1 : (Tq + 15) >> 4, nk = PAIR ? 1 : (Tk + 15) >> 4;
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
    tokk_[t] = toks[1][slot];
    tokq_[t] = toks[0][slot];
    rk[t] = load_row_raw<FR>(rsk, row_off(tokk_[t], (unsigned)ldk * 2u, colb));
    rv[t] = load_row_raw<FR>(rsv, row_off(tokk_[t], (unsigned)ldv * 2u, colb));
    rq[t] = load_row_raw<FR>(rsq, row_off(tokq_[t], (unsigned)ldq * 2u, colb));
    rg[t] = load_row_raw<FR>(rsg, row_off(tokq_[t], (unsigned)lddo * 2u, colb));
    const unsigned loff = tokq_[t] >= 0 ? (unsigned)(tokq_[t] * nhead + head) * 4u : 0xFFFFFFFFu;
    lse_i[t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsl, (int)loff, 0, 0));
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int slot = t * 16 + i;
    float f[FR];
    if (t < nk) {                                    // wave-uniform
      unpack_row<FR>(rk[t], f);
      const float nrm = normalize_frag<FR>(f, 1.0f);
      if constexpr (SPLIT) split_frag<FR>(f, kf[t], kl[t]); else kf[t] = pack_frag<FR>(f);
      if (g == 0) knorm[w][slot] = nrm;
      vr[t] = raw_as_frag<FR>(rv[t]);
    } else {
#pragma unroll
      for (int j = 0; j < FR; ++j) { kf[t][j] = 0; vr[t][j] = 0; if constexpr (SPLIT) kl[t][j] = 0; }
    }
    store_img_frag<DH>(&kimg[w][slot * RB], g, kf[t]);
    if (tokq_[t] < 0 || t >= nq) lse_i[t] = INFINITY;                      // no query: p = exp(s - inf) = 0
    if (t < nq) {
      unpack_row<FR>(rq[t], f);
      const float nrm = normalize_frag<FR>(f, inv_tau);
      if constexpr (SPLIT) split_frag<FR>(f, qf[t], ql[t]); else qf[t] = pack_frag<FR>(f);
      if (g == 0) qnorm[w][slot] = nrm;
      gf[t] = raw_as_frag<FR>(rg[t]);
    }
  }
  __syncthreads();
  f32x4 dKa[NT][CT], dVa[NT][CT];            // first written by query tile 0 (always present; tiles kt >= nk are never read)
  float dtau_acc = 0.f;
  constexpr bool KB = NT <= 2;             // MASKED_LOGIT on absent / foreign key rows through the C operand (see the forward)
  f32x4 kbias[NT];
  if constexpr (KB) {
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) kbias[kt][r] = key_bias<PAIR>(kt * 16 + 4 * g + r, i, wi);
  } else {
    const int first_absent = Tk - 4 * g;       // row kt * 16 + 4 g + r is absent when kt * 16 + r >= Tk - 4 g
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) kbias[kt][r] = (kt * 16 + r >= first_absent) ? MASKED_LOGIT : 0.f;
  }
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int qt = 0; qt < NT; ++qt) {
    if (qt < nq) {
      const int qslot = qt * 16 + i;
      // transposed right-hand fragments of this query tile: its rows go into the wave's one-tile images first (LDS operations
      // of a wave execute in order: the stores land behind the previous tile's transposed reads and in front of these)
      store_img_frag<DH>(&qimg[w][i * RB], g, qf[qt]);
      store_img_frag<DH>(&gimg[w][i * RB], g, gf[qt]);
      __builtin_amdgcn_wave_barrier();
      s16x4 trQ[CT], trG[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const int off = (4 * g + (i >> 2)) * RB + ct * 32 + 8 * (i & 3);
        trQ[ct] = tr_read4(&qimg[w][off]);
        trG[ct] = tr_read4(&gimg[w][off]);
      }
      __builtin_amdgcn_wave_barrier();
      f32x4 dQa[CT];                                  // first written by key tile 0 (always present)
      // pass 1 over the key tiles: logits, probabilities and dP; D_i = sum_j P_ij dP_ij is taken from THESE values
      // (not from dO . O with the bf16-rounded saved output), so that sum_j dS_ij = 0 holds to fp32 rounding --
      // the tau gradient sum_ij dS_ij s_ij is a difference of large terms and is biased otherwise
      // NT == 4: the tiles are recomputed in pass 2 (4 MFMAs + 4 exps per tile pair, far below the budget of this
      // latency-bound kernel) instead of being held in 48 registers, which keeps two waves per SIMD resident.
      constexpr bool RECOMP = NT >= 4;
      f32x4 sTk[RECOMP ? 1 : NT], pTk[NT], dPk[RECOMP ? 1 : NT];   // P is kept in every class (16 registers at NT = 4)
      float dacc = 0.f;
      const float lse_l = lse_i[qt] * LOG2E_F;
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
        if (kt < nk) {
          // swapped: rows = keys 4g+r of this tile, column = query i; masked keys (MASKED_LOGIT through the C operand) and
          // absent queries (lse = +inf) give p = 0
          f32x4 sT;
          if constexpr (SPLIT) {
            sT = mfma_s(kl[kt], qf[qt], kbias[kt]);
            sT = mfma_s(kf[kt], ql[qt], sT);
            sT = mfma_s(kf[kt], qf[qt], sT);
          } else {
            sT = mfma_s(kf[kt], qf[qt], kbias[kt]);
          }
          const f32x4 dP = mfma_s(vr[kt], gf[qt], zero4);
          if constexpr (!RECOMP) { sTk[kt] = sT; dPk[kt] = dP; }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = exp_minus(sT[r], lse_l);
            pTk[kt][r] = p;
            dacc += p * dP[r];
          }
          asm volatile("" : "+v"(dacc));                             // the MFMAs' first readers stay in their block
        }
      }
      dacc = quad_sum(dacc);
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
        if (kt < nk) {
          f32x4 sT, dP, pT;
          if constexpr (RECOMP) {
            if constexpr (SPLIT) {
              sT = mfma_s(kl[kt], qf[qt], kbias[kt]);
              sT = mfma_s(kf[kt], ql[qt], sT);
              sT = mfma_s(kf[kt], qf[qt], sT);
            } else {
              sT = mfma_s(kf[kt], qf[qt], kbias[kt]);
            }
            dP = mfma_s(vr[kt], gf[qt], f32x4{-dacc, -dacc, -dacc, -dacc});   // dP - D: D enters through the C operand
            // Round 4 masked the absent keys per element here and found that restricting the mask to the last tile (a wave-uniform
            // branch between the dP MFMA and `dP - D`) gave NaN in dK and garbage in dQ at the temperature clamp: not arithmetic --
            // the compiler had left the TAKEN side of that branch without the wait states an MFMA result needs before its first
            // VALU reader, so `dP - D` read the registers' previous content (-inf for an absent query).  DESIGN.md section 6h;
            // tools/check_mfma_hazards.py walks both sides of every branch behind every MFMA.  Now the mask rides on the C
            // operand and no branch is left between these MFMAs and their readers.
            pT = pTk[kt];                                  // (the exponentials are not recomputed: 64 v_exp_f32 + 64 FMAs per item)
          } else {
            sT = sTk[kt]; dP = dPk[kt]; pT = pTk[kt];
          }
          float dsv[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            dsv[r] = pT[r] * (RECOMP ? dP[r] : dP[r] - dacc);  // dS = P (dP - D)
            dtau_acc += dsv[r] * sT[r];                    // p = 0 entries: 0 * finite
          }
          const u32x2 dsu = {pack_bf16x2(dsv[0], dsv[1]), pack_bf16x2(dsv[2], dsv[3])};
          const u32x2 ptu = {pack_bf16x2(pT[0], pT[1]), pack_bf16x2(pT[2], pT[3])};
          const s16x4 dsT = __builtin_bit_cast(s16x4, dsu), pTb = __builtin_bit_cast(s16x4, ptu);
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) {
            const s16x4 trK = tr_read4(&kimg[w][(kt * 16 + 4 * g + (i >> 2)) * RB + ct * 32 + 8 * (i & 3)]);
            // (first tile: C = the inline constant 0 instead of a zeroed accumulator -- kt is a compile-time index here)
            dQa[ct] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(trK, dsT, kt == 0 ? zero4 : dQa[ct], 0, 0, 0);   // dQ-hat^T
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
            dVa[kt][ct] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(trG[ct], pU, qt == 0 ? zero4 : dVa[kt][ct], 0, 0, 0);    // dV^T
            dKa[kt][ct] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(trQ[ct], dsU, qt == 0 ? zero4 : dKa[kt][ct], 0, 0, 0);   // dK-hat^T
          }
        }
      }
      // dq = (dq-hat - q-hat (q-hat . dq-hat)) / |q|.  The products are taken transposed (rows = channels) over the
      // permuted images, so lane (g, i) holds channels FR g + 4 ct + r of query qt*16+i: the SAME channels as its own
      // row fragment qf[qt] -- q-hat comes from registers, the dot product is a quad reduction, one wide store per lane
      {
        // with qf = q-hat / tau (bf16 hi part) and dQa = tau * dq-hat: q-hat . dq-hat = qf . dQa, and
        // dq = dQa (1 / (tau |q|)) - qf (tau (q-hat . dq-hat) / |q|): the scale factors are applied once per row, not per channel
        float qb[FR], dot = 0.f;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            qb[4 * ct + r] = bf2f(qf[qt][4 * ct + r]);
            dot += qb[4 * ct + r] * dQa[ct][r];
          }
        dot = quad_sum(dot);
        const float nrm = qnorm[w][qslot];               // slots without a query: stale, the store below is dropped
        if (nrm <= 1e-12f) dot = 0.f;
        const float inv = fast_rcp(nrm);
        const float ca = inv_tau * inv, cb = -(tau_c * dot) * inv;
        float o[FR];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) o[4 * ct + r] = __builtin_fmaf(dQa[ct][r], ca, qb[4 * ct + r] * cb);
        store_row_frag<FR>(rsdq, row_off(tokq_[qt], (unsigned)lddq * 2u, colb), o);
      }
    }
  }
  if constexpr (PAIR) {
    // one partial per WINDOW, not per pair: which two windows share a unit depends on the order of the work list (built with an
    // atomic counter), and a pair's sum would make the tau gradient differ in its last bits from run to run.  Query column i < 8
    // belongs to window A, i >= 8 to window B; both sums run over the same lanes in the same order whoever the partner is.
    const float sa = wave_sum(i < 8 ? dtau_acc : 0.f), sb = wave_sum(i < 8 ? 0.f : dtau_acc);
    if (lane == 0) {
      *dtp = sa;
      if (wi.dwinB >= 0) dtau_partial[(int64_t)wi.dwinB * nhead + head] = sb;
    }
  } else {
    dtau_acc = wave_sum(dtau_acc);
    if (lane == 0) *dtp = dtau_acc;
  }
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
      dot = quad_sum(dot);
      const float nrm = knorm[w][ks];
      if (nrm <= 1e-12f) dot = 0.f;
      const float inv = fast_rcp(nrm);
      const float cb = -dot * inv;
      float o[FR];
#pragma unroll
      for (int j = 0; j < FR; ++j) o[j] = __builtin_fmaf(dkh[j], inv, kh[j] * cb);
      store_row_frag<FR>(rsdk, row_off(tokk_[kt], (unsigned)lddk * 2u, colb), o);
      store_row_frag<FR>(rsdv, row_off(tokk_[kt], (unsigned)lddv * 2u, colb), dvv);
    }
  }
}

template <int DH, int NT, bool PAIR>
__global__ __launch_bounds__(256, 2) void win_attn_bwd_mfma_kernel(
    const __hip_bfloat16* __restrict__ q, int64_t ldq, const __hip_bfloat16* __restrict__ k, int64_t ldk,
    const __hip_bfloat16* __restrict__ v, int64_t ldv, const __hip_bfloat16* __restrict__ outp, int64_t ldo,
    const __hip_bfloat16* __restrict__ dout, int64_t lddo, const float* __restrict__ lse, int nhead,
    const int32_t* __restrict__ grid_q, const int32_t* __restrict__ grid_k, int ny, int nx, int Wy, int Wx, int sy,
    int sx, const float* __restrict__ tau, float tau_min, __hip_bfloat16* __restrict__ dq, int64_t lddq,
    __hip_bfloat16* __restrict__ dk, int64_t lddk, __hip_bfloat16* __restrict__ dv, int64_t lddv,
    float* __restrict__ dtau_partial, const int32_t* __restrict__ wl, int cls, int64_t nwin, AttnBufs nb, int tau_stride) {
  static_assert(!PAIR || NT == 1, "two windows per tile: single-tile class only");
  __shared__ BwdLds<DH, NT> L;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int head = blockIdx.y * 4 + w, hoff = head * DH;
  WinInfo wi;
  if (!find_tokens<PAIR>(wl, cls, nwin, grid_q, grid_k, ny, nx, Wy, Wx, sy, sx, L.toks, lane, w, wi)) return;
  float* dtp = dtau_partial + (int64_t)wi.dwin * nhead + head;
  if (wi.Tq == 0 || wi.Tk == 0) {             // (dense launch only) nothing attended here: zero gradients
    __syncthreads();
    if (lane == 0) *dtp = 0.f;
    const int tq = L.toks[0][lane], tk = L.toks[1][lane];
    if (tq >= 0) {
      __hip_bfloat16* p = dq + (int64_t)tq * lddq + hoff;
#pragma unroll
      for (int c = 0; c < DH; ++c) p[c] = __float2bfloat16(0.f);
    }
    if (tk >= 0 && grid_q != grid_k) {
      __hip_bfloat16* p1 = dk + (int64_t)tk * lddk + hoff;
      __hip_bfloat16* p2 = dv + (int64_t)tk * lddv + hoff;
#pragma unroll
      for (int c = 0; c < DH; ++c) { p1[c] = __float2bfloat16(0.f); p2[c] = __float2bfloat16(0.f); }
    }
    return;
  }
  __syncthreads();                                   // token lists visible
  // one temperature per launch or per head (see the forward kernel): scalar compare, one branch per wave, the same for the four
  // waves of the workgroup
  const TauPick tp = pick_tau(tau, tau_min, tau_stride, w);
  const float tau_c = tp.own;
  if (tp.blk < TAU_SPLIT_BELOW)
    win_attn_bwd_body<DH, NT, PAIR, true>(q, ldq, k, ldk, v, ldv, dout, lddo, lse, nhead, tau_c, dq, lddq, dk, lddk, dv, lddv,
                                          dtau_partial, dtp, nb, wi, L);
  else
    win_attn_bwd_body<DH, NT, PAIR, false>(q, ldq, k, ldk, v, ldv, dout, lddo, lse, nhead, tau_c, dq, lddq, dk, lddk, dv, lddv,
                                           dtau_partial, dtp, nb, wi, L);
}

int tmae_win_attn_bwd_mfma(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                           const void* out, int64_t ldo, const void* dout, int64_t lddo, const float* lse, int64_t mq,
                           int64_t mk, int nhead, int dh, const int32_t* grid_q, const int32_t* grid_k, int batch,
                           int ny, int nx, int do_shift, const float* tau, float tau_min, void* dq, int64_t lddq,
                           void* dk, int64_t lddk, void* dv, int64_t lddv, float* dtau_partial,
                           const int32_t* worklist, int tau_stride, hipStream_t stream) {
  if (nhead % 4 || (dh != 16 && dh != 32)) return TMAE_EARG;
  if (!attn_sizes_ok(mq, ldq) || !attn_sizes_ok(mk, ldk) || !attn_sizes_ok(mk, ldv) || !attn_sizes_ok(mq, lddo) ||
      !attn_sizes_ok(mq, lddq) || !attn_sizes_ok(mk, lddk) || !attn_sizes_ok(mk, lddv) ||
      (int64_t)batch * ny * nx >= ((int64_t)1 << 31))
    return TMAE_EARG;
  if ((lddq % 4) || (lddk % 4) || (lddv % 4) || ((uintptr_t)dq & 7) || ((uintptr_t)dk & 7) || ((uintptr_t)dv & 7))
    return TMAE_EARG;
  const int Wy = (ny + WIN - 1) / WIN + 1, Wx = (nx + WIN - 1) / WIN + 1;
  const int s = do_shift ? WIN / 2 : WIN;
  const int64_t nwin = (int64_t)batch * Wy * Wx;
  const int d = nhead * dh;
  AttnBufs nb = {};
  nb.q = attn_bytes(mq, ldq, d); nb.k = attn_bytes(mk, ldk, d); nb.v = attn_bytes(mk, ldv, d);
  nb.g = attn_bytes(mq, lddo, d); nb.lse = (unsigned)(mq * nhead * 4);
  nb.dq = attn_bytes(mq, lddq, d); nb.dk = attn_bytes(mk, lddk, d); nb.dv = attn_bytes(mk, lddv, d);
#define BWDM(DH, NT, PAIR, CLS, GX)                                                                                  \
  hipLaunchKernelGGL((win_attn_bwd_mfma_kernel<DH, NT, PAIR>), dim3((unsigned)(GX), (unsigned)(nhead / 4)), dim3(256), \
                     0, stream, (const __hip_bfloat16*)q, ldq, (const __hip_bfloat16*)k, ldk, (const __hip_bfloat16*)v, \
                     ldv, (const __hip_bfloat16*)out, ldo, (const __hip_bfloat16*)dout, lddo, lse, nhead, grid_q,     \
                     grid_k, ny, nx, Wy, Wx, s, s, tau, tau_min, (__hip_bfloat16*)dq, lddq, (__hip_bfloat16*)dk, lddk, \
                     (__hip_bfloat16*)dv, lddv, dtau_partial, worklist, CLS, nwin, nb, tau_stride)
  if (!worklist) {
    if (dh == 32) BWDM(32, 4, false, 0, nwin); else BWDM(16, 4, false, 0, nwin);
  } else {
    const int64_t g0 = class_grid(0, nwin, mq, mk), g1 = class_grid(1, nwin, mq, mk), g2 = class_grid(2, nwin, mq, mk),
                  g3 = class_grid(3, nwin, mq, mk);
    if (dh == 32) {
      if (g0 > 0) BWDM(32, 1, true, 0, g0);
      if (g1 > 0) BWDM(32, 1, false, 1, g1);
      if (g2 > 0) BWDM(32, 2, false, 2, g2);
      if (g3 > 0) BWDM(32, 4, false, 3, g3);
    } else {
      if (g0 > 0) BWDM(16, 1, true, 0, g0);
      if (g1 > 0) BWDM(16, 1, false, 1, g1);
      if (g2 > 0) BWDM(16, 2, false, 2, g2);
      if (g3 > 0) BWDM(16, 4, false, 3, g3);
    }
  }
#undef BWDM
  return (int)hipGetLastError();
}
