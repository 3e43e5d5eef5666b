// Weight / bias gradient of a Linear over a token list:  dW[n,k] = sum_m dY[m,n] * X[m,k],  db[n] = sum_m dY[m,n].
//
// Why a kernel of our own: on this path M is 1e5..4e5 tokens while N,K <= 512, so the output is a few tiles and
// the library GEMM runs on a handful of workgroups (measured 15-130 TFLOP/s, 0.25-0.7 TB/s; profiles/).  The work
// is one streaming pass over dY and X, so the bound is HBM.  Here the token axis is split over one resident round of
// workgroups; each streams its 32-row slices of dY and X through LDS once (16-byte coalesced loads, two slices in flight
// in registers, double-buffered LDS) and contracts
// them on the matrix cores.  The contraction index (token m) is the ROW index of both operands, i.e. both MFMA
// operands are needed "transposed"; gfx950's ds_read_b64_tr_b16 delivers exactly that from a row-major LDS image
// (XOR-swizzled so the transposed reads are bank-conflict free), so no transposed copy is ever materialised.
// Partial [N,K] blocks go to fp32 slabs (plain stores) and a second small kernel sums the slabs in a fixed order:
// deterministic, no atomics.
#include "common.h"
#include <cstdlib>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define WG_BN 128   // output rows (n) per workgroup
#define WG_BK 128   // output cols (k) per workgroup
#define WG_MS 32    // tokens per step = one 16x16x32 MFMA k-step

// byte offset of 16-byte chunk `ch` (0..15) of row `row` in a [rows][128 x bf16] LDS image; the XOR makes both the
// row-wise b128 stores and the transposed b64 reads of 16x16x32 operands conflict free (guide T10, layout (b)).
__device__ __forceinline__ int tile_off(int row, int ch) {
  return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}

__device__ __forceinline__ s16x4 tr_read(const char* lds_ptr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)lds_ptr);
}

// Row loads of the token-list (non-gathered) operands: BUFFER loads with a workgroup-uniform descriptor over the workgroup's own
// rows, a per-lane byte offset that never changes (row-in-slice x pitch + column) and the slice's offset as the SCALAR operand.
// Rounds 1-4 used flat loads from clamped 64-bit addresses computed per slice; the register allocator put those address
// temporaries into the destination registers of the loads still in flight for the OTHER parity, and the write-after-write
// hazard became `s_waitcnt vmcnt(1)`, `vmcnt(0)` at the top of every second step: "two slices in flight" was one (this is the
// 4.1 TB/s of the 256-tile kernel).  Here a step's loads need no VALU work at all, rows past the workgroup's range and columns
// past N / K return zeros from the descriptor's range check (no clamps, no selects when the registers are staged into LDS).
#define WG_OOB 0x7FFFFFF0u          // per-lane offset past every buffer (pitch x rows < 2^31, checked by the launcher)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wg_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
// the cell bytes of a lane's 8 tokens: same addressing (fixed per-lane offset 8 g, the slice's offset as the scalar operand)
__device__ __forceinline__ uint2 wg_load_cells(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
  const u32x2 v = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)voff, (int)soff, 0));
  return make_uint2(v[0], v[1]);
}
template <bool NT>
__device__ __forceinline__ u32x4 wg_load(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, (int)soff, NT ? 2 : 0));
}

// fragment (8 consecutive tokens 8g..8g+7 for column cb*16 + (lane&15)) of a [32][128] image
__device__ __forceinline__ bf16x8 load_frag(const char* img, int cb, int lane) {
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  const s16x4 lo = tr_read(img + tile_off(8 * g + q, 2 * cb + (p >> 1)) + 8 * (p & 1));
  const s16x4 hi = tr_read(img + tile_off(8 * g + 4 + q, 2 * cb + (p >> 1)) + 8 * (p & 1));
  s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return *reinterpret_cast<bf16x8*>(&v);
}

// CL (attention in-projections, ops.pos_proj): the kernel also returns the PER-CELL column sums of dY,
//   dcell[c, n] = sum over the tokens whose window cell has xc == c (c < 8) / yc == c - 8 (c >= 8) of dY[m, n],
// for the output rows n < pos_n -- the position part of the weight gradient is dcell^T . E (16 x k), so dY is not
// streamed a second time against a one-hot matrix.  cells [>= round_up(M, 32) + 64] u8 = xc | yc << 3.
// Here: the ones-vector MFMA of the bias gradient becomes a one-hot MFMA (16 cell columns), built per step from the
// cell bytes of the lane's 8 tokens.
template <bool G, bool CL>          // G: X rows are gathered through nbr (sparse-conv rulebook)
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const __hip_bfloat16* __restrict__ dY, int64_t ldy,
                                                   const __hip_bfloat16* __restrict__ X, int64_t ldx, int64_t M,
                                                   int N, int K, int rows_per_split, float* __restrict__ slab,
                                                   int64_t count, bool has_bias, int NB, int KB, int S, int per_xcd,
                                                   const int32_t* __restrict__ nbr, int cin,
                                                   const uint8_t* __restrict__ cells, int pos_n) {
  __shared__ __attribute__((aligned(16))) char lds[2][2][WG_MS * 256];
  __shared__ uint2 ldsC[2][4];         // CL: the slice's 32 cell bytes, staged with the rows (8 bytes per lane group g)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wn = w >> 1, wk = w & 1;
  // XCD-aware 1-D grid: block ids are dealt round-robin over the 8 XCDs (speed only, never correctness), so the
  // (n, k) output blocks of one token chunk s get ids of the same residue: they stream the same dY / X rows at the
  // same time and share them through that XCD's L2 instead of fetching them once per block from HBM.
  // per_xcd > 0 (round 6, the 9-block chunks of the d = 256 sparse convs): an XCD's workgroups are a CONTIGUOUS range of the list
  // (chunk, block) instead of whole chunks dealt in turn -- whole chunks filled 27 of an XCD's 32 slots; a chunk that straddles two
  // XCDs is fetched by both.  Measured per launch (profiles/round6_ab_wgrad_map.txt): 9 blocks -3..-7 %, 5 and 3 blocks nothing,
  // and the 2- / 4-block shapes (where both forms use every slot) +5..+8 % with the contiguous deal: those keep the round-robin one.
  const int per_s = NB * KB;
  int s, blk;
  if (per_xcd > 0) {
    const int L = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (L >= S * per_s) return;
    s = L / per_s; blk = L % per_s;
  } else {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    s = (j / per_s) * 8 + xcd;
    if (s >= S) return;
    blk = j % per_s;
  }
  const int n0 = (blk % NB) * WG_BN, kblk = blk / NB, k0 = kblk * WG_BK;
  const int64_t m_begin = (int64_t)s * rows_per_split;
  const int64_t m_end = min(M, m_begin + (int64_t)rows_per_split);
  const int steps = m_end > m_begin ? (int)((m_end - m_begin + WG_MS - 1) / WG_MS) : 0;
  // bias and per-cell sums: the four row tiles of a wave row (wn) are shared out over its two waves -- wave wk takes the tiles
  // {2 wk, 2 wk + 1} -- and are computed UNCONDITIONALLY (a branch inside the step makes the compiler drain vmcnt at its join):
  // wave wk walks the row tiles in the order a ^ 2 wk, so its two tiles are a = 0, 1 (literal register indices).  Results
  // nobody wants (no bias, k-blocks > 0, rows past pos_n) are simply not stored / multiplied by an all-zero one-hot operand.
  const bool want_bias = !G && has_bias && kblk == 0;          // workgroup-uniform
  const bool cell_wave = CL && kblk == 0;                      // writes the cell block of its rows (zeros if n0 >= pos_n)
  const bool want_cells = cell_wave && n0 < pos_n;
  const int rot = G ? 0 : 2 * wk;                              // row tile of accumulator a: a ^ rot
  // gathered X (sparse conv): column block k0 lies inside tap k0 / cin; row m reads feature row nbr[m, tap]
  const int tap = G ? k0 / cin : 0, c0 = G ? k0 % cin : k0;
  float* __restrict__ slab_w = slab + (int64_t)s * count;
  float* __restrict__ slab_b = slab_w + (int64_t)N * K;
  float* __restrict__ slab_c = slab_b + N;                     // [16][N] per-cell sums (CL)

  // register staging, two slices deep: while slice st is contracted out of LDS, the loads of slices st+1 and st+2
  // are in flight (32 KB per workgroup) -- one slice of prefetch left the kernel latency-bound at ~2.5 TB/s.
  // Slice t uses register set t & 1; the main loop is unrolled by two so that every set index is a constant.
  // EVERY global load is unconditional (clamped address) and its predicate is applied when the registers are written
  // to LDS: a load under a branch makes the compiler wait for it (s_waitcnt vmcnt(0)) before the next one is issued,
  // which serialised the four loads of a slice and left nothing in flight behind the MFMAs.
  // Addresses are a workgroup-uniform base (SGPRs) + a 32-bit byte offset: no 64-bit multiplies, no address VGPR pairs.
  u32x4 ry[2][2], rx[2][2];
  bool oky[2][2], okx[2][2];
  int xi[2][2] = {{0, 0}, {0, 0}};     // gathered mode: feature-row ids, fetched one slice before their row loads so
                                       // the row loads never wait on an index load (no dependent round trip)
  uint2 c8[2] = {{0u, 0u}, {0u, 0u}};  // CL: cell bytes of the lane's 8 tokens (8g .. 8g+7 of the slice)
  // (the cells buffer is padded to round_up(M, 32) + 64 bytes: slices up to one past the end are staged)
  const __amdgpu_buffer_rsrc_t rsC = wg_rsrc(CL ? cells + m_begin : nullptr, 0x7FFFFFFFu);
  const unsigned voc = 8u * (unsigned)(lane >> 4);
  const int rows_here = max((int)(m_end - m_begin), 0);
  const char* __restrict__ baseY = reinterpret_cast<const char*>(dY + m_begin * ldy);
  const char* __restrict__ baseX = reinterpret_cast<const char*>(G ? X : X + m_begin * ldx);
  const int32_t* __restrict__ baseI = G ? nbr + m_begin * 9 + tap : nullptr;
  const unsigned pitchY = (unsigned)ldy * 2u, pitchX = (unsigned)ldx * 2u;
  auto iload = [&](int step, int P) {
    if constexpr (G) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = step * WG_MS + ((tid + 256 * i) >> 4);
        xi[P][i] = baseI[(unsigned)min(r, rows_here - 1) * 9u];
      }
    }
  };
  // non-gathered operands: buffer loads (see wg_load): per-lane offsets of the thread's two chunks, fixed for the whole kernel
  const __amdgpu_buffer_rsrc_t rsY = wg_rsrc(baseY, (unsigned)rows_here * pitchY);
  const __amdgpu_buffer_rsrc_t rsX = wg_rsrc(baseX, G ? 0u : (unsigned)rows_here * pitchX);
  unsigned voy[2], vox[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = tid + 256 * i, row = c >> 4, ch = c & 15;
    const int cy = n0 + ch * 8, cx = k0 + ch * 8;
    voy[i] = cy < N ? (unsigned)row * pitchY + (unsigned)cy * 2u : WG_OOB;
    vox[i] = cx < K ? (unsigned)row * pitchX + (unsigned)(c0 + ch * 8) * 2u : WG_OOB;
  }
  auto gload = [&](int step, int P) {
    // (slices up to one past the end are staged: the cells buffer is padded for that)
    if constexpr (CL) c8[P] = wg_load_cells(rsC, voc, (unsigned)step * WG_MS);
    if constexpr (!G) {
      const unsigned soy = (unsigned)step * (WG_MS * pitchY), sox = (unsigned)step * (WG_MS * pitchX);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ry[P][i] = wg_load<false>(rsY, voy[i], soy);
        rx[P][i] = wg_load<false>(rsX, vox[i], sox);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c = tid + 256 * i, row = c >> 4, ch = c & 15;
        const int r = step * WG_MS + row, rc = min(r, rows_here - 1);
        const int cy = n0 + ch * 8, cx = k0 + ch * 8;
        oky[P][i] = r < rows_here && cy < N;
        ry[P][i] = *reinterpret_cast<const u32x4*>(baseY + ((unsigned)rc * pitchY + (unsigned)(cy < N ? cy : 0) * 2u));
        const unsigned xc = (unsigned)(cx < K ? c0 + ch * 8 : 0) * 2u;
        const int xr = xi[P][i];
        okx[P][i] = r < rows_here && cx < K && xr >= 0;
        rx[P][i] = *reinterpret_cast<const u32x4*>(baseX + ((int64_t)max(xr, 0) * pitchX + xc));
      }
    }
  };
  auto lwrite = [&](int buf, int P) {
    const u32x4 z = {0u, 0u, 0u, 0u};
    // the cell bytes travel like the rows: registers two slices ahead, LDS one slice ahead, read at use.  (Read from the
    // registers at the top of the step, the compiler waited there for loads of the PREVIOUS step: its count at the loop header)
    if constexpr (CL) { if (w == 0 && (lane & 15) == 0) ldsC[buf][lane >> 4] = c8[P]; }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = tid + 256 * i, row = c >> 4, ch = c & 15;
      if constexpr (!G) {
        *reinterpret_cast<u32x4*>(&lds[buf][0][tile_off(row, ch)]) = ry[P][i];
        *reinterpret_cast<u32x4*>(&lds[buf][1][tile_off(row, ch)]) = rx[P][i];
      } else {
        *reinterpret_cast<u32x4*>(&lds[buf][0][tile_off(row, ch)]) = oky[P][i] ? ry[P][i] : z;
        *reinterpret_cast<u32x4*>(&lds[buf][1][tile_off(row, ch)]) = okx[P][i] ? rx[P][i] : z;
      }
    }
  };

  f32x4 acc[4][4], accb[2], accs[CL ? 2 : 1];
#pragma unroll
  for (int a = 0; a < (CL ? 2 : 1); ++a) accs[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < 2; ++a) accb[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < 4; ++a) {
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  s16x8 ones_s = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};   // bf16 1.0
  const bf16x8 ones = *reinterpret_cast<bf16x8*>(&ones_s);

  if (steps > 0) {
    iload(0, 0);
    iload(1, 1);
    gload(0, 0);
    gload(1, 1);
    iload(2, 0);
    lwrite(0, 0);
  }
  __syncthreads();
  auto step = [&](int st, int P) {      // P == st & 1, a literal at both call sites
    // no branches around the loads / LDS writes (slices past the end read clamped rows and stage zeros): the
    // compiler's wait-count bookkeeping turns conservative (vmcnt(0)) at every control-flow join
    iload(st + 3, P ^ 1);     // index loads first: vmcnt retires in order, and the next step's row
    gload(st + 2, P);         // loads must be able to wait for the ids without draining these
    __builtin_amdgcn_sched_barrier(0);   // keep the loads up here and their consumers below the MFMAs: left alone, the
                                         // scheduler sinks the loads and hoists the waits to shorten live ranges
    uint2 ccur = {0u, 0u};
    if constexpr (CL) ccur = ldsC[P][lane >> 4];
    bf16x8 fa[4], fb[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      fa[t] = load_frag(lds[P][0], wn * 4 + (t ^ rot), lane);
      fb[t] = load_frag(lds[P][1], wk * 4 + t, lane);
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
    if constexpr (!G) {
#pragma unroll
      for (int a = 0; a < 2; ++a) accb[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], ones, accb[a], 0, 0, 0);
    }
    if constexpr (CL) {
      // one-hot B fragment of THIS slice (its cell bytes were loaded two steps ago with the rows): column lane & 15
      // = cell slot (0..7 xc, 8..15 yc), the lane's 8 tokens 8g .. 8g+7; a workgroup past the position columns matches nothing
      const int ci_ = lane & 15;
      const unsigned want_ = want_cells ? (unsigned)(ci_ & 7) : 8u, sh_ = ci_ < 8 ? 0u : 3u;
      unsigned wds[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned src = j < 2 ? ccur.x : ccur.y;
        const unsigned ca = (src >> (16 * (j & 1))) & 0xFFu, cb = (src >> (16 * (j & 1) + 8)) & 0xFFu;
        wds[j] = ((((ca >> sh_) & 7u) == want_) ? 0x00003F80u : 0u) | ((((cb >> sh_) & 7u) == want_) ? 0x3F800000u : 0u);
      }
      const u32x4 ohu = {wds[0], wds[1], wds[2], wds[3]};
      const bf16x8 oh = __builtin_bit_cast(bf16x8, ohu);
#pragma unroll
      for (int a = 0; a < 2; ++a) accs[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], oh, accs[a], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    lwrite(P ^ 1, P ^ 1);
    __syncthreads();
  };
  for (int st = 0; st < steps; st += 2) {
    step(st, 0);
    if (st + 1 < steps) step(st + 1, 1);
  }
  // C layout of mfma_f32_16x16x*: lane holds rows 4*(lane>>4)+r, column lane&15
  const int g = lane >> 4, ci = lane & 15;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + (wn * 4 + (a ^ rot)) * 16 + 4 * g + r;
      if (n >= N) continue;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int k = k0 + (wk * 4 + b) * 16 + ci;
        if (k < K) slab_w[(int64_t)n * K + k] = acc[a][b][r];
      }
      if (a < 2) {
        if (want_bias && ci == 0) slab_b[n] = accb[a][r];
        if constexpr (CL) { if (cell_wave) slab_c[(int64_t)ci * N + n] = accs[a][r]; }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// 256 x 256 output tile, 8 waves (wave tile 128 x 64): used when n >= 256 and k >= 256 (the d = 256 stages and their
// sparse convs).  With the 128 x 128 tile those shapes re-read every operand 2-18x through L2 and ran at ~2 TB/s of
// algorithmic bytes; here each workgroup streams 32 KB per slice for four times the MFMA work.  One workgroup per CU
// (236-244 registers, 64 KB LDS), two slices of loads in flight.  Each operand slice is kept as two [32][128] images so
// the swizzle and the fragment reads are those of the small kernel.  The bias gradient is summed on the VALU from the
// staged dY chunks (a thread always stages the same 8 columns), not with an extra MFMA: no accumulator registers.
// ------------------------------------------------------------------------------------------------
#define WG2_B 256
// CL (round 5): the per-cell column sums of dY (see wgrad_kernel) inside this kernel too.  Rounds 2-4 took them from a second, narrow
// pass of the 128-tile kernel over dY ("no registers left for a one-hot MFMA": true for a ninth accumulator column PER WAVE TILE ROW,
// i.e. 32 registers) -- 24 launches and 1.75 ms per step that only re-read dY.  Here the 8 row tiles of a wave row (wn) are shared out
// over its four waves: wave wk adds the one-hot product for the tiles {2 wk, 2 wk + 1} only -- 2 MFMAs on top of 32 per step and 8
// accumulator registers.  To keep every register index a literal without a branch per tile, wave wk walks the row tiles in the
// order a ^ 2 wk (a bijection on 0..7 that puts ITS two tiles at a = 0, 1); in the swizzled image that rotation is one XOR of the
// fragment's byte offset with 64 wk.
// VAR (A/B, csrc/common.h TMAE_AB): 1 = waves 4-7 -- the SIMD partners of waves 0-3 -- stage the next slice into LDS BEFORE their
// MFMA block instead of after it, so that on every SIMD one wave's LDS stores run beside the other's matrix work
// (MI355X_MICROARCH.md, "Two waves per SIMD", item 9: partners running the same program in lockstep); 2 = 1 + raised priority
// around the MFMA block.
template <bool G, int NTL = 0, int VAR = 0, bool CL = false>      // NTL: bit 0 / 1 = dY / X rows are loaded with the non-temporal policy (read by one workgroup only)
__global__ __launch_bounds__(512, 2) void wgrad256_kernel(const __hip_bfloat16* __restrict__ dY, int64_t ldy,
                                                         const __hip_bfloat16* __restrict__ X, int64_t ldx, int64_t M,
                                                         int N, int K, int rows_per_split, float* __restrict__ slab,
                                                         int64_t count, bool has_bias, int NB, int KB, int S, int per_xcd,
                                                         const int32_t* __restrict__ nbr, int cin,
                                                         const uint8_t* __restrict__ cells, int pos_n) {
  static_assert(!(G && CL), "per-cell sums belong to the Linear in-projections, not to the gathered sparse-conv gradient");
  __shared__ __attribute__((aligned(16))) char lds[2][2][2][WG_MS * 256];     // [buffer][operand][column half]
  __shared__ uint2 ldsC[2][4];         // CL: the slice's 32 cell bytes, staged with the rows (8 bytes per lane group g)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wn = w >> 2, wk = w & 3;
  const int per_s = NB * KB;
  int s, blk;
  if (per_xcd > 0) {                                                   // see wgrad_kernel
    const int L = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (L >= S * per_s) return;
    s = L / per_s; blk = L % per_s;
  } else {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    s = (j / per_s) * 8 + xcd;
    if (s >= S) return;
    blk = j % per_s;
  }
  const int n0 = (blk % NB) * WG2_B, kblk = blk / NB, k0 = kblk * WG2_B;
  const int64_t m_begin = (int64_t)s * rows_per_split;
  const int64_t m_end = min(M, m_begin + (int64_t)rows_per_split);
  const int steps = m_end > m_begin ? (int)((m_end - m_begin + WG_MS - 1) / WG_MS) : 0;
  const bool want_bias = has_bias && kblk == 0;                 // workgroup-uniform
  const bool cell_wg = CL && kblk == 0;                         // writes the cell block of its rows (zeros if n0 >= pos_n)
  const bool want_cells = cell_wg && n0 < pos_n;
  float* __restrict__ slab_w = slab + (int64_t)s * count;
  float* __restrict__ slab_b = slab_w + (int64_t)N * K;
  float* __restrict__ slab_c = slab_b + N;                      // [16][N] per-cell sums (CL)
  const int rot = CL ? 2 * wk : 0;                              // row tile of accumulator a: a ^ rot
  // staging: thread owns 16-byte chunk ch (0..31) of rows r0 and r0 + 16 of both operands
  const int r0 = tid >> 5, ch = tid & 31;
  const int cy = n0 + ch * 8, cx = k0 + ch * 8;
  const int tap = G ? cx / cin : 0, xcol = G ? cx % cin : cx;   // gathered X: column cx lies inside tap cx / cin
  // unconditional loads from clamped addresses, predicates applied at the LDS write (see wgrad_kernel)
  u32x4 ry[2][2], rx[2][2];
  bool oky[2][2], okx[2][2];
  int xi[2][2] = {{0, 0}, {0, 0}};
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  uint2 c8[2] = {{0u, 0u}, {0u, 0u}};  // CL: cell bytes of the lane's 8 tokens (8g .. 8g+7 of the slice)
  // (the cells buffer is padded to round_up(M, 32) + 64 bytes: slices up to one past the end are staged)
  const __amdgpu_buffer_rsrc_t rsC = wg_rsrc(CL ? cells + m_begin : nullptr, 0x7FFFFFFFu);
  const unsigned voc = 8u * (unsigned)(lane >> 4);
  const int rows_here = max((int)(m_end - m_begin), 0);
  const unsigned pitchY = (unsigned)ldy * 2u, pitchX = (unsigned)ldx * 2u;
  const char* __restrict__ baseY = reinterpret_cast<const char*>(dY + m_begin * ldy) + (cy < N ? cy : 0) * 2;
  const char* __restrict__ baseX = reinterpret_cast<const char*>(G ? X : X + m_begin * ldx) + (cx < K ? xcol : 0) * 2;
  const int32_t* __restrict__ baseI = G ? nbr + m_begin * 9 + (cx < K ? tap : 0) : nullptr;
  const bool coly = cy < N, colx = cx < K;
  auto iload = [&](int step, int P) {
    if constexpr (G) {
#pragma unroll
      for (int i = 0; i < 2; ++i) xi[P][i] = baseI[(unsigned)min(step * WG_MS + r0 + 16 * i, rows_here - 1) * 9u];
    }
  };
  // non-gathered operands: buffer loads (see wg_load): per-lane offsets of the thread's rows r0 and r0 + 16, fixed for the kernel
  const __amdgpu_buffer_rsrc_t rsY = wg_rsrc(reinterpret_cast<const char*>(dY + m_begin * ldy), (unsigned)rows_here * pitchY);
  const __amdgpu_buffer_rsrc_t rsX = wg_rsrc(reinterpret_cast<const char*>(X + (G ? 0 : m_begin * ldx)), G ? 0u : (unsigned)rows_here * pitchX);
  unsigned voy[2], vox[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    voy[i] = coly ? (unsigned)(r0 + 16 * i) * pitchY + (unsigned)cy * 2u : WG_OOB;
    vox[i] = colx ? (unsigned)(r0 + 16 * i) * pitchX + (unsigned)xcol * 2u : WG_OOB;
  }
  auto gload = [&](int step, int P) {
    // (slices up to one past the end are staged: the cells buffer is padded for that.  Unconditional, like every load of this
    // loop: a branch inside the step makes the compiler drain vmcnt at its join, i.e. wait for the slices meant to stay in flight)
    if constexpr (CL) c8[P] = wg_load_cells(rsC, voc, (unsigned)step * WG_MS);
    if constexpr (!G) {
      const unsigned soy = (unsigned)step * (WG_MS * pitchY), sox = (unsigned)step * (WG_MS * pitchX);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ry[P][i] = wg_load<(NTL & 1) != 0>(rsY, voy[i], soy);
        rx[P][i] = wg_load<(NTL & 2) != 0>(rsX, vox[i], sox);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = step * WG_MS + r0 + 16 * i, rc = min(r, rows_here - 1);
        oky[P][i] = r < rows_here && coly;
        ry[P][i] = *reinterpret_cast<const u32x4*>(baseY + (unsigned)rc * pitchY);
        const int xr = xi[P][i];
        okx[P][i] = r < rows_here && colx && xr >= 0;
        rx[P][i] = *reinterpret_cast<const u32x4*>(baseX + (int64_t)max(xr, 0) * pitchX);
      }
    }
  };
  auto lwrite = [&](int buf, int P) {
    const u32x4 z = {0u, 0u, 0u, 0u};
    // the cell bytes travel like the rows: registers two slices ahead, LDS one slice ahead, read at use.  (Read from the
    // registers at the top of the step, the compiler waited there for loads of the PREVIOUS step: its count at the loop header)
    if constexpr (CL) { if (w == 0 && (lane & 15) == 0) ldsC[buf][lane >> 4] = c8[P]; }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int off = tile_off(r0 + 16 * i, ch & 15);
      const u32x4 vy = G ? (oky[P][i] ? ry[P][i] : z) : ry[P][i];
      *reinterpret_cast<u32x4*>(&lds[buf][0][ch >> 4][off]) = vy;
      *reinterpret_cast<u32x4*>(&lds[buf][1][ch >> 4][off]) = G ? (okx[P][i] ? rx[P][i] : z) : rx[P][i];
      if (want_bias) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          bsum[2 * q] += __uint_as_float(vy[q] << 16);
          bsum[2 * q + 1] += __uint_as_float(vy[q] & 0xFFFF0000u);
        }
      }
    }
  };

  f32x4 acc[8][4], accs[CL ? 2 : 1];
#pragma unroll
  for (int a = 0; a < (CL ? 2 : 1); ++a) accs[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (steps > 0) {
    iload(0, 0);
    iload(1, 1);
    gload(0, 0);
    gload(1, 1);
    iload(2, 0);
    lwrite(0, 0);
  }
  __syncthreads();
  auto step = [&](int st, int P) {
    // no branches around the loads / LDS writes (slices past the end read clamped rows and stage zeros): the
    // compiler's wait-count bookkeeping turns conservative (vmcnt(0)) at every control-flow join
    iload(st + 3, P ^ 1);     // index loads first: vmcnt retires in order, and the next step's row
    gload(st + 2, P);         // loads must be able to wait for the ids without draining these
    __builtin_amdgcn_sched_barrier(0);   // keep the loads up here and their consumers below the MFMAs: left alone, the
                                         // scheduler sinks the loads and hoists the waits to shorten live ranges
    const bool early = VAR >= 1 && w >= 4;         // wave-uniform
    if (early) {
      lwrite(P ^ 1, P ^ 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    uint2 ccur = {0u, 0u};
    if constexpr (CL) ccur = ldsC[P][lane >> 4];
    bf16x8 fb[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) fb[t] = load_frag(lds[P][1][wk >> 1], (wk & 1) * 4 + t, lane);
    bf16x8 oh;
    if constexpr (CL) {
      // one-hot B fragment of THIS slice (wgrad_kernel): column lane & 15 = cell slot (0..7 xc, 8..15 yc), the lane's 8 tokens
      const int ci_ = lane & 15;
      // a workgroup past the position columns (n0 >= pos_n) matches nothing: its products are zeros, no branch in the loop
      const unsigned want_ = want_cells ? (unsigned)(ci_ & 7) : 8u, sh_ = ci_ < 8 ? 0u : 3u;
      unsigned wds[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned src = j < 2 ? ccur.x : ccur.y;
        const unsigned ca = (src >> (16 * (j & 1))) & 0xFFu, cb = (src >> (16 * (j & 1) + 8)) & 0xFFu;
        wds[j] = ((((ca >> sh_) & 7u) == want_) ? 0x00003F80u : 0u) | ((((cb >> sh_) & 7u) == want_) ? 0x3F800000u : 0u);
      }
      const u32x4 ohu = {wds[0], wds[1], wds[2], wds[3]};
      oh = __builtin_bit_cast(bf16x8, ohu);
    }
    if constexpr (VAR == 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      const bf16x8 fa = load_frag(lds[P][0][wn], a ^ rot, lane);
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb[b], acc[a][b], 0, 0, 0);
      if constexpr (CL) {
        if (a < 2) accs[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, oh, accs[a], 0, 0, 0);
      }
    }
    if constexpr (VAR == 2) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    if (!early) lwrite(P ^ 1, P ^ 1);
    __syncthreads();
  };
  for (int st = 0; st < steps; st += 2) {
    step(st, 0);
    if (st + 1 < steps) step(st + 1, 1);
  }
  const int g = lane >> 4, ci = lane & 15;
#pragma unroll
  for (int a = 0; a < 8; ++a) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + wn * 128 + (a ^ rot) * 16 + 4 * g + r;
      if (n >= N) continue;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int k = k0 + wk * 64 + b * 16 + ci;
        if (k < K) slab_w[(int64_t)n * K + k] = acc[a][b][r];
      }
      if constexpr (CL) {
        if (a < 2) { if (cell_wg) slab_c[(int64_t)ci * N + n] = accs[a][r]; }
      }
    }
  }
  if (want_bias) {                       // 16 threads (r0 = 0..15) hold partial sums of the same 8 columns
    float* red = reinterpret_cast<float*>(&lds[0][0][0][0]);     // [16][256] floats = 16 KB, the loop is done with LDS
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; ++q) red[r0 * 256 + ch * 8 + q] = bsum[q];
    __syncthreads();
    if (tid < 256 && n0 + tid < N) {
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) t += red[r * 256 + tid];
      slab_b[n0 + tid] = t;
    }
  }
}

// Slab reduction, two levels in ONE launch, both in a fixed order (deterministic): a 1024-thread workgroup owns 64
// float4 of the output; wavefront y sums the slabs y, y+RG, y+2RG, ... for them (one float4 per lane and slab, all
// loads independent), the 16 partial sums meet in LDS and wavefront 0 adds them in the order y = 0..15 and writes dW /
// db.  (Two launches with the partials in global memory cost a second ~5 us dispatch per weight gradient, 112 per
// step; a last-arriving-block finish across workgroups needs a device-scope fence per block and measured 2-4x slower.)
#define WG_RG 16
// posE (round 5, the in-projections with the position embedding folded into the GEMM): E [16, k] fp32, the separable embedding.
// The weight gradient of (x + pos) W^T is dY^T x + dcell^T E; rounds 2-4 added the second term with a [n,16] x [16,k] library GEMM
// per in-projection (24 launches of 6-17 us per step).  Here the workgroup that finishes the weight elements of output row j
// also sums the 16 per-cell partials of row j over the slabs (one extra 64-byte load per wave and slab) and adds
// sum_c dcell[c, j] E[c, :] to the row before it is written.  Needs 256 % k == 0 or k % 256 == 0 and n k % 256 == 0 (checked by
// the launcher): a workgroup's 256 elements then cover whole rows or a piece of one.
__global__ __launch_bounds__(64 * WG_RG) void wgrad_reduce_kernel(const float* __restrict__ slab, int splits,
                                                                 int64_t count, int n, int k, float* __restrict__ dw,
                                                                 float* __restrict__ db, float* __restrict__ dc,
                                                                 int ldc, const float* __restrict__ posE) {
  __shared__ float4 red[WG_RG][64];
  __shared__ float redc[WG_RG][64], cs[64];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t e0 = (int64_t)blockIdx.x * 256, e = e0 + lane * 4;
  const int64_t nk = (int64_t)n * k;
  // the per-cell partials this workgroup needs: lane rr * 16 + c <-> cell slot c of its rr-th row (rows e0 / k ...)
  const bool pe = posE != nullptr && e0 < nk;                               // workgroup-uniform
  const int nrows = k >= 256 ? 1 : 256 / k, row0 = (int)(e0 / k);
  const bool cl_lane = pe && lane < 16 * nrows && row0 + (lane >> 4) < n;
  const int64_t ce = nk + n + (int64_t)(lane & 15) * n + row0 + (lane >> 4);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float cacc = 0.f;
  if (e < count) {
    // four slabs in flight per wave (round 6: the rolled loop waited for every load before it issued the next -- 8 to 16 dependent
    // round trips per wave); the sums are taken in the same order as before
    int s = w;
    for (; s + 3 * WG_RG < splits; s += 4 * WG_RG) {
      float4 v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const float4*>(slab + (int64_t)(s + q * WG_RG) * count + e);
#pragma unroll
      for (int q = 0; q < 4; ++q) { acc.x += v[q].x; acc.y += v[q].y; acc.z += v[q].z; acc.w += v[q].w; }
    }
    for (; s < splits; s += WG_RG) {
      const float4 v = *reinterpret_cast<const float4*>(slab + (int64_t)s * count + e);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  if (cl_lane)
    for (int s = w; s < splits; s += WG_RG) cacc += slab[(int64_t)s * count + ce];
  red[w][lane] = acc;
  redc[w][lane] = cacc;
  __syncthreads();
  if (w == 0) {
    if (pe) {
      float c = 0.f;
#pragma unroll
      for (int y = 0; y < WG_RG; ++y) c += redc[y][lane];
      cs[lane] = c;                                      // same wave reads it below: LDS operations of a wave run in order
      __builtin_amdgcn_wave_barrier();
    }
    if (e < count) {
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int y = 0; y < WG_RG; ++y) { const float4 v = red[y][lane]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
      if (pe) {
        const int lr = (lane * 4) / k, kc = (int)((e - (int64_t)(row0 + lr) * k));      // local row, column of the lane's 4 elements
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          const float dcv = cs[lr * 16 + c];
          const float4 ev = *reinterpret_cast<const float4*>(posE + (int64_t)c * k + kc);
          t.x += dcv * ev.x; t.y += dcv * ev.y; t.z += dcv * ev.z; t.w += dcv * ev.w;
        }
      }
      const float tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t idx = e + j;
        if (idx < nk) dw[idx] = tv[j];
        else if (idx < nk + n) { if (db) db[idx - nk] = tv[j]; }
        else if (dc && idx < nk + 17 * (int64_t)n) {                             // [16][n] per-cell sums -> pitch ldc
          const int64_t e2 = idx - nk - n;
          dc[(e2 / n) * ldc + e2 % n] = tv[j];
        }
      }
    }
  }
}

// a single 256 x 256 output tile (n = k = 256) stays with the small kernel: there the big tile only doubles the slab
// traffic (256 splits instead of 128) -- measured 5-15 % slower
static bool wgrad_big_tile(int n, int k) {
  return n >= WG2_B && k >= WG2_B && ((n + WG2_B - 1) / WG2_B) * ((k + WG2_B - 1) / WG2_B) >= 2;
}

// Token split.  A workgroup id is dealt to XCD id % 8 and all output tiles of one token chunk go to the same XCD
// (they share dY / X through that L2), so chunks are handed out per XCD: as many as fit the XCD's resident
// workgroups in ONE round (32 CUs x 2 for the 128-tile kernel, x 1 for the 256-tile kernel) -- a second, partly
// filled round costs a whole extra pass (measured: 576 workgroups on 512 slots ran 1.8x longer than 504).
static bool wgrad_contig(int per_xcd, int nb) {
  const int whole = 8 * (per_xcd / nb) * nb, list = (8 * per_xcd) / nb * nb;
  return nb <= per_xcd && list * 10 >= whole * 11;
}

static void wgrad_plan(int64_t m, int n, int k, int& splits, int& rows_per_split) {
  const bool big = wgrad_big_tile(n, k);
  const int bt = big ? WG2_B : WG_BN;
  const int nb = ((n + bt - 1) / bt) * ((k + bt - 1) / bt);
  static const int slots = TMAE_AB_INT("TMAE_WGRAD_SLOTS", 0);
  const int per_xcd = slots > 0 ? slots : (big ? 32 : 64);
  int64_t c = per_xcd / nb;
  if (c < 1) c = 1;
  int64_t s = 8 * c;
  // whole chunks per XCD leave slots idle when nb does not divide per_xcd; where that is >= 10 % of the chip (nb = 9: 8 x 3 x 9 = 216 of
  // 256) the chunks are dealt as one contiguous list instead (wgrad_contig): 28 x 9 = 252
  if (wgrad_contig(per_xcd, nb)) s = (8 * per_xcd) / nb;
  const int64_t max_s = (m + 255) / 256;                  // at least 8 steps per workgroup
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  int64_t rows = (m + s - 1) / s;
  rows = (rows + WG_MS - 1) / WG_MS * WG_MS;
  if (rows < WG_MS) rows = WG_MS;
  splits = (int)((m + rows - 1) / rows);
  if (splits < 1) splits = 1;
  rows_per_split = (int)rows;
}

// slab row = [n*k weight partials | n bias partials | (cells: 16*n per-cell partials) | pad to a multiple of 4 floats]
static int64_t slab_count(int n, int k, bool cl = false) { return (((int64_t)n * k + n + (cl ? 16 * n : 0)) + 3) / 4 * 4; }

size_t tmae_linear_wgrad_workspace(int64_t m, int n, int k) {
  int splits, rows;
  wgrad_plan(m, n, k, splits, rows);
  return tmae_align((size_t)splits * slab_count(n, k, true) * 4) + tmae_align((size_t)n * 8 * 4) + 1024;   // (cells variant too)
}

static int wgrad_launch(const void* dy, int64_t ldy, const void* x, int64_t ldx, int64_t m, int n, int k, float* dw,
                        float* db, const int32_t* nbr, int cin, const uint8_t* cells, int pos_n, float* dc, int ldc,
                        void* wsp, size_t ws_bytes, hipStream_t stream, const float* posE = nullptr) {
  if (m < 0 || n <= 0 || k <= 0 || !dw || (n % 8) || (k % 8) || (ldy % 8) || (ldx % 8)) return TMAE_EARG;
  if (nbr && (cin <= 0 || cin % WG_BK || k != 9 * cin || (((uintptr_t)nbr) & 3))) return TMAE_EARG;
  if (cells && (nbr || (!dc && !posE) || pos_n < 0 || pos_n > n || (dc && ldc < n) || (((uintptr_t)cells) & 7))) return TMAE_EARG;
  if (posE && (!cells || (((uintptr_t)posE) & 15) || !((256 % k) == 0 || (k % 256) == 0) || k < 64 || (((int64_t)n * k) % 256)))
    return TMAE_EARG;
  // the reduction adds dcell^T E to EVERY row of dw and relies on zero cell sums for the rows >= pos_n; the kernels mask the cell sums
  // per output tile, so the position rows must end on a tile boundary (d = 64 cross k | v would have n = 128, pos_n = 64)
  if (posE && pos_n != n && (pos_n % (wgrad_big_tile(n, k) ? WG2_B : WG_BN))) return TMAE_EARG;
  if (m > 0 && (!dy || !x)) return TMAE_EARG;
  if (((uintptr_t)dy & 15) || ((uintptr_t)x & 15)) return TMAE_EARG;
  int splits, rows;
  wgrad_plan(m, n, k, splits, rows);
  const bool cl = cells != nullptr;
  const int64_t count = slab_count(n, k, cl);
  // the kernels address a workgroup's rows with 32-bit byte offsets from its first row
  if ((int64_t)rows * (ldy > ldx ? ldy : ldx) * 2 >= (int64_t)1 << 31 || (nbr && (int64_t)rows * 36 >= (int64_t)1 << 31)) return TMAE_EARG;
  WsCarver ws(wsp, ws_bytes);
  float* slab = ws.take<float>((size_t)splits * count);
  if (!ws.ok) return TMAE_EWS;
  // without a bias the slabs' bias columns stay unwritten; the reduction discards those sums
#define WG_ARGS (const __hip_bfloat16*)dy, ldy, (const __hip_bfloat16*)x, ldx, m, n, k, rows, slab, count, db != nullptr, NB, \
                KB, splits, per_x, nbr, cin
  if (wgrad_big_tile(n, k)) {
    const int NB = (n + WG2_B - 1) / WG2_B, KB = (k + WG2_B - 1) / WG2_B;
    // per_x > 0: blocks per XCD residue, a contiguous range of (chunk, block); 0: whole chunks dealt round-robin
    const int per_x = wgrad_contig(32, NB * KB) ? (splits * NB * KB + 7) / 8 : 0;
    const unsigned nblocks = per_x ? 8u * (unsigned)per_x : 8u * (unsigned)((splits + 7) / 8) * (unsigned)(NB * KB);
    if (nbr) hipLaunchKernelGGL((wgrad256_kernel<true>), dim3(nblocks), dim3(512), 0, stream, WG_ARGS, cells, pos_n);
    else if (cl) {                 // per-cell sums inside the 256-tile kernel (dY non-temporal when nobody re-reads it)
      if (KB == 1) hipLaunchKernelGGL((wgrad256_kernel<false, 1, 0, true>), dim3(nblocks), dim3(512), 0, stream, WG_ARGS, cells, pos_n);
      else hipLaunchKernelGGL((wgrad256_kernel<false, 0, 0, true>), dim3(nblocks), dim3(512), 0, stream, WG_ARGS, cells, pos_n);
    } else {
      // dY columns belong to one n-block each: with a single k-block nobody re-reads them; likewise X with one n-block
      // (measured on the priced shape: -3 %; TMAE_WGRAD_NT=0 in a -DTMAE_AB build turns it off)
      static const int ntl_env = TMAE_AB_INT("TMAE_WGRAD_NT", 3);
      const int ntl = ntl_env & ((KB == 1 ? 1 : 0) | (NB == 1 ? 2 : 0));
#ifdef TMAE_AB
      static const int var = TMAE_AB_INT("TMAE_WGRAD_VAR", 0);
      if (var == 1 || var == 2) {
#define WG_V(NT_, V_) hipLaunchKernelGGL((wgrad256_kernel<false, NT_, V_>), dim3(nblocks), dim3(512), 0, stream, WG_ARGS, cells, pos_n)
        if (var == 1) { if (ntl == 1) WG_V(1, 1); else if (ntl == 2) WG_V(2, 1); else if (ntl == 3) WG_V(3, 1); else WG_V(0, 1); }
        else { if (ntl == 1) WG_V(1, 2); else if (ntl == 2) WG_V(2, 2); else if (ntl == 3) WG_V(3, 2); else WG_V(0, 2); }
#undef WG_V
      } else
#endif
      if (ntl == 1) hipLaunchKernelGGL((wgrad256_kernel<false, 1>), dim3(nblocks), dim3(512), 0, stream, WG_ARGS, cells, pos_n);
      else if (ntl == 2) hipLaunchKernelGGL((wgrad256_kernel<false, 2>), dim3(nblocks), dim3(512), 0, stream, WG_ARGS, cells, pos_n);
      else if (ntl == 3) hipLaunchKernelGGL((wgrad256_kernel<false, 3>), dim3(nblocks), dim3(512), 0, stream, WG_ARGS, cells, pos_n);
      else hipLaunchKernelGGL((wgrad256_kernel<false, 0>), dim3(nblocks), dim3(512), 0, stream, WG_ARGS, cells, pos_n);
    }
  } else {
    const int NB = (n + WG_BN - 1) / WG_BN, KB = (k + WG_BK - 1) / WG_BK;
    const int per_x = wgrad_contig(64, NB * KB) ? (splits * NB * KB + 7) / 8 : 0;
    const unsigned nblocks = per_x ? 8u * (unsigned)per_x : 8u * (unsigned)((splits + 7) / 8) * (unsigned)(NB * KB);
    if (nbr) hipLaunchKernelGGL((wgrad_kernel<true, false>), dim3(nblocks), dim3(256), 0, stream, WG_ARGS, cells, pos_n);
    else if (cl) hipLaunchKernelGGL((wgrad_kernel<false, true>), dim3(nblocks), dim3(256), 0, stream, WG_ARGS, cells, pos_n);
    else hipLaunchKernelGGL((wgrad_kernel<false, false>), dim3(nblocks), dim3(256), 0, stream, WG_ARGS, cells, pos_n);
  }
#undef WG_ARGS
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(tmae_cdiv(count / 4, 64)), dim3(64 * WG_RG), 0, stream, slab, splits,
                     count, n, k, dw, db, dc, ldc, posE);
  return tmae_launch_status();
}

int tmae_linear_wgrad(const void* dy, int64_t ldy, const void* x, int64_t ldx, int64_t m, int n, int k, float* dw,
                      float* db, void* wsp, size_t ws_bytes, void* stream_) {
  (void)hipGetLastError();
  return wgrad_launch(dy, ldy, x, ldx, m, n, k, dw, db, nullptr, 0, nullptr, 0, nullptr, 0, wsp, ws_bytes,
                      (hipStream_t)stream_);
}

int tmae_linear_wgrad_cells(const void* dy, int64_t ldy, const void* x, int64_t ldx, int64_t m, int n, int k,
                            const uint8_t* cells, int pos_n, const float* pos_e, float* dw, float* db, float* dcell,
                            void* wsp, size_t ws_bytes, void* stream_) {
  (void)hipGetLastError();
  if (!cells || (!dcell && !pos_e)) return TMAE_EARG;
  return wgrad_launch(dy, ldy, x, ldx, m, n, k, dw, db, nullptr, 0, cells, pos_n, dcell, n, wsp, ws_bytes,
                      (hipStream_t)stream_, pos_e);
}

int tmae_spconv_wgrad(const void* dy, int64_t ldy, const void* feat, int64_t ldf, const int32_t* nbr, int64_t m_out,
                      int cout, int cin, float* dw, void* wsp, size_t ws_bytes, void* stream_) {
  (void)hipGetLastError();
  if (!nbr) return TMAE_EARG;
  return wgrad_launch(dy, ldy, feat, ldf, m_out, cout, 9 * cin, dw, nullptr, nbr, cin, nullptr, 0, nullptr, 0, wsp,
                      ws_bytes, (hipStream_t)stream_);
}
