// A9: the sparse convolution itself as an implicit GEMM (spconv's gather-GEMM-scatter, pcdet/utils/spconv_utils.py:37-56,
// spt_backbone.py:280-304), forward and input gradient:
//     out[o, n] = sum_{t < 9} sum_{c < CIN}  in[nbr[o, t], c] * W[n, t * CIN + c]          (nbr < 0: no such neighbour)
// The [m, 9 CIN] im2col matrix of the gather + library-GEMM formulation (2.1 GB written and read back for a d = 256
// stage) never exists: a workgroup owns a 128-row x 128-column output tile, walks over the 9 taps x CIN / 64 slices of
// the contraction and stages, per slice, the 128 gathered input rows (64 channels each: one 128-byte piece per row)
// and the 128 weight rows through LDS -- register-staged one slice ahead (issue the next slice's global loads, contract
// the current one, write the next one into the other LDS buffer, one barrier per slice).
// bf16 in, fp32 accumulate (v_mfma_f32_16x16x32_bf16), bf16 out.  The products are taken "swapped" (rows = output
// channels, column = token) over weight rows permuted in LDS so that a pair of column tiles gives a lane 8 CONSECUTIVE
// output channels and the four lanes of a token 32: two 16-byte stores per lane, each instruction writing a contiguous
// 64-byte segment per token (no holes inside a line: csrc/token_gemm.hip, csrc/token_gemm_wreg.hip).  The input gradient is the same kernel on the transposed rulebook and the transposed weight.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define IG_BM 128
#define IG_BN 128
#define IG_BK 64
#define IG_PITCH (IG_BK * 2 + 16)      // bytes per staged row: +16 spreads 16 consecutive rows over all banks

__device__ __forceinline__ unsigned ig_bf16_bits(float v) {
  return (unsigned)__builtin_bit_cast(unsigned short, __float2bfloat16(v));
}

#ifdef TMAE_AB   // the retired register-staged 128 x 128 kernel: only in the A/B debug build (common.h)
template <int CIN>
__global__ __launch_bounds__(256, 2) void spconv_igemm_kernel(const __hip_bfloat16* __restrict__ feat, int64_t ldf,
                                                             const int32_t* __restrict__ nbr, int64_t m_out,
                                                             const __hip_bfloat16* __restrict__ W, int cout,
                                                             __hip_bfloat16* __restrict__ out, int64_t ldo) {
  constexpr int KC = CIN / IG_BK;                      // slices per tap
  constexpr int STEPS = 9 * KC;
  __shared__ __attribute__((aligned(16))) char As[2][IG_BM * IG_PITCH];
  __shared__ __attribute__((aligned(16))) char Bs[2][IG_BN * IG_PITCH];
  __shared__ int nb[IG_BM * 9];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, i = lane & 15;
  // column tiles of one row tile run on adjacent block ids of one XCD residue (blockIdx.x % 8 constant): they gather the
  // same input rows through that XCD's L2
  const int nct = cout / IG_BN;
  const int64_t bid = blockIdx.x;
  const int64_t rt = (bid / (8 * nct)) * 8 + (bid & 7);
  const int ct = (int)((bid >> 3) % nct);
  const int64_t row0 = rt * IG_BM;
  if (row0 >= m_out) return;
  const int n0 = ct * IG_BN;
  for (int e = tid; e < IG_BM * 9; e += 256) {
    const int64_t r = row0 + e / 9;
    nb[e] = r < m_out ? nbr[r * 9 + (e % 9)] : -1;
  }
  __syncthreads();
  const int piece = tid & 7, lrow = tid >> 3;          // 16-byte piece of the 128-byte slice row; rows lrow + 32 j
  int brow[4];                                         // LDS row of weight row lrow + 32 j (16 consecutive columns per lane)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = lrow + 32 * j, s = n & 63;
    brow[j] = (n & 64) + 16 * (2 * (s >> 5) + ((s >> 2) & 1)) + 4 * ((s >> 3) & 3) + (s & 3);
  }
  u32x4 ra[4], rb[4];
  int src[4];
  auto gload = [&](int step) {
    const int t = step / KC, kc = step % KC;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      src[j] = nb[(lrow + 32 * j) * 9 + t];
      // unconditional (clamped) loads, masked when they are written to LDS: a load under a predicate is waited for
      // before the next one is issued
      ra[j] = *reinterpret_cast<const u32x4*>(feat + (int64_t)(src[j] < 0 ? 0 : src[j]) * ldf + kc * IG_BK + piece * 8);
      rb[j] = *reinterpret_cast<const u32x4*>(W + (int64_t)(n0 + lrow + 32 * j) * (9 * CIN) + t * CIN + kc * IG_BK + piece * 8);
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32x4 z = {0u, 0u, 0u, 0u};
      *reinterpret_cast<u32x4*>(&As[buf][(lrow + 32 * j) * IG_PITCH + piece * 16]) = src[j] < 0 ? z : ra[j];
      *reinterpret_cast<u32x4*>(&Bs[buf][brow[j] * IG_PITCH + piece * 16]) = rb[j];
    }
  };
  const int wm = w & 1, wn = w >> 1;                   // wave tile: rows wm*64.., columns wn*64..
  f32x4 acc[4][4];                                     // [column tile nt][row tile mt]
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  gload(0);
  lstore(0);
  __syncthreads();
  for (int step = 0; step < STEPS; ++step) {
    const int buf = step & 1;
    if (step + 1 < STEPS) gload(step + 1);
    // keep the next slice's loads ABOVE the MFMA block (left alone, the scheduler sinks them into it and the wait for
    // them lands ~250 cycles later: less than one L2 round trip under load)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < IG_BK / 32; ++ks) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        af[nt] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(&Bs[buf][(wn * 64 + nt * 16 + i) * IG_PITCH + (ks * 4 + g) * 16]));
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
        bfr[mt] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(&As[buf][(wm * 64 + mt * 16 + i) * IG_PITCH + (ks * 4 + g) * 16]));
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
          acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[nt], bfr[mt], acc[nt][mt], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (step + 1 < STEPS) lstore(buf ^ 1);
    __syncthreads();
  }
  // rows 4g + r of column tile nt = output channels wn*64 + 32 (nt / 2) + 8g + 4 (nt % 2) + r; column i = token wm*64 + mt*16 + i
  // (a tile pair = 8 consecutive channels per lane: every store instruction writes 64 contiguous bytes per token)
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int64_t r = row0 + wm * 64 + mt * 16 + i;
    if (r < m_out) {
      __hip_bfloat16* p = out + r * ldo + n0 + wn * 64 + 8 * g;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        u32x4 v;
        v[0] = ig_bf16_bits(acc[2 * h][mt][0]) | (ig_bf16_bits(acc[2 * h][mt][1]) << 16);
        v[1] = ig_bf16_bits(acc[2 * h][mt][2]) | (ig_bf16_bits(acc[2 * h][mt][3]) << 16);
        v[2] = ig_bf16_bits(acc[2 * h + 1][mt][0]) | (ig_bf16_bits(acc[2 * h + 1][mt][1]) << 16);
        v[3] = ig_bf16_bits(acc[2 * h + 1][mt][2]) | (ig_bf16_bits(acc[2 * h + 1][mt][3]) << 16);
        *reinterpret_cast<u32x4*>(p + 32 * h) = v;
      }
    }
  }
}

#endif
// ------------------------------------------------------------------------------------------------
// Ring variant (the one that runs): 256-row x 128-column tile, 8 waves, the slices travel global -> LDS by LDS-DMA
// (global_load_lds_dwordx4: no staging registers, no ds_write pass) into a 3-slot ring, two slices in flight behind a
// COUNTED vmcnt, one raw s_barrier per slice (a __syncthreads() would drain the DMA queue with its vmcnt(0)).
// LDS image: plain 128-byte rows (64 channels of one slice) with the 16-byte chunks XOR-swizzled by (row & 7); an
// LDS-DMA instruction writes lane l at base + 16 l (8 rows x 8 chunks), so the swizzle is applied to the SOURCE chunk a
// lane fetches and again to the chunk a fragment read addresses (conflict-free ds_read_b128 for the MFMA operand
// layout).  The gather is the per-lane source address of the DMA; an absent neighbour fetches a row of zeros.
// ------------------------------------------------------------------------------------------------
#define IR_BM 256
__device__ __attribute__((aligned(256))) unsigned ig_zero_row[256];  // 1 KB of zeros (static storage: zero-initialised)

__device__ __forceinline__ void ig_glds16(const void* g, char* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// BN = 128: 3 ring slots of 48 KB, two slices in flight.  BN = 256 (the 256-channel stages): the gathered rows are
// fetched ONCE for all 256 output channels -- the per-CU gather rate (30-70 GB/s, MI355X_MICROARCH.md "Indexed rows"),
// not the MFMAs, bounds this kernel -- with 2 slots of 64 KB and one slice in flight behind 64 MFMAs per wave.
template <int CIN, int BN>
__global__ __launch_bounds__(512, 1) void spconv_igemm_ring_kernel(const __hip_bfloat16* __restrict__ feat, int64_t ldf,
                                                                  const int32_t* __restrict__ nbr, int64_t m_out,
                                                                  const __hip_bfloat16* __restrict__ W, int cout,
                                                                  __hip_bfloat16* __restrict__ out, int64_t ldo) {
  constexpr int KC = CIN / IG_BK;
  constexpr int STEPS = 9 * KC;
  constexpr int NSLOT = BN == 128 ? 3 : 2;
  constexpr int AHEAD = NSLOT - 1;                      // slices in flight ahead of the one being contracted
  constexpr int STAGE = IR_BM * 128 + BN * 128;         // bytes per ring slot
  constexpr int NT = BN / 32;                           // 16-column tiles per wave (wave tile 64 rows x BN/2 columns)
  constexpr int BI = BN / 64;                           // weight-row DMA instructions per wave and slice
  constexpr int NDMA = 4 + BI;                          // DMA instructions per wave and slice
  extern __shared__ __attribute__((aligned(1024))) char ring[];        // NSLOT slots, then the rulebook tile
  int* nb = reinterpret_cast<int*>(ring + NSLOT * STAGE);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, i = lane & 15;
  const int nct = cout / BN;
  const int64_t bid = blockIdx.x;
  // ids of one residue mod 8 share an XCD (and its L2): give every XCD a CONTIGUOUS range of row tiles -- a tile's
  // neighbour rows (y +- 1: two tiles away in a 468-wide dense grid) then come from the same L2 -- and keep the column
  // tiles of one row tile on adjacent ids of that XCD
  const int64_t per_xcd = (int64_t)gridDim.x / (8 * nct);
  const int64_t rt = (bid & 7) * per_xcd + (bid >> 3) / nct;
  const int ct = (int)((bid >> 3) % nct);
  const int64_t row0 = rt * IR_BM;
  if (row0 >= m_out) return;
  const int n0 = ct * BN;
  for (int e = tid; e < IR_BM * 9; e += 512) {
    const int64_t r = row0 + e / 9;
    nb[e] = r < m_out ? nbr[r * 9 + (e % 9)] : -1;
  }
  __syncthreads();
  // ---- what this lane fetches in every slice: 4 gathered-row pieces (A) and BI weight-row pieces (B)
  const int r8 = lane >> 3, chunk = (lane & 7) ^ r8;                   // row of the 8-row group, source chunk (swizzled)
  const char* wsrc[BI];
#pragma unroll
  for (int j = 0; j < BI; ++j) {
    const int L = w * (8 * BI) + j * 8 + r8, s64 = L & 63;             // LDS row L holds weight column n (16 consecutive
    const int n = (L & ~63) + 32 * (s64 >> 5) + 8 * ((s64 >> 2) & 3) + 4 * ((s64 >> 4) & 1) + (s64 & 3);  // output channels per lane, see above)
    wsrc[j] = reinterpret_cast<const char*>(W + (int64_t)(n0 + n) * (9 * CIN)) + chunk * 16;
  }
  const char* fbase = reinterpret_cast<const char*>(feat) + chunk * 16;
  const char* zrow = reinterpret_cast<const char*>(ig_zero_row) + chunk * 16;
  const int arow = w * 32 + r8;                                        // + 8 j
  int src[4];
  auto fetch_src = [&](int step) {
    const int t = step / KC;
#pragma unroll
    for (int j = 0; j < 4; ++j) src[j] = nb[(arow + 8 * j) * 9 + t];     // four LDS reads, one wait
  };
  auto issue = [&](int step) {
    const int t = step / KC, kc = step % KC, slot = step % NSLOT;
    char* sa = ring + slot * STAGE;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // branch-free: both addresses are formed, a select picks (a branch around the 64-bit multiply costs an exec-mask
      // round trip per piece)
      const uintptr_t pa = (uintptr_t)fbase + (uintptr_t)(((int64_t)(src[j] < 0 ? 0 : src[j]) * ldf + kc * IG_BK) * 2);
      const uintptr_t p = src[j] < 0 ? (uintptr_t)zrow : pa;
      ig_glds16(reinterpret_cast<const void*>(p), sa + (w * 32 + 8 * j) * 128);
    }
#pragma unroll
    for (int j = 0; j < BI; ++j)
      ig_glds16(wsrc[j] + (t * CIN + kc * IG_BK) * 2, sa + IR_BM * 128 + (w * (8 * BI) + 8 * j) * 128);
  };
  const int wm = w & 3, wn = w >> 2;                   // wave tile: rows wm*64.., columns wn*(BN/2)..
  f32x4 acc[NT][4];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int p = 0; p < AHEAD; ++p) {
    fetch_src(p);
    issue(p);
  }
  for (int step = 0; step < STEPS; ++step) {
    // slice `step` has landed once every wave's NDMA transfers of it are done; with two slices ahead (BN = 128) the
    // NDMA transfers of slice step+1 may stay in flight
    if (AHEAD == 2 && step + 1 < STEPS) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const bool more = step + AHEAD < STEPS;
    if (more) fetch_src(step + AHEAD);                 // rulebook entries of the slice issued below: their LDS latency
    const char* sa = ring + (step % NSLOT) * STAGE;    // hides behind the first half of this slice's MFMAs
    const char* sb = sa + IR_BM * 128;
#pragma unroll
    for (int ks = 0; ks < IG_BK / 32; ++ks) {
      if (ks == 1) {
        __builtin_amdgcn_sched_barrier(0);
        if (more) issue(step + AHEAD);                 // into the slot every wave finished reading before this barrier
        __builtin_amdgcn_sched_barrier(0);
      }
      bf16x8 bfr[4];
      const int c = ks * 4 + g;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const int R = wm * 64 + mt * 16 + i;
        bfr[mt] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(sa + R * 128 + ((c ^ (R & 7)) << 4)));
      }
#pragma unroll
      for (int nh = 0; nh < NT / 4; ++nh) {
        bf16x8 af[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int L = wn * (BN / 2) + (nh * 4 + q) * 16 + i;
          af[q] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(sb + L * 128 + ((c ^ (L & 7)) << 4)));
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
            acc[nh * 4 + q][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[q], bfr[mt], acc[nh * 4 + q][mt], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of the slot are done before it moves on
  }
  // rows 4g + r of column tile nt = output channels (slab nt/4) * 64 + 32 ((nt%4) / 2) + 8g + 4 (nt % 2) + r; column i = token
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int64_t r = row0 + wm * 64 + mt * 16 + i;
    if (r < m_out) {
#pragma unroll
      for (int sbk = 0; sbk < NT / 4; ++sbk) {
        __hip_bfloat16* p = out + r * ldo + n0 + wn * (BN / 2) + sbk * 64 + 8 * g;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int t0 = sbk * 4 + 2 * h;
          u32x4 v;
          v[0] = ig_bf16_bits(acc[t0][mt][0]) | (ig_bf16_bits(acc[t0][mt][1]) << 16);
          v[1] = ig_bf16_bits(acc[t0][mt][2]) | (ig_bf16_bits(acc[t0][mt][3]) << 16);
          v[2] = ig_bf16_bits(acc[t0 + 1][mt][0]) | (ig_bf16_bits(acc[t0 + 1][mt][1]) << 16);
          v[3] = ig_bf16_bits(acc[t0 + 1][mt][2]) | (ig_bf16_bits(acc[t0 + 1][mt][3]) << 16);
          *reinterpret_cast<u32x4*>(p + 32 * h) = v;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The 256-column form (the d = 256 stages: cout = 256), second generation.  Same tile (256 rows x 256 columns, 8 waves,
// wave tile 64 x 128), same LDS image and swizzle as spconv_igemm_ring_kernel<CIN, 256>, two slots of 64 KB; what changed
// is everything around the MFMAs (the first generation spent 1.1 VALU instructions per MFMA on addresses and waited for
// every operand read right before its use; MFMA busy 0.44):
//  * transfers are `buffer_load_dwordx4 ... lds` with a 32-bit per-lane offset and a SCALAR slice offset: the rulebook tile in
//    LDS holds byte offsets (row * pitch; an absent neighbour = an offset past the buffer, which the range check turns
//    into zeros), a gathered piece costs one v_add, a weight piece nothing;
//  * the step loop is unrolled over the two slots, so every LDS operand address is a per-lane base + an immediate;
//  * operand reads are inline asm, issued one 16-MFMA group ahead into alternating register sets, with counted waits;
//  * one barrier per step, placed before the step's LAST MFMA group: a wave drains its reads of the slot, waits for the next
//    slice (issued a whole step earlier), meets the others, issues the slice after next into the slot just drained and the
//    first reads of the next step, then runs the last group.
// ------------------------------------------------------------------------------------------------
#define IR2_OOB 0x7FFFFFF0u
template <int CIN>
__global__ __launch_bounds__(512, 1) void spconv_igemm_ring256_kernel(const __hip_bfloat16* __restrict__ feat, int64_t ldf,
                                                                     unsigned fbytes, const int32_t* __restrict__ nbr,
                                                                     int64_t m_out, const __hip_bfloat16* __restrict__ W,
                                                                     int cout, __hip_bfloat16* __restrict__ out, int64_t ldo) {
  constexpr int KC = CIN / IG_BK, STEPS = 9 * KC, BN = 256;
  // LDS: gathered rows of slot 0, of slot 1, weight rows of slot 0, of slot 1 (32 KB each), then the rulebook tile: the two
  // slots of an operand lie 32 KB apart, within the 16-bit offset field of a DS instruction
  constexpr int HALF = IR_BM * 128, BOFF = 2 * HALF;
  static_assert(STEPS % 2 == 0, "the step loop is unrolled over the two slots");
  extern __shared__ __attribute__((aligned(1024))) char ring[];
  unsigned* nb = reinterpret_cast<unsigned*>(ring + 4 * HALF);           // [9][256] byte offsets of the neighbour rows
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, i = lane & 15;
  const int nct = cout / BN;
  const int64_t bid = blockIdx.x;
  const int64_t per_xcd = (int64_t)gridDim.x / (8 * nct);               // XCD-contiguous row tiles (see the kernel above)
  const int64_t rt = (bid & 7) * per_xcd + (bid >> 3) / nct;
  const int ct = (int)((bid >> 3) % nct);
  const int64_t row0 = rt * IR_BM;
  if (row0 >= m_out) return;
  const int n0 = ct * BN;
  const unsigned pitch = (unsigned)ldf * 2u;
  for (int e = tid; e < IR_BM * 9; e += 512) {
    const int64_t r = row0 + e / 9;
    const int v = r < m_out ? nbr[r * 9 + (e % 9)] : -1;
    nb[(e % 9) * IR_BM + e / 9] = v < 0 ? IR2_OOB : (unsigned)v * pitch;
  }
  __syncthreads();
  const __amdgpu_buffer_rsrc_t fr = __builtin_amdgcn_make_buffer_rsrc((void*)feat, 0, (int)fbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)(W + (int64_t)n0 * (9 * CIN)), 0,
                                                                      BN * 9 * CIN * 2, 0x00020000);
  const int r8 = lane >> 3, chunk16 = ((lane & 7) ^ r8) << 4;
  // weight row of this lane's piece j: LDS row L = 32 w + 8 j + r8 holds column n(L) (the permutation of the kernel above);
  // n(32 w + 8 j + r8) = n(32 w + r8) + {0, 16, 4, 20}[j], so one per-lane offset serves the four pieces
  unsigned woff;
  {
    const int L = w * 32 + r8, s64 = L & 63;
    const int n = (L & ~63) + 32 * (s64 >> 5) + 8 * ((s64 >> 2) & 3) + 4 * ((s64 >> 4) & 1) + (s64 & 3);
    woff = (unsigned)n * (9 * CIN * 2) + chunk16;
  }
  const unsigned* nbl = nb + w * 32 + r8;                                // gathered row of piece j: 32 w + r8 + 8 j
  unsigned goff[4];
  auto fetch_goff = [&](int t) {
#pragma unroll
    for (int j = 0; j < 4; ++j) goff[j] = nbl[t * IR_BM + 8 * j] + chunk16;
  };
  auto issue = [&](int slot, int t, int kc) {                            // uses goff (of tap t)
    char* sa = ring + slot * HALF + (w * 32) * 128;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(fr, (__attribute__((address_space(3))) void*)(sa + 8 * j * 128), 16, goff[j],
                                               kc * 128, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (__attribute__((address_space(3))) void*)(sa + BOFF + 8 * j * 128), 16,
                                               woff, (t * CIN + kc * IG_BK) * 2 + (16 * (j & 1) + 4 * (j >> 1)) * (9 * CIN * 2),
                                               0, 0);
  };
  const int wm = w & 3, wn = w >> 2;                   // wave tile: rows wm*64.., columns wn*128..
  unsigned vA[2], vB[2];                               // LDS byte addresses of the operand reads (slot 0) per channel half
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int sw = ((ks * 4 + g) ^ (i & 7)) << 4;
    vB[ks] = (unsigned)((wm * 64 + i) * 128 + sw);
    vA[ks] = (unsigned)(BOFF + (wn * 128 + i) * 128 + sw);
  }
  f32x4 acc[8][4];
#pragma unroll
  for (int nt = 0; nt < 8; ++nt)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 af[2][4], bfr[2][4];
  // 4 reads of 16-byte fragments, rows 16 apart (2048 bytes), from `addr` + OFF into registers R[0..3]
#define IR2_READ4(R, addr, OFF)                                                                                        \
  asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\t"                                    \
               "ds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %4 offset:%8"                                        \
               : "=&v"(R[0]), "=&v"(R[1]), "=&v"(R[2]), "=&v"(R[3])                                                    \
               : "v"(addr), "n"(OFF), "n"((OFF) + 2048), "n"((OFF) + 4096), "n"((OFF) + 6144) : "memory")
#define IR2_WAIT(N, RA, RB)                                                                                            \
  asm volatile("s_waitcnt lgkmcnt(" #N ")"                                                                             \
               : "+v"(RA[0]), "+v"(RA[1]), "+v"(RA[2]), "+v"(RA[3]), "+v"(RB[0]), "+v"(RB[1]), "+v"(RB[2]), "+v"(RB[3])  \
               ::"memory")
#define IR2_MFMAS(NH, RA, RB)                                                                                          \
  do {                                                                                                                 \
    __builtin_amdgcn_s_setprio(1);                                                                                     \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_)                                                                   \
      _Pragma("unroll") for (int mt_ = 0; mt_ < 4; ++mt_)                                                              \
        acc[(NH) * 4 + q_][mt_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, RA[q_]),          \
                                                                          __builtin_bit_cast(bf16x8, RB[mt_]),         \
                                                                          acc[(NH) * 4 + q_][mt_], 0, 0, 0);           \
    __builtin_amdgcn_s_setprio(0);                                                                                     \
  } while (0)
  // one step on slot SL; on entry the reads of its first group (bfr[0], af[0]) are in flight.  (tq, kq) = the slice that the
  // transfer issued in this step fetches (step + 2).
#define IR2_STEP(SL)                                                                                                   \
  do {                                                                                                                 \
    IR2_READ4(af[1], vA[0], (SL) * HALF + 8192);                                  /* columns 64..127 of the wave, channels 0..31 */  \
    IR2_WAIT(4, af[0], bfr[0]);                                                                                        \
    IR2_MFMAS(0, af[0], bfr[0]);                                                                                       \
    IR2_READ4(bfr[1], vB[1], (SL) * HALF);                                                                            \
    IR2_READ4(af[0], vA[1], (SL) * HALF);                                                                             \
    IR2_WAIT(8, af[1], bfr[0]);                                                                                        \
    IR2_MFMAS(1, af[1], bfr[0]);                                                                                       \
    IR2_READ4(af[1], vA[1], (SL) * HALF + 8192);                                                                      \
    IR2_WAIT(4, af[0], bfr[1]);                                                                                        \
    IR2_MFMAS(0, af[0], bfr[1]);                                                                                       \
    IR2_WAIT(0, af[1], bfr[1]);                                         /* all reads of this slot are done */              \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    /* the next slice (issued a step ago) landed */   \
    __builtin_amdgcn_s_barrier();                                                                                      \
    asm volatile("" ::: "memory");                                                                                     \
    if (tq < 9) {                                                                                                      \
      issue(SL, tq, kq);                                                                                               \
      if (++kq == KC) { kq = 0; if (++tq < 9) fetch_goff(tq); }                                                       \
    }                                                                                                                  \
    IR2_READ4(bfr[0], vB[0], (1 - (SL)) * HALF);                                                                      \
    IR2_READ4(af[0], vA[0], (1 - (SL)) * HALF);                                                                       \
    IR2_MFMAS(1, af[1], bfr[1]);                                                                                       \
  } while (0)
  int tq = 0, kq = 0;
  fetch_goff(0);
  issue(0, 0, 0);
  if (++kq == KC) { kq = 0; ++tq; fetch_goff(tq); }
  issue(1, tq, kq);
  if (++kq == KC) { kq = 0; ++tq; fetch_goff(tq); }
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                       // slice 0 (slice 1's 8 transfers may be in flight)
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  IR2_READ4(bfr[0], vB[0], 0);
  IR2_READ4(af[0], vA[0], 0);
  for (int step = 0; step < STEPS; step += 2) {
    IR2_STEP(0);
    IR2_STEP(1);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#undef IR2_STEP
#undef IR2_MFMAS
#undef IR2_WAIT
#undef IR2_READ4
  // rows 4g + r of column tile nt = output channels (slab nt/4) * 64 + 32 ((nt%4) / 2) + 8g + 4 (nt % 2) + r; column i = token
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int64_t r = row0 + wm * 64 + mt * 16 + i;
    if (r < m_out) {
#pragma unroll
      for (int sbk = 0; sbk < 2; ++sbk) {
        __hip_bfloat16* p = out + r * ldo + n0 + wn * 128 + sbk * 64 + 8 * g;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int t0 = sbk * 4 + 2 * h;
          u32x4 v;
          v[0] = ig_bf16_bits(acc[t0][mt][0]) | (ig_bf16_bits(acc[t0][mt][1]) << 16);
          v[1] = ig_bf16_bits(acc[t0][mt][2]) | (ig_bf16_bits(acc[t0][mt][3]) << 16);
          v[2] = ig_bf16_bits(acc[t0 + 1][mt][0]) | (ig_bf16_bits(acc[t0 + 1][mt][1]) << 16);
          v[3] = ig_bf16_bits(acc[t0 + 1][mt][2]) | (ig_bf16_bits(acc[t0 + 1][mt][3]) << 16);
          *reinterpret_cast<u32x4*>(p + 32 * h) = v;
        }
      }
    }
  }
}

// out [m_out, cout] bf16 = sparse conv of feat [m_in, cin] bf16 through the rulebook nbr [m_out, 9] with the weight
// matrix W [cout, 9 * cin] bf16 (the spconv-2 layout [cout, 3, 3, cin] flattened).  cin in {128, 256, 384}, cout % 128 == 0.
static int igemm_launch(const void* feat, int64_t ldf, int64_t m_in, int cin, const int32_t* nbr, int64_t m_out,
                        const void* w, int cout, void* out, int64_t ldo, hipStream_t stream) {
  if (m_in < 0 || m_out < 0 || (cin != 128 && cin != 256 && cin != 384) || cout <= 0 || (cout % IG_BN) || ldf < cin || ldo < cout ||
      (ldf % 8) || (ldo % 8))
    return TMAE_EARG;
  if (m_out == 0) return TMAE_OK;
  if (!feat || !nbr || !w || !out || m_in == 0) return TMAE_EARG;
  if (((uintptr_t)feat & 15) || ((uintptr_t)w & 15) || ((uintptr_t)out & 15)) return TMAE_EARG;
  const int nct = cout / IG_BN;
  static const int impl = TMAE_AB_INT("TMAE_IGEMM", 3);   // 2 (debug build only): the retired register-staged 128x128 kernel
  if (impl == 3) {
    const int64_t rts = (m_out + IR_BM - 1) / IR_BM;
    static const int wide_off = TMAE_AB_INT("TMAE_IGEMM_BN256", 1) == 0;
    const int bn = (cout % 256 == 0 && !wide_off) ? 256 : 128;
    const int nctr = cout / bn;
    const int64_t grid = ((rts + 7) / 8) * 8 * nctr;     // = 8 XCD residues x ceil(rts / 8) row tiles x column tiles
    const int lds = (bn == 128 ? 3 : 2) * (IR_BM * 128 + bn * 128) + IR_BM * 9 * 4;
#define IR_LAUNCH(C, B)                                                                                               \
  do {                                                                                                                \
    static TmaeLdsAttr attr;                                                                                          \
    if (int e_ = tmae_allow_lds(attr, (const void*)spconv_igemm_ring_kernel<C, B>, lds)) return e_;                   \
    hipLaunchKernelGGL((spconv_igemm_ring_kernel<C, B>), dim3((unsigned)grid), dim3(512), lds, stream,                \
                       (const __hip_bfloat16*)feat, ldf, nbr, m_out, (const __hip_bfloat16*)w, cout,                  \
                       (__hip_bfloat16*)out, ldo);                                                                    \
  } while (0)
    const int64_t fbytes = ((m_in - 1) * ldf + cin) * 2;
    static const int gen2 = TMAE_AB_INT("TMAE_IGEMM_RING2", 1);
    if (bn == 256 && gen2 != 0 && fbytes < (int64_t)IR2_OOB && ldf * 2 < (1 << 24) && m_in < (1 << 24)) {
      // (Round 6 tried to run a last, mostly empty round of row tiles -- 86 k rows = 337 tiles = 1.3 rounds of 256 workgroups -- on the
      //  128-column kernel in a second launch: that kernel gathers the same rows for half the columns and took 0.7 of a full round plus
      //  its own launch; ring256 2.66 -> 2.34 ms, tail launches +0.29 ms: nothing.  profiles/round6_ab_spconv_tail.txt.)
      const int64_t m_main = m_out, grid_main = grid;
#define IR2_LAUNCH(C)                                                                                                 \
  do {                                                                                                                \
    static TmaeLdsAttr attr;                                                                                          \
    if (int e_ = tmae_allow_lds(attr, (const void*)spconv_igemm_ring256_kernel<C>, lds)) return e_;                   \
    hipLaunchKernelGGL((spconv_igemm_ring256_kernel<C>), dim3((unsigned)grid_main), dim3(512), lds, stream,           \
                       (const __hip_bfloat16*)feat, ldf, (unsigned)fbytes, nbr, m_main, (const __hip_bfloat16*)w, cout, \
                       (__hip_bfloat16*)out, ldo);                                                                    \
  } while (0)
      if (cin == 128) IR2_LAUNCH(128); else if (cin == 256) IR2_LAUNCH(256); else IR2_LAUNCH(384);
#undef IR2_LAUNCH
    } else if (bn == 256) {
      if (cin == 128) IR_LAUNCH(128, 256); else if (cin == 256) IR_LAUNCH(256, 256); else IR_LAUNCH(384, 256);
    } else {
      if (cin == 128) IR_LAUNCH(128, 128); else if (cin == 256) IR_LAUNCH(256, 128); else IR_LAUNCH(384, 128);
    }
#undef IR_LAUNCH
    return tmae_launch_status();
  }
#ifdef TMAE_AB
  const int64_t rts = (m_out + IG_BM - 1) / IG_BM;
  const int64_t grid = ((rts + 7) / 8) * 8 * nct;       // row tiles padded to a multiple of 8 (the XCD-aware id map)
  if (cin == 128)
    hipLaunchKernelGGL((spconv_igemm_kernel<128>), dim3((unsigned)grid), dim3(256), 0, stream, (const __hip_bfloat16*)feat,
                       ldf, nbr, m_out, (const __hip_bfloat16*)w, cout, (__hip_bfloat16*)out, ldo);
  else if (cin == 256)
    hipLaunchKernelGGL((spconv_igemm_kernel<256>), dim3((unsigned)grid), dim3(256), 0, stream, (const __hip_bfloat16*)feat,
                       ldf, nbr, m_out, (const __hip_bfloat16*)w, cout, (__hip_bfloat16*)out, ldo);
  else
    hipLaunchKernelGGL((spconv_igemm_kernel<384>), dim3((unsigned)grid), dim3(256), 0, stream, (const __hip_bfloat16*)feat,
                       ldf, nbr, m_out, (const __hip_bfloat16*)w, cout, (__hip_bfloat16*)out, ldo);
  return tmae_launch_status();
#else
  (void)nct;
  return TMAE_EARG;
#endif
}

int tmae_spconv_fwd(const void* feat, int64_t ldf, int64_t m_in, int cin, const int32_t* nbr, int64_t m_out,
                    const void* weight, int cout, void* out, int64_t ldo, void* stream_) {
  (void)hipGetLastError();
  return igemm_launch(feat, ldf, m_in, cin, nbr, m_out, weight, cout, out, ldo, (hipStream_t)stream_);
}

// din [m_in, cin] = sum_t dout[nbr_t[i, t], :] . W[:, t, :]: the forward kernel on the transposed rulebook with
// weight_t [cin, 9 * cout], weight_t[c, t * cout + n] = W[n, t * cin + c] (the caller keeps that copy).
int tmae_spconv_bwd_data(const void* dout, int64_t lddo, int64_t m_out, int cout, const int32_t* nbr_t, int64_t m_in,
                         const void* weight_t, int cin, void* din, int64_t lddi, void* stream_) {
  (void)hipGetLastError();
  return igemm_launch(dout, lddo, m_out, cout, nbr_t, m_in, weight_t, cin, din, lddi, (hipStream_t)stream_);
}

// ------------------------------------------------------------------------------------------------
// Dense 3x3 convolution (padding 1) on a channels-last grid [B, Y, X, CIN] -- the decoder's conv (SiamWCA_MAE.py:100-115)
// and its input gradient -- as a HALO-tiled implicit GEMM.  The ring kernel above re-gathers every input row once per tap
// and is bound by the per-CU gather rate (~40 GB/s); here a workgroup owns a 16 x 16 block of cells and stages, per
// 64-channel slice, the 18 x 18 halo of input rows ONCE (41 KB) -- the 9 taps read shifted rows of that LDS image -- so
// only the weight slices (16 KB per tap, L2-resident) stream per step.  Same MFMA tiling, swizzle, LDS-DMA / counted
// vmcnt / raw barrier scheme as the ring kernel; every wave issues exactly 3 transfers per tap step (2 weight pieces +
// 1 halo piece of the NEXT channel slice, a dummy piece once the halo is complete) so that the counted wait is uniform.
//   out[b, y, x, n] = sum_{ky,kx,c} in[b, y+ky-1, x+kx-1, c] * W[n, (ky*3+kx)*CIN + c]
// ------------------------------------------------------------------------------------------------
// DIL = dilation (1: 18 x 18 halo = 41 pieces of 8 rows, 41 KB per image; 2 -- the third conv of SSTBEVBackbone,
// sst_bev_backbone.py:20-30 with t_mae.yaml:107-112 -- 20 x 20 = 50 pieces, 50 KB: 2 images + 3 weight slots = 149 KB).
// sum over the 16 lanes of a DPP row (lanes 16 k .. 16 k + 15), result in every lane
__device__ __forceinline__ float ig_row16_sum(float v) {
#define IG_ROR(n) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + (n), 0xF, 0xF, false))
  v += IG_ROR(8);
  v += IG_ROR(4);
  v += IG_ROR(2);
  v += IG_ROR(1);
#undef IG_ROR
  return v;
}

// Reduce-scatter of 16 per-lane values over the 16 lanes of a DPP row: returns, in lane i, the sum over the row's lanes of a[i].
// Steps s = 1, 2, 4, 8: a lane keeps the values whose index agrees with its own lane id in bit s and adds what lane i + s (row
// rotation: same low bits, bit s flipped) holds for them -- every (lane, value) term is counted once; 15 selects-and-adds.
__device__ __forceinline__ float ig_row16_scatter_sum(const float* a, int i) {
  float v8[8], v4[4], v2[2];
#define IG_RS(dst, lo, hi, S)                                                                                   \
  {                                                                                                             \
    const bool up_ = (i & (S)) != 0;                                                                            \
    const float keep_ = up_ ? (hi) : (lo), send_ = up_ ? (lo) : (hi);                                           \
    dst = keep_ + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send_), 0x120 + (S), 0xF, 0xF, false)); \
  }
  // (row_ror:S reads lane (i + S) mod 16 of the row.)  value index j = 8 b3 + 4 b2 + 2 b1 + b0; step 1 pairs (j, j ^ 1) ...
#pragma unroll
  for (int q = 0; q < 8; ++q) IG_RS(v8[q], a[2 * q], a[2 * q + 1], 1)        // v8[q]: index bits 3..1 = q, bit 0 = lane's
#pragma unroll
  for (int q = 0; q < 4; ++q) IG_RS(v4[q], v8[2 * q], v8[2 * q + 1], 2)
#pragma unroll
  for (int q = 0; q < 2; ++q) IG_RS(v2[q], v4[2 * q], v4[2 * q + 1], 4)
  float r;
  IG_RS(r, v2[0], v2[1], 8)
#undef IG_RS
  return r;
}

// out[c] = sum over the `rows` (block, wave) rows of part [rows][ncol], in a fixed order: 16 columns x 64 row lanes per workgroup,
// four loads in flight per thread, double accumulation (the 2 048 x 384 partials of a launch: ~5 us)
__global__ __launch_bounds__(1024) void dense_colsum_finish_kernel(const float* __restrict__ part, int rows, int ncol,
                                                                  float* __restrict__ out) {
  __shared__ double red[64][17];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  double acc = 0.0;
  if (c < ncol) {
    int r = rl;
    for (; r + 3 * 64 < rows; r += 4 * 64) {
      float v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = part[(int64_t)(r + q * 64) * ncol + c];
#pragma unroll
      for (int q = 0; q < 4; ++q) acc += (double)v[q];
    }
    for (; r < rows; r += 64) acc += (double)part[(int64_t)r * ncol + c];
  }
  red[rl][cl] = acc;
  __syncthreads();
  if (rl == 0 && c < ncol) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < 64; ++q) t += red[q][cl];
    out[c] = (float)t;
  }
}

template <int DIL> struct HaloGeom {
  static constexpr int HW = 16 + 2 * DIL, NH = HW * HW, NP = (NH + 7) / 8, ABYTES = NP * 8 * 128, PQ = (NP + 7) / 8;
};

// CS (round 6): the kernel also leaves, per wave, the COLUMN SUMS of what it stored (the bf16-rounded outputs; CS = 2: and of their
// squares) in cs_part [blocks][8 waves][CS][cout] -- the decoder conv's input gradient is column-summed by the BatchNorm backward
// of the three deconvolutions in front of it (a 1.35 GB pass over [cells, 384]: tmae_column_sums), and its forward output by the
// statistics pass of the norm behind it (448 MB).  The sums ride in an LDS row per wave (every wave owns its row: no atomics, fixed
// order); dense_colsum_finish_kernel adds the rows up in a fixed order.
template <int CIN, int DIL = 1, int CS = 0>
__global__ __launch_bounds__(512, 1) void dense_conv3x3_halo_kernel(const __hip_bfloat16* __restrict__ in, int B, int Y,
                                                                   int X, const __hip_bfloat16* __restrict__ W, int cout,
                                                                   __hip_bfloat16* __restrict__ out, int nunits,
                                                                   const __hip_bfloat16* __restrict__ post,
                                                                   float* __restrict__ cs_part) {
  // post (optional, the shape of out): out = conv + post -- the gradient that reaches a residual block's input through its
  // shortcut, added where the input gradient of its conv is in registers (sst_bev_backbone.py:35-41 backwards)
  constexpr int KC = CIN / 64;
  constexpr int HW = HaloGeom<DIL>::HW, NH = HaloGeom<DIL>::NH, NP = HaloGeom<DIL>::NP, HC_ABYTES = HaloGeom<DIL>::ABYTES;
  static_assert(NP <= 64, "the piece issued in the last tap step of a slice must be a dummy (it may still be in flight)");
  static_assert(KC * 128 + 128 <= (int)sizeof(ig_zero_row), "zero-row sources advance with the channel slice");
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  constexpr int BRING = 2 * HC_ABYTES;                 // 2 halo images (channel slices kc, kc+1), then 3 weight slots of 16 KB
  constexpr int SCRATCH = BRING + 3 * (IG_BN * 128);   // 1 KB sink of the dummy transfers
  constexpr int CSOFF = SCRATCH + 1024;                // CS: [8 waves][CS][cout] floats
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, i = lane & 15;
  // ALLCT (cin = 128: the two halo images hold the WHOLE input of the tile): one unit of work = a cell block with ALL its column
  // tiles -- the halo is fetched once instead of once per column tile (the decoder conv's input gradient, 128 -> 384: 1.50 ->
  // 1.26 ms); otherwise a unit = (cell block, column tile).
  // PERSISTENT: a workgroup walks over the units u = blockIdx.x, + gridDim.x, ... (one workgroup per CU); the next unit's first
  // halo slice and first weight slices are fetched during the last channel slice of the current one, exactly like the next
  // channel slice inside a unit, so a unit boundary costs its output stores and nothing else (one workgroup per unit paid the
  // HBM latency of its prologue and the workgroup turnaround once per 18-54 tap steps).
  constexpr bool ALLCT = KC == 2;
  const int nct = cout / IG_BN;
  const int bx = (X + 15) / 16, by = (Y + 15) / 16;
  const int r8 = lane >> 3, slot8 = lane & 7;
  int ct = 0, b = 0, y0 = 0, x0 = 0;                   // the unit whose transfers are being issued (decode()), then computed
  auto decode = [&](int u) {
    if (!ALLCT) { ct = u % nct; u /= nct; }
    const int tx = u % bx, t2 = u / bx;
    b = t2 / by; y0 = (t2 - b * by) * 16; x0 = tx * 16;
  };
  // All addresses of the main loop are set up per unit (the loop itself then spends ~10 VALU instructions per 32 MFMAs; computed
  // in the loop they were 77 -- 3.6 VALU per MFMA, as much SIMD time as the MFMAs themselves: profiles/round4_pmc_kernels.md).
  // Halo image: row h = hy * HW + hx of the (16 + 2 DIL)^2 halo, 128 bytes; the 16-byte chunk c of a row sits at chunk
  // position c ^ (hx & 7) -- a swizzle by the COLUMN only, so that a tap shift (ky, kx) of a reader is an immediate offset
  // plus one of three per-lane bases.  Piece p = t * 8 + w (tap step t, wave w) = halo rows 8 p .. 8 p + 7.
  const uintptr_t zlane = (uintptr_t)ig_zero_row + slot8 * 16;
  uintptr_t hsrc[9];
  auto halo_sources = [&]() {                          // of slice 0 of the unit in (b, y0, x0)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int p = t * 8 + w, h = 8 * p + r8;
      const int hy = h / HW, hx = h - hy * HW;
      const int y = y0 - DIL + hy, x = x0 - DIL + hx;
      const bool ok = p < NP && h < NH && y >= 0 && y < Y && x >= 0 && x < X;
      hsrc[t] = ok ? (uintptr_t)in + ((uintptr_t)(((int64_t)b * Y + y) * X + x) * CIN) * 2 + ((slot8 ^ (hx & 7)) << 4) : zlane;
    }
  };
  uintptr_t wrow[2], wk[2];                            // this lane's weight rows: column tile 0 / the current channel slice
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int L = w * 16 + j * 8 + r8, s64 = L & 63;
    const int n = (L & ~63) + 32 * (s64 >> 5) + 8 * ((s64 >> 2) & 3) + 4 * ((s64 >> 4) & 1) + (s64 & 3);
    wrow[j] = (uintptr_t)(W + (int64_t)n * (9 * CIN)) + ((slot8 ^ r8) << 4);
  }
  int unit = blockIdx.x;
  if (unit >= nunits) return;
  float* colacc = reinterpret_cast<float*>(lds + CSOFF) + (CS ? w * CS * cout : 0);      // this wave's row(s)
  if constexpr (CS > 0) {
    for (int c = lane; c < CS * cout; c += 64) colacc[c] = 0.f;
  }
  decode(unit);
  halo_sources();
#pragma unroll
  for (int j = 0; j < 2; ++j) wk[j] = wrow[j] + (uintptr_t)ct * (IG_BN * 9 * CIN * 2);
  const int wm = w & 3, wn = w >> 2;
  int vB[3][2], vA[2];                                 // LDS byte offsets of this lane's operand reads (image 0, slot 0)
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    vA[ks] = BRING + (wn * 64 + i) * 128 + (((ks * 4 + g) ^ (i & 7)) << 4);
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
      vB[kx][ks] = (4 * wm * HW + i + kx * DIL) * 128 + (((ks * 4 + g) ^ ((i + kx * DIL) & 7)) << 4);
  }
  // every tap step issues exactly 3 transfers per wave (weight piece j = 1 of step + 2, one halo piece of the next channel
  // slice, weight piece j = 0 of step + 3; dummies into the scratch KB where there is nothing to fetch), so one counted wait
  // serves all steps.  They are spread over the MFMAs of the step (see HC_STEP): issued back to back they stall the wave
  // for several hundred cycles in which its SIMD partner, at the same point behind the same barrier, does the same.
  const int wdst = (w * 16) * 128;
  auto issue_halo = [&](int t, int img, bool real) {
    const int p = t * 8 + w;
    ig_glds16(reinterpret_cast<const void*>(real ? hsrc[t] : zlane), lds + ((p < NP && real) ? img + p * 1024 : SCRATCH));
  };
  uintptr_t wnx[2] = {0, 0};                           // the slice that follows the current one (next slice / column tile / unit)
  auto issue_w = [&](int j, int t, bool next, bool real) {    // piece j of tap t of the current or the following slice
    ig_glds16(reinterpret_cast<const void*>((next ? wnx[j] : wk[j]) + (real ? t * CIN * 2 : 0)),
              lds + (real ? BRING + (t % 3) * (IG_BN * 128) + wdst + 8 * j * 128 : SCRATCH));
  };
#pragma unroll
  for (int q = 0; q < HaloGeom<DIL>::PQ; ++q) issue_halo(q, 0, true);
#pragma unroll
  for (int t = 0; t < 9; ++t) hsrc[t] += 128;
  issue_w(0, 0, false, true);
  issue_w(1, 0, false, true);
  issue_w(0, 1, false, true);
  issue_w(1, 1, false, true);
  ig_glds16(reinterpret_cast<const void*>(zlane), lds + SCRATCH);
  issue_w(0, 2, false, true);
  f32x4 acc[4][4];
  // Operand registers are double-buffered: the reads of a half step (32 channels of a tap) are issued one half step ahead
  // and land under the 16 MFMAs of the current one.  One barrier per tap step, in its MIDDLE: after the first half's MFMAs
  // a wave drains its LDS reads (issued a half step ago: done), issues the step's 3 transfers, waits until everything older
  // than those has landed (= the next step's weight slice, issued a step ago) and meets the others; the next step's first
  // half is read right behind the barrier.  Write-after-read: slot (s+2)%3 and the other halo image were last read by
  // reads that every wave drained before the PREVIOUS barrier.
  // The reads are inline asm with hand-counted waits: left to the compiler, every MFMA group waited for lgkmcnt(0), i.e. also
  // for the reads issued just before it for the NEXT group.
  u32x4 af[2][4], bfr[2][4];
  // operand reads of tap T (a constant expression), channel half ks, halo image at byte offset `img` -> register buffer buf
#define HC_LOAD_OPS(buf, T, ks, img)                                                                                   \
  do {                                                                                                                 \
    constexpr int ky_ = (T) / 3, kx_ = (T) - ky_ * 3, so_ = ((T) % 3) * (IG_BN * 128), ro_ = ky_ * DIL * HW * 128;       \
    const unsigned aa_ = (unsigned)vA[ks], ab_ = (unsigned)(vB[kx_][ks] + (img));                                      \
    asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\t"                                 \
                 "ds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %4 offset:%8"                                      \
                 : "=&v"(af[buf][0]), "=&v"(af[buf][1]), "=&v"(af[buf][2]), "=&v"(af[buf][3])                           \
                 : "v"(aa_), "n"(so_), "n"(so_ + 2048), "n"(so_ + 4096), "n"(so_ + 6144) : "memory");                   \
    asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\t"                                 \
                 "ds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %4 offset:%8"                                      \
                 : "=&v"(bfr[buf][0]), "=&v"(bfr[buf][1]), "=&v"(bfr[buf][2]), "=&v"(bfr[buf][3])                       \
                 : "v"(ab_), "n"(ro_), "n"(ro_ + HW * 128), "n"(ro_ + 2 * HW * 128), "n"(ro_ + 3 * HW * 128) : "memory"); \
  } while (0)
  // wait until at most N LDS reads are outstanding; the operand registers of buffer `buf` pass through the statement so
  // that no MFMA on them can be scheduled above it
#define HC_WAIT_OPS(buf, N)                                                                                            \
  asm volatile("s_waitcnt lgkmcnt(" #N ")"                                                                             \
               : "+v"(af[buf][0]), "+v"(af[buf][1]), "+v"(af[buf][2]), "+v"(af[buf][3]), "+v"(bfr[buf][0]),            \
                 "+v"(bfr[buf][1]), "+v"(bfr[buf][2]), "+v"(bfr[buf][3])::"memory")
  // 16 MFMAs of operand buffer `buf`; `mid` runs after the 5th and `late` after the 11th (a transfer each)
#define HC_MFMAS(buf, mid, late)                                                                                       \
  do {                                                                                                                 \
    __builtin_amdgcn_s_setprio(1);                                                                                     \
    _Pragma("unroll") for (int e_ = 0; e_ < 16; ++e_) {                                                                \
      const int nt_ = e_ >> 2, mt_ = e_ & 3;                                                                           \
      acc[nt_][mt_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[buf][nt_]),                \
                                                              __builtin_bit_cast(bf16x8, bfr[buf][mt_]), acc[nt_][mt_], 0, 0, 0); \
      if (e_ == 4) { __builtin_amdgcn_sched_barrier(0); mid; __builtin_amdgcn_sched_barrier(0); }                     \
      if (e_ == 10) { __builtin_amdgcn_sched_barrier(0); late; __builtin_amdgcn_sched_barrier(0); }                   \
    }                                                                                                                  \
    __builtin_amdgcn_s_setprio(0);                                                                                     \
  } while (0)
  // slot (s+2)%3 [(s+3)%3] was last read by reads that every wave drained before the barrier of step s-1 [s]; the other
  // halo image in the previous slice
#define HC_STEP(T)                                                                                                     \
  do {                                                                                                                 \
    HC_LOAD_OPS(1, T, 1, img);                                                                                         \
    HC_WAIT_OPS(0, 8);                                                                                                 \
    HC_MFMAS(0, issue_w(1, ((T) + 2) % 9, (T) >= 7, (T) < 7 || more_w),                                                \
             (issue_halo(T, img ^ HC_ABYTES, more_h), hsrc[T] += 128));                                                \
    HC_WAIT_OPS(1, 0);                                                                                                 \
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                                                   \
    __builtin_amdgcn_s_barrier();                                                                                      \
    asm volatile("" ::: "memory");                                                                                     \
    HC_LOAD_OPS(0, ((T) + 1) % 9, 0, (T) < 8 ? img : imgn);                                                            \
    HC_MFMAS(1, (void)0, issue_w(0, ((T) + 3) % 9, (T) >= 6, (T) < 6 || more_w));                                      \
  } while (0)
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");       // the halo of slice 0 and weight slice 0
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  HC_LOAD_OPS(0, 0, 0, 0);
  const int nloop = ALLCT ? nct : 1;
  for (; unit < nunits; unit += gridDim.x) {
    const int ub = b, uy0 = y0, ux0 = x0, uct = ct;        // this unit (decode() moves on to the next one below)
    const bool more_u = unit + (int)gridDim.x < nunits;
    for (int cti = 0; cti < nloop; ++cti) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
      const bool more_ct = cti + 1 < nloop;
      for (int kc = 0; kc < KC; ++kc) {
        const bool more_s = kc + 1 < KC;
        const bool last = !more_s && !more_ct;               // the unit's last slice: the slot of "the next slice" is the next unit's
        if (last && more_u) {                                // first one (every piece of this unit's halo has been issued by now)
          decode(unit + (int)gridDim.x);
          halo_sources();
        }
        // the halo is streamed once per unit (ALLCT: later column tiles find it in LDS); weights of the taps that wrap around
        const bool more_h = (more_s && cti == 0) || (last && more_u);
        const bool more_w = more_s || more_ct || more_u;
#pragma unroll
        for (int j = 0; j < 2; ++j)
          wnx[j] = more_s ? wk[j] + 128 : (more_ct ? wk[j] + (IG_BN * 9 * CIN * 2 - (KC - 1) * 128)
                                                   : wrow[j] + (uintptr_t)ct * (IG_BN * 9 * CIN * 2));
        const int img = (kc & 1) * HC_ABYTES, imgn = (((kc + 1) % KC) & 1) * HC_ABYTES;
        HC_STEP(0); HC_STEP(1); HC_STEP(2); HC_STEP(3); HC_STEP(4); HC_STEP(5); HC_STEP(6); HC_STEP(7); HC_STEP(8);
        wk[0] = wnx[0];
        wk[1] = wnx[1];
      }
      const int n0 = (ALLCT ? cti : uct) * IG_BN;
      float cs1[4][4], cs2[CS == 2 ? 4 : 1][4];
      if constexpr (CS > 0) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) { cs1[nt][r] = 0.f; if constexpr (CS == 2) cs2[nt][r] = 0.f; }
      }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const int y = uy0 + 4 * wm + mt, x = ux0 + i;
        if (y < Y && x < X) {
          const int64_t eo = (((int64_t)ub * Y + y) * X + x) * cout + n0 + wn * 64 + 8 * g;
          __hip_bfloat16* p = out + eo;
          if (post) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const u32x4 pv = *reinterpret_cast<const u32x4*>(post + eo + 32 * h);
#pragma unroll
              for (int q2 = 0; q2 < 2; ++q2) {
                acc[2 * h + q2][mt][0] += __uint_as_float(pv[2 * q2] << 16);
                acc[2 * h + q2][mt][1] += __uint_as_float(pv[2 * q2] & 0xFFFF0000u);
                acc[2 * h + q2][mt][2] += __uint_as_float(pv[2 * q2 + 1] << 16);
                acc[2 * h + q2][mt][3] += __uint_as_float(pv[2 * q2 + 1] & 0xFFFF0000u);
              }
            }
          }
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            u32x4 v;
            v[0] = ig_bf16_bits(acc[2 * h][mt][0]) | (ig_bf16_bits(acc[2 * h][mt][1]) << 16);
            v[1] = ig_bf16_bits(acc[2 * h][mt][2]) | (ig_bf16_bits(acc[2 * h][mt][3]) << 16);
            v[2] = ig_bf16_bits(acc[2 * h + 1][mt][0]) | (ig_bf16_bits(acc[2 * h + 1][mt][1]) << 16);
            v[3] = ig_bf16_bits(acc[2 * h + 1][mt][2]) | (ig_bf16_bits(acc[2 * h + 1][mt][3]) << 16);
            *reinterpret_cast<u32x4*>(p + 32 * h) = v;
            if constexpr (CS > 0) {                              // sums of the STORED (bf16-rounded) values
#pragma unroll
              for (int q2 = 0; q2 < 2; ++q2) {
                const float f0 = __uint_as_float(v[2 * q2] << 16), f1 = __uint_as_float(v[2 * q2] & 0xFFFF0000u);
                const float f2 = __uint_as_float(v[2 * q2 + 1] << 16), f3 = __uint_as_float(v[2 * q2 + 1] & 0xFFFF0000u);
                cs1[2 * h + q2][0] += f0; cs1[2 * h + q2][1] += f1; cs1[2 * h + q2][2] += f2; cs1[2 * h + q2][3] += f3;
                if constexpr (CS == 2) {
                  cs2[2 * h + q2][0] += f0 * f0; cs2[2 * h + q2][1] += f1 * f1; cs2[2 * h + q2][2] += f2 * f2; cs2[2 * h + q2][3] += f3 * f3;
                }
              }
            }
          }
        }
      }
      if constexpr (CS > 0) {
        // over the 16 cells (lanes i) of the wave's rows: a reduce-scatter -- lane i ends up with the total of value i = (nt, r) =
        // (i >> 2, i & 3) -- in 15 DPP adds instead of 16 full reductions (64), then ONE add per lane into the wave's own LDS row
        const int chl = n0 + wn * 64 + 32 * (i >> 3) + 8 * g + 4 * ((i >> 2) & 1) + (i & 3);
        colacc[chl] += ig_row16_scatter_sum(&cs1[0][0], i);
        if constexpr (CS == 2) colacc[cout + chl] += ig_row16_scatter_sum(&cs2[0][0], i);
      }
    }
  }
  if constexpr (CS > 0) {
    // (LDS operations of a wave execute in order: the row is complete)
    for (int c = lane; c < CS * cout; c += 64) cs_part[((int64_t)blockIdx.x * 8 + w) * (CS * cout) + c] = colacc[c];
  }
#undef HC_MFMAS
#undef HC_STEP
#undef HC_WAIT_OPS
#undef HC_LOAD_OPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the dummy transfers of the last steps
}

// out [B, Y, X, cout] = conv3x3(in [B, Y, X, cin], padding = dilation) with weight [cout, 9 * cin] (taps ky-major, then
// kx, then channel: the [cout, 3, 3, cin] layout flattened), bf16, fp32 accumulation.  cin in {128, 256, 384},
// cout % 128 == 0, dilation in {1, 2}.  The input gradient of such a conv is the same call on dout with
// weight_t[c, (2-ky)*3 + (2-kx), n] = weight[n, ky, kx, c].
static int dense_conv3x3_launch(const void* in, int batch, int ny, int nx, int cin, const void* weight, int cout, int dil,
                                const void* post, void* out, hipStream_t stream, int cs = 0, float* sums = nullptr,
                                void* wsp = nullptr, size_t ws_bytes = 0) {
  if (batch <= 0 || ny <= 0 || nx <= 0 || (cin != 128 && cin != 256 && cin != 384) || cout <= 0 || (cout % IG_BN) ||
      (dil != 1 && dil != 2))
    return TMAE_EARG;
  if (!in || !weight || !out || ((uintptr_t)in & 15) || ((uintptr_t)weight & 15) || ((uintptr_t)out & 15)) return TMAE_EARG;
  // units of work: cell blocks (cin = 128: with all their column tiles, see ALLCT in the kernel) or (cell block, column tile);
  // one persistent workgroup per CU walks over them
  const int64_t units = (int64_t)batch * ((ny + 15) / 16) * ((nx + 15) / 16) * (cin == 128 ? 1 : cout / IG_BN);
  if (units >= ((int64_t)1 << 31)) return TMAE_EARG;
  const int ncu = tmae_num_cus();
  const int64_t blocks = units < ncu ? units : ncu;
  float* part = nullptr;
  if (cs) {                       // the two shapes of the decoder conv (SiamWCA_MAE.py:100-115): forward 384 -> 128, input gradient 128 -> 384
    if (dil != 1 || !sums || !((cs == 2 && cin == 384 && cout == 128) || (cs == 1 && cin == 128 && cout == 384))) return TMAE_EARG;
    WsCarver ws(wsp, ws_bytes);
    part = ws.take<float>((size_t)blocks * 8 * cs * cout);
    if (!ws.ok) return TMAE_EWS;
  }
#define HC_LAUNCH(C, D, S)                                                                                            \
  do {                                                                                                                \
    const int lds = 2 * HaloGeom<D>::ABYTES + 3 * (IG_BN * 128) + 1024 + 8 * (S) * cout * 4;                          \
    static TmaeLdsAttr attr;                                                                                          \
    if (int e_ = tmae_allow_lds(attr, (const void*)dense_conv3x3_halo_kernel<C, D, S>, lds)) return e_;               \
    hipLaunchKernelGGL((dense_conv3x3_halo_kernel<C, D, S>), dim3((unsigned)blocks), dim3(512), lds, stream,         \
                       (const __hip_bfloat16*)in, batch, ny, nx, (const __hip_bfloat16*)weight, cout,                 \
                       (__hip_bfloat16*)out, (int)units, (const __hip_bfloat16*)post, part);                          \
  } while (0)
  if (cs == 2) HC_LAUNCH(384, 1, 2);
  else if (cs == 1) HC_LAUNCH(128, 1, 1);
  else if (dil == 1) {
    if (cin == 128) HC_LAUNCH(128, 1, 0); else if (cin == 256) HC_LAUNCH(256, 1, 0); else HC_LAUNCH(384, 1, 0);
  } else {
    if (cin == 128) HC_LAUNCH(128, 2, 0); else if (cin == 256) HC_LAUNCH(256, 2, 0); else HC_LAUNCH(384, 2, 0);
  }
#undef HC_LAUNCH
  if (cs)
    hipLaunchKernelGGL(dense_colsum_finish_kernel, dim3(tmae_cdiv(cs * cout, 16)), dim3(1024), 0, stream, part, (int)blocks * 8,
                       cs * cout, sums);
  return tmae_launch_status();
}

size_t tmae_dense_conv3x3_sums_workspace(int cout) { return tmae_align((size_t)tmae_num_cus() * 8 * 2 * (cout > 0 ? cout : 0) * 4) + 256; }

// tmae_dense_conv3x3 / _add that also returns column sums of the stored output, taken in the conv's epilogue:
//   moments = 1: sums [cout]       = sum over all cells of out[., n]             (cin = 128, cout = 384: the decoder conv's input gradient)
//   moments = 2: sums [2][cout]    = ... and of out[., n]^2                      (cin = 384, cout = 128: its forward)
// post may be NULL.  ws: tmae_dense_conv3x3_sums_workspace(cout) bytes.  Other shapes: TMAE_EARG (run the conv, then tmae_column_sums).
int tmae_dense_conv3x3_sums(const void* in, int batch, int ny, int nx, int cin, const void* weight, int cout, const void* post,
                            int moments, void* out, float* sums, void* ws, size_t ws_bytes, void* stream_) {
  (void)hipGetLastError();
  if ((moments != 1 && moments != 2) || !sums || (post && ((uintptr_t)post & 15))) return TMAE_EARG;
  return dense_conv3x3_launch(in, batch, ny, nx, cin, weight, cout, 1, post, out, (hipStream_t)stream_, moments, sums, ws, ws_bytes);
}

int tmae_dense_conv3x3(const void* in, int batch, int ny, int nx, int cin, const void* weight, int cout, void* out,
                       void* stream_) {
  (void)hipGetLastError();
  return dense_conv3x3_launch(in, batch, ny, nx, cin, weight, cout, 1, nullptr, out, (hipStream_t)stream_);
}

int tmae_dense_conv3x3_dilated(const void* in, int batch, int ny, int nx, int cin, const void* weight, int cout,
                               int dilation, void* out, void* stream_) {
  (void)hipGetLastError();
  return dense_conv3x3_launch(in, batch, ny, nx, cin, weight, cout, dilation, nullptr, out, (hipStream_t)stream_);
}

int tmae_dense_conv3x3_add(const void* in, int batch, int ny, int nx, int cin, const void* weight, int cout, int dilation,
                           const void* post, void* out, void* stream_) {
  (void)hipGetLastError();
  if (!post || ((uintptr_t)post & 15)) return TMAE_EARG;
  return dense_conv3x3_launch(in, batch, ny, nx, cin, weight, cout, dilation, post, out, (hipStream_t)stream_);
}
