// Token-list GEMM  y[m, N] = x[m, K] . W[N, KA]^T + bias  (bf16, fp32 accumulation) -- the "W in registers" kernel.
// Same operation and operand conventions as csrc/token_gemm.hip (the Linear layers of sst_basic_block.py:45-83 and their
// input gradients); this file is the variant for the heavy shapes, where HBM bytes decide:
//
//  * The register files of a CU hold 512 KB, its LDS 160 KB.  token_gemm_res_kernel keeps W in LDS (one 256-column
//    slice per workgroup), so N = 512 / 768 run as 2 / 3 column groups that each read x again (PMC traffic 1.2x / 1.5x
//    the algorithmic bytes).  Here the workgroup's 8 waves split the OUTPUT COLUMNS: wave w keeps the MFMA A-fragments
//    of its 16 * NTC columns x KA in registers for the whole kernel (K = 256, 64 columns: 128 VGPRs), so one workgroup
//    covers N = 512 (or 2 teams of 4 waves cover N = 256 on alternate token groups) and x is read from HBM ONCE.
//  * x does not pass through registers at all: 64-token steps travel global -> LDS by `buffer_load_dwordx4 ... lds`
//    (LDS-DMA) into a ring of 4 slots, three steps (96 KB per CU) in flight behind a COUNTED s_waitcnt, one raw
//    s_barrier per step; out-of-range rows are zero-filled by the buffer range check (no clamps, no branches).
//    Every wave reads every x fragment of the step from LDS (ds_read_b128, conflict-free through an XOR swizzle that
//    is applied on the SOURCE side of the DMA: LDS position c' of row R holds global 16-byte chunk c' ^ (R & 15)).
//  * products "swapped" (rows = output columns, column = token) over W rows permuted at load time so that a lane ends
//    with 4 * NTC CONSECUTIVE columns of one token: 16-byte stores, four lanes per 128-byte line (token_gemm.hip).
//  * vmcnt bookkeeping: DMA pieces and stores retire in order.  Step q issues DMA(q+3) right after its barrier and its
//    stores at its end, so when step q+1 waits for DMA(q+1) the younger operations are stores(q-2), DMA(q+2),
//    stores(q-1), DMA(q+3), stores(q): vmcnt(3 NST + 2 ND).  The prologue issues dropped dummy stores so that the count
//    is uniform from step 0, and the tail issues out-of-range DMAs (zeros, no traffic) for the same reason.
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned tgw_pack2(float lo, float hi) {       // one v_cvt_pk_bf16_f32 (round to nearest even)
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}

// one-hot B fragment of the 32 position columns (see pos_onehot in token_gemm.hip): k-group g = 0/2 -> x cell, 1/3 -> y cell
__device__ __forceinline__ bf16x8 tgw_pos_onehot(unsigned cell, int g) {
  const unsigned sel = (g & 1) ? (cell >> 3) & 7u : cell & 7u;
  const unsigned v = (sel & 1u) ? 0x3F800000u : 0x00003F80u, wi = sel >> 1;
  const u32x4 u = {wi == 0u ? v : 0u, wi == 1u ? v : 0u, wi == 2u ? v : 0u, wi == 3u ? v : 0u};
  return __builtin_bit_cast(bf16x8, u);
}

// exact (erf) GELU of a bf16-rounded pre-activation, the forward twin of token_gemm.hip's dgelu(): erf through
// Abramowitz-Stegun 7.1.26 (|error| < 1.5e-7, far below the bf16 rounding of the result), one exp, one rcp, a few FMAs
__device__ __forceinline__ float tgw_gelu(float x) {
  const float e = __expf(-0.5f * x * x);
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * 0.70710678f * fabsf(x));
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erf_abs = 1.0f - poly * e;
  return x * (0.5f * (1.0f + copysignf(erf_abs, x)));
}
// the two bf16 halves of a packed dword through GELU, packed again
__device__ __forceinline__ unsigned tgw_gelu2(unsigned u) {
  return tgw_pack2(tgw_gelu(__uint_as_float(u << 16)), tgw_gelu(__uint_as_float(u & 0xFFFF0000u)));
}

// d/dx of the exact GELU, Phi(x) + x phi(x): the same arithmetic as token_gemm.hip's dgelu()
__device__ __forceinline__ float tgw_dgelu(float x) {
  const float e = __expf(-0.5f * x * x);
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * 0.70710678f * fabsf(x));
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erf_abs = 1.0f - poly * e;
  return 0.5f * (1.0f + copysignf(erf_abs, x)) + x * e * 0.39894228040143267794f;
}

// tokens per ring slot: 32 KB of x (DG: 16 KB); contraction 384 / 768 (the attention in-projections' input gradients,
// W = 96 / 192 registers per lane): 24 KB
__host__ __device__ constexpr int tgw_step(int k, bool dg) { return k == 768 ? 16 : (k == 384 ? 32 : (dg ? 8192 : 16384) / k); }
// ACC at contraction 128 takes the half step too: its y tile (128 columns) is as large as the x tile
__host__ __device__ constexpr bool tgw_half(int k, bool acc, bool dg) { return dg || (acc && k == 128); }

#define TGW_OOB 0x7FFFFFF0u          // voffset past every buffer: loads return zeros, stores are dropped
#define TGW_NT 2                     // cache policy of the y stores (written once, streamed)

// GELU2 (the FFN's first Linear, sst_basic_block.py:81): the kernel ALSO writes y2 = gelu(y) (same shape and pitch) from
// the bf16-rounded tile it is about to store -- a second trip through the staging patch and a second set of full-line
// stores.  The separate GELU pass (read [m, dff], write [m, dff]: 84 us per layer at 466 k tokens, as fast as a copy)
// becomes one extra write here; this kernel's step has ~3x the VALU slack the ~13 instructions per element need.
// DG (the FFN's first Linear backwards, d h_pre = (dy W2) * gelu'(h_pre), sst_basic_block.py:81-82): y2 is the READ-ONLY
// [m, N] tile source h_pre (same pitch as y); its step tile rides in the ring slot like ACC's y tile and the epilogue multiplies
// by gelu'() of it.  Steps of half the tokens (16 KB of x + 32 KB of h_pre per slot, 3 slots).
template <int K, int NTC, int NWC, bool POS, bool ACC = false, bool GELU2 = false, bool DG = false>
__global__ __launch_bounds__(512, 2) void token_gemm_wreg_kernel(const __hip_bfloat16* __restrict__ x, int ldx,
                                                                const __hip_bfloat16* __restrict__ W,
                                                                const __hip_bfloat16* __restrict__ bias,
                                                                __hip_bfloat16* __restrict__ y, int ldy, int m, int ncg,
                                                                unsigned xbytes, unsigned ybytes,
                                                                const uint8_t* __restrict__ cells,
                                                                __hip_bfloat16* __restrict__ y2) {
  static_assert(!(DG && (ACC || GELU2 || POS)), "DG is a mode of its own");
  constexpr int TEAMS = 8 / NWC;                  // teams of NWC waves; a team covers all columns of the group
  constexpr int STEP = tgw_step(K, tgw_half(K, ACC, DG));   // tokens per ring slot (32 KB of x; DG 16 KB): 128 / 64 / 32 for K = 128 / 256 / 512
  constexpr int TGW = (STEP / 16) / TEAMS;        // 16-token groups per wave and step
  constexpr int KX = K / 32, KA = K + (POS ? 32 : 0), KS = KA / 32;
  constexpr int ROWB = K * 2;                     // bytes per x row
  constexpr int PPW = STEP * ROWB / 1024 / 8;     // 1-KiB DMA pieces per wave and step
  constexpr int CELLB = POS ? STEP * 4 : 0;       // one dword per token of the step (its window cell byte)
  // ACC (y += x W^T, the in-place input gradient of a Linear whose input has a second consumer): the step's tile of y rides
  // in the ring slot beside x (a register load of y in the epilogue would have to be waited for IN ORDER, i.e. together
  // with the three steps of x prefetched before it) and the accumulators start from it
  constexpr int YROWB = NWC * NTC * 32;           // bytes per row of the y tile (the group's columns)
  constexpr int YB = (ACC || DG) ? STEP * YROWB : 0, PPY = YB / 1024 / 8;
  constexpr int SLOT = STEP * ROWB + CELLB + YB;
  constexpr int NS = (ACC || DG) ? 3 : 4;
  constexpr int ND = PPW + (POS ? STEP / 64 : 0) + PPY; // DMA instructions per wave and step
  static_assert(!(ACC || DG) || YB % 8192 == 0, "y tile must divide into 1-KiB pieces per wave");
  constexpr int NST = TGW * (NTC / 2) * (GELU2 ? 2 : 1);   // 16-byte store instructions per wave and step
  constexpr int WAITN = (NS - 1) * NST + (NS - 2) * ND;
  static_assert(TGW >= 1 && (NTC == 2 || NTC == 4) && WAITN < 64 && ROWB <= 1536 && (STEP * ROWB) % 8192 == 0, "shape");
  extern __shared__ __attribute__((aligned(1024))) char ring[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, i = lane & 15;
  const int team = w / NWC, wc = w % NWC;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int cg = j % ncg, tg = (j / ncg) * 8 + xcd, ntg = gridDim.x / ncg;
  const int nsteps = (m + STEP - 1) / STEP;
  const int mine = tg < nsteps ? (nsteps - tg + ntg - 1) / ntg : 0;     // steps of this workgroup: tg, tg + ntg, ...
  const int colbase = cg * (NWC * NTC * 16) + wc * (NTC * 16);

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, (int)ybytes, 0x00020000);
  // ACC with y2 != NULL: the accumulators start from y2 (same pitch as y) instead of y itself -- y = y2 + x W^T + b, the residual add
  // of the encoder FFN riding on its second Linear (sst_basic_block.py:81-83) out of place
  const __amdgpu_buffer_rsrc_t y2r = __builtin_amdgcn_make_buffer_rsrc((void*)((GELU2 || DG || (ACC && y2 != nullptr)) ? y2 : y), 0,
                                                                       (int)ybytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t cr = __builtin_amdgcn_make_buffer_rsrc((void*)(POS ? cells : (const uint8_t*)x), 0, POS ? m : 0, 0x00020000);

  // ---- what this lane fetches in every step: PPW pieces; piece p = w * PPW + jj covers LDS bytes [1024 p, +1024) of the slot
  int drow[PPW];
  unsigned doff[PPW];
#pragma unroll
  for (int jj = 0; jj < PPW; ++jj) {
    const int lb = (w * PPW + jj) * 1024 + lane * 16;
    const int R = lb / ROWB, cpos = (lb % ROWB) / 16;
    drow[jj] = R;
    doff[jj] = (unsigned)(R * ldx * 2 + ((cpos ^ (R & 15)) << 4));
  }
  auto issue = [&](int q) {                        // local step q -> slot q % NS (q may lie past the end: zeros)
    const int tok0 = q < mine ? (tg + q * ntg) * STEP : m;     // past the end: every row fails the range test below
    char* slot = ring + (q % NS) * SLOT;
    const unsigned tb = (unsigned)tok0 * (unsigned)(ldx * 2);
#pragma unroll
    for (int jj = 0; jj < PPW; ++jj) {
      const unsigned vo = tok0 + drow[jj] < m ? tb + doff[jj] : TGW_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (__attribute__((address_space(3))) void*)(slot + (w * PPW + jj) * 1024),
                                               16, vo, 0, 0, 0);
    }
    if constexpr (ACC || DG) {
#pragma unroll
      for (int jj = 0; jj < PPY; ++jj) {
        const int lb = (w * PPY + jj) * 1024 + lane * 16;
        const int R = lb / YROWB, cpos = (lb % YROWB) / 16;
        const unsigned vo = tok0 + R < m ? (unsigned)(tok0 + R) * (unsigned)(ldy * 2) +
                                               (unsigned)(cg * YROWB + ((cpos ^ (R & 15)) << 4)) : TGW_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(y2r, (__attribute__((address_space(3))) void*)(slot + STEP * ROWB + CELLB +
                                                                                                (w * PPY + jj) * 1024),
                                                 16, vo, 0, 0, 0);
      }
    }
    if constexpr (POS) {                           // every wave fetches the step's cell bytes (same values: benign)
#pragma unroll
      for (int c = 0; c < STEP / 64; ++c) {
        const unsigned vo = tok0 + c * 64 + lane < m ? (unsigned)(tok0 + c * 64 + lane) : TGW_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(cr, (__attribute__((address_space(3))) void*)(slot + STEP * ROWB + c * 256), 1,
                                                 vo, 0, 0, 0);
      }
    }
  };
  auto dummy_stores = [&]() {
#pragma unroll
    for (int q = 0; q < NST; ++q)
      __builtin_amdgcn_raw_buffer_store_b128(u32x4{0u, 0u, 0u, 0u}, yr, (int)TGW_OOB, 0, 0);
  };

  // ---- prologue: this wave's W fragments (tile ct, row i of the tile = column 4 NTC (i >> 2) + 4 ct + (i & 3)) and bias
  u32x4 wf[NTC][KS];
#pragma unroll
  for (int ct = 0; ct < NTC; ++ct) {
    const int col = colbase + 4 * NTC * (i >> 2) + 4 * ct + (i & 3);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      wf[ct][ks] = *reinterpret_cast<const u32x4*>(W + (int64_t)col * KA + ks * 32 + g * 8);
  }
  f32x4 bl[NTC];                                  // bias of rows 4g .. 4g+3 of every tile: the accumulators start from it
#pragma unroll
  for (int ct = 0; ct < NTC; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) bl[ct][r] = __bfloat162float(bias[colbase + 4 * NTC * g + 4 * ct + r]);
#pragma unroll
  for (int q = 0; q < NS - 1; ++q) {
    issue(q);
    dummy_stores();
  }
  // LDS read addresses of this lane's B fragments: row i of a token group, chunk (4 ks + g) ^ i
  int ardr[4];
#pragma unroll
  for (int kl = 0; kl < 4; ++kl) ardr[kl] = i * ROWB + ((g ^ (i & 3)) << 4) + (((kl ^ (i >> 2)) & 3) << 6);

  // epilogue staging: CPL 16-byte chunks per line of this wave's 16 NTC columns, RPI token rows per store instruction
  constexpr int CPL = NTC * 2, RPI = 64 / CPL;
  char* stg = ring + NS * SLOT + w * (16 * NTC * 32);
  const int srow = lane / CPL, schunk = lane % CPL;
  const unsigned soff = (unsigned)((srow * ldy + colbase) * 2 + schunk * 16);
  // PT token groups (16 tokens each x this wave's 16 NTC columns) at a time: M = their MFMAs, E = their epilogue.
  // (The non-temporal policy on the x DMA loads: 149 against 127-131 us on one box, not used.  PT = 2 -- two groups share every W fragment register read -- and a half-step stagger of waves 4-7 against their
  // SIMD partners were both measured: within the run-to-run noise of this HBM-bound kernel, and PT = 2 spills at KA = 288.)
  constexpr int PT = 1, NPR = TGW / PT;
  f32x4 acc[NTC][PT];
  auto mfma_tg = [&](const char* slot, int tgi) {
    u32x4 bf[2][PT];
    unsigned cellv[PT];
#pragma unroll
    for (int t = 0; t < PT; ++t) {
      bf[0][t] = *reinterpret_cast<const u32x4*>(slot + ardr[0] + (tgi + t) * 16 * ROWB);
      if constexpr (POS) cellv[t] = *reinterpret_cast<const unsigned*>(slot + STEP * ROWB + ((tgi + t) * 16 + i) * 4);
    }
    f32x4 c0[NTC][PT];                              // what the accumulators start from: the bias (+ the y tile)
#pragma unroll
    for (int t = 0; t < PT; ++t) {
#pragma unroll
      for (int ct = 0; ct < NTC; ++ct) c0[ct][t] = bl[ct];
      if constexpr (ACC) {                          // this lane's 4 NTC consecutive columns of token i: chunk index ^ i
        const char* yt = slot + STEP * ROWB + CELLB + ((tgi + t) * 16 + i) * YROWB;
#pragma unroll
        for (int h = 0; h < NTC / 2; ++h) {
          const int ch = wc * (NTC / 2) * 4 + (NTC / 2) * g + h;          // 16-byte chunk of the row
          const u32x4 yv = *reinterpret_cast<const u32x4*>(yt + ((ch ^ i) << 4));
#pragma unroll
          for (int q2 = 0; q2 < 2; ++q2) {
            c0[2 * h + q2][t][0] += __uint_as_float(yv[2 * q2] << 16);
            c0[2 * h + q2][t][1] += __uint_as_float(yv[2 * q2] & 0xFFFF0000u);
            c0[2 * h + q2][t][2] += __uint_as_float(yv[2 * q2 + 1] << 16);
            c0[2 * h + q2][t][3] += __uint_as_float(yv[2 * q2 + 1] & 0xFFFF0000u);
          }
        }
      }
    }
#pragma unroll
    for (int ks = 0; ks < KX; ++ks) {
      if (ks + 1 < KX) {
#pragma unroll
        for (int t = 0; t < PT; ++t)
          bf[(ks + 1) & 1][t] = *reinterpret_cast<const u32x4*>(slot + ardr[(ks + 1) & 3] + (tgi + t) * 16 * ROWB + ((ks + 1) >> 2) * 256);
      }
#pragma unroll
      for (int ct = 0; ct < NTC; ++ct)
#pragma unroll
        for (int t = 0; t < PT; ++t)
          acc[ct][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[ct][ks]),
                                                               __builtin_bit_cast(bf16x8, bf[ks & 1][t]),
                                                               ks == 0 ? c0[ct][t] : acc[ct][t], 0, 0, 0);
    }
    if constexpr (POS) {
#pragma unroll
      for (int t = 0; t < PT; ++t) {
        const bf16x8 oh = tgw_pos_onehot(cellv[t], g);
#pragma unroll
        for (int ct = 0; ct < NTC; ++ct)
          acc[ct][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[ct][KX]), oh, acc[ct][t], 0, 0, 0);
      }
    }
  };
  // rows 4g + r of tile ct = columns colbase + 4 NTC g + 4 ct + r, column i = token.  Stored straight from this layout a
  // wave instruction writes 64 separate 16-byte pieces (half of every 128-byte line, with holes): the L2 request rate,
  // not bytes, then bounds the kernel (stores alone: 314 us for 477 MB).  So the tile takes a turn through a wave-private
  // LDS patch (2 KB, 16-byte chunks XOR-swizzled by the row: conflict-free both ways) and leaves as FULL 128-byte
  // lines: lane l stores chunk l & 7 of token l >> 3, 8 whole lines per instruction (stores alone: 109 us).
  auto epi_tg = [&](int tokb, const char* slot, int tgi) {   // tokb = first token of the PT groups (tgi: their index in the step)
#pragma unroll
    for (int t = 0; t < PT; ++t) {
      u32x4 pk[NTC / 2];
#pragma unroll
      for (int h = 0; h < NTC / 2; ++h) {
        f32x4 v0 = acc[2 * h][t], v1 = acc[2 * h + 1][t];
        if constexpr (DG) {                             // this lane's 8 columns of token i of h_pre (layout: see ACC above)
          const char* yt = slot + STEP * ROWB + CELLB + ((tgi + t) * 16 + i) * YROWB;
          const int ch = wc * (NTC / 2) * 4 + (NTC / 2) * g + h;
          const u32x4 a = *reinterpret_cast<const u32x4*>(yt + ((ch ^ i) << 4));
          v0[0] *= tgw_dgelu(__uint_as_float(a[0] << 16)); v0[1] *= tgw_dgelu(__uint_as_float(a[0] & 0xFFFF0000u));
          v0[2] *= tgw_dgelu(__uint_as_float(a[1] << 16)); v0[3] *= tgw_dgelu(__uint_as_float(a[1] & 0xFFFF0000u));
          v1[0] *= tgw_dgelu(__uint_as_float(a[2] << 16)); v1[1] *= tgw_dgelu(__uint_as_float(a[2] & 0xFFFF0000u));
          v1[2] *= tgw_dgelu(__uint_as_float(a[3] << 16)); v1[3] *= tgw_dgelu(__uint_as_float(a[3] & 0xFFFF0000u));
        }
        pk[h] = u32x4{tgw_pack2(v0[0], v0[1]), tgw_pack2(v0[2], v0[3]), tgw_pack2(v1[0], v1[1]), tgw_pack2(v1[2], v1[3])};
        *reinterpret_cast<u32x4*>(stg + i * (NTC * 32) + ((((NTC / 2) * g + h) ^ (i & (CPL - 1))) << 4)) = pk[h];
      }
#pragma unroll
      for (int hh = 0; hh < NTC / 2; ++hh) {
        const int r = hh * RPI + srow;                   // token row of the group this lane stores
        const u32x4 v = *reinterpret_cast<const u32x4*>(stg + r * (NTC * 32) + ((schunk ^ (r & (CPL - 1))) << 4));
        const unsigned vo = tokb + t * 16 + r < m ? (unsigned)(tokb + t * 16 + hh * RPI) * (unsigned)(ldy * 2) + soff : TGW_OOB;
        __builtin_amdgcn_raw_buffer_store_b128(v, yr, (int)vo, 0, TGW_NT);
      }
      if constexpr (GELU2) {                             // the same tile through GELU: LDS operations of one wave are ordered,
#pragma unroll                                           // so the patch can be rewritten right behind the reads above
        for (int h = 0; h < NTC / 2; ++h)
          *reinterpret_cast<u32x4*>(stg + i * (NTC * 32) + ((((NTC / 2) * g + h) ^ (i & (CPL - 1))) << 4)) =
              u32x4{tgw_gelu2(pk[h][0]), tgw_gelu2(pk[h][1]), tgw_gelu2(pk[h][2]), tgw_gelu2(pk[h][3])};
#pragma unroll
        for (int hh = 0; hh < NTC / 2; ++hh) {
          const int r = hh * RPI + srow;
          const u32x4 v = *reinterpret_cast<const u32x4*>(stg + r * (NTC * 32) + ((schunk ^ (r & (CPL - 1))) << 4));
          const unsigned vo = tokb + t * 16 + r < m ? (unsigned)(tokb + t * 16 + hh * RPI) * (unsigned)(ldy * 2) + soff : TGW_OOB;
          __builtin_amdgcn_raw_buffer_store_b128(v, y2r, (int)vo, 0, TGW_NT);
        }
      }
    }
  };
  for (int q = 0; q < mine; ++q) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    issue(q + NS - 1);                               // into the slot every wave finished reading before this barrier
    __builtin_amdgcn_sched_barrier(0);
    const char* slot = ring + (q % NS) * SLOT;
    const int tok0 = (tg + q * ntg) * STEP + team * TGW * 16;
#pragma unroll
    for (int k2 = 0; k2 < NPR; ++k2) {
      mfma_tg(slot, team * TGW + k2 * PT);
      epi_tg(tok0 + k2 * PT * 16, slot, team * TGW + k2 * PT);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of the slot are done before the next barrier
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // no LDS-DMA may land after the workgroup has released its LDS
}

template <int K, int NTC, int NWC, bool POS, bool ACC = false, bool GELU2 = false, bool DG = false>
static int tgw_launch(const void* x, int64_t ldx, int64_t m, const void* w, int n, const void* bias, void* y, int64_t ldy,
                      const void* cells, hipStream_t stream, void* y2 = nullptr) {
  constexpr int NG = NWC * NTC * 16;
  constexpr int STEP = tgw_step(K, tgw_half(K, ACC, DG));
  constexpr int lds = ((ACC || DG) ? 3 : 4) * (STEP * K * 2 + (POS ? STEP * 4 : 0) + ((ACC || DG) ? STEP * NG * 2 : 0)) + 8 * 16 * NTC * 32;
  const int ncg = n / NG;
  static TmaeLdsAttr attr;
  if (int e = tmae_allow_lds(attr, (const void*)token_gemm_wreg_kernel<K, NTC, NWC, POS, ACC, GELU2, DG>, lds)) return e;
  const int64_t xbytes = ((m - 1) * ldx + K) * 2, ybytes = ((m - 1) * ldy + n) * 2;
  const int grid = 8 * ncg * (32 / ncg);
  hipLaunchKernelGGL((token_gemm_wreg_kernel<K, NTC, NWC, POS, ACC, GELU2, DG>), dim3(grid), dim3(512), lds, stream,
                     (const __hip_bfloat16*)x, (int)ldx, (const __hip_bfloat16*)w, (const __hip_bfloat16*)bias,
                     (__hip_bfloat16*)y, (int)ldy, (int)m, ncg, (unsigned)xbytes, (unsigned)ybytes, (const uint8_t*)cells,
                     (__hip_bfloat16*)y2);
  return tmae_launch_status();
}

// Returns TMAE_EARG for shapes this kernel does not cover (the caller falls back to csrc/token_gemm.hip's kernels).
int tmae_token_gemm_wreg(const void* x, int64_t ldx, int64_t m, int k, const void* w, int n, const void* bias,
                         const uint8_t* cells, void* y, int64_t ldy, int accumulate, void* stream_) {
  (void)hipGetLastError();
  hipStream_t stream = (hipStream_t)stream_;
  if (m <= 0 || !x || !w || !y || !bias || ldx < k || ldy < n || (ldx % 8) || (ldy % 8)) return TMAE_EARG;
  if (((uintptr_t)x & 15) || ((uintptr_t)w & 15) || ((uintptr_t)y & 15) || ((uintptr_t)bias & 1)) return TMAE_EARG;
  if (((m - 1) * ldx + k) * 2 >= (int64_t)TGW_OOB || ((m - 1) * ldy + n) * 2 >= (int64_t)TGW_OOB) return TMAE_EARG;
  if (k == 512) {          // FFN-2 forward (sst_basic_block.py:82) and FFN-1's input gradient: 8 waves x 32 columns, W[256, 512]
    if (n != 256 || cells) return TMAE_EARG;       // = 128 registers per lane
    if (accumulate) return tgw_launch<512, 2, 8, false, true>(x, ldx, m, w, n, bias, y, ldy, nullptr, stream);
    return tgw_launch<512, 2, 8, false>(x, ldx, m, w, n, bias, y, ldy, nullptr, stream);
  }
  if (accumulate) {        // the d = 128 FFN-1 input gradient [m, 128] += [m, 256] W; the in-projections' input gradients
    if (cells) return TMAE_EARG;                                    // [m, 256] += [m, 768] W and [m, 128] += [m, 384] W
    if (k == 256 && n == 128) return tgw_launch<256, 4, 2, false, true>(x, ldx, m, w, n, bias, y, ldy, nullptr, stream);
    if (k == 768 && n == 256) return tgw_launch<768, 2, 8, false, true>(x, ldx, m, w, n, bias, y, ldy, nullptr, stream);
    if (k == 384 && n == 128) return tgw_launch<384, 2, 4, false, true>(x, ldx, m, w, n, bias, y, ldy, nullptr, stream);
    // the square ones (the cross layers' query input gradient into the alias gradient, the out-projections'): 256 -> 256 as two
    // column groups of the 256 -> 128 instance (x read twice, the second time from the cache), 128 -> 128 on half steps
    if (k == 256 && n == 256) return tgw_launch<256, 4, 2, false, true>(x, ldx, m, w, n, bias, y, ldy, nullptr, stream);
    if (k == 128 && n == 128) return tgw_launch<128, 4, 2, false, true>(x, ldx, m, w, n, bias, y, ldy, nullptr, stream);
    return TMAE_EARG;
  }
  if (n % 128 || (k != 128 && k != 256)) return TMAE_EARG;
  // column blocks of 512 (8 waves x 64 columns), then one of 256 (2 teams of 4 waves) or 128 (4 teams of 2 waves);
  // x is re-read per block (from the Infinity Cache where it fits): N = 768 = 512 + 256
  const int ka = k + (cells ? 32 : 0);
  for (int n0 = 0; n0 < n;) {
    const int nb = n - n0 >= 512 ? 512 : (n - n0 >= 256 ? 256 : 128);
    const void* wb = (const char*)w + (int64_t)n0 * ka * 2;
    const void* bb = (const char*)bias + (int64_t)n0 * 2;
    void* yb = (char*)y + (int64_t)n0 * 2;
    int rc;
#define TGW_GO(KK, NWC, P) rc = tgw_launch<KK, 4, NWC, P>(x, ldx, m, wb, nb, bb, yb, ldy, cells, stream)
    if (k == 256) {
      if (cells) { if (nb == 512) TGW_GO(256, 8, true); else if (nb == 256) TGW_GO(256, 4, true); else TGW_GO(256, 2, true); }
      else { if (nb == 512) TGW_GO(256, 8, false); else if (nb == 256) TGW_GO(256, 4, false); else TGW_GO(256, 2, false); }
    } else {
      if (cells) { if (nb == 512) TGW_GO(128, 8, true); else if (nb == 256) TGW_GO(128, 4, true); else TGW_GO(128, 2, true); }
      else { if (nb == 512) TGW_GO(128, 8, false); else if (nb == 256) TGW_GO(128, 4, false); else TGW_GO(128, 2, false); }
    }
#undef TGW_GO
    if (rc != TMAE_OK) return rc;
    n0 += nb;
  }
  return TMAE_OK;
}

// y = x W^T + bias AND y_gelu = gelu(y) in one pass (tmae_token_gemm_gelu): the FFN's first Linear, (k, n) = (256, 512) or
// (128, 256).  TMAE_EARG for anything else (the caller runs tmae_token_gemm and its own GELU pass).
int tmae_token_gemm_wreg_gelu(const void* x, int64_t ldx, int64_t m, int k, const void* w, int n, const void* bias, void* y,
                              void* y_gelu, int64_t ldy, void* stream_) {
  (void)hipGetLastError();
  hipStream_t stream = (hipStream_t)stream_;
  if (m <= 0 || !x || !w || !y || !y_gelu || !bias || ldx < k || ldy < n || (ldx % 8) || (ldy % 8)) return TMAE_EARG;
  if (((uintptr_t)x & 15) || ((uintptr_t)w & 15) || ((uintptr_t)y & 15) || ((uintptr_t)y_gelu & 15) || ((uintptr_t)bias & 1)) return TMAE_EARG;
  if (((m - 1) * ldx + k) * 2 >= (int64_t)TGW_OOB || ((m - 1) * ldy + n) * 2 >= (int64_t)TGW_OOB) return TMAE_EARG;
  if (k == 256 && n == 512) return tgw_launch<256, 4, 8, false, false, true>(x, ldx, m, w, n, bias, y, ldy, nullptr, stream, y_gelu);
  if (k == 128 && n == 256) return tgw_launch<128, 4, 4, false, false, true>(x, ldx, m, w, n, bias, y, ldy, nullptr, stream, y_gelu);
  return TMAE_EARG;
}

// y = (x W^T + bias) * gelu'(aux) in one pass (tmae_token_gemm_dgelu on the heavy shapes): the input gradient of the FFN's
// first Linear through its GELU, (k, n) = (256, 512) or (128, 256); aux [m, n] has y's pitch.  TMAE_EARG for anything else.
int tmae_token_gemm_wreg_res(const void* x, int64_t ldx, int64_t m, int k, const void* w, int n, const void* bias, const void* res,
                             void* y, int64_t ldy, void* stream_) {
  (void)hipGetLastError();
  hipStream_t stream = (hipStream_t)stream_;
  if (m <= 0 || !x || !w || !y || !res || !bias || ldx < k || ldy < n || (ldx % 8) || (ldy % 8)) return TMAE_EARG;
  if (((uintptr_t)x & 15) || ((uintptr_t)w & 15) || ((uintptr_t)y & 15) || ((uintptr_t)res & 15) || ((uintptr_t)bias & 1)) return TMAE_EARG;
  if (((m - 1) * ldx + k) * 2 >= (int64_t)TGW_OOB || ((m - 1) * ldy + n) * 2 >= (int64_t)TGW_OOB) return TMAE_EARG;
  if (k == 512 && n == 256)
    return tgw_launch<512, 2, 8, false, true>(x, ldx, m, w, n, bias, y, ldy, nullptr, stream, const_cast<void*>(res));
  if (k == 256 && n == 128)
    return tgw_launch<256, 4, 2, false, true>(x, ldx, m, w, n, bias, y, ldy, nullptr, stream, const_cast<void*>(res));
  return TMAE_EARG;
}

int tmae_token_gemm_wreg_dgelu(const void* x, int64_t ldx, int64_t m, int k, const void* w, int n, const void* bias,
                               const void* aux, void* y, int64_t ldy, void* stream_) {
  (void)hipGetLastError();
  hipStream_t stream = (hipStream_t)stream_;
  if (m <= 0 || !x || !w || !y || !aux || !bias || ldx < k || ldy < n || (ldx % 8) || (ldy % 8)) return TMAE_EARG;
  if (((uintptr_t)x & 15) || ((uintptr_t)w & 15) || ((uintptr_t)y & 15) || ((uintptr_t)aux & 15) || ((uintptr_t)bias & 1)) return TMAE_EARG;
  if (((m - 1) * ldx + k) * 2 >= (int64_t)TGW_OOB || ((m - 1) * ldy + n) * 2 >= (int64_t)TGW_OOB) return TMAE_EARG;
  if (k == 256 && n == 512)
    return tgw_launch<256, 4, 8, false, false, false, true>(x, ldx, m, w, n, bias, y, ldy, nullptr, stream, const_cast<void*>(aux));
  if (k == 128 && n == 256)
    return tgw_launch<128, 4, 4, false, false, false, true>(x, ldx, m, w, n, bias, y, ldy, nullptr, stream, const_cast<void*>(aux));
  return TMAE_EARG;
}
