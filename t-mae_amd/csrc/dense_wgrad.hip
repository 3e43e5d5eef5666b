// Weight gradient of the dense 3x3 convolution (padding = dilation, no bias) on a channels-last grid -- the decoder's conv
// (SiamWCA_MAE.py:100-115) and the convs of SSTBEVBackbone (sst_bev_backbone.py:20-30):
//     dW[n, t, c] = sum over cells p of dY[p, n] * X[p + delta_t, c]          (zero outside the grid)
// The first implementation ran it as a sparse-conv weight gradient through the rulebook of a FULL grid (csrc/wgrad.hip,
// wgrad_kernel<true, false>): every (tap, 128-channel) block of the [128, 9 cin] output streamed dY again and gathered its own
// shifted copy of X -- 24 GB through L2 per call at cin = 384, 2.2 ms, bound by L2 bandwidth.  Here the nine shifted copies
// come out of ONE staged image, like the forward conv's halo kernel (csrc/spconv_igemm.hip):
//   * a workgroup owns all 128 output channels x one 64-channel slice of X x all 9 taps (128 x 576 fp32 accumulators = 144
//     registers per lane) and walks over half tiles of 8 x 16 cells; per half tile it stages dY (128 cells x 128 n, 32 KB)
//     and the (8 + 2 d) x (16 + 2 d) halo of X (64 channels, 23 KB at d = 1) by LDS-DMA, double-buffered: 14 kFLOP per staged
//     byte instead of 0.5;
//   * both MFMA operands have the contraction index (the cell) as their ROW index, i.e. both are needed transposed:
//     ds_read_b64_tr_b16 (csrc/wgrad.hip): its 16 lanes fetch 4 consecutive cells x 32 bytes (16 channels).  The images are
//     plain rows -- X [halo cell][128 B], dY [cell][256 B], filled in pieces of whole 128-byte lines (a first version with
//     32-byte records, whose tap shifts were all immediates, ran 2.6 ms: the LDS-DMA path is bound by its line REQUESTS, and
//     that layout made four of them per line) -- with the 32-byte chunk index XOR-swizzled by the cell's COLUMN ((x >> 1) & 3
//     for X, x & 7 for dY): conflict-free reads, a tap row shift ky stays an immediate, a tap column shift kx selects one of
//     three per-lane bases;
//   * contraction step = 32 cells = two tile rows; MFMA k index 8 g + j <-> cell (row 2 s + (j >> 2), x = 4 g + (j & 3)) for
//     both operands (any bijection will do as long as it is the same one);
//   * partial sums go to fp32 slabs [split][128][9 cin], summed in a fixed order by a second small kernel (deterministic).
#include "common.h"
#include <cstdio>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DW_OOB 0x7FFFFFF0u
#define DW_N 128

template <int DIL> struct DwGeom {
  static constexpr int HW = 16 + 2 * DIL, HR = 8 + 2 * DIL, NHC = HR * HW;   // halo of a half tile: HR rows x HW cells
  static constexpr int NPX = ((NHC + 7) / 8 + 7) / 8 * 8;                   // 1-KiB pieces (8 cells x 128 B) of the X image, padded to 8 n
  static constexpr int XBYTES = NPX * 1024;                                  // [halo cell][64 channels]
  static constexpr int YBYTES = 128 * 256;                                   // [cell][128 output channels]: 32 pieces of 4 cells
  static constexpr int STAGE = XBYTES + YBYTES;
  static constexpr int NPIECE = NPX + 32, PPW = NPIECE / 8;                 // transfers per stage, per wave
};

__device__ __forceinline__ s16x4 dw_tr(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p);
}
__device__ __forceinline__ bf16x8 dw_frag(const char* lo, const char* hi) {
  const s16x4 a = dw_tr(lo), b = dw_tr(hi);
  const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, v);
}

#ifdef TMAE_AB
// diagnostic stamps (A/B build only, TMAE_DW_VAR bit 0): per wave, summed over its stages -- s_memtime cycles from the stage's top
// to [0] the end of its vmcnt(0) wait, [1] the barrier's release, [2] the end of group 6 (the last one that issues a transfer),
// [3] the end of the stage; [4] stages; [5] s_memrealtime ticks (100 MHz) over the whole loop.  Written to a buffer of their own
// that nothing else reads (MI355X_MICROARCH.md, DVFS give-back item 6).
__device__ unsigned long long dw_stamps[256 * 8 * 8];
#endif

// VAR (A/B build only; the shipped build instantiates VAR = 0): bit 0 = stamps, bit 1 = raised priority around the MFMA groups
template <int DIL, int VAR = 0>
__global__ __launch_bounds__(512, 1) void dense_wgrad_halo_kernel(const __hip_bfloat16* __restrict__ dY,
                                                                 const __hip_bfloat16* __restrict__ Xin, int B, int Y, int X,
                                                                 int cin, unsigned dybytes, unsigned xbytes, int nsplit,
                                                                 int sps, int tot, float* __restrict__ slab) {
  using G = DwGeom<DIL>;
  constexpr int HW = G::HW, PPW = G::PPW;
  extern __shared__ __attribute__((aligned(1024))) char lds[];         // two stages
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  const int KC = cin / 64;
  // blocks of one split (its KC channel slices stream the same dY half tiles) sit on adjacent ids of one XCD
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int kc = j % KC, s = (j / KC) * 8 + xcd;
  if (s >= nsplit) return;
  const int u0 = s * sps, u1 = min(tot, u0 + sps);
  const int tyN = (Y + 15) / 16, txN = (X + 15) / 16;
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)dY, 0, (int)dybytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)Xin, 0, (int)xbytes, 0x00020000);

  // ---- this lane's PPW transfers per stage: piece P = w PPW + jj; P < NPX: 8 halo cells x 128 B of X (lane l: cell l >> 3,
  // 16-byte slot l & 7), else 4 cells x 256 B of dY (cell l >> 4, slot l & 15).  LDS slot sl of a row holds source chunk
  // (((sl >> 1) ^ key) << 1) | (sl & 1), key = the 32-byte swizzle of the cell's column.
  // desc = (row offset + 8) | (column offset + 8) << 8 | (channel byte offset) << 16 | static-valid << 31
  unsigned desc[PPW];
#pragma unroll
  for (int jj = 0; jj < PPW; ++jj) {
    const int P = w * PPW + jj;
    int ry, rx, chan, sv;
    if (P < G::NPX) {
      const int h = P * 8 + (lane >> 3), sl = lane & 7, hy = h / HW, hx = h - hy * HW;
      ry = hy - DIL; rx = hx - DIL; sv = h < G::NHC;
      chan = ((((sl >> 1) ^ ((hx >> 1) & 3)) << 1) | (sl & 1)) * 16;
    } else {
      const int c = (P - G::NPX) * 4 + (lane >> 4), sl = lane & 15;
      ry = c >> 4; rx = c & 15; sv = 1;
      chan = ((((sl >> 1) ^ (rx & 7)) << 1) | (sl & 1)) * 16;
    }
    desc[jj] = (unsigned)(ry + 8) | ((unsigned)(rx + 8) << 8) | ((unsigned)chan << 16) | ((unsigned)sv << 31);
  }
  // stage u = (tile u >> 1, half u & 1): its cell origin, then one transfer per call (spread over the MFMA groups of the
  // stage before: issued back to back at the stage's top they kept both waves of every SIMD off the matrix core at once)
  int ob = 0, oy = 0, ox = 0;                                            // origin of the stage the transfers are issued for
  int ctx = 0, cty = 0;                                                  // its tile coordinates (walked, not divided)
  auto origin_first = [&](int u) {
    const int tile = u >> 1, t2 = tile / txN;
    ctx = tile - t2 * txN; cty = t2 % tyN; ob = t2 / tyN;
    oy = cty * 16 + 8 * (u & 1); ox = ctx * 16;
  };
  auto origin_next = [&](int u) {                                        // stage u, given the values of stage u - 1
    if (u & 1) { oy += 8; return; }
    if (++ctx == txN) { ctx = 0; if (++cty == tyN) { cty = 0; ++ob; } }
    oy = cty * 16; ox = ctx * 16;
  };
  auto issue_piece = [&](int jj, int buf) {
    const int P = w * PPW + jj;
    const bool isx = P < G::NPX;                                         // wave-uniform
    unsigned d = desc[jj];
    asm volatile("" : "+v"(d));                                          // keep the packed form live across the loop, not its fields
    const int gy = oy + (int)(d & 255u) - 8, gx = ox + (int)((d >> 8) & 255u) - 8;
    const bool ok = (d >> 31) && (unsigned)gy < (unsigned)Y && (unsigned)gx < (unsigned)X;
    const unsigned cell = (unsigned)((ob * Y + gy) * X + gx);
    const unsigned off = cell * (isx ? (unsigned)cin * 2u : (unsigned)(DW_N * 2)) + ((d >> 16) & 0x7FFFu);
    char* dst = lds + buf * G::STAGE + P * 1024;                          // (the dY image follows the X image)
    if (isx)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (__attribute__((address_space(3))) void*)dst, 16, ok ? off : DW_OOB,
                                               kc * 128, 0, 0);
    else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(yr, (__attribute__((address_space(3))) void*)dst, 16, ok ? off : DW_OOB, 0, 0, 0);
  };

  // wave (wn, cg): output channels wn * 64 .. + 63 (4 row tiles) x channels cg * 16 .. + 15 of the slice x 9 taps
  const int wn = w >> 2, cg = w & 3;
  // read bases (stage buffer 0): cell column x = 4 g + q of a tile row, 8-byte quad p of the lane's 32-byte chunk
  unsigned aaddr[4], baddr[3];
#pragma unroll
  for (int a = 0; a < 4; ++a) aaddr[a] = (unsigned)(G::XBYTES + (4 * g + q) * 256 + (((wn * 4 + a) ^ ((4 * g + q) & 7)) << 5) + p * 8);
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    const int x = 4 * g + q + kx * DIL;
    baddr[kx] = (unsigned)(x * 128 + ((cg ^ ((x >> 1) & 3)) << 5) + p * 8);
  }
  f32x4 acc[4][9];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Operand reads are inline asm with hand-counted waits.  (Through the compiler's own LDS reads every stage serialised: it
  // orders an LDS read behind the youngest LDS-DMA write -- s_waitcnt vmcnt(0) right after the NEXT stage's transfers were
  // issued: 3.0 ms per call instead of 1.x.)  A stage = 12 groups (contraction step ks = 0..3 x tap row ky = 0..2) of 3 taps x 4
  // row tiles = 12 MFMAs; the reads of group G + 1 (6 B reads; + 8 A reads when it opens a new contraction step) are issued
  // before the MFMAs of group G into the other B register set.
  s16x4 blo[2][3], bhi[2][3], alo[4], ahi[4];
  bf16x8 af[4];
#define DW_TR(dst, addr, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory")
#define DW_READS(GI)                                                                                                   \
  do {                                                                                                                 \
    constexpr int ks_ = (GI) / 3, ky_ = (GI) % 3, set_ = (GI) & 1;                                                      \
    if constexpr (ky_ == 0) {                                                                                          \
      DW_TR(alo[0], aaddr[0], (2 * ks_) * 4096); DW_TR(ahi[0], aaddr[0], (2 * ks_ + 1) * 4096);                        \
      DW_TR(alo[1], aaddr[1], (2 * ks_) * 4096); DW_TR(ahi[1], aaddr[1], (2 * ks_ + 1) * 4096);                        \
      DW_TR(alo[2], aaddr[2], (2 * ks_) * 4096); DW_TR(ahi[2], aaddr[2], (2 * ks_ + 1) * 4096);                        \
      DW_TR(alo[3], aaddr[3], (2 * ks_) * 4096); DW_TR(ahi[3], aaddr[3], (2 * ks_ + 1) * 4096);                        \
    }                                                                                                                  \
    DW_TR(blo[set_][0], baddr[0], (2 * ks_ + ky_ * DIL) * HW * 128);                                                   \
    DW_TR(bhi[set_][0], baddr[0], (2 * ks_ + 1 + ky_ * DIL) * HW * 128);                                               \
    DW_TR(blo[set_][1], baddr[1], (2 * ks_ + ky_ * DIL) * HW * 128);                                                   \
    DW_TR(bhi[set_][1], baddr[1], (2 * ks_ + 1 + ky_ * DIL) * HW * 128);                                               \
    DW_TR(blo[set_][2], baddr[2], (2 * ks_ + ky_ * DIL) * HW * 128);                                                   \
    DW_TR(bhi[set_][2], baddr[2], (2 * ks_ + 1 + ky_ * DIL) * HW * 128);                                               \
  } while (0)
  // wait until at most N reads are outstanding; the registers group GI consumes pass through the statement
#define DW_WAIT(GI, N)                                                                                                 \
  do {                                                                                                                 \
    constexpr int set_ = (GI) & 1;                                                                                     \
    if constexpr ((GI) % 3 == 0)                                                                                       \
      asm volatile("s_waitcnt lgkmcnt(" #N ")"                                                                         \
                   : "+v"(blo[set_][0]), "+v"(bhi[set_][0]), "+v"(blo[set_][1]), "+v"(bhi[set_][1]), "+v"(blo[set_][2]), \
                     "+v"(bhi[set_][2]), "+v"(alo[0]), "+v"(ahi[0]), "+v"(alo[1]), "+v"(ahi[1]), "+v"(alo[2]),           \
                     "+v"(ahi[2]), "+v"(alo[3]), "+v"(ahi[3])::"memory");                                              \
    else                                                                                                               \
      asm volatile("s_waitcnt lgkmcnt(" #N ")"                                                                         \
                   : "+v"(blo[set_][0]), "+v"(bhi[set_][0]), "+v"(blo[set_][1]), "+v"(bhi[set_][1]), "+v"(blo[set_][2]), \
                     "+v"(bhi[set_][2])::"memory");                                                                    \
  } while (0)
#define DW_MFMAS(GI)                                                                                                   \
  do {                                                                                                                 \
    constexpr int ky_ = (GI) % 3, set_ = (GI) & 1;                                                                     \
    if constexpr ((VAR & 2) != 0) __builtin_amdgcn_s_setprio(1);                                                       \
    if constexpr (ky_ == 0) {                                                                                          \
      _Pragma("unroll") for (int a_ = 0; a_ < 4; ++a_) {                                                               \
        const s16x8 v_ = {alo[a_][0], alo[a_][1], alo[a_][2], alo[a_][3], ahi[a_][0], ahi[a_][1], ahi[a_][2], ahi[a_][3]}; \
        af[a_] = __builtin_bit_cast(bf16x8, v_);                                                                       \
      }                                                                                                                \
    }                                                                                                                  \
    _Pragma("unroll") for (int kx_ = 0; kx_ < 3; ++kx_) {                                                              \
      const s16x8 v_ = {blo[set_][kx_][0], blo[set_][kx_][1], blo[set_][kx_][2], blo[set_][kx_][3],                    \
                        bhi[set_][kx_][0], bhi[set_][kx_][1], bhi[set_][kx_][2], bhi[set_][kx_][3]};                    \
      const bf16x8 bfr_ = __builtin_bit_cast(bf16x8, v_);                                                              \
      _Pragma("unroll") for (int a_ = 0; a_ < 4; ++a_)                                                                 \
        acc[a_][ky_ * 3 + kx_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a_], bfr_, acc[a_][ky_ * 3 + kx_], 0, 0, 0); \
    }                                                                                                                  \
    if constexpr ((VAR & 2) != 0) __builtin_amdgcn_s_setprio(0);                                                       \
  } while (0)
  // group GI: prefetch GI + 1 (NEXT reads: 6, or 14 when GI + 1 opens a contraction step), then wait for GI's own
#define DW_GROUP(GI, NEXT)                                                                                             \
  do {                                                                                                                 \
    DW_READS((GI) + 1);                                                                                                \
    DW_WAIT(GI, NEXT);                                                                                                 \
    DW_MFMAS(GI);                                                                                                      \
    if ((GI) < PPW && more) issue_piece(GI, buf ^ 1);                    /* the other buffer: read in the previous stage */ \
  } while (0)
  if (u0 < u1) {
    origin_first(u0);
#pragma unroll
    for (int jj = 0; jj < PPW; ++jj) issue_piece(jj, 0);
  }
  int buf = 0;
  constexpr bool STAMP = (VAR & 1) != 0;
  unsigned long long st_acc[4] = {0ull, 0ull, 0ull, 0ull}, rt0 = 0ull;
  if constexpr (STAMP) rt0 = __builtin_amdgcn_s_memrealtime();
  for (int u = u0; u < u1; ++u) {
    unsigned long long t0 = 0ull;
    if constexpr (STAMP) t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // this stage's transfers (issued a stage ago)
    if constexpr (STAMP) st_acc[0] += __builtin_amdgcn_s_memtime() - t0;
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if constexpr (STAMP) st_acc[1] += __builtin_amdgcn_s_memtime() - t0;
    const bool more = u + 1 < u1;
    if (more) origin_next(u + 1);
    DW_READS(0);
    DW_GROUP(0, 6); DW_GROUP(1, 6); DW_GROUP(2, 14);
    DW_GROUP(3, 6); DW_GROUP(4, 6); DW_GROUP(5, 14);
    DW_GROUP(6, 6);
    if constexpr (STAMP) st_acc[2] += __builtin_amdgcn_s_memtime() - t0;
    DW_GROUP(7, 6); DW_GROUP(8, 14);
    DW_GROUP(9, 6); DW_GROUP(10, 6);
    DW_WAIT(11, 0);
    DW_MFMAS(11);
    if constexpr (STAMP) st_acc[3] += __builtin_amdgcn_s_memtime() - t0;
    const unsigned step = buf ? 0u - (unsigned)G::STAGE : (unsigned)G::STAGE;     // the read bases follow the buffer
#pragma unroll
    for (int a = 0; a < 4; ++a) aaddr[a] += step;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) baddr[kx] += step;
    buf ^= 1;
  }
#undef DW_GROUP
#undef DW_MFMAS
#undef DW_WAIT
#undef DW_READS
#undef DW_TR
#ifdef TMAE_AB
  if constexpr (STAMP) {
    if (lane == 0 && blockIdx.x < 256) {
      unsigned long long* d = dw_stamps + ((size_t)blockIdx.x * 8 + w) * 8;
      d[0] = st_acc[0]; d[1] = st_acc[1]; d[2] = st_acc[2]; d[3] = st_acc[3];
      d[4] = (unsigned long long)(u1 - u0); d[5] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
  }
#endif
  // rows 4 g + r of tile a = output channel (wn * 4 + a) * 16 + 4 g + r, column i = channel kc * 64 + cg * 16 + i of tap t
  float* __restrict__ sl = slab + (int64_t)s * (DW_N * 9 * cin);
  const int K = 9 * cin;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = (wn * 4 + a) * 16 + 4 * g + r;
#pragma unroll
      for (int t = 0; t < 9; ++t) sl[(int64_t)n * K + t * cin + kc * 64 + cg * 16 + i] = acc[a][t][r];
    }
}

// dw[e] = slab[0][e] + slab[1][e] + ... in that order
__global__ __launch_bounds__(256) void dense_wgrad_reduce_kernel(const float* __restrict__ slab, int nsplit, int count,
                                                                float* __restrict__ dw) {
  const int e = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (e >= count) return;
  float4 acc = *reinterpret_cast<const float4*>(slab + e);
  int s = 1;
  for (; s + 3 < nsplit; s += 4) {                 // four slabs in flight, summed in slab order (the rolled loop: one round trip each)
    float4 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const float4*>(slab + (int64_t)(s + q) * count + e);
#pragma unroll
    for (int q = 0; q < 4; ++q) { acc.x += v[q].x; acc.y += v[q].y; acc.z += v[q].z; acc.w += v[q].w; }
  }
  for (; s < nsplit; ++s) {
    const float4 v = *reinterpret_cast<const float4*>(slab + (int64_t)s * count + e);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  *reinterpret_cast<float4*>(dw + e) = acc;
}

// One workgroup per CU, and the blocks of one split share an XCD (block ids are dealt round-robin over the 8 XCDs): splits x channel
// slices <= 256 is not enough, EVERY XCD must get at most its 32 CUs' worth -- the split count is a multiple of 8.  Until round 6
// this was 256 / KC = 42 for the decoder conv (KC = 6): XCDs 0 and 1 were dealt 6 splits = 36 workgroups for their 32 CUs, four of
// them waited for a free CU and the launch took two rounds (2.07 ms for a loop whose in-kernel stamps add up to 1.15 ms:
// profiles/round6_dense_wgrad_stamps.txt); with 40 splits every XCD runs 30 workgroups in one round.
static int dw_splits(int cin) { return 256 / (cin / 64) / 8 * 8; }

size_t tmae_dense_conv3x3_wgrad_workspace(int cin, int cout) {
  if (cout != DW_N || cin <= 0 || cin % 64) return 0;
  return tmae_align((size_t)dw_splits(cin) * DW_N * 9 * cin * 4) + 1024;
}

// dw [cout, 9 * cin] fp32 (taps ky-major, kx, channel: the layout of tmae_dense_conv3x3's weight) from dy [B, Y, X, cout] and
// x [B, Y, X, cin], both bf16 and contiguous.  cout = 128, cin a multiple of 64 (<= 512), dilation in {1, 2}; TMAE_EARG for
// anything else (callers then take tmae_spconv_wgrad over the full-grid rulebook).
int tmae_dense_conv3x3_wgrad(const void* dy, const void* x, int batch, int ny, int nx, int cin, int cout, int dilation,
                             float* dw, void* wsp, size_t ws_bytes, void* stream_) {
  (void)hipGetLastError();
  hipStream_t stream = (hipStream_t)stream_;
  if (batch <= 0 || ny <= 0 || nx <= 0 || cout != DW_N || cin <= 0 || (cin % 64) || cin > 512 || (dilation != 1 && dilation != 2))
    return TMAE_EARG;
  if (!dy || !x || !dw || !wsp || ((uintptr_t)dy & 15) || ((uintptr_t)x & 15) || ((uintptr_t)dw & 15)) return TMAE_EARG;
  const int64_t cells = (int64_t)batch * ny * nx;
  const int64_t xbytes = cells * cin * 2, dybytes = cells * DW_N * 2;
  if (xbytes >= (int64_t)DW_OOB || dybytes >= (int64_t)DW_OOB) return TMAE_EARG;
  const int64_t tot = (int64_t)batch * ((ny + 15) / 16) * ((nx + 15) / 16) * 2;     // half tiles
  if (tot >= ((int64_t)1 << 30)) return TMAE_EARG;
  const int KC = cin / 64;
  int nsplit = dw_splits(cin);
  if (nsplit > tot) nsplit = (int)tot;
  const int sps = (int)((tot + nsplit - 1) / nsplit);
  nsplit = (int)((tot + sps - 1) / sps);                                            // no empty splits
  const int count = DW_N * 9 * cin;
  WsCarver ws(wsp, ws_bytes);
  float* slab = ws.take<float>((size_t)nsplit * count);
  if (!ws.ok) return TMAE_EWS;
  const unsigned grid = 8u * (unsigned)((nsplit + 7) / 8) * (unsigned)KC;
#define DW_LAUNCH(D)                                                                                                  \
  do {                                                                                                                \
    const int lds = 2 * DwGeom<D>::STAGE;                                                                             \
    static TmaeLdsAttr attr;                                                                                          \
    if (int e_ = tmae_allow_lds(attr, (const void*)dense_wgrad_halo_kernel<D>, lds)) return e_;                       \
    hipLaunchKernelGGL((dense_wgrad_halo_kernel<D>), dim3(grid), dim3(512), lds, stream, (const __hip_bfloat16*)dy,   \
                       (const __hip_bfloat16*)x, batch, ny, nx, cin, (unsigned)dybytes, (unsigned)xbytes, nsplit, sps, \
                       (int)tot, slab);                                                                               \
  } while (0)
#ifdef TMAE_AB
  static const int var = TMAE_AB_INT("TMAE_DW_VAR", 0);
  if (var != 0 && dilation == 1) {
#define DW_LAUNCH_V(V)                                                                                                \
  do {                                                                                                                \
    const int lds = 2 * DwGeom<1>::STAGE;                                                                             \
    static TmaeLdsAttr attr;                                                                                          \
    if (int e_ = tmae_allow_lds(attr, (const void*)dense_wgrad_halo_kernel<1, V>, lds)) return e_;                    \
    hipLaunchKernelGGL((dense_wgrad_halo_kernel<1, V>), dim3(grid), dim3(512), lds, stream, (const __hip_bfloat16*)dy, \
                       (const __hip_bfloat16*)x, batch, ny, nx, cin, (unsigned)dybytes, (unsigned)xbytes, nsplit, sps, \
                       (int)tot, slab);                                                                               \
  } while (0)
    if (var == 1) DW_LAUNCH_V(1); else if (var == 2) DW_LAUNCH_V(2); else DW_LAUNCH_V(3);
#undef DW_LAUNCH_V
    if (var & 1) {                       // print the stamps (diagnostic run: synchronises)
      static int printed = 0;
      (void)hipStreamSynchronize(stream);
      if (printed++ == 3) {
        static unsigned long long h[256 * 8 * 8];
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(dw_stamps), sizeof(h));
        for (int wv = 0; wv < 8; wv += 4) {
          double a[6] = {0, 0, 0, 0, 0, 0};
          int nb = 0;
          for (unsigned b = 0; b < grid && b < 256; ++b) {
            const unsigned long long* d = h + ((size_t)b * 8 + wv) * 8;
            if (d[4] == 0) continue;
            for (int q = 0; q < 4; ++q) a[q] += (double)d[q] / (double)d[4];
            a[4] += (double)d[4]; a[5] += (double)d[3] / ((double)d[5] * 10.0);      // cycles per ns = GHz
            ++nb;
          }
          fprintf(stderr, "dw stamps wave %d over %d blocks: stages %.0f; cycles per stage: vmcnt wait %.0f, +barrier %.0f, "
                          "through group 6 %.0f, whole stage %.0f; in-loop clock %.2f GHz\n", wv, nb, a[4] / nb, a[0] / nb,
                  a[1] / nb, a[2] / nb, a[3] / nb, a[5] / nb);
        }
      }
    }
  } else
#endif
  if (dilation == 1) DW_LAUNCH(1); else DW_LAUNCH(2);
#undef DW_LAUNCH
  hipLaunchKernelGGL(dense_wgrad_reduce_kernel, dim3(tmae_cdiv(count / 4, 256)), dim3(256), 0, stream, slab, nsplit, count, dw);
  return tmae_launch_status();
}
