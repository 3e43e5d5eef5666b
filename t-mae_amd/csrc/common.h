// Shared helpers for the gfx950 kernels of libtmae_hip.so (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/tmae_hip.h"

#define TMAE_WAVE 64

static inline int tmae_launch_status() { return (int)hipGetLastError(); }
static inline size_t tmae_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }
static inline unsigned tmae_cdiv(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute: remember per (call site, device) that it has been
// raised (a process-wide flag would leave the second device of a process at the 64 KB default), thread-safe, and hand
// the error back instead of dropping it.
struct TmaeLdsAttr { unsigned long long done = 0; };
static inline int tmae_allow_lds(TmaeLdsAttr& a, const void* func, int bytes) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  const unsigned long long bit = 1ull << (dev & 63);
  if (__atomic_load_n(&a.done, __ATOMIC_ACQUIRE) & bit) return 0;
  e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return (int)e;
  __atomic_fetch_or(&a.done, bit, __ATOMIC_RELEASE);
  return 0;
}

// CU count of the CURRENT device, cached per device ordinal (a process-wide value would be the first device's: wrong the day
// one process drives two device types), relaxed atomics: two threads racing on the first call store the same value
static inline int tmae_num_cus() {
  static int cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  int v = __atomic_load_n(&cache[dev & 63], __ATOMIC_RELAXED);
  if (v > 0) return v;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
  __atomic_store_n(&cache[dev & 63], v, __ATOMIC_RELAXED);
  return v;
}

// Carves 256-byte aligned sub-buffers out of the caller's workspace.
struct WsCarver {
  char* base;
  size_t size, used;
  bool ok;
  WsCarver(void* p, size_t n) : base((char*)p), size(n), used(0), ok(p != nullptr || n == 0) {}
  template <class T>
  T* take(size_t count) {
    size_t bytes = tmae_align(count * sizeof(T));
    if (!ok || used + bytes > size) { ok = false; return nullptr; }
    T* r = (T*)(base + used);
    used += bytes;
    return r;
  }
};

// ---- element type helpers (feature tensors are f32 or bf16; arithmetic is f32) --------------
template <class T> __device__ __forceinline__ float ld_f(const T* p);
template <> __device__ __forceinline__ float ld_f<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld_f<__hip_bfloat16>(const __hip_bfloat16* p) { return __bfloat162float(*p); }
template <class T> __device__ __forceinline__ void st_f(T* p, float v);
template <> __device__ __forceinline__ void st_f<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st_f<__hip_bfloat16>(__hip_bfloat16* p, float v) { *p = __float2bfloat16(v); }

__device__ __forceinline__ float bf16_bits_to_f(unsigned short b) { return __uint_as_float(((unsigned)b) << 16); }

// ---- correctly-rounded single fp32 ops that the compiler may NOT contract into an fma.  hipcc defaults to
// -ffp-contract=fast and HIP's __fmul_rn/__fadd_rn are plain operators, so parity-critical arithmetic
// (voxel indices, voxel centres: the reference rounds every op separately) goes through these.
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float add_rn(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ float sub_rn(float a, float b) {
#pragma clang fp contract(off)
  return a - b;
}
__device__ __forceinline__ float div_rn(float a, float b) { return __fdiv_rn(a, b); }

// ---- wave reductions ---------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- 8 consecutive elements per lane (one 16-byte access in bf16, two in fp32): the layout of the row-streaming
// kernels (LayerNorm, BatchNorm).  A row of C channels is held by C/8 adjacent lanes, a wavefront holds 512/C rows.
template <class T> __device__ __forceinline__ void load8(const T* p, float* v);
template <> __device__ __forceinline__ void load8<float>(const float* p, float* v) {
  const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <> __device__ __forceinline__ void load8<__hip_bfloat16>(const __hip_bfloat16* p, float* v) {
  const uint4 u = *reinterpret_cast<const uint4*>(p);
  const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int q = 0; q < 4; ++q) { v[2 * q] = __uint_as_float(w[q] << 16); v[2 * q + 1] = __uint_as_float(w[q] & 0xFFFF0000u); }
}
template <class T> __device__ __forceinline__ void store8(T* p, const float* v);
template <> __device__ __forceinline__ void store8<float>(float* p, const float* v) {
  reinterpret_cast<float4*>(p)[0] = make_float4(v[0], v[1], v[2], v[3]);
  reinterpret_cast<float4*>(p)[1] = make_float4(v[4], v[5], v[6], v[7]);
}
template <> __device__ __forceinline__ void store8<__hip_bfloat16>(__hip_bfloat16* p, const float* v) {
  __hip_bfloat16 t[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) t[i] = __float2bfloat16(v[i]);
  *reinterpret_cast<uint4*>(p) = *reinterpret_cast<const uint4*>(t);
}
// the same with the nt (streaming) cache policy: for tensors that are written now and read much later (saved for the
// backward pass), so that they do not push the next kernel's inputs out of L2 / the Infinity Cache
typedef unsigned tmae_u32x4 __attribute__((ext_vector_type(4)));
typedef float tmae_f32x4 __attribute__((ext_vector_type(4)));
template <class T> __device__ __forceinline__ void store8_nt(T* p, const float* v);
template <> __device__ __forceinline__ void store8_nt<float>(float* p, const float* v) {
  __builtin_nontemporal_store(tmae_f32x4{v[0], v[1], v[2], v[3]}, reinterpret_cast<tmae_f32x4*>(p));
  __builtin_nontemporal_store(tmae_f32x4{v[4], v[5], v[6], v[7]}, reinterpret_cast<tmae_f32x4*>(p) + 1);
}
template <> __device__ __forceinline__ void store8_nt<__hip_bfloat16>(__hip_bfloat16* p, const float* v) {
  __hip_bfloat16 t[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) t[i] = __float2bfloat16(v[i]);
  __builtin_nontemporal_store(*reinterpret_cast<const tmae_u32x4*>(t), reinterpret_cast<tmae_u32x4*>(p));
}
// sum over the aligned group of G adjacent lanes (G a power of two <= 64)
template <int G> __device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// sum over the lanes that hold the same channels of different rows (stride G, G a power of two <= 64)
template <int G> __device__ __forceinline__ float cross_group_sum(float v) {
#pragma unroll
  for (int o = 32; o >= G; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- csrc/token_gemm_wreg.hip (internal: reached through tmae_token_gemm / tmae_token_gemm_pos)
int tmae_token_gemm_wreg(const void* x, int64_t ldx, int64_t m, int k, const void* w, int n, const void* bias,
                         const uint8_t* cells, void* y, int64_t ldy, int accumulate, void* stream);
int tmae_token_gemm_wreg_gelu(const void* x, int64_t ldx, int64_t m, int k, const void* w, int n, const void* bias, void* y,
                              void* y_gelu, int64_t ldy, void* stream);
int tmae_token_gemm_wreg_res(const void* x, int64_t ldx, int64_t m, int k, const void* w, int n, const void* bias, const void* res,
                             void* y, int64_t ldy, void* stream);
int tmae_token_gemm_wreg_dgelu(const void* x, int64_t ldx, int64_t m, int k, const void* w, int n, const void* bias,
                               const void* aux, void* y, int64_t ldy, void* stream);

// ---- device-wide exclusive scan of int32 (three small launches per level) ----------------------
// out[i] = sum_{j<i} in[i];  *total (device, optional) = sum of all.  in != out.
size_t tmae_scan_i32_workspace(int64_t n);
int tmae_scan_i32(const int32_t* in, int32_t* out, int64_t n, int32_t* total, void* ws, size_t ws_bytes,
                  hipStream_t stream);

// ---- A/B switches ------------------------------------------------------------------------------
// Only a debug build made with -DTMAE_AB (python t-mae_amd/build.py --ab -> t-mae_amd/build_ab/libtmae_ab.so, used by
// profiles/scripts/ab_env.sh through TMAE_LIB_PATH) reads the environment, and only such a build carries the retired
// kernel variants behind those switches.  The shipped libtmae_hip.so picks its kernels from its arguments alone: a C-ABI
// library must not change behaviour with the caller's environment.
#ifdef TMAE_AB
#include <cstdlib>
#define TMAE_AB_INT(name, dflt) ([]() -> int { const char* e_ = getenv(name); return e_ ? atoi(e_) : (dflt); }())
#else
#define TMAE_AB_INT(name, dflt) (dflt)
#endif
