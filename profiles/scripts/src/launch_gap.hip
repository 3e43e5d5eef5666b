// What one more kernel on the stream costs when the host is far ahead: N back-to-back launches of a kernel that spins for a fixed
// number of clock ticks, for two spin lengths; the slope of total time over the spin is the kernel time, the intercept the
// per-launch cost (dispatch + the cache write-back / invalidate at the kernel boundary).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/launch_gap profiles/scripts/src/launch_gap.hip && /tmp/launch_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void spin_kernel(long long ticks, int* sink) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (ticks < 0) *sink = 1;
}
__global__ void touch_kernel(float* p, int n) {            // a kernel that dirties memory (so the boundary has lines to write back)
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] += 1.0f;
}
int main() {
  int* sink; hipMalloc(&sink, 4);
  float* buf; const int nb = 64 << 20; hipMalloc(&buf, nb * 4); hipMemset(buf, 0, nb * 4);
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int N = 2000;
  for (int grid : {256, 2048}) {
    for (long long us : {10, 20, 40}) {
      const long long ticks = us * 100;                      // wall_clock64 runs at 100 MHz
      for (int w = 0; w < 50; ++w) hipLaunchKernelGGL(spin_kernel, dim3(grid), dim3(256), 0, s, ticks, sink);
      hipStreamSynchronize(s);
      hipEventRecord(e0, s);
      for (int i = 0; i < N; ++i) hipLaunchKernelGGL(spin_kernel, dim3(grid), dim3(256), 0, s, ticks, sink);
      hipEventRecord(e1, s);
      hipStreamSynchronize(s);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("spin grid %4d  %3lld us requested: %.3f us per launch\n", grid, us, ms * 1e3 / N);
    }
  }
  for (int n : {1 << 20, 16 << 20, 64 << 20}) {           // 4 MB, 64 MB, 256 MB read + written per launch
    for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(touch_kernel, dim3(n / 256), dim3(256), 0, s, buf, n);
    hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int i = 0; i < 500; ++i) hipLaunchKernelGGL(touch_kernel, dim3(n / 256), dim3(256), 0, s, buf, n);
    hipEventRecord(e1, s);
    hipStreamSynchronize(s);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("touch %3d MB: %.3f us per launch (%.2f TB/s)\n", n >> 18, ms * 1e3 / 500, 8.0 * n / (ms * 1e-3 / 500) * 1e-12);
  }
  return 0;
}
