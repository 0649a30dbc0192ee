// v_mfma_f32_4x4x1_16B_f32 (16 independent 4x4 outer products per instruction): which lane feeds which element, and what does it cost
// next to v_mfma_f32_16x16x4_f32?  Prints, for every lane and output register, the (A lane, B lane) whose product lands there, then
// the time per instruction of a chain of independent MFMAs of both shapes (4 waves per SIMD, one workgroup of 1024 threads per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k_map(float* out) {
  const int lane = threadIdx.x;
  f32x4 z{0, 0, 0, 0};
  const f32x4 da = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(lane + 1), 1.0f, z, 0, 0, 0);   // -> index of the A lane + 1
  const f32x4 db = __builtin_amdgcn_mfma_f32_4x4x1f32(1.0f, (float)(lane + 1), z, 0, 0, 0);   // -> index of the B lane + 1
  for (int v = 0; v < 4; ++v) {
    out[(lane * 4 + v) * 2] = da[v];
    out[(lane * 4 + v) * 2 + 1] = db[v];
  }
}

template <int SHAPE>
__global__ __launch_bounds__(1024) void k_rate(float* out, int iters) {
  const int lane = threadIdx.x & 63;
  const float a = (float)lane * 0.001f, b = 1.0f - (float)lane * 0.002f;
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      acc[i] = SHAPE == 4 ? __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0) : __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * 1024 + threadIdx.x] = s;
}

template <int SHAPE>
void rate(int iters) {
  float* out;
  CK(hipMalloc(&out, 256 * 1024 * sizeof(float)));
  hipLaunchKernelGGL(k_rate<SHAPE>, dim3(256), dim3(1024), 0, 0, out, 10);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(k_rate<SHAPE>, dim3(256), dim3(1024), 0, 0, out, iters);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double per_simd = 4.0 * 8.0 * iters;           // instructions per SIMD (4 waves x 8 per iteration)
  printf("{\"shape\": \"%s\", \"ns_per_mfma_per_simd\": %.3f}\n", SHAPE == 4 ? "4x4x1_16B" : "16x16x4", ms * 1e6 / per_simd);
  CK(hipFree(out));
}

int main() {
  float* out;
  CK(hipMalloc(&out, 64 * 4 * 2 * sizeof(float)));
  hipLaunchKernelGGL(k_map, dim3(1), dim3(64), 0, 0, out);
  CK(hipDeviceSynchronize());
  float h[64 * 4 * 2];
  CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
  for (int lane = 0; lane < 64; ++lane) {
    printf("lane %2d:", lane);
    for (int v = 0; v < 4; ++v) printf("  d[%d] = A(lane %2d) x B(lane %2d)", v, (int)h[(lane * 4 + v) * 2] - 1, (int)h[(lane * 4 + v) * 2 + 1] - 1);
    printf("\n");
  }
  rate<4>(20000);
  rate<16>(20000);
  return 0;
}
