// Micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 and v_mfma_f32_32x32x2_f32 on gfx950 as a function of the
// number of independent accumulators and of wavefronts per SIMD.  hipcc --offload-arch=gfx950 -O3 mfma_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void k_f64(double* out, int iters) {
  v4d acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = v4d{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ void k_f32(float* out, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0;
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_fma32(float* out, int iters) {
  float x[16];
  for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 1e-3f + i;
  const float a = 1.0001f, b = 1e-7f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = x[i] * a + b;
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_fma64(double* out, int iters) {
  double x[16];
  for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 1e-3 + i;
  const double a = 1.0001, b = 1e-7;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = x[i] * a + b;
  }
  double s = 0;
  for (int i = 0; i < 16; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class F> float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  double* d; hipMalloc(&d, 1 << 24);
  const int iters = 20000;
  const int ncu = 256;
  for (int wpsimd = 1; wpsimd <= 2; ++wpsimd) {
    const int threads = 256 * wpsimd;   // 4 SIMDs x wpsimd waves per workgroup, one workgroup per CU
    float ms;
    ms = timeit([&] { hipLaunchKernelGGL(k_f64<4>, dim3(ncu), dim3(threads), 0, 0, d, iters); });
    printf("f64 16x16x4  nacc=4 waves/SIMD=%d: %.1f ns/MFMA/wave  %.1f TFLOP/s\n", wpsimd, ms * 1e6 / (iters * 4.0), ncu * (threads / 64) * iters * 4.0 * 2048 / (ms * 1e-3) / 1e12);
    ms = timeit([&] { hipLaunchKernelGGL(k_f64<8>, dim3(ncu), dim3(threads), 0, 0, d, iters); });
    printf("f64 16x16x4  nacc=8 waves/SIMD=%d: %.1f ns/MFMA/wave  %.1f TFLOP/s\n", wpsimd, ms * 1e6 / (iters * 8.0), ncu * (threads / 64) * iters * 8.0 * 2048 / (ms * 1e-3) / 1e12);
    ms = timeit([&] { hipLaunchKernelGGL(k_f64<1>, dim3(ncu), dim3(threads), 0, 0, d, iters); });
    printf("f64 16x16x4  nacc=1 waves/SIMD=%d: %.1f ns/MFMA/wave  %.1f TFLOP/s\n", wpsimd, ms * 1e6 / (iters * 1.0), ncu * (threads / 64) * iters * 1.0 * 2048 / (ms * 1e-3) / 1e12);
    ms = timeit([&] { hipLaunchKernelGGL(k_f32<4>, dim3(ncu), dim3(threads), 0, 0, (float*)d, iters); });
    printf("f32 32x32x2  nacc=4 waves/SIMD=%d: %.1f ns/MFMA/wave  %.1f TFLOP/s\n", wpsimd, ms * 1e6 / (iters * 4.0), ncu * (threads / 64) * iters * 4.0 * 4096 / (ms * 1e-3) / 1e12);
    ms = timeit([&] { hipLaunchKernelGGL(k_fma32, dim3(ncu), dim3(threads), 0, 0, (float*)d, iters); });
    printf("f32 v_fma    x16    waves/SIMD=%d: %.2f ns/FMA/wave  %.1f TFLOP/s\n", wpsimd, ms * 1e6 / (iters * 16.0), ncu * (threads / 64) * iters * 16.0 * 128 / (ms * 1e-3) / 1e12);
    ms = timeit([&] { hipLaunchKernelGGL(k_fma64, dim3(ncu), dim3(threads), 0, 0, d, iters); });
    printf("f64 v_fma    x16    waves/SIMD=%d: %.2f ns/FMA/wave  %.1f TFLOP/s\n", wpsimd, ms * 1e6 / (iters * 16.0), ncu * (threads / 64) * iters * 16.0 * 128 / (ms * 1e-3) / 1e12);
  }
  return 0;
}
