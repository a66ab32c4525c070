// semantics check of v_permlane16_swap_b32 as kernels_feature.hip uses it (x16_sum): out[l] = in[l] + in[l ^ 16]
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* o, const float* in) {
  const float v = in[threadIdx.x];
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  o[threadIdx.x] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  o[64 + threadIdx.x] = __uint_as_float(r[0]);
  o[128 + threadIdx.x] = __uint_as_float(r[1]);
}
int main() {
  float h[64], o[192], *di, *dout;
  for (int i = 0; i < 64; ++i) h[i] = (float)(1 << (i % 16)) + 1000.0f * (i / 16);
  hipMalloc(&di, sizeof(h)); hipMalloc(&dout, sizeof(o));
  hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dout, di);
  hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 64; ++i) if (o[i] != h[i] + h[i ^ 16]) ++bad;
  printf("permlane16_swap x16_sum: %s (%d mismatches)\n", bad ? "FAIL" : "ok", bad);
  if (bad) for (int i = 0; i < 64; i += 8) printf("lane %d: in %g r0 %g r1 %g\n", i, h[i], o[64 + i], o[128 + i]);
  return bad != 0;
}
