// Micro-benchmark (diagnostic): issue cost of VALU transcendentals on gfx950, one wave per SIMD.
// hipcc --offload-arch=gfx950 -O3 trans_rate.hip -o trans_rate && ./trans_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP>
__global__ void k(float* out, long long* cyc, int iters) {
  float a = threadIdx.x * 1e-3f + 0.1f, b = a + 0.01f, c = a + 0.02f, d = a + 0.03f;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (OP == 0) { REP16(asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %2, %2, %1\n v_add_f32 %3, %3, %1\n v_add_f32 %4, %4, %1" : "+v"(a) : "v"(1e-6f), "v"(b), "v"(c), "v"(d));) }
    if (OP == 1) { REP16(asm volatile("v_sin_f32 %0, %0\n v_sin_f32 %1, %1\n v_sin_f32 %2, %2\n v_sin_f32 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
    if (OP == 2) { REP16(asm volatile("v_sin_f16 %0, %0\n v_sin_f16 %1, %1\n v_sin_f16 %2, %2\n v_sin_f16 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
    if (OP == 3) { REP16(asm volatile("v_cos_f16 %0, %0\n v_cos_f16 %1, %1\n v_cos_f16 %2, %2\n v_cos_f16 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
    if (OP == 4) { REP16(asm volatile("v_fract_f32 %0, %0\n v_fract_f32 %1, %1\n v_fract_f32 %2, %2\n v_fract_f32 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
    if (OP == 5) { REP16(asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %1, %1, %2\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
    if (OP == 6) { REP16(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
    if (OP == 7) { REP16(asm volatile("v_sin_f32 %0, %0\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(1e-6f));) }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int OP> void run(const char* name, float* out, long long* cyc) {
  const int iters = 2000;
  hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256), 0, 0, out, cyc, iters);  // 4 waves per CU = 1 per SIMD
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256), 0, 0, out, cyc, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  const double n = (double)iters * 64;
  printf("%-34s %6.2f memtime ticks / instr   %6.2f ns / instr (wall)\n", name, (double)h / n, ms * 1e6 / n);
}
int main() {
  float* out; long long* cyc; hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
  run<0>("v_add_f32", out, cyc); run<4>("v_fract_f32", out, cyc); run<5>("v_cvt_pk_bf16_f32", out, cyc);
  run<1>("v_sin_f32", out, cyc); run<2>("v_sin_f16", out, cyc); run<3>("v_cos_f16", out, cyc); run<6>("v_exp_f32", out, cyc);
  run<7>("1 v_sin_f32 + 3 v_add_f32 (avg)", out, cyc);
  return 0;
}
