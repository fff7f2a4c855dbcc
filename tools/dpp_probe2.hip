// dpp_probe2.hip -- cycle cost of dependent in-place DPP add chains (design probe, not product).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP7(x) x x x x x x x
#define REP63(x) REP7(REP7(x)) REP7(x) REP7(x)
#define REP15(x) x x x x x x x x x x x x x x x
template <int MODE>
__global__ void k(float *out, int iters, float a0) {
  float s = threadIdx.x * 0.5f, s2 = threadIdx.x * 0.25f, a = a0 + threadIdx.x * 1e-3f;
  for (int i = 0; i < iters; i++) {
    if (MODE == 0) asm volatile(REP63("s_nop 1\n\tv_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t") : "+v"(s) : "v"(a));
    if (MODE == 1) asm volatile(REP63("v_add_f32_dpp %0, %0, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t") : "+v"(s), "+v"(s2) : "v"(a));
    if (MODE == 2) asm volatile(REP15("s_nop 1\n\tv_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t") REP15("s_nop 1\n\tv_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t") REP15("s_nop 1\n\tv_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t") REP15("s_nop 1\n\tv_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t") "s_nop 1\n\tv_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\tv_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\tv_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t" : "+v"(s) : "v"(a));
    if (MODE == 3) asm volatile(REP63("v_add_f32 %0, %0, %1\n\t") : "+v"(s) : "v"(a));  // plain dependent adds
    if (MODE == 4) asm volatile(REP63("s_nop 0\n\tv_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t") : "+v"(s) : "v"(a));
  }
  out[threadIdx.x] = s + s2;
}
template <int MODE> void run(const char *name, float *d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, d, 100, 1.0f);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, d, iters, 1.0f);
  hipEventRecord(e1, 0); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s %.2f ns per 63-add chain step => %.2f ns per dependent add\n", name, ms * 1e6 / iters, ms * 1e6 / iters / 63);
}
int main() {
  float *d; hipMalloc(&d, 256);
  run<3>("plain v_add_f32 dependent", d);
  run<0>("wave_shr dpp add + s_nop 1", d);
  run<4>("wave_shr dpp add + s_nop 0", d);
  run<1>("two interleaved wave_shr chains (no nop)", d);
  run<2>("row_shr dpp add + s_nop 1", d);
  return 0;
}
