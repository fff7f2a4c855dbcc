// issue_probe.hip -- what one wave can issue when it has a SIMD (nearly) to itself on gfx950: the
// shader clock under a light load, dependent / independent fp32 adds, the five-instruction root,
// LDS round trips, workgroup barriers.  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/issue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ float sqrt_fast(float x) {
  const float s = __builtin_amdgcn_sqrtf(x);
  const float h = 0.5f * __builtin_amdgcn_rsqf(x);
  const float e = fmaf(-s, s, x);
  return fmaf(e, h, s);
}

// out[0] = shader cycles (s_memtime), out[1] = wall ticks (100 MHz), per test
template <int TEST>
__global__ void probe(float *sink, long long *out, int iters) {
  __shared__ float lds[4096];
  const int lane = threadIdx.x;
  lds[lane] = lane;
  __syncthreads();
  float a = lane * 1e-3f + 1.0f, b = 1.0f + lane, c = 2.0f, d = 3.0f;
  const long long w0 = wall_clock64();
  const long long t0 = clock64();
  for (int i = 0; i < iters; i++) {
    if (TEST == 0) {  // 16 dependent adds
#pragma unroll
      for (int j = 0; j < 16; j++) a = a + b;
    } else if (TEST == 1) {  // 16 adds on 4 independent chains
#pragma unroll
      for (int j = 0; j < 4; j++) { a = a + 1.5f; b = b + 2.5f; c = c + 3.5f; d = d + 4.5f; }
    } else if (TEST == 2) {  // 4 independent exact roots
      a = sqrt_fast(a + 1.0f); b = sqrt_fast(b + 1.0f); c = sqrt_fast(c + 1.0f); d = sqrt_fast(d + 1.0f);
    } else if (TEST == 3) {  // dependent LDS round trip
      a = lds[(static_cast<int>(a) + lane) & 1023] + 1.0f;
    } else if (TEST == 4) {  // barrier
      __syncthreads();
      a = a + 1.0f;
    } else if (TEST == 5) {  // one dependent root chain
      a = sqrt_fast(a + 1.0f);
    }
  }
  const long long t1 = clock64();
  const long long w1 = wall_clock64();
  sink[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; }
}

int main() {
  float *sink; long long *out;
  hipMalloc(&sink, 4 << 20); hipMalloc(&out, 64);
  const int iters = 20000;
  const char *names[] = {"16 dependent v_add_f32", "16 v_add_f32 on 4 chains", "4 independent 5-instr roots",
                         "dependent LDS read", "barrier (block of 256)", "1 dependent 5-instr root"};
  for (int grid : {1, 256, 1024, 4096}) {
    for (int block : {64, 256}) {
      for (int t = 0; t < 6; t++) {
        if (t == 4 && block == 64) continue;
        for (int rep = 0; rep < 2; rep++) {
          switch (t) {
            case 0: probe<0><<<grid, block>>>(sink, out, iters); break;
            case 1: probe<1><<<grid, block>>>(sink, out, iters); break;
            case 2: probe<2><<<grid, block>>>(sink, out, iters); break;
            case 3: probe<3><<<grid, block>>>(sink, out, iters); break;
            case 4: probe<4><<<grid, block>>>(sink, out, iters); break;
            case 5: probe<5><<<grid, block>>>(sink, out, iters); break;
          }
          hipDeviceSynchronize();
        }
        long long h[2];
        hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
        printf("grid %5d block %3d  %-28s  %8.1f cycles/iter  %8.1f ns/iter  clock %.0f MHz\n", grid, block, names[t],
               double(h[0]) / iters, double(h[1]) * 10.0 / iters, double(h[0]) / (double(h[1]) * 10.0) * 1000.0);
      }
    }
  }
  return 0;
}
