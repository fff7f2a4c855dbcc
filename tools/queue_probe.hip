// queue_probe.hip -- what the command processor charges between two dependent kernels of one stream
// when events and a second, busy queue are involved (the step boundary of the engine):
//   hipcc --offload-arch=gfx950 -O3 tools/queue_probe.hip -o /tmp/queue_probe && /tmp/queue_probe
// Kernels stamp wall_clock64() (100 MHz) at their first and last instruction into a table; the host
// enqueues everything ahead and reads the table afterwards, so nothing but the queues is measured.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__global__ void work(long long *stamps, int slot, int ticks) {
  const long long t0 = wall_clock64();
  if (blockIdx.x == 0 && threadIdx.x == 0) stamps[2 * slot] = t0;
  while (wall_clock64() - t0 < ticks) {}
  if (blockIdx.x == 0 && threadIdx.x == 0) stamps[2 * slot + 1] = wall_clock64();
}
__global__ __launch_bounds__(1024) void pass(int *p, int ticks) {
  __shared__ int lds[5248];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (p && lds[threadIdx.x] == -1) *p = 1;
}
__global__ void tiny(int *p) { if (p && threadIdx.x == 1000) *p = 1; }

struct Stat { double mean, p10, p90; };
static Stat gaps(const std::vector<long long> &st, int n) {
  std::vector<double> g;
  for (int i = 1; i < n; i++) g.push_back((st[2 * i] - st[2 * (i - 1) + 1]) / 100.0);
  std::sort(g.begin(), g.end());
  double s = 0;
  for (double x : g) s += x;
  return {s / g.size(), g[g.size() / 10], g[g.size() * 9 / 10]};
}

int main() {
  const int N = 60, TICKS = 10000;  // 100 us of work per kernel
  hipStream_t a, b, c, d;
  CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&c, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&d, hipStreamNonBlocking));
  long long *stamps;
  CK(hipMalloc(&stamps, sizeof(long long) * 2 * N));
  CK(hipMemset(stamps, 0, sizeof(long long) * 2 * N));
  int *sink;
  CK(hipMalloc(&sink, 4096));
  std::vector<hipEvent_t> ev(4 * N);
  for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  std::vector<long long> h(2 * N);
  auto report = [&](const char *name) {
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), stamps, sizeof(long long) * 2 * N, hipMemcpyDeviceToHost));
    const Stat s = gaps(h, N);
    std::printf("%-78s gap mean %6.1f us  p10 %6.1f  p90 %6.1f\n", name, s.mean, s.p10, s.p90);
  };
  const int GRID = 1024;
  for (int rep = 0; rep < 2; rep++) {
    // 1: back to back on one stream
    for (int i = 0; i < N; i++) hipLaunchKernelGGL(work, dim3(GRID), dim3(256), 0, a, stamps, i, TICKS);
    report("1 kernel, kernel, ... on one stream");
    // 2: an event record between them
    for (int i = 0; i < N; i++) {
      hipLaunchKernelGGL(work, dim3(GRID), dim3(256), 0, a, stamps, i, TICKS);
      CK(hipEventRecord(ev[i], a));
    }
    report("2 kernel, record, kernel, record, ...");
    // 3: two records + two waits on long-satisfied events of another stream
    CK(hipEventRecord(ev[N], b));
    CK(hipEventRecord(ev[N + 1], b));
    CK(hipStreamSynchronize(b));
    for (int i = 0; i < N; i++) {
      hipLaunchKernelGGL(work, dim3(GRID), dim3(256), 0, a, stamps, i, TICKS);
      CK(hipEventRecord(ev[i], a));
      CK(hipEventRecord(ev[2 * N + i], a));
      CK(hipStreamWaitEvent(a, ev[N], 0));
      CK(hipStreamWaitEvent(a, ev[N + 1], 0));
    }
    report("3 kernel, 2 records, 2 waits on events that completed long ago");
    // 4: the second stream waits for each kernel and runs a chain of tiny kernels beside the next one
    for (int chain : {1, 8, 20}) {
      for (int i = 0; i < N; i++) {
        hipLaunchKernelGGL(work, dim3(GRID), dim3(256), 0, a, stamps, i, TICKS);
        CK(hipEventRecord(ev[i], a));
        CK(hipStreamWaitEvent(b, ev[i], 0));
        for (int k = 0; k < chain; k++) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, b, sink);
      }
      char name[128];
      std::snprintf(name, sizeof name, "4 kernel, record; stream B: wait, %d tiny kernels", chain);
      report(name);
    }
    // 5: like 4 with 20, and the main stream also waits for B's chain of the iteration before last
    for (int i = 0; i < N; i++) {
      hipLaunchKernelGGL(work, dim3(GRID), dim3(256), 0, a, stamps, i, TICKS);
      CK(hipEventRecord(ev[i], a));
      CK(hipStreamWaitEvent(b, ev[i], 0));
      for (int k = 0; k < 20; k++) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, b, sink);
      CK(hipEventRecord(ev[N + i], b));
      if (i >= 2) CK(hipStreamWaitEvent(a, ev[N + i - 2], 0));
    }
    report("5 ... + main waits for B's chain of two iterations ago");
    // 6: like 4 with memsets instead of tiny kernels
    for (int i = 0; i < N; i++) {
      hipLaunchKernelGGL(work, dim3(GRID), dim3(256), 0, a, stamps, i, TICKS);
      CK(hipEventRecord(ev[i], a));
      CK(hipStreamWaitEvent(b, ev[i], 0));
      for (int k = 0; k < 10; k++) CK(hipMemsetAsync(sink, 0, 256, b));
    }
    report("6 kernel, record; stream B: wait, 10 memsets");
    // 7: fork / join: kernel on A, then kernels on C and D that wait for it, A waits for both, next kernel
    for (int i = 0; i < N; i++) {
      hipLaunchKernelGGL(work, dim3(GRID), dim3(256), 0, a, stamps, i, TICKS);
      CK(hipEventRecord(ev[i], a));
      CK(hipStreamWaitEvent(c, ev[i], 0));
      CK(hipStreamWaitEvent(d, ev[i], 0));
      hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, c, sink);
      hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, d, sink);
      CK(hipEventRecord(ev[N + i], c));
      CK(hipEventRecord(ev[2 * N + i], d));
      CK(hipStreamWaitEvent(a, ev[N + i], 0));
      CK(hipStreamWaitEvent(a, ev[2 * N + i], 0));
    }
    report("7 kernel -> fork to two streams (tiny kernel each) -> join -> kernel");
    // 8: the same, and B runs its chain of 20 beside it
    for (int i = 0; i < N; i++) {
      hipLaunchKernelGGL(work, dim3(GRID), dim3(256), 0, a, stamps, i, TICKS);
      CK(hipEventRecord(ev[i], a));
      CK(hipStreamWaitEvent(c, ev[i], 0));
      CK(hipStreamWaitEvent(d, ev[i], 0));
      CK(hipStreamWaitEvent(b, ev[i], 0));
      for (int k = 0; k < 20; k++) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, b, sink);
      hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, c, sink);
      hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, d, sink);
      CK(hipEventRecord(ev[N + i], c));
      CK(hipEventRecord(ev[2 * N + i], d));
      CK(hipStreamWaitEvent(a, ev[N + i], 0));
      CK(hipStreamWaitEvent(a, ev[2 * N + i], 0));
    }
    report("8 fork / join as 7 + stream B's chain of 20 tiny kernels");
    // 9: the engine's FM step.  main: wait copied(i), wait grouped(i), row kernel, update kernel, tiny, record
    // set_free(i), record trained(i).  look-ahead stream, two blocks ahead: upload-like kernel, record
    // copied(i+2), wait set_free(i-1), 10 tiny kernels + 3 passes of twenty 1024-thread workgroups, record grouped(i+2)
    for (int variant = 0; variant < 3; variant++) {
      const int M = N / 2;
      auto prep = [&](int blk) {
        hipLaunchKernelGGL(work, dim3(24), dim3(256), 0, b, stamps + 2 * N - 2, 0, 8000);
        CK(hipEventRecord(ev[blk], b));                                     // copied
        if (blk >= 3) CK(hipStreamWaitEvent(b, ev[2 * N + blk - 3], 0));     // set_free of block blk - 3
        for (int k = 0; k < 10; k++) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, b, sink);
        if (variant != 1) for (int k = 0; k < 3; k++) {
          hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, b, sink);
          hipLaunchKernelGGL(pass, dim3(20), dim3(1024), 0, b, sink, 1000);
        }
        CK(hipEventRecord(ev[N + blk], b));                                 // grouped
      };
      prep(0); prep(1);
      for (int i = 0; i < M; i++) {
        if (i + 2 < M) prep(i + 2);
        if (variant != 2) CK(hipStreamWaitEvent(a, ev[i], 0));
        CK(hipStreamWaitEvent(a, ev[N + i], 0));
        hipLaunchKernelGGL(work, dim3(GRID), dim3(256), 0, a, stamps, i, TICKS);
        hipLaunchKernelGGL(work, dim3(GRID), dim3(256), 0, a, stamps + 2 * M, i, 8000);
        hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, a, sink);
        CK(hipEventRecord(ev[2 * N + i], a));                               // set_free
        if (variant != 2) CK(hipEventRecord(ev[3 * N + i], a));             // trained
      }
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(h.data(), stamps, sizeof(long long) * 2 * N, hipMemcpyDeviceToHost));
      std::vector<double> g;
      for (int i = 1; i < M - 1; i++) g.push_back((h[2 * i] - h[2 * M + 2 * (i - 1) + 1]) / 100.0);
      std::sort(g.begin(), g.end());
      double sum = 0;
      for (double x : g) sum += x;
      std::printf("9.%d engine-like FM step%-58s gap mean %6.1f us  p10 %6.1f  p90 %6.1f\n", variant,
                  variant == 0 ? "" : variant == 1 ? " (no 1024-thread passes)" : " (one wait, one record)", sum / g.size(), g[g.size() / 10], g[g.size() * 9 / 10]);
    }
    // 10: hipExtAnyOrderLaunch: does a kernel launched "in any order" start before its predecessor on the
    // same stream ends?  (two small grids: room for both)
    for (int i = 0; i < N / 2; i++) {
      hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, a, stamps, 2 * i, TICKS);
      hipExtLaunchKernelGGL(work, dim3(64), dim3(256), 0, a, nullptr, nullptr, hipExtAnyOrderLaunch, stamps, 2 * i + 1, TICKS);
    }
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), stamps, sizeof(long long) * 2 * N, hipMemcpyDeviceToHost));
    {
      double ov = 0;
      for (int i = 1; i < N / 2; i++) ov += (h[2 * (2 * i + 1)] - h[2 * (2 * i)]) / 100.0;  // start of the second - start of the first
      std::printf("10 any-order launch: second kernel starts %.1f us after the first STARTS (kernels last 100 us)\n", ov / (N / 2 - 1));
    }
  }
  return 0;
}
