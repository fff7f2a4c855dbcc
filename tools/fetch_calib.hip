// fetch_calib.hip -- known-bytes microbenchmarks for the access patterns of the FFM kernels.
//
// Why: rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 are calibrated (MI355X_MICROARCH.md, HBM
// section) only for wide coalesced streams (FETCH_SIZE = 1/2 of the bytes).  The row kernel mixes
// record streams (16 B per lane, 2496 contiguous bytes), 64-byte slot gathers (one thread = one
// slot, or four lanes = one slot) and 4-byte-per-lane gathers; VERDICT r02 asked for a
// calibration on THIS mix before any counter figure is trusted.  Every kernel below moves a byte
// count that is known by construction; the program prints it (and the bandwidth it reached) as one
// JSON line per kernel, tools/fetch_calib.sh runs it under `rocprofv3 --pmc FETCH_SIZE` /
// `--pmc WRITE_SIZE`, and tools/fetch_calib_summary.py divides.
//
// Second part: VALU issue cost of the instructions the update chains are made of (is a wave64
// v_fma_f32 2 or 4 cycles on a gfx950 SIMD?  v_pk_fma_f32?  v_sqrt_f32?  DPP adds?), which decides
// what the "VALU floor" of a kernel is.
//
// build: hipcc --offload-arch=gfx950 -O3 -o tools/fetch_calib tools/fetch_calib.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                      \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));       \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

constexpr int kRecFloats = 3 * 624;  // one FFM 39x16 record: n | z | w rows of 624 floats
constexpr int kRowF4 = 156;          // 16-byte vectors per row of a record

// bijective scramble of [0, 2^bits): distinct inputs -> distinct outputs (no reuse inside a pass)
__device__ __forceinline__ uint32_t scramble(uint32_t i, uint32_t mask) {
  return (i * 2654435761u + 12345u) & mask;
}

__global__ void sink_kernel(float *out, float v) {
  if (v == 123456.0f) out[0] = v;
}

// K1: coalesced float4 stream over `n4` vectors
__global__ __launch_bounds__(256) void stream_read_f4(const float4 *src, size_t n4, float *out) {
  float acc = 0.0f;
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < n4; i += stride) {
    const float4 v = src[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 123456.0f) out[0] = acc;
}

// K2: the refresh pattern: thread t -> (record, vector): reads the n and z rows (2 x 2496 B) of
// `n_rec` records picked by a scramble of their index
__global__ __launch_bounds__(256) void record_read_nz(const float4 *lat, uint32_t n_rec, uint32_t mask, float *out) {
  float acc = 0.0f;
  const uint32_t total = n_rec * kRowF4;
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
    const uint32_t j = t / kRowF4, c = t - j * kRowF4;
    const float4 *rec = lat + static_cast<size_t>(scramble(j, mask)) * (kRecFloats / 4);
    const float4 n = rec[c], z = rec[kRowF4 + c];
    acc += n.x + z.y;
  }
  if (acc == 123456.0f) out[0] = acc;
}

// K3: the pair phase's `vb` operand: one thread reads ONE 64-byte slot (4 x float4) of the w row of
// a scrambled record; consecutive threads hit different records
__global__ __launch_bounds__(256) void slot_read_thread64(const float4 *lat, uint32_t n_items, uint32_t mask, float *out) {
  float acc = 0.0f;
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n_items; t += gridDim.x * blockDim.x) {
    const uint32_t r = scramble(t, mask);
    const uint32_t slot = (t * 7u) % 39u;
    const float4 *p = lat + static_cast<size_t>(r) * (kRecFloats / 4) + 2 * kRowF4 + slot * 4;
    const float4 a = p[0], b = p[1], c = p[2], d = p[3];
    acc += a.x + b.y + c.z + d.w;
  }
  if (acc == 123456.0f) out[0] = acc;
}

// K4: the same 64-byte slots, four adjacent lanes per slot (16 B each)
__global__ __launch_bounds__(256) void slot_read_quad16(const float4 *lat, uint32_t n_items, uint32_t mask, float *out) {
  float acc = 0.0f;
  const uint32_t total = n_items * 4;
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
    const uint32_t it = t >> 2, q = t & 3;
    const uint32_t r = scramble(it, mask);
    const uint32_t slot = (it * 7u) % 39u;
    const float4 a = lat[static_cast<size_t>(r) * (kRecFloats / 4) + 2 * kRowF4 + slot * 4 + q];
    acc += a.x;
  }
  if (acc == 123456.0f) out[0] = acc;
}

// K5: the chain kernels' partner gathers: 16 lanes read one 64-byte slot, 4 bytes each
__global__ __launch_bounds__(256) void slot_read_lane4(const float *lat, uint32_t n_items, uint32_t mask, float *out) {
  float acc = 0.0f;
  const uint32_t total = n_items * 16;
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
    const uint32_t it = t >> 4, q = t & 15;
    const uint32_t r = scramble(it, mask);
    const uint32_t slot = (it * 7u) % 39u;
    acc += lat[static_cast<size_t>(r) * kRecFloats + 2 * 624 + slot * 16 + q];
  }
  if (acc == 123456.0f) out[0] = acc;
}

// K6: whole w rows (2496 contiguous bytes) of scrambled records, as the LDS-staging variant of the
// pair phase would read them
__global__ __launch_bounds__(256) void record_read_w(const float4 *lat, uint32_t n_rec, uint32_t mask, float *out) {
  float acc = 0.0f;
  const uint32_t total = n_rec * kRowF4;
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
    const uint32_t j = t / kRowF4, c = t - j * kRowF4;
    const float4 w = lat[static_cast<size_t>(scramble(j, mask)) * (kRecFloats / 4) + 2 * kRowF4 + c];
    acc += w.x;
  }
  if (acc == 123456.0f) out[0] = acc;
}

// W1: coalesced float4 store stream
__global__ __launch_bounds__(256) void stream_write_f4(float4 *dst, size_t n4) {
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < n4; i += stride)
    dst[i] = make_float4(1.0f, 2.0f, 3.0f, static_cast<float>(i));
}

// W2: the update's store pattern: n and z rows of scrambled records
__global__ __launch_bounds__(256) void record_write_nz(float4 *lat, uint32_t n_rec, uint32_t mask) {
  const uint32_t total = n_rec * kRowF4;
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
    const uint32_t j = t / kRowF4, c = t - j * kRowF4;
    float4 *rec = lat + static_cast<size_t>(scramble(j, mask)) * (kRecFloats / 4);
    rec[c] = make_float4(1.0f, 0.0f, 0.0f, 0.0f);
    rec[kRowF4 + c] = make_float4(0.0f, 1.0f, 0.0f, 0.0f);
  }
}

// W3: 16-byte stores scattered at (occurrence, field) granularity -- the fact stream (haux)
__global__ __launch_bounds__(256) void scatter_write16(float4 *dst, uint32_t n_items, uint32_t mask) {
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n_items; t += gridDim.x * blockDim.x)
    dst[scramble(t, mask)] = make_float4(1.0f, 2.0f, 3.0f, 4.0f);
}

// RW: read-modify-write of n, z rows (the update of once-only records): reads 2 rows + w, writes 2
__global__ __launch_bounds__(256) void record_update_nz(float4 *lat, uint32_t n_rec, uint32_t mask) {
  const uint32_t total = n_rec * kRowF4;
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
    const uint32_t j = t / kRowF4, c = t - j * kRowF4;
    float4 *rec = lat + static_cast<size_t>(scramble(j, mask)) * (kRecFloats / 4);
    float4 n = rec[c], z = rec[kRowF4 + c];
    const float4 w = rec[2 * kRowF4 + c];
    n.x += w.x; z.y += w.y;
    rec[c] = n;
    rec[kRowF4 + c] = z;
  }
}

// ---- VALU issue cost: `iters` x 8 independent instructions per lane, every wave slot filled ----
template <int OP>
__global__ __launch_bounds__(256) void valu_kernel(float *out, int iters, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,
        a6 = a0 + 6, a7 = a0 + 7;
  const float b = seed * 0.5f, c = seed * 0.25f;
  for (int i = 0; i < iters; i++) {
    if (OP == 0) {  // v_fma_f32
      asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\t"
                   "v_fma_f32 %3, %3, %8, %9\n\tv_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\t"
                   "v_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                   : "v"(b), "v"(c));
    } else if (OP == 1) {  // v_add_f32
      asm volatile("v_add_f32 %0, %0, %8\n\tv_add_f32 %1, %1, %8\n\tv_add_f32 %2, %2, %8\n\t"
                   "v_add_f32 %3, %3, %8\n\tv_add_f32 %4, %4, %8\n\tv_add_f32 %5, %5, %8\n\t"
                   "v_add_f32 %6, %6, %8\n\tv_add_f32 %7, %7, %8"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                   : "v"(b));
    } else if (OP == 2) {  // v_sqrt_f32
      asm volatile("v_sqrt_f32 %0, %0\n\tv_sqrt_f32 %1, %1\n\tv_sqrt_f32 %2, %2\n\tv_sqrt_f32 %3, %3\n\t"
                   "v_sqrt_f32 %4, %4\n\tv_sqrt_f32 %5, %5\n\tv_sqrt_f32 %6, %6\n\tv_sqrt_f32 %7, %7"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if (OP == 3) {  // v_rcp_f32
      asm volatile("v_rcp_f32 %0, %0\n\tv_rcp_f32 %1, %1\n\tv_rcp_f32 %2, %2\n\tv_rcp_f32 %3, %3\n\t"
                   "v_rcp_f32 %4, %4\n\tv_rcp_f32 %5, %5\n\tv_rcp_f32 %6, %6\n\tv_rcp_f32 %7, %7"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if (OP == 4) {  // dependent v_add_f32 chain (latency of one add)
      asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\t"
                   "v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\t"
                   "v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1"
                   : "+v"(a0)
                   : "v"(b));
    } else if (OP == 5) {  // v_add_f32_dpp row_shr:1 chain (the ordered prefix of the update chains)
      asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                   "s_nop 1\n\tv_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                   "s_nop 1\n\tv_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                   "s_nop 1\n\tv_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                   "s_nop 1\n\tv_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                   "s_nop 1\n\tv_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                   "s_nop 1\n\tv_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                   "s_nop 1\n\tv_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf"
                   : "+v"(a0)
                   : "v"(b));
    }
  }
  if (OP == 6) {  // v_pk_fma_f32 on register pairs (two floats per lane per instruction)
    typedef float float2v __attribute__((ext_vector_type(2)));
    float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    float2v q0 = p0 + 1.0f, q1 = p1 + 1.0f, q2 = p2 + 1.0f, q3 = p3 + 1.0f;
    const float2v bb = {b, b}, cc = {c, c};
    for (int i = 0; i < iters; i++) {
      asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n\tv_pk_fma_f32 %1, %1, %8, %9\n\tv_pk_fma_f32 %2, %2, %8, %9\n\t"
                   "v_pk_fma_f32 %3, %3, %8, %9\n\tv_pk_fma_f32 %4, %4, %8, %9\n\tv_pk_fma_f32 %5, %5, %8, %9\n\t"
                   "v_pk_fma_f32 %6, %6, %8, %9\n\tv_pk_fma_f32 %7, %7, %8, %9"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3)
                   : "v"(bb), "v"(cc));
    }
    a0 = p0.x + p1.y + p2.x + p3.y + q0.x + q1.y + q2.x + q3.y;
  }
  const float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (r == 123456.0f) out[0] = r;
}

struct Timer {
  hipEvent_t a, b;
  Timer() { CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b)); }
  void start() { CHECK(hipEventRecord(a, 0)); }
  double stop_us() {
    CHECK(hipEventRecord(b, 0));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e3;
  }
};

static void report(const char *name, const char *kind, double bytes, double us) {
  printf("{\"kernel\": \"%s\", \"kind\": \"%s\", \"known_bytes\": %.0f, \"us\": %.1f, \"GBps\": %.1f}\n", name,
         kind, bytes, us, bytes / us / 1e3);
  fflush(stdout);
}

int main(int argc, char **argv) {
  // 2^20 records of 7488 B = 7.85 GB: 30x the Infinity Cache, nothing is re-read inside a pass
  const uint32_t rec_bits = 20, n_records = 1u << rec_bits, mask = n_records - 1;
  const size_t lat_floats = static_cast<size_t>(n_records) * kRecFloats;
  float *lat, *out;
  CHECK(hipMalloc(&lat, lat_floats * sizeof(float)));
  CHECK(hipMalloc(&out, 4096));
  CHECK(hipMemset(lat, 0, lat_floats * sizeof(float)));
  const float4 *lat4 = reinterpret_cast<const float4 *>(lat);
  float4 *lat4w = reinterpret_cast<float4 *>(lat);
  const int grid = 256 * 8;  // eight 256-thread workgroups per CU
  Timer tm;
  const int reps = argc > 1 ? atoi(argv[1]) : 3;
  for (int rep = 0; rep < reps; rep++) {
    const size_t n4 = static_cast<size_t>(1) << 28;  // 4 GiB stream
    tm.start();
    hipLaunchKernelGGL(stream_read_f4, dim3(grid), dim3(256), 0, 0, lat4, n4, out);
    report("stream_read_f4", "read", 16.0 * n4, tm.stop_us());

    const uint32_t n_rec = 200000;  // 2 x 2496 B each ~ 1 GB
    tm.start();
    hipLaunchKernelGGL(record_read_nz, dim3(grid), dim3(256), 0, 0, lat4, n_rec, mask, out);
    report("record_read_nz", "read", 4992.0 * n_rec, tm.stop_us());

    tm.start();
    hipLaunchKernelGGL(record_read_w, dim3(grid), dim3(256), 0, 0, lat4, 2 * n_rec, mask, out);
    report("record_read_w", "read", 2496.0 * 2 * n_rec, tm.stop_us());

    const uint32_t n_slots = 1u << 23;  // 8 M slots x 64 B = 0.5 GB
    tm.start();
    hipLaunchKernelGGL(slot_read_thread64, dim3(grid), dim3(256), 0, 0, lat4, n_slots, mask, out);
    report("slot_read_thread64", "read", 64.0 * n_slots, tm.stop_us());

    tm.start();
    hipLaunchKernelGGL(slot_read_quad16, dim3(grid), dim3(256), 0, 0, lat4, n_slots, mask, out);
    report("slot_read_quad16", "read", 64.0 * n_slots, tm.stop_us());

    tm.start();
    hipLaunchKernelGGL(slot_read_lane4, dim3(grid), dim3(256), 0, 0, lat, n_slots, mask, out);
    report("slot_read_lane4", "read", 64.0 * n_slots, tm.stop_us());

    tm.start();
    hipLaunchKernelGGL(stream_write_f4, dim3(grid), dim3(256), 0, 0, lat4w, n4 / 4);
    report("stream_write_f4", "write", 16.0 * (n4 / 4), tm.stop_us());

    tm.start();
    hipLaunchKernelGGL(record_write_nz, dim3(grid), dim3(256), 0, 0, lat4w, n_rec, mask);
    report("record_write_nz", "write", 4992.0 * n_rec, tm.stop_us());

    tm.start();
    hipLaunchKernelGGL(scatter_write16, dim3(grid), dim3(256), 0, 0, lat4w, n_slots, (1u << 26) - 1);
    report("scatter_write16", "write", 16.0 * n_slots, tm.stop_us());

    tm.start();
    hipLaunchKernelGGL(record_update_nz, dim3(grid), dim3(256), 0, 0, lat4w, n_rec, mask);
    report("record_update_nz", "read+write", (7488.0 + 4992.0) * n_rec, tm.stop_us());

    // the Infinity Cache: the same 96 MB (12800 records) streamed twice -- second pass on-die?
    for (int pass = 0; pass < 2; pass++) {
      tm.start();
      hipLaunchKernelGGL(stream_read_f4, dim3(grid), dim3(256), 0, 0, lat4, static_cast<size_t>(6) << 20, out);
      report(pass ? "stream_read_f4_96MB_again" : "stream_read_f4_96MB_first", "read", 16.0 * (6 << 20), tm.stop_us());
    }
  }
  // VALU issue cost: 1024 SIMDs x 8 waves, 8 instructions x iters per wave
  const int iters = 20000;
  const char *names[] = {"v_fma_f32", "v_add_f32", "v_sqrt_f32", "v_rcp_f32", "v_add_f32_dependent",
                         "v_add_f32_dpp_row_shr1_dependent", "v_pk_fma_f32"};
  for (int op = 0; op < 7; op++) {
    for (int wpe = 8; wpe >= 1; wpe /= 8) {  // 8 waves per SIMD, then 1
      const int g = 256 * wpe;  // 256-thread workgroups: 4 waves, one per SIMD; wpe of them per CU
      tm.start();
      switch (op) {
        case 0: hipLaunchKernelGGL(valu_kernel<0>, dim3(g), dim3(256), 0, 0, out, iters, 1.0f); break;
        case 1: hipLaunchKernelGGL(valu_kernel<1>, dim3(g), dim3(256), 0, 0, out, iters, 1.0f); break;
        case 2: hipLaunchKernelGGL(valu_kernel<2>, dim3(g), dim3(256), 0, 0, out, iters, 1.0f); break;
        case 3: hipLaunchKernelGGL(valu_kernel<3>, dim3(g), dim3(256), 0, 0, out, iters, 1.0f); break;
        case 4: hipLaunchKernelGGL(valu_kernel<4>, dim3(g), dim3(256), 0, 0, out, iters, 1.0f); break;
        case 5: hipLaunchKernelGGL(valu_kernel<5>, dim3(g), dim3(256), 0, 0, out, iters, 1.0f); break;
        default: hipLaunchKernelGGL(valu_kernel<6>, dim3(g), dim3(256), 0, 0, out, iters, 1.0f); break;
      }
      const double us = tm.stop_us();
      // wave-instructions per SIMD = wpe * 8 * iters; at clk GHz: cycles per instruction per SIMD
      const double per_simd = static_cast<double>(wpe) * 8.0 * iters;
      printf("{\"valu\": \"%s\", \"waves_per_simd\": %d, \"us\": %.1f, \"ns_per_wave_instr_per_simd\": %.3f, "
             "\"cycles_at_2.4GHz\": %.2f}\n", names[op], wpe, us, us * 1e3 / per_simd, us * 1e3 / per_simd * 2.4);
      fflush(stdout);
    }
  }
  CHECK(hipDeviceSynchronize());
  CHECK(hipFree(lat));
  CHECK(hipFree(out));
  return 0;
}
