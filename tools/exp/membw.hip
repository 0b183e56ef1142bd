// Memory-pipeline calibration for the RoI kernel's access shape (tools/exp, not part of the library):
//   read_seq      every workgroup streams a contiguous slice (16 B per lane)
//   write_seq     the same, stores (nontemporal)
//   copy          read + write
//   read_rows     1 KB rows (64 lanes x 16 B = one RoI tap of 256 float32 channels) in a PERMUTED order, XCD x only
//                 touching region x (the image -> XCD pinning of k_roi_pool), rows_per_wave rows in flight per wave
//   read_rows_w   the same with a 1 KB store per 7.93 / 1 rows read (the kernel's read : write ratio = 2 : 1 in bytes)
// Prints GB/s of each.  hipcc --offload-arch=gfx950 -O3 tools/exp/membw.hip -o /tmp/membw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_read_seq(const f4* __restrict__ src, size_t n_per_wg, float* sink) {
  const f4* p = src + (size_t)blockIdx.x * n_per_wg;
  f4 acc = {0, 0, 0, 0};
  for (size_t i = threadIdx.x; i < n_per_wg; i += 256 * 4) {
    f4 a = p[i], b = p[i + 256], c = p[i + 512], d = p[i + 768];
    acc += a + b + c + d;
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}
// grid-stride form: consecutive workgroups read consecutive 4 KB pieces, DEPTH loads in flight per lane
template <int DEPTH, bool NT>
__global__ void __launch_bounds__(256) k_read_gs(const f4* __restrict__ src, size_t n, float* sink) {
  const size_t stride = (size_t)gridDim.x * 256;
  f4 acc = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + (DEPTH - 1) * stride < n; i += DEPTH * stride) {
    f4 v[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) v[d] = NT ? __builtin_nontemporal_load(src + i + d * stride) : src[i + d * stride];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) acc += v[d];
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}
__global__ void __launch_bounds__(256) k_write_seq(f4* __restrict__ dst, size_t n_per_wg) {
  f4* p = dst + (size_t)blockIdx.x * n_per_wg;
  const f4 v = {1.0f, 2.0f, 3.0f, (float)blockIdx.x};
  for (size_t i = threadIdx.x; i < n_per_wg; i += 256) __builtin_nontemporal_store(v, p + i);
}
__global__ void __launch_bounds__(256) k_copy(const f4* __restrict__ src, f4* __restrict__ dst, size_t n_per_wg) {
  const f4* p = src + (size_t)blockIdx.x * n_per_wg;
  f4* q = dst + (size_t)blockIdx.x * n_per_wg;
  for (size_t i = threadIdx.x; i < n_per_wg; i += 256 * 4) {
    f4 a = p[i], b = p[i + 256], c = p[i + 512], d = p[i + 768];
    __builtin_nontemporal_store(a, q + i); __builtin_nontemporal_store(b, q + i + 256);
    __builtin_nontemporal_store(c, q + i + 512); __builtin_nontemporal_store(d, q + i + 768);
  }
}
// order[]: row indices (1 KB rows) region by region; workgroup g of XCD x (= blockIdx.x & 7) takes the rows
// [ (g*W + wave) * rows_per_wave ... ) of region x.  DEPTH rows in flight per wave.
template <int DEPTH, int STORE_EVERY>
__global__ void __launch_bounds__(448) k_read_rows(const f4* __restrict__ src, const unsigned* __restrict__ order,
                                                   unsigned rows_per_region, unsigned rows_per_wave, f4* __restrict__ dst,
                                                   float* sink) {
  const unsigned xcd = blockIdx.x & 7, g = blockIdx.x >> 3;
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned first = (g * 7 + wave) * rows_per_wave;
  if (first >= rows_per_region) return;
  const unsigned* ord = order + (size_t)xcd * rows_per_region + first;
  const unsigned n = min(rows_per_wave, rows_per_region - first);
  f4 acc = {0, 0, 0, 0};
  f4* out = dst + ((size_t)xcd * rows_per_region + first) / (STORE_EVERY > 0 ? STORE_EVERY : 1) * 64 + lane;
  unsigned stored = 0;
  for (unsigned i = 0; i + DEPTH <= n; i += DEPTH) {
    f4 v[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) v[d] = src[(size_t)ord[i + d] * 64 + lane];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) acc += v[d];
    if (STORE_EVERY > 0) { __builtin_nontemporal_store(acc, out + (size_t)stored * 64); ++stored; }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}

// the read : write mix of k_roi_pool (2 rows read per 1 KB row stored) with buffer instructions and explicit cache-policy
// bits (aux: 1 = sc0, 2 = nt, 16 = sc1) on the loads and on the stores
typedef unsigned u4v __attribute__((ext_vector_type(4)));
template <int LAUX, int SAUX>
__global__ void __launch_bounds__(448) k_mix(const f4* __restrict__ src, const unsigned* __restrict__ order,
                                             unsigned rows_per_region, unsigned rows_per_wave, f4* __restrict__ dst) {
  const unsigned xcd = blockIdx.x & 7, g = blockIdx.x >> 3;
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned first = (g * 7 + wave) * rows_per_wave;
  if (first >= rows_per_region) return;
  const unsigned* ord = order + (size_t)xcd * rows_per_region + first;
  const unsigned n = min(rows_per_wave, rows_per_region - first);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<f4*>(src) + (size_t)xcd * rows_per_region * 64, 0,
                                                                      (int)(rows_per_region * 1024u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dst + (size_t)xcd * rows_per_region * 32, 0,
                                                                      (int)(rows_per_region * 512u), 0x00020000);
  unsigned so = first / 2 * 1024u;
  for (unsigned i = 0; i + 8 <= n; i += 8) {
    u4v v[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      const unsigned row = __builtin_amdgcn_readfirstlane(ord[i + d]) - xcd * rows_per_region;
      v[d] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(lane * 16u), (int)(row * 1024u), LAUX);
    }
#pragma unroll
    for (int d = 0; d < 8; d += 2) {
      u4v o = v[d] ^ v[d + 1];
      __builtin_amdgcn_raw_buffer_store_b128(o, rd, (int)(lane * 16u), (int)so, SAUX);
      so += 1024u;
    }
  }
}

int main(int argc, char** argv) {
  const size_t region_bytes = 91ull << 20;                 // one image's pyramid (P2..P5 x 256 x 4 B) ~ 91 MB
  const size_t bytes = 8 * region_bytes;
  const unsigned rows_per_region = (unsigned)(region_bytes / 1024);
  f4 *a, *b; float* sink; unsigned* order; char* flush;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&flush, 1ull << 30));
  CK(hipMalloc(&order, 8ull * rows_per_region * 4));
  CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  // flush_mode 0: a 1 GiB WRITE between launches (the caches are left full of DIRTY lines, which the measured launch
  // has to evict); 1: a 1 GiB nontemporal-free READ of another buffer (clean lines); 2: nothing (back to back)
  int flush_mode = argc > 1 ? atoi(argv[1]) : 0;
  printf("flush mode %d\n", flush_mode);
  auto timeit = [&](const char* name, double gb, auto launch) {
    float best = 1e9f, sum = 0; const int reps = 6;
    for (int r = 0; r < reps; ++r) {
      if (flush_mode == 0) CK(hipMemsetAsync(flush, r, 1ull << 30, 0));
      if (flush_mode == 1) hipLaunchKernelGGL((k_read_gs<8, false>), dim3(4096), dim3(256), 0, 0, (const f4*)flush, (1ull << 30) / 16, sink);
      CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r > 0) { best = std::min(best, ms); sum += ms; }
    }
    printf("%-46s %8.1f us mean  %8.1f us min  %7.0f GB/s (mean)\n", name, sum / (reps - 1) * 1e3, best * 1e3, gb / (sum / (reps - 1) * 1e-3));
  };
  const double GB = bytes / 1e9;
  const int wgs = 6656;                                    // 7 x 1024 vectors per workgroup, exactly
  const size_t n_per_wg = bytes / 16 / wgs;
  if (n_per_wg * wgs * 16 != bytes || n_per_wg % 1024 != 0) { printf("slice sizes do not divide\n"); return 1; }
  timeit("read_seq", GB, [&] { hipLaunchKernelGGL(k_read_seq, dim3(wgs), dim3(256), 0, 0, a, n_per_wg, sink); });
  for (int g : {1024, 2048, 4096, 16384}) {
    char nm[96];
    snprintf(nm, sizeof nm, "read grid-stride depth 4, %d workgroups", g);
    timeit(nm, GB, [&] { hipLaunchKernelGGL((k_read_gs<4, false>), dim3(g), dim3(256), 0, 0, a, bytes / 16, sink); });
    snprintf(nm, sizeof nm, "read grid-stride depth 8, %d workgroups", g);
    timeit(nm, GB, [&] { hipLaunchKernelGGL((k_read_gs<8, false>), dim3(g), dim3(256), 0, 0, a, bytes / 16, sink); });
  }
  timeit("read grid-stride depth 8 nontemporal, 4096", GB, [&] { hipLaunchKernelGGL((k_read_gs<8, true>), dim3(4096), dim3(256), 0, 0, a, bytes / 16, sink); });
  timeit("write_seq (nt)", GB, [&] { hipLaunchKernelGGL(k_write_seq, dim3(wgs), dim3(256), 0, 0, b, n_per_wg); });
  timeit("copy (bytes read + written)", 2 * GB, [&] { hipLaunchKernelGGL(k_copy, dim3(wgs), dim3(256), 0, 0, a, b, n_per_wg); });
  std::vector<unsigned> ord(8ull * rows_per_region);
  std::mt19937 rng(1);
  for (int mode = 0; mode < 3; ++mode) {
    // mode 0: sequential rows; 1: fully permuted inside the region; 2: permuted inside 4 MB windows (a band of the map)
    for (unsigned x = 0; x < 8; ++x) {
      unsigned* o = ord.data() + (size_t)x * rows_per_region;
      for (unsigned i = 0; i < rows_per_region; ++i) o[i] = x * rows_per_region + i;
      if (mode == 1) std::shuffle(o, o + rows_per_region, rng);
      if (mode == 2) for (unsigned s = 0; s < rows_per_region; s += 4096) std::shuffle(o + s, o + std::min(rows_per_region, s + 4096), rng);
    }
    CK(hipMemcpy(order, ord.data(), ord.size() * 4, hipMemcpyHostToDevice));
    const char* mn[3] = {"sequential", "permuted in region", "permuted in 4 MB windows"};
    char name[128];
    const unsigned rpw = 56;                                // rows per wave (~ one output row's 7 bins x 8 cells)
    const unsigned groups = (rows_per_region + 7 * rpw - 1) / (7 * rpw);
    snprintf(name, sizeof name, "read_rows depth 8, %s", mn[mode]);
    timeit(name, GB, [&] { hipLaunchKernelGGL((k_read_rows<8, 0>), dim3(groups * 8), dim3(448), 0, 0, a, order, rows_per_region, rpw, b, sink); });
    snprintf(name, sizeof name, "read_rows depth 4, %s", mn[mode]);
    timeit(name, GB, [&] { hipLaunchKernelGGL((k_read_rows<4, 0>), dim3(groups * 8), dim3(448), 0, 0, a, order, rows_per_region, rpw, b, sink); });
    if (mode == 2) {
#define MIX(LA, SA) snprintf(name, sizeof name, "mix 2:1 buffer ops, load aux %d, store aux %d", LA, SA); \
      timeit(name, GB * 1.5, [&] { hipLaunchKernelGGL((k_mix<LA, SA>), dim3(groups * 8), dim3(448), 0, 0, a, order, rows_per_region, rpw, b); });
      MIX(0, 0) MIX(0, 2) MIX(0, 1) MIX(0, 16) MIX(0, 17) MIX(0, 18) MIX(0, 19) MIX(0, 3)
      MIX(2, 2) MIX(2, 0) MIX(16, 2) MIX(1, 2) MIX(17, 2) MIX(17, 17) MIX(2, 17) MIX(2, 19)
    }
    snprintf(name, sizeof name, "read_rows depth 8 + 1 KB store / 2 rows, %s", mn[mode]);
    timeit(name, GB * 1.5, [&] { hipLaunchKernelGGL((k_read_rows<2, 2>), dim3(groups * 8), dim3(448), 0, 0, a, order, rows_per_region, rpw, b, sink); });
  }
  return 0;
}
