// Experiment: how busy can the matrix pipe get from LDS-resident operands, by wave-tile shape?
//   variant A: 8 waves per workgroup, a wave = 8 x 4 tiles of 16 x 16 (the conv3x3 kernel's shape: 12 fragment reads per 32 MFMAs)
//   variant B: 4 waves per workgroup, a wave = 8 x 8 tiles (8 reads per 32 MFMAs, one wave per SIMD, accumulators in AGPRs)
// No global traffic in the loop; the LDS image is never rewritten (no DMA): the ceiling of the inner loop alone.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_loop_probe tools/exp/mfma_loop_probe.hip && ./mfma_loop_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

typedef __attribute__((address_space(3))) void* lds_ptr;
// DMA = 1: every K-step the workgroup also copies a fresh 64 KB stage (256 + 256 rows of 128 B) global -> LDS with LDS-DMA
// from an L2-resident buffer, into the stage not being read, and waits for it before the barrier (the real kernel's traffic)
template <int WAVES, int MT, int NT, int DMA>
__global__ void __launch_bounds__(WAVES * 64) k_probe(float* out, int ksteps, const char* src) {
  extern __shared__ __align__(16) char lds[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  // fill LDS (64 KB stage: 512 rows of 128 B) with something finite
  for (int i = threadIdx.x; i < 65536 / 16; i += WAVES * 64) {
    h8 v;
    for (int e = 0; e < 8; ++e) v[e] = (_Float16)(0.001f * (float)((i + e) & 31));
    *reinterpret_cast<h8*>(lds + i * 16) = v;
  }
  __syncthreads();
  f4 acc[MT][NT];
  for (int m = 0; m < MT; ++m) for (int n = 0; n < NT; ++n) acc[m][n] = (f4){0.f, 0.f, 0.f, 0.f};
  // rows: A tiles (pixels) from rows [0, 256), B tiles (channels) from rows [256, 512); 128-byte rows, XOR-swizzled slots
  const uint32_t arow = (uint32_t)((wv % (WAVES == 8 ? 2 : 2)) * 16 * MT + l15);
  const uint32_t brow = 256u + (uint32_t)((wv / 2) * 16 * NT % 256 + l15);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, 1 << 22, 0x00020000);
  const uint32_t voff = (uint32_t)(blockIdx.x & 15) * 65536u + (uint32_t)lane * 16u;
  for (int ks = 0; ks < ksteps; ++ks) {
    const uint32_t cur = DMA ? (uint32_t)(ks & 1) * 65536u : 0u;
    if (DMA) {
#pragma unroll
      for (int i = 0; i < 64 / WAVES; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(lds + (65536u - cur) + (uint32_t)(wv + WAVES * i) * 1024u), 16,
                                                 (int)(voff + (uint32_t)(wv + WAVES * i) * 1024u), (int)((uint32_t)(ks & 31) * 128u), 0, 0);
    }
    const char* sb = lds + cur;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      h8 af[MT], bf[NT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const uint32_t r = arow + 16u * m;
        af[m] = *reinterpret_cast<const h8*>(sb + r * 128u + ((((uint32_t)(kk * 4 + lq)) ^ (r & 7u)) * 16u));
      }
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const uint32_t r = brow + 16u * n;
        bf[n] = *reinterpret_cast<const h8*>(sb + (r & 511u) * 128u + ((((uint32_t)(kk * 4 + lq)) ^ (r & 7u)) * 16u));
      }
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[n], af[m], acc[m][n], 0, 0, 0);
    }
    if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                        // the real kernel has one barrier per K-step
  }
  float s = 0.f;
  for (int m = 0; m < MT; ++m) for (int n = 0; n < NT; ++n) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
  if (s == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int WAVES, int MT, int NT, int DMA>
void run(const char* name, int lds_bytes) {
  float* out; hipMalloc(&out, 1 << 24);
  char* src; hipMalloc(&src, 1 << 23); hipMemset(src, 0, 1 << 23);
  const int grid = 256 * 4, ksteps = 2000;
  hipFuncSetAttribute((const void*)k_probe<WAVES, MT, NT, DMA>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL((k_probe<WAVES, MT, NT, DMA>), dim3(grid), dim3(WAVES * 64), lds_bytes, 0, out, ksteps, src);
    hipEventRecord(b); hipEventSynchronize(b);
  }
  float ms; hipEventElapsedTime(&ms, a, b);
  const double flops = (double)grid * WAVES * MT * NT * 2.0 * ksteps * 16384.0;
  printf("%-44s %8.3f ms  %7.1f TFLOP/s  (lds %d KB per workgroup)\n", name, ms, flops / ms / 1e9, lds_bytes >> 10);
  hipFree(out); hipFree(src);
}

int main() {
  run<8, 8, 4, 0>("A: 8 waves x (8 x 4 tiles), no copies", 128 << 10);
  run<8, 8, 4, 1>("A+DMA: the same + 64 KB of LDS-DMA per K-step", 128 << 10);
  run<4, 8, 8, 0>("B: 4 waves x (8 x 8 tiles), no copies", 128 << 10);
  run<4, 8, 8, 1>("B+DMA", 128 << 10);
  return 0;
}
