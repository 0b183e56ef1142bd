// v_cvt_pk_f16_f32 (gfx950) against two v_cvt_f16_f32 + pack: bit-for-bit over 2^32 float32 inputs?
//   hipcc --offload-arch=gfx950 -O3 tools/exp/cvt_pk_probe.hip -o /tmp/cvt_pk_probe && /tmp/cvt_pk_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned long long* bad, unsigned* first) {
  const unsigned long long n = 1ull << 32;
  for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
    const float x = __uint_as_float((unsigned)i), y = __uint_as_float((unsigned)(i * 2654435761ull));
    unsigned r;
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    h2 w; w[0] = (_Float16)x; w[1] = (_Float16)y;
    const unsigned want = __builtin_bit_cast(unsigned, w);
    if (r != want) { atomicAdd(bad, 1ull); atomicMin(first, (unsigned)i); }
  }
}
int main() {
  unsigned long long* bad; unsigned* first;
  hipMalloc(&bad, 8); hipMalloc(&first, 4);
  hipMemset(bad, 0, 8); hipMemset(first, 0xFF, 4);
  hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, bad, first);
  unsigned long long hb; unsigned hf;
  hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&hf, first, 4, hipMemcpyDeviceToHost);
  printf("mismatches over 2^32 inputs: %llu (first input bits 0x%08x)\n", hb, hf);
  return hb != 0;
}
