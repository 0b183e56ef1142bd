// Memory-pipeline calibration for bench.py's roofline object (measurement infrastructure, not a stage of the path):
// a kernel that does NOTHING but move the RoI launch's bytes -- it reads `read_bytes` of `src` once, in 1 KB rows
// (= one tap of 256 float32 channels, the access unit of k_roi_pool), XCD x reading only the x-th eighth (the
// image -> XCD pinning of the batched RoI launch), and writes `write_bytes` of `dst` with the same nontemporal
// 16-byte-per-lane stores, the stores interleaved with the loads at the same byte ratio.  Its duration, measured
// next to the RoI kernel's under the same cache conditions, is the time the memory system of THIS box needs for
// that read : write mix with the same instructions and cache policy: on MI355X mixed traffic runs well below the
// 8 TB/s of the data sheet (tools/exp/membw.hip, profiles/r02_membw_calibration.txt), so "fraction of the calibrated
// rate" says how much of the kernel's time is the kernel's fault.
#include <hip/hip_ext.h>

#include "odet_internal.h"

typedef unsigned cal_u4 __attribute__((ext_vector_type(4)));

struct CalibParams {
  const char* src; char* dst;
  unsigned rows_read_per_xcd, rows_written_per_xcd;   // 1 KB rows
  unsigned rows_read_per_wave;
};

__global__ void __launch_bounds__(448) k_calib_stream_mix(CalibParams p) {
  const unsigned xcd = blockIdx.x & 7, g = blockIdx.x >> 3;
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned first = (g * 7 + wave) * p.rows_read_per_wave;
  if (first >= p.rows_read_per_xcd) return;
  const unsigned n = min(p.rows_read_per_wave, p.rows_read_per_xcd - first);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(p.src) + (size_t)xcd * p.rows_read_per_xcd * 1024u, 0, (int)(p.rows_read_per_xcd * 1024u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
      p.dst + (size_t)xcd * p.rows_written_per_xcd * 1024u, 0, (int)(p.rows_written_per_xcd * 1024u), 0x00020000);
  // this wave's share of the written rows: [w0, w1)
  const unsigned long long R = p.rows_read_per_xcd, Wn = p.rows_written_per_xcd;
  unsigned w = (unsigned)((unsigned long long)first * Wn / R);
  cal_u4 acc = {0u, 0u, 0u, 0u};
  for (unsigned i = 0; i < n; i += 8) {
    cal_u4 v[8];
#pragma unroll
    for (int d = 0; d < 8; ++d)      // (rows beyond the region read as zero: the descriptor's bounds check)
      v[d] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(lane * 16u), (int)((first + i + d) * 1024u), 0);
#pragma unroll
    for (int d = 0; d < 8; ++d) acc ^= v[d];
    const unsigned w_end = (unsigned)((unsigned long long)(first + min(i + 8, n)) * Wn / R);
    for (; w < w_end; ++w) __builtin_amdgcn_raw_buffer_store_b128(acc, rd, (int)(lane * 16u), (int)(w * 1024u), /*nt*/ 2);
  }
}

extern "C" int odet_calib_stream_mix(const void* src, unsigned long long read_bytes, void* dst,
                                     unsigned long long write_bytes, odet_stream_t stream, void* start_event,
                                     void* stop_event) {
  ODET_REQUIRE(src && dst, "odet_calib_stream_mix: null pointer");
  ODET_REQUIRE(read_bytes >= 8192 && read_bytes % 8192 == 0 && write_bytes % 8192 == 0,
               "odet_calib_stream_mix: byte counts must be multiples of 8 KiB (8 XCD regions of 1 KB rows)");
  ODET_REQUIRE(read_bytes / 8 < 0x7FFFFFFFull && write_bytes / 8 < 0x7FFFFFFFull, "odet_calib_stream_mix: region above 2 GiB");
  CalibParams p;
  p.src = (const char*)src; p.dst = (char*)dst;
  p.rows_read_per_xcd = (unsigned)(read_bytes / 8 / 1024);
  p.rows_written_per_xcd = (unsigned)(write_bytes / 8 / 1024);
  p.rows_read_per_wave = 56;                 // ~ one output row of k_roi_pool: 7 bins x 8 cell loads
  const unsigned groups = (p.rows_read_per_xcd + 7 * p.rows_read_per_wave - 1) / (7 * p.rows_read_per_wave);
  hipExtLaunchKernelGGL(k_calib_stream_mix, dim3(groups * 8), dim3(448), 0, (hipStream_t)stream, (hipEvent_t)start_event,
                        (hipEvent_t)stop_event, 0, p);
  ODET_LAUNCH_CHECK();
  return ODET_OK;
}

// Counter calibration (tools/pmc_roi_forms.sh): reads `bytes` of src exactly once, BPL bytes per lane (16: the float32
// RoI kernel's cell loads; 8: the float16 one's), one 64 * BPL-byte row per wave instruction, rows in a permuted order --
// a known byte count in the RoI kernels' access shape, so that FETCH_SIZE's gfx950 factor (MI355X guide, HBM section:
// exactly 1/2 for 16-byte lanes, other widths uncalibrated) is MEASURED for both widths in the same profiler pass.
template <int BPL>
__global__ void __launch_bounds__(256) k_calib_read(const char* src, unsigned long long rows, unsigned* sink) {
  const unsigned lane = threadIdx.x & 63;
  const unsigned long long wave = (unsigned long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const unsigned long long nw = (unsigned long long)gridDim.x * 4;
  unsigned acc = 0;
  for (unsigned long long r = wave; r < rows; r += nw) {
    const unsigned long long row = (r * 2654435761ull) % rows;          // (rows is odd: a permutation)
    const char* ptr = src + row * (64ull * BPL) + lane * BPL;
    if (BPL == 16) {
      const cal_u4 v = *reinterpret_cast<const cal_u4*>(ptr);
      acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
    } else {
      const uint2 v = *reinterpret_cast<const uint2*>(ptr);
      acc ^= v.x ^ v.y;
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;                                 // (keeps the loads alive)
}

extern "C" int odet_calib_read_rows(const void* src, unsigned long long bytes, int bytes_per_lane, void* sink,
                                    odet_stream_t stream) {
  ODET_REQUIRE(src && sink && (bytes_per_lane == 8 || bytes_per_lane == 16), "odet_calib_read_rows: bad arguments");
  unsigned long long rows = bytes / (64ull * bytes_per_lane);
  if (rows % 2 == 0) rows -= 1;
  ODET_REQUIRE(rows > 0, "odet_calib_read_rows: too few bytes");
  if (bytes_per_lane == 16)
    hipLaunchKernelGGL(k_calib_read<16>, dim3(256 * 16), dim3(256), 0, (hipStream_t)stream, (const char*)src, rows, (unsigned*)sink);
  else
    hipLaunchKernelGGL(k_calib_read<8>, dim3(256 * 16), dim3(256), 0, (hipStream_t)stream, (const char*)src, rows, (unsigned*)sink);
  ODET_LAUNCH_CHECK();
  return ODET_OK;
}
