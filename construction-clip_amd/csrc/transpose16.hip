// Batched out-of-place transpose of 16-bit matrices (bf16 / fp16 alike: elements are moved, never interpreted).
//
// Why it exists: the backward of an nn.Linear layer needs dX = dY . W with W stored [out, in] - for the GEMM that is a
// K-STRIDED B operand (contraction over `out`), read from LDS through ds_read_b64_tr_b16 at twice the LDS instructions of the
// forward layout.  The weights are tiny next to the activations (ViT-B/32: 85 M elements against 51 200 x 768 activations PER
// GEMM), so the arena keeps a second 16-bit shadow holding every block weight TRANSPOSED ([in, out]); it is rebuilt once per
// optimiser step by this one launch (~340 MB of traffic, HBM-bound) and every dgrad GEMM then runs in the forward layout
// - which also opens the persistent streamed-epilogue configuration to it.
//
// One workgroup per 64 x 64 tile; grid.y = matrix index, grid.x = the largest tile count (surplus workgroups exit).
// table[i] = {src element offset, dst element offset, rows, cols} (int64, device memory); dst is [cols, rows].
#include "cclip_common.h"
#include "../../include/cclip_hip.h"

__global__ __launch_bounds__(256) void transpose16_batched_kernel(const unsigned short* __restrict__ src_base,
                                                                  unsigned short* __restrict__ dst_base,
                                                                  const long* __restrict__ table) {
  __shared__ unsigned short tile[64][66];
  const long* e = table + 4L * blockIdx.y;
  const long R = e[2], C = e[3];
  const int tiles_c = (int)((C + 63) / 64), tiles_r = (int)((R + 63) / 64);
  if ((int)blockIdx.x >= tiles_c * tiles_r) return;
  const unsigned short* src = src_base + e[0];
  unsigned short* dst = dst_base + e[1];
  const long r0 = 64L * (blockIdx.x / tiles_c), c0 = 64L * (blockIdx.x % tiles_c);
  const int t = threadIdx.x, row = t >> 2, seg = (t & 3) * 16;
  const bool fast = r0 + 64 <= R && c0 + 64 <= C && !(C & 7) && !(R & 7) && !(((uintptr_t)src | (uintptr_t)dst) & 15);
  if (fast) {
    const uint4* s = (const uint4*)(src + (r0 + row) * C + c0 + seg);
    const uint4 a = s[0], b = s[1];
    unsigned short v[16];
    *(uint4*)&v[0] = a; *(uint4*)&v[8] = b;
#pragma unroll
    for (int i = 0; i < 16; ++i) tile[row][seg + i] = v[i];
  } else {
    for (int i = 0; i < 16; ++i) {
      const long r = r0 + row, c = c0 + seg + i;
      tile[row][seg + i] = (r < R && c < C) ? src[r * C + c] : (unsigned short)0;
    }
  }
  __syncthreads();
  // output row = source column c0 + row; 16 consecutive source rows r0 + seg .. + 15 become 32 contiguous bytes
  if (fast) {
    unsigned short v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = tile[seg + i][row];
    uint4* o = (uint4*)(dst + (c0 + row) * R + r0 + seg);
    o[0] = *(uint4*)&v[0]; o[1] = *(uint4*)&v[8];
  } else {
    for (int i = 0; i < 16; ++i) {
      const long c = c0 + row, r = r0 + seg + i;
      if (c < C && r < R) dst[c * R + r] = tile[seg + i][row];
    }
  }
}

extern "C" int cclip_transpose16_batched(const void* src_base, void* dst_base, const int64_t* table_dev, int32_t n_matrices,
                                         int32_t max_tiles, hipStream_t stream) {
  if (!src_base || !dst_base || !table_dev || n_matrices <= 0 || max_tiles <= 0) return CCLIP_ERR_ARG;
  hipLaunchKernelGGL(transpose16_batched_kernel, dim3(max_tiles, n_matrices), dim3(256), 0, stream,
                     (const unsigned short*)src_base, (unsigned short*)dst_base, (const long*)table_dev);
  return cclip_launch_status();
}
