// Micro-benchmark: does the GEMM epilogue's lane->address map (8-column runs per lane, 16 rows per instruction)
// cost HBM bandwidth against a row-contiguous map?  fp32 read-modify-write of [M, N] and 16-bit store of [M, N].
//   hipcc --offload-arch=gfx950 -O3 tools/micro/epi_pattern.hip -o tools/micro/_bin/epi_pattern && tools/micro/_bin/epi_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) unsigned short us8;

// one workgroup = 128x128 tile, 4 waves as 2x2 of 64x64; wave tile = 4 m-tiles x 2 halves of 8-col runs (as gemm_bf16_impl.h)
template <int MODE, int F32>
__global__ __launch_bounds__(256) void tile_kernel(float* __restrict__ x, unsigned short* __restrict__ y, int M, int N) {
  const int tiles_n = N / 128;
  const int bm0 = (blockIdx.x / tiles_n) * 128, bn0 = (blockIdx.x % tiles_n) * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (MODE == 0) {
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64, li = lane & 15, g = lane >> 4;
    float4 r[4][2][2];
    if (F32) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const float* p = x + (long)(bm0 + wm0 + 16 * mt + li) * N + bn0 + wn0 + 32 * h + 8 * g;
          r[mt][h][0] = *(const float4*)p; r[mt][h][1] = *(const float4*)(p + 4);
        }
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const long off = (long)(bm0 + wm0 + 16 * mt + li) * N + bn0 + wn0 + 32 * h + 8 * g;
        if (F32) {
          float4 a = r[mt][h][0], b = r[mt][h][1];
          a.x += 1.f; a.y += 1.f; a.z += 1.f; a.w += 1.f; b.x += 1.f; b.y += 1.f; b.z += 1.f; b.w += 1.f;
          *(float4*)(x + off) = a; *(float4*)(x + off + 4) = b;
        } else {
          us8 t; for (int i = 0; i < 8; ++i) t[i] = (unsigned short)(lane + i);
          *(us8*)(y + off) = t;
        }
      }
  } else {
    // row-contiguous: fp32: a wave covers 2 rows x 128 cols per instruction (float4 per lane); 16-bit: 4 rows x 128 cols (us8 per lane)
    if (F32) {
      float4 r[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = wave * 32 + i * 2 + (lane >> 5);
        r[i] = *(const float4*)(x + (long)(bm0 + row) * N + bn0 + (lane & 31) * 4);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = wave * 32 + i * 2 + (lane >> 5);
        float4 a = r[i]; a.x += 1.f; a.y += 1.f; a.z += 1.f; a.w += 1.f;
        *(float4*)(x + (long)(bm0 + row) * N + bn0 + (lane & 31) * 4) = a;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = wave * 32 + i * 4 + (lane >> 4);
        us8 t; for (int j = 0; j < 8; ++j) t[j] = (unsigned short)(lane + j);
        *(us8*)(y + (long)(bm0 + row) * N + bn0 + (lane & 15) * 8) = t;
      }
    }
  }
}

template <int MODE, int F32>
static float run(float* x, unsigned short* y, int M, int N) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = (M / 128) * (N / 128);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((tile_kernel<MODE, F32>), dim3(grid), dim3(256), 0, 0, x, y, M, N);
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((tile_kernel<MODE, F32>), dim3(grid), dim3(256), 0, 0, x, y, M, N);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 20;
}

int main() {
  const int M = 51200;
  for (int N : {768, 2304, 3072}) {
    float* x; unsigned short* y;
    hipMalloc(&x, (size_t)M * N * 4); hipMalloc(&y, (size_t)M * N * 2);
    hipMemset(x, 0, (size_t)M * N * 4);
    const float a = run<0, 1>(x, y, M, N), b = run<1, 1>(x, y, M, N), c = run<0, 0>(x, y, M, N), d = run<1, 0>(x, y, M, N);
    const double rmw = 2.0 * M * N * 4, st = 1.0 * M * N * 2;
    printf("N=%4d fp32 RMW: gemm-map %.1f us %.2f TB/s | row-map %.1f us %.2f TB/s || 16-bit store: gemm-map %.1f us %.2f TB/s | row-map %.1f us %.2f TB/s\n",
           N, a * 1e3, rmw / a / 1e9, b * 1e3, rmw / b / 1e9, c * 1e3, st / c / 1e9, d * 1e3, st / d / 1e9);
    hipFree(x); hipFree(y);
  }
  return 0;
}
