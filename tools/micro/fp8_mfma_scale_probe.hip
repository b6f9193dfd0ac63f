// Which (lane, byte) of the scale operands of v_mfma_scale_f32_16x16x128_f8f6f4 scales which (row / column, k block)?
// All data = 1.0 (e4m3 0x38), all scales 1.0 (E8M0 0x7F) except ONE byte of ONE lane = 2.0 (0x80); D is 128 everywhere except
// where that scale acts (+32 per affected 32-deep k block).  Prints, per (operand, op_sel, lane, byte), what changed.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/fp8_mfma_scale_probe.hip -o tools/micro/_bin/fp8_mfma_scale_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int OPSEL, int WHICH>
__global__ void k(float* D, int L, int bp) {
  const int l = threadIdx.x;
  v8i ones;
  for (int q = 0; q < 8; ++q) ones[q] = 0x38383838;
  int sc = 0x7F7F7F7F;
  if (l == L) sc = (sc & ~(0xFF << (8 * bp))) | (0x80 << (8 * bp));
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  if (WHICH == 0) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ones, ones, acc, 0, 0, OPSEL, sc, 0, 0x7F7F7F7F);
  else acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ones, ones, acc, 0, 0, 0, 0x7F7F7F7F, OPSEL, sc);
  for (int q = 0; q < 4; ++q) D[(4 * (l >> 4) + q) * 16 + (l & 15)] = acc[q];
}

template <int OPSEL, int WHICH>
void run(float* dd) {
  for (int bp = 0; bp < 4; ++bp)
    for (int L = 0; L < 64; ++L) {
      hipLaunchKernelGGL((k<OPSEL, WHICH>), dim3(1), dim3(64), 0, 0, dd, L, bp);
      float hd[256]; hipMemcpy(hd, dd, sizeof hd, hipMemcpyDeviceToHost);
      int nrow = 0, ncol = 0, cnt = 0; float delta = 0; int r0 = -1, c0 = -1;
      bool rows[16] = {}, cols[16] = {};
      for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) if (hd[i * 16 + j] != 128.f) { ++cnt; rows[i] = cols[j] = true; delta = hd[i * 16 + j] - 128.f; }
      for (int i = 0; i < 16; ++i) { if (rows[i]) { ++nrow; r0 = i; } if (cols[i]) { ++ncol; c0 = i; } }
      if (cnt) printf("operand %d op_sel %d byte %d lane %2d: %3d entries changed by %+g : %s %d\n", WHICH, OPSEL, bp, L, cnt, delta,
                      nrow == 1 ? "ROW" : ncol == 1 ? "COL" : "??", nrow == 1 ? r0 : c0);
    }
}

int main() {
  float* dd; hipMalloc(&dd, 256 * 4);
  run<0, 0>(dd); run<1, 0>(dd); run<2, 0>(dd); run<3, 0>(dd);
  run<0, 1>(dd); run<1, 1>(dd);
  return 0;
}
