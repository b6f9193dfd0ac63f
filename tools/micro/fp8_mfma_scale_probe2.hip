// Follow-up of fp8_mfma_scale_probe.hip: WHICH 32 k-values does the scale byte of lane group gs act on?  The second operand is
// 1.0 only in ONE 16-byte half (registers 4h..4h+3) of ONE lane group gd and 0 elsewhere (first operand all ones): D = 16
// everywhere; the scale of lane group gs is 2.0.  D = 32 <=> group gs' scale acts on that slice.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/fp8_mfma_scale_probe2.hip -o tools/micro/_bin/fp8_mfma_scale_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int WHICH>
__global__ void k(float* D, int gs, int gd, int h) {
  const int l = threadIdx.x, g = l >> 4;
  v8i ones, sl;
  for (int q = 0; q < 8; ++q) { ones[q] = 0x38383838; sl[q] = (g == gd && (q >> 2) == h) ? 0x38383838 : 0; }
  const int sc = g == gs ? 0x80808080 : 0x7F7F7F7F;
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  if (WHICH == 0) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(sl, ones, acc, 0, 0, 0, sc, 0, 0x7F7F7F7F);
  else acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ones, sl, acc, 0, 0, 0, 0x7F7F7F7F, 0, sc);
  for (int q = 0; q < 4; ++q) D[(4 * g + q) * 16 + (l & 15)] = acc[q];
}

int main() {
  float* dd; (void)hipMalloc(&dd, 256 * 4);
  for (int which = 0; which < 2; ++which)
    for (int gd = 0; gd < 4; ++gd)
      for (int h = 0; h < 2; ++h) {
        printf("operand %d data slice (lane group %d, registers %d..%d): scaled by lane group", which, gd, 4 * h, 4 * h + 3);
        for (int gs = 0; gs < 4; ++gs) {
          if (which == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, dd, gs, gd, h);
          else hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, dd, gs, gd, h);
          float hd[256]; (void)hipMemcpy(hd, dd, sizeof hd, hipMemcpyDeviceToHost);
          bool all32 = true, all16 = true;
          for (int i = 0; i < 256; ++i) { all32 &= hd[i] == 32.f; all16 &= hd[i] == 16.f; }
          if (all32) printf(" %d", gs);
          else if (!all16) printf(" (%d: mixed, D[0]=%g)", gs, hd[0]);
        }
        printf("\n");
      }
  return 0;
}
