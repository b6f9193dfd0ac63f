// Which (lane, byte) holds which (row, k) of the operands of v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 data?
// Exact small-integer data, two candidate maps, both operand roles; prints the combination that reproduces X * Y^T.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/fp8_mfma_layout.hip -o tools/micro/_bin/fp8_mfma_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

static unsigned char enc_e4m3(float v) {                 // exact for the small integers used here
  if (v == 0.f) return 0;
  unsigned char s = v < 0 ? 0x80 : 0; v = fabsf(v);
  int e; float m = frexpf(v, &e);                        // v = m * 2^e, m in [0.5, 1)
  int ee = e - 1 + 7; int mant = (int)roundf((m * 2.f - 1.f) * 8.f);
  if (mant == 8) { mant = 0; ++ee; }
  return s | (unsigned char)(ee << 3) | (unsigned char)mant;
}

__global__ void k(const unsigned char* X, const unsigned char* Y, float* D, int map) {
  const int l = threadIdx.x, r = l & 15, g = l >> 4;
  unsigned char xa[32], ya[32];
  for (int j = 0; j < 32; ++j) {
    const int kk = map == 0 ? 32 * g + j : 16 * g + (j & 15) + 64 * (j >> 4);
    xa[j] = X[r * 128 + kk]; ya[j] = Y[r * 128 + kk];
  }
  v8i xv, yv;
  for (int q = 0; q < 8; ++q) {
    xv[q] = xa[4 * q] | (xa[4 * q + 1] << 8) | (xa[4 * q + 2] << 16) | (xa[4 * q + 3] << 24);
    yv[q] = ya[4 * q] | (ya[4 * q + 1] << 8) | (ya[4 * q + 2] << 16) | (ya[4 * q + 3] << 24);
  }
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(xv, yv, acc, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
  for (int q = 0; q < 4; ++q) D[(4 * g + q) * 16 + r] = acc[q];     // assume row = 4g + reg, col = lane & 15
}

int main() {
  unsigned char hx[16 * 128], hy[16 * 128]; float fx[16 * 128], fy[16 * 128];
  srand(3);
  for (int i = 0; i < 16 * 128; ++i) { fx[i] = (float)(rand() % 9 - 4); fy[i] = (float)(rand() % 7 - 3); hx[i] = enc_e4m3(fx[i]); hy[i] = enc_e4m3(fy[i]); }
  unsigned char *dx, *dy; float* dd;
  hipMalloc(&dx, sizeof hx); hipMalloc(&dy, sizeof hy); hipMalloc(&dd, 256 * 4);
  hipMemcpy(dx, hx, sizeof hx, hipMemcpyHostToDevice); hipMemcpy(dy, hy, sizeof hy, hipMemcpyHostToDevice);
  for (int map = 0; map < 2; ++map) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, dy, dd, map);
    float hd[256]; hipMemcpy(hd, dd, sizeof hd, hipMemcpyDeviceToHost);
    int ok_xy = 1, ok_yx = 1;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
      float sxy = 0, syx = 0;
      for (int kk = 0; kk < 128; ++kk) { sxy += fx[i * 128 + kk] * fy[j * 128 + kk]; syx += fy[i * 128 + kk] * fx[j * 128 + kk]; }
      if (hd[i * 16 + j] != sxy) ok_xy = 0;
      if (hd[i * 16 + j] != syx) ok_yx = 0;
    }
    printf("map %d (%s): D[row][col] == sum_k first[row][k]*second[col][k]: %s ; == sum_k second[row][k]*first[col][k]: %s ; D[0][0]=%g D[1][0]=%g\n", map,
           map == 0 ? "k = 32*(lane>>4)+j" : "k = 16*(lane>>4)+(j&15)+64*(j>>4)", ok_xy ? "YES" : "no", ok_yx ? "YES" : "no", hd[0], hd[16]);
  }
  return 0;
}
