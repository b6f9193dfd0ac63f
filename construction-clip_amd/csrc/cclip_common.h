// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of the CLIP hot path.
// Wave = 64 lanes; MFMA 16x16x32 bf16; LDS 160 KiB/CU.  No other target is supported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// The 16-bit element type of GEMM / attention operands.  Every translation unit that touches 16-bit data is
// compiled twice: once with __bf16 (default, BASELINE configs[1]) and once with -DCCLIP_F16 for IEEE fp16
// (the reference's own CUDA dtype; same MFMA rate, 3 more mantissa bits -> the <= 1e-3 parity mode).
// The historical name `bf16` is kept for the element type inside the kernels; CCLIP_NS keeps the two
// builds' device code apart at link time and CCLIP_FN() names the exported twins.
#ifdef CCLIP_F16
typedef _Float16 bf16;
#define CCLIP_NS cclip_f16
#define CCLIP_FN(n) n##_f16
#define CCLIP_GEMM_FN cclip_gemm_f16
#define CCLIP_CAST_FN cclip_cast_f32_to_f16
#define CCLIP_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
typedef __attribute__((ext_vector_type(4))) __fp16 cclip_h4;
#define CCLIP_TR16(p) __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) cclip_h4*)(p)))
#else
typedef __bf16 bf16;
#define CCLIP_NS cclip_bf16
#define CCLIP_FN(n) n
#define CCLIP_GEMM_FN cclip_gemm_bf16
#define CCLIP_CAST_FN cclip_cast_f32_to_bf16
#define CCLIP_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define CCLIP_TR16(p) __builtin_amdgcn_ds_read_tr16_b64_v4bf16(p)
#endif
typedef __attribute__((ext_vector_type(8))) bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define CCLIP_OK 0
#define CCLIP_ERR_ARG 1      // shape / alignment contract violated
#define CCLIP_ERR_LAUNCH 2   // hipGetLastError() != hipSuccess after launch

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// 16-byte async global -> LDS copy (global_load_lds_dwordx4).  LDS destination is
// wave-uniform base + lane*16; the global source address is per lane.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(gsrc, LDS_PTR(lds_wave_base), 16, 0, 0);
}

// ds_read_b64_tr_b16: per 16-lane group a 4-row x 16-col block of 16-bit elements is returned
// column-major (lane i gets column i, row q in element q); lane 4q+p supplies the address of
// row q, columns 4p..4p+3.  EXEC must be all ones.
__device__ __forceinline__ bf16x4 lds_read_tr16(const void* lds_addr) {
  return CCLIP_TR16((__attribute__((address_space(3))) bf16x4*)(lds_addr));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Bijective XCD-aware block remap: blocks b and b+8 share an XCD (round-robin dispatch), so give
// each XCD a contiguous chunk of the logical tile order (neighbouring tiles share operand panels
// in that XCD's L2).  Speed only; any placement is correct.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (orig >> 3);
}

static inline int cclip_launch_status() {
  return hipGetLastError() == hipSuccess ? CCLIP_OK : CCLIP_ERR_LAUNCH;
}
