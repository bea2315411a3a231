// Do fp32 / bf16 matrix instructions of one wave overlap with the VALU work of OTHER waves on the same SIMD? (diagnostic,
// not product code).  One 1024-thread workgroup per CU (16 waves, 4 per SIMD, wave w on SIMD w % 4 or
// w / 4 -- both splits are timed); `mfma_waves` of the 4 waves of every SIMD run a chain of matrix instructions, the others
// a chain of dependent-free v_fma_f32.  Prints kernel time for: matrix only, VALU only, both.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

template <int KIND>  // 0: f32 32x32x2, 1: bf16 32x32x16, 2: f32 16x16x4
__global__ __launch_bounds__(1024, 1) void mix(float* out, int iters, int mfma_mask, int do_mfma, int do_valu, int split, int mode) {
  const int wave = threadIdx.x >> 6;
  const int slot = split ? (wave >> 2) : (wave & 3);  // position of the wave among the 4 of its SIMD (under either mapping)
  const bool is_mfma = (mfma_mask >> slot) & 1;
  float s = 0;
  if (mode == 4) {  // every wave: matrix instructions with independent VALU work of the SAME wave between them
    f16v a0, a1;
    for (int r = 0; r < 16; ++r) { a0[r] = 0; a1[r] = 0; }
    float a = threadIdx.x * 0.5f, b = threadIdx.x * 0.25f;
    bf8 ab, bb;
    for (int r = 0; r < 8; ++r) { ab[r] = (__bf16)(float)(threadIdx.x + r); bb[r] = (__bf16)(float)(r); }
    float x[8];
    for (int r = 0; r < 8; ++r) x[r] = threadIdx.x + r;
    const float m = 1.0001f, c = 0.5f;
    for (int it = 0; it < iters; ++it) {
      if (do_mfma) { if (KIND == 0) a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, a0, 0, 0, 0); else { a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, a1, 0, 0, 0);} }
      if (do_valu) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 8; ++r) x[r] = __builtin_fmaf(x[r], m, c);
      }
      if (do_mfma) { if (KIND == 0) a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, a1, 0, 0, 0); else { a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, a1, 0, 0, 0);} }
      if (do_valu) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 8; ++r) x[r] = __builtin_fmaf(x[r], m, c);
      }
    }
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r];
    for (int r = 0; r < 8; ++r) s += x[r];
    out[blockIdx.x * 1024 + threadIdx.x] = s;
    return;
  }
  if (is_mfma) {
    if (!do_mfma) return;
    f16v a0, a1;
    for (int r = 0; r < 16; ++r) { a0[r] = 0; a1[r] = 0; }
    float a = threadIdx.x * 0.5f, b = threadIdx.x * 0.25f;
    bf8 ab, bb;
    for (int r = 0; r < 8; ++r) { ab[r] = (__bf16)(float)(threadIdx.x + r); bb[r] = (__bf16)(float)(r); }
    f4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    if (mode == 1) {
      for (int it = 0; it < iters; ++it) {  // ONE dependent chain: the next instruction waits for its accumulator, not for the pipe
        if (KIND == 0) { a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, a0, 0, 0, 0); a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, a0, 0, 0, 0); }
        else { for (int q = 0; q < 4; ++q) a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, a0, 0, 0, 0); }
      }
    } else if (mode == 2) {
      for (int it = 0; it < iters; ++it) {  // two chains, the wave idles ~48 clocks after each instruction
        if (KIND == 0) {
          a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, a0, 0, 0, 0); asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15");
          a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, a1, 0, 0, 0); asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15");
        } else { for (int q = 0; q < 2; ++q) { a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, a0, 0, 0, 0); asm volatile("s_nop 15"); a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, a1, 0, 0, 0); asm volatile("s_nop 15"); } }
      }
    } else
    for (int it = 0; it < iters; ++it) {
      if (KIND == 0) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, a1, 0, 0, 0);
      } else if (KIND == 1) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, a1, 0, 0, 0);
      } else {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
      }
    }
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r];
    s += c0[0] + c1[0];
  } else {
    if (!do_valu) return;
    if (mode == 3) __builtin_amdgcn_s_setprio(3);
    if (mode == 5) {
      unsigned y[8];
      for (int r = 0; r < 8; ++r) y[r] = threadIdx.x + r;
      unsigned k1 = 0x9e3779b9u + threadIdx.x;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int r = 0; r < 8; ++r) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(y[r]) : "v"(k1));
      }
      for (int r = 0; r < 8; ++r) s += (float)y[r];
      out[blockIdx.x * 1024 + threadIdx.x] = s;
      return;
    }
    float x[8];
    for (int r = 0; r < 8; ++r) x[r] = threadIdx.x + r;
    const float m = 1.0001f, c = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int r = 0; r < 8; ++r) x[r] = __builtin_fmaf(x[r], m, c);  // 32 independent-enough v_fma_f32 per iteration
    }
    for (int r = 0; r < 8; ++r) s += x[r];
  }
  out[blockIdx.x * 1024 + threadIdx.x] = s;
}

template <int KIND>
static float run(float* out, int iters, int mask, int dm, int dv, int split, int mode) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  mix<KIND><<<256, 1024>>>(out, iters, mask, dm, dv, split, mode);
  hipEventRecord(a);
  mix<KIND><<<256, 1024>>>(out, iters, mask, dm, dv, split, mode);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms;
}

int main() {
  float* out; hipMalloc(&out, 256 * 1024 * 4);
  const int iters = 20000;
  const char* names[3] = {"f32 32x32x2 ", "bf16 32x32x16", "f32 16x16x4 "};
  const char* modes[6] = {"two chains", "one dependent chain", "two chains + s_nop", "VALU waves at prio 3", "SAME wave: 2 matrix + 32 VALU per iteration, 16 waves", "two chains, VALU waves run v_xor_b32"};
  for (int mode = 5; mode < 6; ++mode)
    for (int mask : {0x3, 0x1}) {
      if (mode == 4 && mask == 0x1) continue;
      printf("%s, matrix waves per SIMD mask 0x%x\n", modes[mode], mask);
      for (int k = 0; k < 2; ++k) {
        float m, v, both;
        if (k == 0) { m = run<0>(out, iters, mask, 1, 0, 1, mode); v = run<0>(out, iters, mask, 0, 1, 1, mode); both = run<0>(out, iters, mask, 1, 1, 1, mode); }
        else { m = run<1>(out, iters, mask, 1, 0, 1, mode); v = run<1>(out, iters, mask, 0, 1, 1, mode); both = run<1>(out, iters, mask, 1, 1, 1, mode); }
        printf("  %s  matrix only %7.3f ms   VALU only %7.3f ms   both %7.3f ms   (sum %7.3f, max %7.3f)\n", names[k], m, v, both, m + v, m > v ? m : v);
      }
    }
  return 0;
}
