// Does VGPR relative addressing (s_set_gpr_idx_on, M0) apply to the accumulator operands of v_mfma_f32_16x16x4_f32 on gfx950?
// (diagnostic for splat D's class dispatch)  Prints the 8 tiles after accumulating into tile `sel`.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x32 __attribute__((ext_vector_type(32)));

__global__ void probe(float* out, int sel) {
  float a = 1.0f, b = 1.0f;
  int off = 4 * sel;
  float r[8];
  // tiles in v[32:63] (fixed registers, clobbered); dst / src2 relative: mode bits 1 = src0, 2 = src1, 4 = src2, 8 = dst
  asm volatile(
      "v_mov_b32 v32, 0\n\tv_mov_b32 v33, 0\n\tv_mov_b32 v34, 0\n\tv_mov_b32 v35, 0\n\t"
      "v_mov_b32 v36, 0\n\tv_mov_b32 v37, 0\n\tv_mov_b32 v38, 0\n\tv_mov_b32 v39, 0\n\t"
      "v_mov_b32 v40, 0\n\tv_mov_b32 v41, 0\n\tv_mov_b32 v42, 0\n\tv_mov_b32 v43, 0\n\t"
      "v_mov_b32 v44, 0\n\tv_mov_b32 v45, 0\n\tv_mov_b32 v46, 0\n\tv_mov_b32 v47, 0\n\t"
      "v_mov_b32 v48, 0\n\tv_mov_b32 v49, 0\n\tv_mov_b32 v50, 0\n\tv_mov_b32 v51, 0\n\t"
      "v_mov_b32 v52, 0\n\tv_mov_b32 v53, 0\n\tv_mov_b32 v54, 0\n\tv_mov_b32 v55, 0\n\t"
      "v_mov_b32 v56, 0\n\tv_mov_b32 v57, 0\n\tv_mov_b32 v58, 0\n\tv_mov_b32 v59, 0\n\t"
      "v_mov_b32 v60, 0\n\tv_mov_b32 v61, 0\n\tv_mov_b32 v62, 0\n\tv_mov_b32 v63, 0\n\t"
      "s_nop 4\n\t"
      "s_set_gpr_idx_on %10, 0xc\n\t"
      "v_mfma_f32_16x16x4_f32 v[32:35], %8, %9, v[32:35]\n\t"
      "s_set_gpr_idx_off\n\t"
      "s_nop 7\n\ts_nop 7\n\t"
      "v_mov_b32 %0, v32\n\tv_mov_b32 %1, v36\n\tv_mov_b32 %2, v40\n\tv_mov_b32 %3, v44\n\t"
      "v_mov_b32 %4, v48\n\tv_mov_b32 %5, v52\n\tv_mov_b32 %6, v56\n\tv_mov_b32 %7, v60\n\t"
      : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7])
      : "v"(a), "v"(b), "s"(off)
      : "memory", "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47",
        "v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63");
  if (threadIdx.x == 0)
    for (int i = 0; i < 8; ++i) out[i] = r[i];
}

int main() {
  float* d; hipMalloc(&d, 32 * 4);
  for (int sel : {0, 3, 7}) {
    probe<<<1, 64>>>(d, sel);
    float h[8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("sel %d:", sel);
    for (int i = 0; i < 8; ++i) printf(" %g", h[i]);
    printf("\n");
  }
  return 0;
}
