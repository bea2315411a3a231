// Layout + issue-rate probe for the multi-block f32 MFMA forms on gfx950 (diagnostic, not product code).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void layout_4x4(const float* a, const float* b, float* d) {
  int l = threadIdx.x;
  f4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) d[r * 64 + l] = acc[r];
}
__global__ void layout_16x16x1(const float* a, const float* b, float* d) {
  int l = threadIdx.x;
  f16v acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0;
  acc = __builtin_amdgcn_mfma_f32_16x16x1f32(a[l], b[l], acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) d[r * 64 + l] = acc[r];
}
template <int NACC>
__global__ void rate_4x4(float* out, int iters) {
  f4 acc[NACC];
  float a = threadIdx.x * 0.5f, b = threadIdx.x * 0.25f;
  for (int i = 0; i < NACC; ++i) acc[i] = f4{0, 0, 0, 0};
  long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
  }
  long t1 = clock64();
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = float(t1 - t0) / (float(iters) * NACC);
}
template <int NACC>
__global__ void rate_16x16x4(float* out, int iters) {
  f4 acc[NACC];
  float a = threadIdx.x * 0.5f, b = threadIdx.x * 0.25f;
  for (int i = 0; i < NACC; ++i) acc[i] = f4{0, 0, 0, 0};
  long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  long t1 = clock64();
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = float(t1 - t0) / (float(iters) * NACC);
}
__global__ void rate_16x16x1(float* out, int iters) {
  f16v acc0, acc1;
  float a = threadIdx.x * 0.5f, b = threadIdx.x * 0.25f;
  for (int r = 0; r < 16; ++r) { acc0[r] = 0; acc1[r] = 0; }
  long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    acc0 = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, acc1, 0, 0, 0);
  }
  long t1 = clock64();
  float s = 0;
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = float(t1 - t0) / (float(iters) * 2);
}

int main() {
  float *a, *b, *d;
  hipMalloc(&a, 64 * 4); hipMalloc(&b, 64 * 4); hipMalloc(&d, 4 * 1024 * 1024 * 4);
  std::vector<float> ha(64), hb(64), hd(16 * 64);
  // A: value encodes lane as (lane+1); B: value encodes lane as 100*(lane+1): D = a*b identifies (la, lb).
  for (int l = 0; l < 64; ++l) { ha[l] = float(l + 1); hb[l] = float(1000 * (l + 1)); }
  hipMemcpy(a, ha.data(), 256, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 256, hipMemcpyHostToDevice);
  layout_4x4<<<1, 64>>>(a, b, d);
  hipMemcpy(hd.data(), d, 4 * 64 * 4, hipMemcpyDeviceToHost);
  printf("4x4x1_16B: D[reg][lane] = A[lane la] * B[lane lb]\n");
  for (int r = 0; r < 4; ++r) for (int l = 0; l < 64; l += 1) {
    long v = (long)hd[r * 64 + l];
    // find la, lb with (la+1)*(1000*(lb+1)) == v
    int fla = -1, flb = -1;
    for (int la = 0; la < 64 && fla < 0; ++la) for (int lb = 0; lb < 64; ++lb) if ((long)(la + 1) * 1000 * (lb + 1) == v) {
      // disambiguate: prefer same block (la/4 == lb/4 == l/4)
      if (la / 4 == l / 4 && lb / 4 == l / 4) { fla = la; flb = lb; break; } }
    if (l < 8 || l >= 60) printf("  reg %d lane %2d: la=%2d lb=%2d (v=%ld)\n", r, l, fla, flb, v);
  }
  layout_16x16x1<<<1, 64>>>(a, b, d);
  hipMemcpy(hd.data(), d, 16 * 64 * 4, hipMemcpyDeviceToHost);
  printf("16x16x1_4B:\n");
  for (int r = 0; r < 16; ++r) for (int l : {0, 1, 15, 16, 17, 33, 63}) {
    long v = (long)hd[r * 64 + l];
    int fla = -1, flb = -1;
    for (int la = 0; la < 64 && fla < 0; ++la) for (int lb = 0; lb < 64; ++lb) if ((long)(la + 1) * 1000 * (lb + 1) == v) { fla = la; flb = lb; break; }
    printf("  reg %2d lane %2d: la=%2d lb=%2d\n", r, l, fla, flb);
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](auto launch, const char* name, int nacc, int waves) {
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // 256 CUs x 4 SIMDs, `waves` waves per SIMD, each 100000*nacc instrs
    double per_simd = double(ms) * 1e-3 * 2.4e9 / (100000.0 * nacc * waves);
    printf("%s NACC=%d waves/SIMD=%d: %.2f clk(2.4GHz)/instr/SIMD  (%.3f ms)\n", name, nacc, waves, per_simd, ms);
  };
  for (int waves = 1; waves <= 4; waves *= 2) {
    int thr = 256 * waves; if (thr > 1024) thr = 1024;
    int blocks = 256 * (256 * waves / thr);
    timeit([&] { rate_4x4<1><<<blocks, thr>>>(d, 100000); }, "4x4x1", 1, waves);
    timeit([&] { rate_4x4<4><<<blocks, thr>>>(d, 100000); }, "4x4x1", 4, waves);
    timeit([&] { rate_4x4<8><<<blocks, thr>>>(d, 100000); }, "4x4x1", 8, waves);
    timeit([&] { rate_16x16x4<4><<<blocks, thr>>>(d, 100000); }, "16x16x4", 4, waves);
    timeit([&] { rate_16x16x1<<<blocks, thr>>>(d, 100000); }, "16x16x1_4B", 2, waves);
  }
  return 0;
}
