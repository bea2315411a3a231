#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ int wave_inclusive_max(int v) {
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false));
    return v;
}
__global__ void k(const int* in, int* out) { out[threadIdx.x] = wave_inclusive_max(in[threadIdx.x]); }
int main() {
    int h[64], o[64], *d, *e;
    for (int i = 0; i < 64; ++i) h[i] = (i % 7 == 3) ? i : 0;
    hipMalloc(&d, 256); hipMalloc(&e, 256);
    hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, e);
    hipMemcpy(o, e, 256, hipMemcpyDeviceToHost);
    int ref = 0, bad = 0;
    for (int i = 0; i < 64; ++i) { ref = h[i] > ref ? h[i] : ref; if (o[i] != ref) ++bad; printf("%d:%d/%d ", i, o[i], ref); }
    printf("\nbad=%d\n", bad);
    return 0;
}
