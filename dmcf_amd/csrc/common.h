// Internal helpers shared by the gfx950 kernels of libdmcf_hip.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dmcf_hip.h"

namespace dmcf {

constexpr int kWave = 64;  // CDNA4 wavefront

extern thread_local int g_last_hip_error;

inline int check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return DMCF_ELAUNCH;
    }
    return DMCF_OK;
}

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// compute units of the current device (persistent kernels launch one workgroup, or a few, per CU); 256 when the query fails
inline int device_cu_count() {
    static thread_local int ncu[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
    if (ncu[dev] == 0) {
        int n = 0;
        ncu[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return ncu[dev];
}

// un-fused float arithmetic for everything that feeds a comparison which decides set membership
__device__ __forceinline__ float dist2_unfused(float ax, float ay, float az, float bx, float by, float bz) {
    const float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by), dz = __fsub_rn(az, bz);
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// exclusive prefix scans (device-wide, three launches, any length)
int scan_exclusive_u32(const uint32_t* in, uint32_t* out, int64_t n, void* tmp, size_t tmp_bytes,
                       hipStream_t stream);  // out[i] = sum(in[0..i)), out has n entries
int scan_counts_to_row_splits(const int32_t* counts, int64_t* row_splits, int64_t n, void* tmp,
                              size_t tmp_bytes, hipStream_t stream);  // row_splits has n+1 entries
size_t scan_tmp_bytes(int64_t n);

}  // namespace dmcf
