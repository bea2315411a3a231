// What bounds the per-tile tail of the CConv kernels (B rows -> LDS, contraction with the packed filter, sum over waves,
// store) on gfx950 -- diagnostic, not product code.  The 32 -> 32 layer with EMPTY neighbour lists takes 2.37 ms at 1.12M
// points (tools/bench_overhead.py) against 0.83 ms of matrix clocks; this isolates that tail (16 points per tile, K = 64 cells
// x 32 channels in two 16-channel chunks, 32 output channels) and removes one cost at a time:
//   bit 0  persistent workgroups (one per CU, tiles by stride) instead of one workgroup per tile
//   bit 1  filter fragments loaded ONCE per workgroup (filter-stationary) instead of per tile
//   bit 2  no matrix instructions          bit 3  no B-row stores / A-fragment reads from LDS
//   bit 4  no sum over waves / output store
//   hipcc -O3 --offload-arch=gfx950 -o tools/ubench/contract tools/ubench/contract.hip && tools/ubench/contract
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kRow = 1024;
constexpr int NT = 2;

// bit 6: the kernel reads its arguments from a module-scope device variable (cached memory, address in the code) instead of
// the kernarg segment
struct TailArgs {
    const float* W;
    float* out;
    int ntiles;
    float seed;
};
__device__ TailArgs g_args;

template <int WAVES, int VAR>
__global__ __launch_bounds__(64 * WAVES, 1) void tail(const float* __restrict__ Wp_arg, float* __restrict__ out_arg, int ntiles_arg, float seed_arg) {
    const float* Wp = (VAR & 64) ? g_args.W : Wp_arg;
    float* out = (VAR & 64) ? g_args.out : out_arg;
    const int ntiles = (VAR & 64) ? g_args.ntiles : ntiles_arg;
    const float seed = (VAR & 64) ? g_args.seed : seed_arg;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr bool kPersist = VAR & 1, kStationary = VAR & 2, kNoMfma = VAR & 4, kNoLds = VAR & 8, kNoEpi = VAR & 16, kReload = VAR & 32;
    constexpr int kIt = 64 / WAVES;  // blocks of a chunk per wave
    constexpr int PPW = 16 / WAVES;  // points per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mi = lane & 15, mg = lane >> 4;
    const int jn = lane & 31, hk = lane >> 5;
    float* Bt = smem;
    f32x4 bw[2][kIt][NT];
    auto w_issue = [&]() {
#pragma unroll
        for (int chunk = 0; chunk < 2; ++chunk)
#pragma unroll
            for (int it = 0; it < kIt; ++it) {
                const int blk = wave + WAVES * it;
                const float* wb = Wp + (size_t)chunk * 64 * (4 * NT * 16 * 4) + ((size_t)(blk * 4 + mg) * NT * 16 + mi) * 4;
#pragma unroll
                for (int n = 0; n < NT; ++n) bw[chunk][it][n] = *(const f32x4*)(wb + n * 64);
            }
    };
    if (kStationary) w_issue();
    const int tile0 = kPersist ? blockIdx.x : (int)(blockIdx.x % 8) * ((ntiles + 7) / 8) + (int)(blockIdx.x / 8);
    const int step = kPersist ? gridDim.x : ntiles;
    for (int tile = tile0; tile < ntiles; tile += step) {
        f32x4 acc[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[n] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        if (kReload) {  // the filter pointer fetched from the argument block again, as a fresh workgroup has to
            typedef const __attribute__((address_space(4))) uint64_t* KP;
            KP kp = (KP)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(kp));
            Wp = (const float*)kp[0];
        }
        if (!kStationary) w_issue();
        const f32x4 val = {seed * tile, seed, seed + 1.0f, seed * lane};
#pragma unroll
        for (int chunk = 0; chunk < 2; ++chunk) {
            if (!kNoLds) {
#pragma unroll
                for (int pp = 0; pp < PPW; ++pp) {
                    const int row = wave + WAVES * pp;
                    if ((jn >> 4) == chunk) {
                        float* q = Bt + row * kRow + hk * 64 + (((jn & 15) ^ (row & 15)) << 2);
#pragma unroll
                        for (int r = 0; r < 8; ++r) *(f32x4*)(q + 128 * r) = val;
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < kIt; ++it) {
                const int blk = wave + WAVES * it;
                f32x4 av = val;
                if (!kNoLds) av = *(const f32x4*)(Bt + (size_t)mi * kRow + ((blk * 16 + mg * 4) ^ (mi << 2)));
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const f32x4 bv = bw[chunk][it][n];
                    if (!kNoMfma) {
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc[n], 0, 0, 0);
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc[n], 0, 0, 0);
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc[n], 0, 0, 0);
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc[n], 0, 0, 0);
                    } else {
                        acc[n] += av * bv;
                    }
                }
            }
            __syncthreads();
        }
        if (kNoEpi) {
            if (acc[0].x + acc[1].y == 12345.678f) out[tile] = acc[0].z;
            continue;
        }
        float* red = Bt;
        constexpr int ncol = 16 * NT;
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((size_t)wave * 16 + 4 * mg + r) * ncol + n * 16 + mi] = acc[n][r];
        __syncthreads();
        for (int e = tid; e < 16 * 32; e += 64 * WAVES) {
            const int ptt = e / 32, o = e % 32;
            float v = 0.0f;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) v += red[((size_t)w * 16 + ptt) * ncol + o];
            out[((size_t)tile * 16 + ptt) * 32 + o] = v;
        }
        if (kPersist) __syncthreads();
    }
}

template <int WAVES, int VAR>
static void run(const float* W, float* out, int ntiles, int ncu, const char* what) {
    const size_t lds = (size_t)16 * kRow * 4 + 16 * 1024;  // B tile + what the splat stages beside it
    hipFuncSetAttribute((const void*)tail<WAVES, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const unsigned grid = (VAR & 1) ? (unsigned)ncu : (unsigned)(((ntiles + 7) / 8) * 8);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((tail<WAVES, VAR>), dim3(grid), dim3(64 * WAVES), lds, 0, W, out, ntiles, 0.0f);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (rep && ms < best) best = ms;
    }
    printf("%2d waves  var %2d  %-60s %6.3f ms  (%5.0f clocks per tile and CU at 2.4 GHz)\n", WAVES, VAR, what, best,
           best * 1e-3 * 2.4e9 / (ntiles / (double)ncu));
    if (VAR == 0) {  // the same launch as a one-node graph (where do a graph's kernel arguments live?)
        hipStream_t st;
        hipStreamCreate(&st);
        hipGraph_t graph;
        hipGraphExec_t exec;
        hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
        hipLaunchKernelGGL((tail<WAVES, VAR>), dim3(grid), dim3(64 * WAVES), lds, st, W, out, ntiles, 0.0f);
        hipStreamEndCapture(st, &graph);
        hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        float gbest = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(a, st);
            hipGraphLaunch(exec, st);
            hipEventRecord(b, st);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            if (rep && ms < gbest) gbest = ms;
        }
        printf("%2d waves  var %2d  %-60s %6.3f ms\n", WAVES, VAR, "  ... launched as a graph node", gbest);
    }
}

int main() {
    const int ntiles = 70304;  // 1,124,864 points / 16
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount;
    std::vector<float> hw(2 * 64 * 4 * NT * 16 * 4);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = (float)((i * 2654435761u) & 1023) / 1024.0f - 0.5f;
    float *W, *out;
    hipMalloc(&W, hw.size() * 4);
    hipMalloc(&out, (size_t)ntiles * 16 * 32 * 4);
    hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    printf("%d CUs, %d tiles of 16 points, K = 2048, 32 output channels; matrix floor %.3f ms\n", ncu, ntiles,
           ntiles * 1024.0 * 32 / 4 / ncu / 2.4e9 * 1e3);
    run<16, 0>(W, out, ntiles, ncu, "as the product (workgroup per tile, filter per tile)");
    {
        TailArgs ha = {W, out, ntiles, 0.0f};
        hipMemcpyToSymbol(HIP_SYMBOL(g_args), &ha, sizeof(ha));
    }
    run<16, 64>(W, out, ntiles, ncu, "workgroup per tile, arguments from a device variable");
    run<8, 64>(W, out, ntiles, ncu, "8 waves: workgroup per tile, arguments from a device variable");
    run<16, 1>(W, out, ntiles, ncu, "persistent");
    run<16, 33>(W, out, ntiles, ncu, "persistent, arguments fetched again per tile");
    run<16, 2>(W, out, ntiles, ncu, "filter once per workgroup (= per tile here)");
    run<16, 3>(W, out, ntiles, ncu, "persistent + filter-stationary");
    run<16, 4>(W, out, ntiles, ncu, "no matrix instructions");
    run<16, 8>(W, out, ntiles, ncu, "no LDS traffic for B");
    run<16, 16>(W, out, ntiles, ncu, "no sum over waves / store");
    run<16, 7>(W, out, ntiles, ncu, "persistent + stationary, no matrix");
    run<16, 11>(W, out, ntiles, ncu, "persistent + stationary, no LDS for B");
    run<16, 19>(W, out, ntiles, ncu, "persistent + stationary, no epilogue");
    run<16, 27>(W, out, ntiles, ncu, "persistent + stationary, matrix only");
    run<8, 0>(W, out, ntiles, ncu, "8 waves: as splat F");
    run<8, 1>(W, out, ntiles, ncu, "8 waves: persistent");
    run<8, 3>(W, out, ntiles, ncu, "8 waves: persistent + filter-stationary");
    run<8, 27>(W, out, ntiles, ncu, "8 waves: persistent + stationary, matrix only");
    return 0;
}
