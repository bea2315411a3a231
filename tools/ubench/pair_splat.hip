// Probe + rate of splat F's inner loop (dmcf_amd/csrc/cconv_pair_splat.inc, tools/gen_pair_splat.py) on gfx950 -- diagnostic, not
// product code.  (1) does M0-relative addressing pick the accumulator tile of v_mfma_f32_4x4x1_16B_f32, with ds_read /
// v_readlane inside the indexed region?  (2) clocks per pair and SIMD at two waves per SIMD (512-thread workgroups).
//   hipcc -O3 --offload-arch=gfx950 -o tools/ubench/pair_splat tools/ubench/pair_splat.hip && tools/ubench/pair_splat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

#define R4(a) "v" #a
#define PAIR_REGS                                                                                                          \
    "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", \
        "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144",      \
        "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158",      \
        "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172",      \
        "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186",      \
        "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200",      \
        "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214",      \
        "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228",      \
        "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242",      \
        "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255"

constexpr int kWaves = 8;
constexpr int kRecF = 16 * 36;   // floats per wave: 16 groups x (8 products x 4 pairs, padded to 36) -- or 32 x 8 x 2
constexpr int kFstF = 32 * 64;   // 32 groups x 32 channels x 2 pairs
constexpr int kWaveF = kRecF + kFstF;

__device__ __forceinline__ uint32_t lds_addr(const void* q) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)q;
}

__device__ __forceinline__ void zero_tiles() {
    asm volatile(
        ".irp r,148,149,150,151,152,153,154,155,156,157,158,159,160,161,162,163,164,165,166,167,168,169,170,171,172,173,174,175,"
        "176,177,178,179,180,181,182,183,184,185,186,187,188,189,190,191,192,193,194,195,196,197,198,199,200,201,202,203,204,"
        "205,206,207,208,209,210,211,212,213,214,215,216,217,218,219,220,221,222,223,224,225,226,227,228,229,230,231,232,233,"
        "234,235,236,237,238,239,240,241,242,243,244,245,246,247,248,249,250,251,252,253,254,255\n\t"
        "v_mov_b32 v\\r, 0\n\t.endr" ::: "memory", PAIR_REGS);
}

// VARIANT 0: the product block (groups of 4 pairs); 1 .. 3: UBENCH_VARIANTS of tools/gen_pair_splat.py (b64, noreads, noclass)
#define SPLAT_OPERANDS                                                                                                        \
        : [s0] "=&s"(s0)                                                                                                      \
        : [pa] "v"(pa), [pf] "v"(pf), [nb] "s"(nb), [c0] "s"(c[0]), [c1] "s"(c[1]), [c2] "s"(c[2]), [c3] "s"(c[3]),            \
          [c4] "s"(c[4]), [c5] "s"(c[5]), [c6] "s"(c[6]), [c7] "s"(c[7]), [c8] "s"(c[8]), [c9] "s"(c[9]), [c10] "s"(c[10]),    \
          [c11] "s"(c[11]), [c12] "s"(c[12]), [c13] "s"(c[13]), [c14] "s"(c[14]), [c15] "s"(c[15])                             \
        : "scc", "m0", "memory", PAIR_REGS)

template <int VARIANT>
__global__ __launch_bounds__(64 * kWaves, 1) __attribute__((amdgpu_num_vgpr(58))) void splat_loop(float* out, long long* clk, int iters, int npairs, int cshift) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int GP = VARIANT == 1 ? 2 : 4;  // pairs per staging group
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* Rec = smem + wave * kWaveF;  // [64 / GP][8][GP]
    float* Fst = Rec + kRecF;           // [64 / GP][32][GP]
    constexpr int RG = VARIANT == 0 ? 36 : 8 * GP;  // floats per record group (the product layout pads to 36)
    static_assert(kFstF >= 64 * 32, "");
    for (int q = 0; q < 8; ++q) Rec[(lane / GP) * RG + q * GP + (lane % GP)] = (lane < npairs) ? (lane + 1) * 0.25f + q * 16.0f : 0.0f;
    for (int c = 0; c < 32; ++c) {
        const float f = (float)(c + 1) + 64.0f * (lane % 3);
        if (VARIANT == 2) Fst[lane * 32 + c] = f;  // row-major: [pair][32 channels]
        else Fst[(lane / GP) * 32 * GP + c * GP + (lane % GP)] = f;
    }
    __syncthreads();
    // class bytes (4 * class) of the 64 pairs, four per scalar register: the owner lanes' values, packed inside each quad with
    // two DPP steps and read out of lanes 0, 4, 8, ...
    const int cls4 = 4 * (((lane >> cshift) * 7) % 27);
    int pk = cls4 | (__builtin_amdgcn_mov_dpp(cls4, 0xb1, 0xf, 0xf, true) << 8);        // quad_perm [1,0,3,2]
    pk = pk | (__builtin_amdgcn_mov_dpp(pk, 0x4e, 0xf, 0xf, true) << 16);               // quad_perm [2,3,0,1]
    uint32_t c[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) c[m] = __builtin_amdgcn_readlane(pk, 4 * m);
    const uint32_t pa = lds_addr(Rec) + 4 * GP * (4 * (lane >> 5) + (lane & 3));
    const uint32_t pf = lds_addr(Fst) + 4 * (VARIANT == 2 ? 1 : GP) * (lane & 31);
    const int nb = (npairs + 7) >> 3;
    zero_tiles();
    uint32_t s0;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if constexpr (VARIANT == 0) {
            asm volatile(
#include "../../dmcf_amd/csrc/cconv_pair_splat.inc"
            SPLAT_OPERANDS;
        } else if constexpr (VARIANT == 1) {
            asm volatile(
#include "pair_splat_v1.inc"
            SPLAT_OPERANDS;
        } else if constexpr (VARIANT == 2) {
            asm volatile(
#include "pair_splat_v2.inc"
            SPLAT_OPERANDS;
        } else {
            asm volatile(
#include "pair_splat_v3.inc"
            SPLAT_OPERANDS;
        }
    }
    const long long t1 = clock64();
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory", PAIR_REGS);
    float* dump = smem;  // [108][64] per wave 0
    __syncthreads();
    if (wave == 0) {
        const uint32_t da = lds_addr(dump) + 4 * lane;
        asm volatile(
            ".set off_, 0\n\t"
            ".irp r,148,149,150,151,152,153,154,155,156,157,158,159,160,161,162,163,164,165,166,167,168,169,170,171,172,173,174,175,"
            "176,177,178,179,180,181,182,183,184,185,186,187,188,189,190,191,192,193,194,195,196,197,198,199,200,201,202,203,204,"
            "205,206,207,208,209,210,211,212,213,214,215,216,217,218,219,220,221,222,223,224,225,226,227,228,229,230,231,232,233,"
            "234,235,236,237,238,239,240,241,242,243,244,245,246,247,248,249,250,251,252,253,254,255\n\t"
            "ds_write_b32 %0, v\\r offset:off_\n\t.set off_, off_ + 256\n\t.endr\n\ts_waitcnt lgkmcnt(0)"
            :: "v"(da) : "memory", PAIR_REGS);
    }
    __syncthreads();
    if (blockIdx.x == 0 && wave == 0)
        for (int r = 0; r < 108; ++r) out[r * 64 + lane] = dump[r * 64 + lane];
    if (tid == 0) clk[blockIdx.x] = t1 - t0;
}

typedef void (*kern_t)(float*, long long*, int, int, int);
constexpr int kNV = 4;
static const kern_t kKernels[kNV] = {splat_loop<0>, splat_loop<1>, splat_loop<2>, splat_loop<3>};
static const char* kNames[kNV] = {"product (products and features in groups of 4)", "groups of 2, ds_read_b64", "features row-major, ds_read_b32", "noclass"};

int main(int argc, char** argv) {
    const int npairs = argc > 1 ? atoi(argv[1]) : 64;
    float* out;
    long long* clk;
    (void)hipMalloc(&out, 108 * 64 * 4);
    (void)hipMalloc(&clk, 1024 * 8);
    const size_t lds = kWaves * kWaveF * 4;
    for (int v = 0; v < kNV; ++v) (void)hipFuncSetAttribute((const void*)kKernels[v], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    // ---- correctness: one iteration, one workgroup; consecutive pairs share a class in runs of 1, 2, 4, 64
    for (int variant = 0; variant < kNV; ++variant) {
        for (int cshift : {0, 1, 2, 6}) {
            hipLaunchKernelGGL(kKernels[variant], dim3(1), dim3(64 * kWaves), lds, 0, out, clk, 1, npairs, cshift);
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
            std::vector<float> h(108 * 64);
            (void)hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
            std::vector<double> ref(108 * 64, 0.0);
            for (int k = 0; k < 64 && k < 8 * ((npairs + 7) / 8); ++k) {
                const int cls = variant == 3 ? 0 : ((k >> cshift) * 7) % 27;
                for (int lane = 0; lane < 64; ++lane)
                    for (int i = 0; i < 4; ++i) {
                        // D[block][i][j]: lane = 4 block + j, register i;  A[block][i] = products[k][4 z' + i], z' = block >> 3;  B[block][j] = f[k][lane & 31]
                        const int q = 4 * (lane >> 5) + i;
                        const double a = k < npairs ? (k + 1) * 0.25 + q * 16.0 : 0.0;
                        const double f = (double)((lane & 31) + 1) + 64.0 * (k % 3);
                        ref[(4 * cls + i) * 64 + lane] += a * f;
                    }
            }
            double err = 0.0, mag = 0.0;
            for (size_t e = 0; e < ref.size(); ++e) { err = fmax(err, fabs(ref[e] - h[e])); mag = fmax(mag, fabs(ref[e])); }
            printf("variant %d (%s), %d pairs, class runs of %d: max |err| %g (max |ref| %g)  %s\n", variant, kNames[variant], npairs,
                   1 << cshift, err, mag, err <= 1e-6 * mag ? "OK" : "WRONG");
        }
    }
    // ---- rate: one 8-wave workgroup per CU (two waves per SIMD), all CUs
    for (int variant = 0; variant < kNV; ++variant) {
        const int iters = 2000, grid = 256;
        hipEvent_t a, b;
        (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(a);
            hipLaunchKernelGGL(kKernels[variant], dim3(grid), dim3(64 * kWaves), lds, 0, out, clk, iters, 64, 0);
            (void)hipEventRecord(b);
            (void)hipDeviceSynchronize();
        }
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        printf("variant %d (%s): %.3f ms for %d iterations; %.2f ns per pair and SIMD = %.1f clk at 2.4 GHz\n",
               variant, kNames[variant], ms, iters, 1e6 * ms / (2.0 * iters * 64), 1e6 * ms / (2.0 * iters * 64) * 2.4);
    }
    return 0;
}
