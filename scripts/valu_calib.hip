// scripts/valu_calib.hip -- how many int32 vector instructions a gfx950 SIMD issues per cycle (measurement only, not part of the
// product).  bench.py's `limiter` / `issue_frac` price SeedExtend's SQ_INSTS_VALU against this figure: the guide
// (/opt/skills/guides/MI355X_MICROARCH.md, "Wave scheduling") gives 2 cycles per wave64 VALU instruction on the SIMD-32 units for
// v_fma_f32; this program measures v_add_u32 / v_and_b32 / v_lshlrev_b32, the mix of the seed kernels, as
//   dependent chain   one accumulator, every instruction waits for the one before it (latency)
//   independent       eight accumulators round robin (throughput of ONE wave)
// at 1 ... 8 wavefronts per SIMD (one block of 256 x w threads per CU; above w = 4 two blocks of 128 x w), timed per wave with
// s_memtime (tick = shader cycle) and over the launch with HIP events.  Prints one JSON object.
//   hipcc --offload-arch=gfx950 -O3 scripts/valu_calib.hip -o parsnp_amd/bin/valu_calib && parsnp_amd/bin/valu_calib
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

constexpr int kOpsPerIter = 96;      // vector instructions per loop iteration (3 opcodes x 32)
constexpr int kIters = 4096;

#define OP3(x, a, m)                                              \
    asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(a));      \
    asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(m));      \
    asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(x));

template <bool kDependent>
__global__ void __launch_bounds__(1024) chain(uint32_t* out, uint64_t* cycles, uint32_t a, uint32_t m) {
    uint32_t x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIters; it++) {
        if (kDependent) {
#pragma unroll
            for (int k = 0; k < kOpsPerIter / 3; k++) { OP3(x0, a, m) }
        } else {
#pragma unroll
            for (int k = 0; k < kOpsPerIter / 24; k++) {
                // the same three opcodes, eight chains interleaved instruction by instruction
                asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                             "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                             "v_and_b32 %0, %0, %9\n v_and_b32 %1, %1, %9\n v_and_b32 %2, %2, %9\n v_and_b32 %3, %3, %9\n"
                             "v_and_b32 %4, %4, %9\n v_and_b32 %5, %5, %9\n v_and_b32 %6, %6, %9\n v_and_b32 %7, %7, %9\n"
                             "v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 1, %3\n"
                             "v_lshlrev_b32 %4, 1, %4\n v_lshlrev_b32 %5, 1, %5\n v_lshlrev_b32 %6, 1, %6\n v_lshlrev_b32 %7, 1, %7\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                             : "v"(a), "v"(m));
            }
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
    if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint32_t* d_out; uint64_t* d_cyc;
    const size_t max_threads = (size_t)cus * 2048;
    CK(hipMalloc(&d_out, 4 * max_threads)); CK(hipMalloc(&d_cyc, 8 * max_threads / 64));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double insts = (double)kOpsPerIter * kIters;
    printf("{\"device\": \"%s\", \"compute_units\": %d, \"clock_mhz_reported\": %.0f, \"instructions_per_wave\": %.0f, \"opcodes\": \"v_add_u32, v_and_b32, v_lshlrev_b32 in equal parts\", \"runs\": [",
           prop.gcnArchName, cus, prop.clockRate / 1e3, insts);
    bool first = true;
    double best_rate = 0;
    for (int dep = 1; dep >= 0; dep--)
        for (int w = 1; w <= 8; w++) {
            // 4 w wavefronts per CU: one block of 256 w threads up to w = 4, two blocks of 128 w threads above (a block holds 1024 at most)
            const int blocks = w > 4 ? 2 * cus : cus, threads = w > 4 ? 128 * w : 256 * w;
            float ms = 0;
            for (int rep = 0; rep < 3; rep++) {      // (the last repetition counts: clocks have settled)
                CK(hipEventRecord(e0, 0));
                if (dep) hipLaunchKernelGGL(chain<true>, dim3(blocks), dim3(threads), 0, 0, d_out, d_cyc, 3u, 0x7fffffffu);
                else hipLaunchKernelGGL(chain<false>, dim3(blocks), dim3(threads), 0, 0, d_out, d_cyc, 3u, 0x7fffffffu);
                CK(hipGetLastError());
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
            }
            const size_t nw = (size_t)blocks * threads / 64;
            std::vector<uint64_t> cyc(nw);
            CK(hipMemcpy(cyc.data(), d_cyc, 8 * nw, hipMemcpyDeviceToHost));
            std::sort(cyc.begin(), cyc.end());
            const double med = (double)cyc[nw / 2];
            // w waves share a SIMD for (about) the median wave's cycles: instructions issued per cycle and SIMD
            const double per_simd = w * insts / med;
            if (!dep && per_simd > best_rate) best_rate = per_simd;
            printf("%s{\"chain\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_wave_median\": %.0f, \"cycles_per_instruction_one_wave\": %.3f, \"instructions_per_cycle_per_simd\": %.4f, "
                   "\"launch_ms\": %.4f, \"effective_ghz\": %.3f}",
                   first ? "" : ", ", dep ? "dependent" : "independent", w, med, med / insts, per_simd, ms, med / (ms * 1e6));
            first = false;
        }
    printf("], \"valu_int32_wave64_instructions_per_cycle_per_simd\": %.4f, \"cycles_per_wave64_int32_valu_instruction\": %.3f}\n", best_rate, best_rate > 0 ? 1.0 / best_rate : 0.0);
    return 0;
}
