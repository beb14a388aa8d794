// scripts/hbm_calib.hip -- calibration of rocprofv3's FETCH_SIZE for the access patterns of the engine's kernels
// (MI355X_MICROARCH.md "HBM": FETCH_SIZE is exact only after calibrating on a known byte count in your own pattern).
// Three kernels over a 2 GiB buffer (8x the 256 MiB Infinity Cache), each with a known number of requested bytes:
//   calib_stream16   every lane reads 16 B, fully coalesced (the guide's reference case: counter = 1/2 of the bytes)
//   calib_gather16   every lane reads one 16-B element from its own random 128-B line  (SeqBlock loads of SeedExtend)
//   calib_gather8    every lane reads one 8-B element from its own random 128-B line   (hash slot / next[] probes)
// Build: hipcc --offload-arch=gfx950 -O3 scripts/hbm_calib.hip -o parsnp_amd/bin/hbm_calib ; run under
// rocprofv3 --pmc FETCH_SIZE --kernel-trace.  Prints the requested bytes per kernel.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__global__ void calib_stream16(const uint4* __restrict__ buf, size_t n, uint32_t* sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 v = buf[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345u) *sink = acc;
}
__device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
__global__ void calib_gather16(const uint4* __restrict__ buf, size_t lines, size_t n, uint32_t* sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    size_t line = mix(i) % lines;                      // 128-B line; element 3 of its 8
    uint4 v = buf[line * 8 + 3];
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345u) *sink = v.x;
}
__global__ void calib_gather8(const uint64_t* __restrict__ buf, size_t lines, size_t n, uint32_t* sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    size_t line = mix(i + 0x9e3779b9ull) % lines;
    uint64_t v = buf[line * 16 + 5];
    if (v == 0x12345u) *sink = (uint32_t)v;
}
int main() {
    const size_t bytes = 2ull << 30;
    void* buf; uint32_t* sink;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(buf, 1, bytes);
    (void)hipDeviceSynchronize();
    const size_t n16 = bytes / 16, lines = bytes / 128, ng = 1ull << 24;
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(calib_stream16, dim3(256 * 16), dim3(256), 0, 0, (const uint4*)buf, n16, sink);
        hipLaunchKernelGGL(calib_gather16, dim3((unsigned)(ng / 256)), dim3(256), 0, 0, (const uint4*)buf, lines, ng, sink);
        hipLaunchKernelGGL(calib_gather8, dim3((unsigned)(ng / 256)), dim3(256), 0, 0, (const uint64_t*)buf, lines, ng, sink);
    }
    (void)hipDeviceSynchronize();
    printf("{\"calib_stream16\": %zu, \"calib_gather16\": %zu, \"calib_gather8\": %zu, \"gather_lanes\": %zu}\n", bytes, ng * 16, ng * 8, ng);
    return 0;
}
