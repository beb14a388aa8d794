#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <chrono>
using cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
    rocprim::radix_sort_onesweep_config<rocprim::kernel_config<PM_HBS, 12>, rocprim::kernel_config<PM_BS, PM_IPT>, PM_BITS, rocprim::block_radix_rank_algorithm::match>>;
int main() {
    const size_t n = 7700000; const unsigned bits = 32;
    std::vector<KEY_T> k(n); std::vector<uint64_t> v(n);
    uint64_t x = 88172645463325252ull;
    for (size_t i = 0; i < n; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; k[i] = (KEY_T)(x & 0xffffffffull); v[i] = i; }
    KEY_T *ki, *ko; uint64_t *vi, *vo; hipMalloc(&ki, sizeof(KEY_T)*n); hipMalloc(&ko, sizeof(KEY_T)*n); hipMalloc(&vi, 8*n); hipMalloc(&vo, 8*n);
    hipMemcpy(ki, k.data(), sizeof(KEY_T)*n, hipMemcpyHostToDevice); hipMemcpy(vi, v.data(), 8*n, hipMemcpyHostToDevice);
    size_t bytes = 0; void* tmp = nullptr;
    rocprim::radix_sort_pairs<cfg>(nullptr, bytes, ki, ko, vi, vo, n, 0, bits, 0);
    hipMalloc(&tmp, bytes);
    for (int it = 0; it < 3; it++) {
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        rocprim::radix_sort_pairs<cfg>(tmp, bytes, ki, ko, vi, vo, n, 0, bits, 0);
        hipDeviceSynchronize();
        printf("bits %d bs %d ipt %d: %.1f us\n", PM_BITS, PM_BS, PM_IPT, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    std::vector<KEY_T> o(n); hipMemcpy(o.data(), ko, sizeof(KEY_T)*n, hipMemcpyDeviceToHost);
    printf("sorted: %d\n", (int)std::is_sorted(o.begin(), o.end()));
    return 0;
}
