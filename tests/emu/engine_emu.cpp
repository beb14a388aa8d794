// tests/emu/engine_emu.cpp -- TEST INFRASTRUCTURE ONLY.
// Runs the engine's kernel functors (parsnp_amd/csrc/engine/kernels.h) one "thread" after another on the host, behind
// the same C ABI, so the kernel logic and the orchestration can be checked against the oracle without a GPU.
// Built into tests/emu/libpm_emu.so by tests; never built or loaded by the product.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../parsnp_amd/csrc/engine/engine_core.h"

struct HostBackend {
    void* alloc(size_t n) { return malloc(n ? n : 1); }
    void free(void* p) { ::free(p); }
    static void* host_alloc(size_t n) { return malloc(n ? n : 1); }
    static void host_free(void* p) { ::free(p); }
    void memset(void* p, int v, size_t n) { ::memset(p, v, n); }
    void clear_many(const pm::ClearJob* jobs, int n) { for (int i = 0; i < n; i++) if (jobs[i].bytes) ::memset(jobs[i].p, jobs[i].value, jobs[i].bytes); }
    uint64_t bytes_h2d = 0, bytes_d2h = 0;      // what a device backend would have moved over the host link
    void h2d(void* d, const void* s, size_t n) { bytes_h2d += n; memcpy(d, s, n); }
    void d2h(void* d, const void* s, size_t n) { bytes_d2h += n; memcpy(d, s, n); }
    void d2h_async(void* d, const void* s, size_t n) { bytes_d2h += n; memcpy(d, s, n); }
    void d2h_async_pinned(void* d, const void* s, size_t n) { d2h_async(d, s, n); }
    void drop_landings() {}
    void d2d(void* d, const void* s, size_t n) { memcpy(d, s, n); }
    void bind() {}
    void* pinned_alloc(size_t n) { return malloc(n ? n : 1); }
    void pinned_free(void* p) { ::free(p); }
    void sync() {}
    void bind_thread() {}
    void* event_record() { return nullptr; }
    static void event_wait(void*) {}
    void event_release(void*) {}
    void fill32(int32_t* p, int32_t v) { *p = v; }
    int allreduce_min_i32_dev(int32_t*, int64_t) { return 1; }      // no device collectives in the emulation
    int allgather_dev(const void*, int64_t, void*) { return 1; }
    std::vector<char> stage;
    void* staging(size_t n) { if (stage.size() < n) stage.resize(n); return stage.data(); }
    void h2d_staged(void* d, const void* s, size_t n) { bytes_h2d += n; memcpy(d, s, n); }
    template <class F> void launch(const char*, int64_t n, F f) { for (int64_t i = 0; i < n; i++) f(i); }
    bool stage_genomes(int n, const uint8_t* const* seqs, const int64_t* lens, const std::vector<char>& take, const std::vector<int64_t>& goff,
                       pm::SeqBlock* blk, int64_t) {
        for (int g = 0; g < n; g++)
            if (take[(size_t)g])
                for (int s = 0; s < 2; s++)
                    launch("pack", (lens[g] + 31) / 32, pm::PackStrand{seqs[g], lens[g], s, blk, goff[2 * (size_t)g + (size_t)s] / 32});
        return true;
    }
    template <class F> void launch_wave(const char*, int64_t n, F f) { for (int64_t i = 0; i < n; i++) f.wave(i); }
    void exclusive_scan(const int64_t* in, int64_t* out, size_t n) { int64_t a = 0; for (size_t i = 0; i < n; i++) { int64_t v = in[i]; out[i] = a; a += v; } }
    void sort_pairs(uint64_t* ki, uint64_t* ko, uint64_t* vi, uint64_t* vo, size_t n, int) {
        std::vector<size_t> idx(n);
        for (size_t i = 0; i < n; i++) idx[i] = i;
        std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return ki[a] < ki[b]; });
        for (size_t i = 0; i < n; i++) { ko[i] = ki[idx[i]]; vo[i] = vi[idx[i]]; }
    }
    bool timing_on = true;
    void mark(const char*) {}
    std::vector<pm::PhaseTime> collect() { return {}; }
    bool ok() const { return true; }
    std::string error() const { return ""; }
};
typedef HostBackend PmBackend;
static const char* pm_backend_name = "emu";
static PmBackend* pm_backend_open(int, std::string*) { return new HostBackend; }
#include "../../parsnp_amd/csrc/engine/abi_glue.h"

// the device gap aligner is a HIP kernel with no host emulation: decline every job, the host aligner takes them
extern "C" int pm_gap_align_batch(int, int64_t n_jobs, const int32_t*, const int64_t*, const uint8_t*, const int32_t*, const int64_t*, uint8_t*, int64_t, int32_t* cols) {
    for (int64_t j = 0; j < n_jobs; j++) cols[j] = -1;
    return PM_OK;
}
extern "C" int pm_gap_align_groups(int, int64_t n_jobs, const int32_t*, const int64_t*, const uint8_t*, const int32_t*, const int64_t*, uint8_t*, int64_t, int32_t* cols,
                                   int n_groups, const int64_t*, void (*done)(void*, int), void* ctx) {
    for (int64_t j = 0; j < n_jobs; j++) cols[j] = -1;
    for (int g = 0; done && g < n_groups; g++) done(ctx, g);
    return PM_OK;
}
extern "C" const char* pm_gap_last_error(void) { return ""; }
extern "C" int pm_warmup(int) { return PM_OK; }
extern "C" int pm_rccl_unique_id(uint8_t*) { return PM_EINVAL; }
extern "C" int pm_session_rccl_ranks(const pm_session*) { return 0; }
extern "C" int pm_session_create_rccl(pm_session**, int, int, const uint8_t* const*, const int64_t*, int, int, const uint8_t*) { return PM_EINVAL; }
