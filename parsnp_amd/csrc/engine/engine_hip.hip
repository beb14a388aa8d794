// engine_hip.hip -- libparsnp_hip.so: the MI355X (gfx950) backend of the multi-MUM engine and its C ABI.
// One HIP stream per session; every kernel of kernels.h is launched as 256-thread workgroups (4 wavefronts) with
// one functor call per thread; radix sort / exclusive scan are rocPRIM device primitives on the same stream.
// Phase timing uses HIP events recorded on that stream (pm_last_timing).
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <unistd.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>          // types only: the library is loaded on demand (see Rccl below), never linked
#include <rocprim/rocprim.hpp>

#include <atomic>
#include <chrono>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "engine_core.h"

namespace {

template <class F>
__global__ __launch_bounds__(256) void pm_kernel(F f, int64_t n) {
    int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (tid < n) f(tid);
}

// clearing of the large tables (64 MB of index slots, the 16 MB coarse table, the layout images): 16 bytes per lane and step,
// every workgroup a contiguous stretch.  (hipMemsetAsync's fill kernel was measured at 0.43 TB/s on the slot table: 149 us.)
__global__ __launch_bounds__(256) void pm_fill16(uint4* p, uint4 v, int64_t n16) {
    const int64_t per = 16 * 256;      // 16 stores per thread
    int64_t i = (int64_t)blockIdx.x * per + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 16; k++, i += 256) if (i < n16) p[i] = v;
}

// the tables and counters a search clears before it starts, in ONE launch (a dozen hipMemsetAsync calls are a dozen 5 us kernels):
// every workgroup takes 64 KB of one job
struct FillJobs { enum { kMax = 12 }; void* p[kMax]; unsigned long long bytes[kMax]; unsigned int value[kMax]; unsigned int first_block[kMax + 1]; int n; };
__global__ __launch_bounds__(256) void pm_fill_many(FillJobs jobs) {
    int j = 0;
    while (j + 1 < jobs.n && blockIdx.x >= jobs.first_block[j + 1]) j++;
    const unsigned long long off = (unsigned long long)(blockIdx.x - jobs.first_block[j]) << 16;
    const unsigned long long n = jobs.bytes[j];
    unsigned char* base = (unsigned char*)jobs.p[j];
    const unsigned int w = jobs.value[j];
    if ((((uintptr_t)base) & 15) == 0) {
        uint4* q = (uint4*)(base + off);
        const unsigned long long n16 = (n > off ? (n - off < 65536 ? n - off : 65536) : 0) >> 4;
#pragma unroll
        for (int k = 0; k < 16; k++) { const unsigned long long i = (unsigned long long)k * 256 + threadIdx.x; if (i < n16) q[i] = make_uint4(w, w, w, w); }
        for (unsigned long long i = off + (n16 << 4) + threadIdx.x; i < n && i < off + 65536; i += 256) base[i] = (unsigned char)w;      // (tail)
    } else {
        for (unsigned long long i = off + threadIdx.x; i < n && i < off + 65536; i += 256) base[i] = (unsigned char)w;
    }
}

// one wavefront per work item: 64-thread workgroups, f.wave(item) with the lanes cooperating (shuffles, LDS)
template <class F>
__global__ __launch_bounds__(64) void pm_wave_kernel(F f, int64_t n) {
    const int64_t w = (int64_t)blockIdx.x;
    if (w < n) f.wave(w);
}

// RCCL, loaded by absolute path when a sharded session asks for device collectives.  Not a link-time dependency: (1) a
// single-GPU run never maps the 570 MB library; (2) a process that also holds PyTorch has torch's own bundled librccl.so.1
// loaded -- bound to torch's private HIP runtime -- and a NEEDED entry with that soname would resolve to it; dlopen of the
// system file by path gives this library its own copy, bound to the HIP runtime the engine's streams belong to.
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
    static Rccl& get() {
        static Rccl r;
        if (r.lib || !r.err.empty()) return r;
        const char* path = getenv("PARSNP_RCCL_LIB");
        if (!path || !*path) path = "/opt/rocm/lib/librccl.so.1";
        r.lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
        if (!r.lib) { r.err = std::string("cannot load RCCL (") + path + "): " + dlerror(); return r; }
        auto sym = [&](const char* n) { void* p = dlsym(r.lib, n); if (!p && r.err.empty()) r.err = std::string("RCCL symbol missing: ") + n; return p; };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
        r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.CommCount = (decltype(r.CommCount))sym("ncclCommCount");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
        return r;
    }
    bool ok() const { return lib && err.empty(); }
};

struct HipBackend {
    hipStream_t stream = nullptr;
    int device = 0;                     // the session's GPU; bind() makes it the calling thread's (a call may come from a helper thread)
    void bind() { check(hipSetDevice(device), "hipSetDevice"); }
    void bind_thread() { bind(); }
    ncclComm_t comm = nullptr;          // device collectives of a sharded session (pm_session_create_rccl)
    std::string err;
    void* tmp = nullptr;
    size_t tmp_cap = 0;
    std::vector<std::pair<const char*, hipEvent_t>> marks;
    std::vector<hipEvent_t> pool;

    bool check(hipError_t e, const char* what) {
        if (e == hipSuccess) return true;
        if (err.empty()) err = std::string(what) + ": " + hipGetErrorString(e);
        return false;
    }
    bool ok() const { return err.empty(); }
    std::string error() const { return err; }

    ~HipBackend() {
        for (auto& m : marks) pool.push_back(m.second);
        for (auto e : pool) (void)hipEventDestroy(e);
        if (tmp) (void)hipFree(tmp);
        if (stage_p) (void)hipHostFree(stage_p);
        if (ring) (void)hipHostFree(ring);
        if (dring) (void)hipHostFree(dring);
        if (comm) (void)Rccl::get().CommDestroy(comm);
        if (stream) (void)hipStreamDestroy(stream);
    }
    bool nccl_check(ncclResult_t r, const char* what) {
        if (r == ncclSuccess) return true;
        if (err.empty()) err = std::string(what) + ": " + Rccl::get().GetErrorString(r);
        return false;
    }
    // one communicator per session, ranks = the processes of the sharded run (one per GPU), over xGMI inside a node
    bool comm_init(int rank, int world, const uint8_t* id128) {
        Rccl& R = Rccl::get();
        if (!R.ok()) { if (err.empty()) err = R.err; return false; }
        ncclUniqueId id;
        static_assert(sizeof(ncclUniqueId) == 128, "PM_RCCL_ID_BYTES");
        memcpy(&id, id128, sizeof id);
        // ranks that hold different ids (a stale id file, a rank of another launch) wait for each other forever inside
        // ncclCommInitRank: a watchdog ends the process with a message instead (PARSNP_RCCL_TIMEOUT seconds, default 180; 0 = none)
        const int limit = getenv("PARSNP_RCCL_TIMEOUT") ? atoi(getenv("PARSNP_RCCL_TIMEOUT")) : 180;
        std::atomic<bool> done{false};
        std::thread dog;
        if (limit > 0)
            dog = std::thread([&done, limit, rank, world] {
                for (int t = 0; t < limit * 10 && !done.load(); t++) std::this_thread::sleep_for(std::chrono::milliseconds(100));
                if (!done.load()) {
                    fprintf(stderr, "parsnp engine: rank %d of %d: the RCCL communicator did not come up within %d s (ranks with different ids? a rank that never started?)\n", rank, world, limit);
                    _exit(4);
                }
            });
        const ncclResult_t r = R.CommInitRank(&comm, world, id, rank);
        done.store(true);
        if (dog.joinable()) dog.join();
        return nccl_check(r, "ncclCommInitRank");
    }
    int comm_ranks() { int n = 0; if (comm && Rccl::get().CommCount(comm, &n) == ncclSuccess) return n; return 0; }      // what RCCL itself says
    void fill32(int32_t* p, int32_t v) { check(hipMemsetD32Async((hipDeviceptr_t)p, v, 1, stream), "hipMemsetD32Async"); }
    // in place on a device buffer, on the engine's stream (stream-ordered with the kernels around it: no host round trip)
    int allreduce_min_i32_dev(int32_t* d, int64_t count) {
        if (!comm) { if (err.empty()) err = "no RCCL communicator"; return 1; }
        return nccl_check(Rccl::get().AllReduce(d, d, (size_t)count, ncclInt32, ncclMin, comm, stream), "ncclAllReduce(min)") ? 0 : 1;
    }
    int allgather_dev(const void* send, int64_t bytes, void* recv) {
        if (!comm) { if (err.empty()) err = "no RCCL communicator"; return 1; }
        return nccl_check(Rccl::get().AllGather(send, recv, (size_t)bytes, ncclUint8, comm, stream), "ncclAllGather") ? 0 : 1;
    }
    void* alloc(size_t n) { void* p = nullptr; if (!check(hipMalloc(&p, n ? n : 1), "hipMalloc")) return nullptr; return p; }
    void free(void* p) { check(hipFree(p), "hipFree"); }
    // host memory for result downloads; static: results may outlive the session.  Ordinary (pageable) memory: the blocks are
    // recycled, so nothing is faulted in per call, and the host's passes over the candidate rows were measured ~25 % faster than
    // over page-locked blocks (112 vs 132 ms per step in round 1) while the download itself took the same time.
    static void* host_alloc(size_t n) { return malloc(n ? n : 1); }
    static void host_free(void* p) { ::free(p); }
    void memset(void* p, int v, size_t n) {
        if (n >= ((size_t)1 << 20) && ((uintptr_t)p & 15) == 0) {      // the large tables: own fill kernel
            const int64_t n16 = (int64_t)(n >> 4);
            const uint32_t w = 0x01010101u * (uint32_t)(uint8_t)v;
            hipLaunchKernelGGL(pm_fill16, dim3((unsigned)((n16 + 4095) / 4096)), dim3(256), 0, stream, (uint4*)p, make_uint4(w, w, w, w), n16);
            check(hipGetLastError(), "pm_fill16");
            const size_t done = (size_t)n16 << 4;
            if (done < n) check(hipMemsetAsync((char*)p + done, v, n - done, stream), "hipMemsetAsync");
            return;
        }
        check(hipMemsetAsync(p, v, n, stream), "hipMemsetAsync");
    }
    // several clears in one launch (engine_core.h: the tables and counters a search clears before it starts)
    void clear_many(const pm::ClearJob* jobs, int n) {
        FillJobs f; f.n = 0; unsigned int blocks = 0;
        for (int i = 0; i < n; i++) {
            if (!jobs[i].bytes) continue;
            if (f.n == FillJobs::kMax) { launch_fill(f, blocks); f.n = 0; blocks = 0; }
            f.p[f.n] = jobs[i].p; f.bytes[f.n] = jobs[i].bytes; f.value[f.n] = 0x01010101u * (unsigned int)(uint8_t)jobs[i].value; f.first_block[f.n] = blocks;
            blocks += (unsigned int)((jobs[i].bytes + 65535) >> 16);
            f.n++;
        }
        if (f.n) launch_fill(f, blocks);
    }
    void launch_fill(FillJobs& f, unsigned int blocks) {
        f.first_block[f.n] = blocks;
        hipLaunchKernelGGL(pm_fill_many, dim3(blocks), dim3(256), 0, stream, f);
        check(hipGetLastError(), "pm_fill_many");
    }
    // bytes moved over the host link by this session (pm_session_traffic): every copy below adds its size
    std::atomic<uint64_t> bytes_h2d{0}, bytes_d2h{0};
    // PARSNP_COPY_LOG=1: every copy over the host link with its size and the time the caller spent in it (stderr)
    static bool copy_log() { static const bool on = getenv("PARSNP_COPY_LOG") != nullptr; return on; }
    static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    // Small uploads (row lists, counters: a few KB, ~10 per step) go through a page-locked ring and are queued without waiting: from
    // ordinary memory the runtime stages the copy itself and the call returns only when it is done -- 15-25 us each on an idle
    // stream.  The ring is reused from its start when full, after one wait for the stream.
    uint8_t* ring = nullptr; size_t ring_at = 0; static constexpr size_t kRing = (size_t)4 << 20;
    void h2d(void* d, const void* s, size_t n) {
        const double t0 = copy_log() ? now_us() : 0;
        bytes_h2d += n;
        if (n && n <= kRing / 4) {
            if (!ring && hipHostMalloc((void**)&ring, kRing, hipHostMallocDefault) != hipSuccess) ring = nullptr;
            if (ring) {
                const size_t need = (n + 63) & ~(size_t)63;
                if (ring_at + need > kRing) { check(hipStreamSynchronize(stream), "sync"); ring_at = 0; }
                memcpy(ring + ring_at, s, n);
                check(hipMemcpyAsync(d, ring + ring_at, n, hipMemcpyHostToDevice, stream), "hipMemcpy H2D (ring)");
                ring_at += need;
                if (copy_log()) fprintf(stderr, "[copy] h2d (ring) %8zu B %8.1f us\n", n, now_us() - t0);
                return;
            }
        }
        if (n) check(hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, stream), "hipMemcpy H2D"); check(hipStreamSynchronize(stream), "sync");
        if (copy_log()) fprintf(stderr, "[copy] h2d       %9zu B %8.1f us\n", n, now_us() - t0);
    }
    // Downloads.  The callers' result blocks are ordinary memory (host_alloc), and a hipMemcpyAsync into ordinary memory is not
    // asynchronous: the runtime waits for the stream, copies into a staging block of its own and from there to the destination, one
    // copy after the other -- the six result arrays of a search cost six round trips of ~30 us with the device idle (rocprofv3
    // --kernel-trace, round 6: 34 copies, 0.7 ms per step).  So a download lands in a page-locked ring, queued behind the kernels
    // like any other command, and sync() -- the one wait of a call -- moves what has arrived to where the caller wants it.
    uint8_t* dring = nullptr; size_t dring_at = 0; static constexpr size_t kDRing = (size_t)48 << 20;
    struct Landing { void* dst; const uint8_t* at; size_t n; };
    std::vector<Landing> landings;
    void land() { for (const Landing& l : landings) memcpy(l.dst, l.at, l.n); landings.clear(); dring_at = 0; }
    // at the start of every call of the C ABI: what an earlier call queued and never waited for (it returned an error) is forgotten
    void drop_landings() { if (!landings.empty()) { (void)hipStreamSynchronize(stream); landings.clear(); dring_at = 0; } }
    void d2h_async(void* d, const void* s, size_t n) {
        const double t0 = copy_log() ? now_us() : 0;
        bytes_d2h += n;
        if (!n) return;
        const size_t need = (n + 255) & ~(size_t)255;
        if (!dring && hipHostMalloc((void**)&dring, kDRing, hipHostMallocDefault) != hipSuccess) dring = nullptr;
        if (dring && dring_at + need <= kDRing) {
            check(hipMemcpyAsync(dring + dring_at, s, n, hipMemcpyDeviceToHost, stream), "hipMemcpy D2H (ring)");
            landings.push_back(Landing{d, dring + dring_at, n});
            dring_at += need;
        } else check(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, stream), "hipMemcpy D2H");      // (too large for the ring: the runtime stages it)
        if (copy_log()) fprintf(stderr, "[copy] d2h_async %9zu B %8.1f us\n", n, now_us() - t0);
    }
    // ... into a block that IS page-locked (the chain's result block): straight there, for a caller that waits on an event
    void d2h_async_pinned(void* d, const void* s, size_t n) {
        bytes_d2h += n; if (n) check(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, stream), "hipMemcpy D2H (pinned)");
    }
    void d2h(void* d, const void* s, size_t n) {
        const double t0 = copy_log() ? now_us() : 0;
        d2h_async(d, s, n); sync();
        if (copy_log()) fprintf(stderr, "[copy] d2h       %9zu B %8.1f us (with the queued work before it)\n", n, now_us() - t0);
    }
    void sync() { check(hipStreamSynchronize(stream), "hipStreamSynchronize"); land(); }
    // an event on the engine's stream that any host thread may wait for (the slices of a row table in flight)
    void* event_record() {
        hipEvent_t e = nullptr;
        if (!pool.empty()) { e = pool.back(); pool.pop_back(); }
        else if (!check(hipEventCreate(&e), "hipEventCreate")) return nullptr;
        check(hipEventRecord(e, stream), "hipEventRecord");
        return (void*)e;
    }
    static void event_wait(void* e) { if (e) (void)hipEventSynchronize((hipEvent_t)e); }
    void event_release(void* e) { if (e) pool.push_back((hipEvent_t)e); }
    // page-locked block for the request rows of a call (kept, grown on demand); h2d_staged queues the DMA without waiting:
    // the block is not written again before the call's last synchronisation
    void* stage_p = nullptr; size_t stage_cap = 0;
    void* staging(size_t n) {
        if (n > stage_cap) {
            if (stage_p) (void)hipHostFree(stage_p);
            stage_p = nullptr; stage_cap = 0;
            if (!check(hipHostMalloc(&stage_p, n + n / 4 + 4096, hipHostMallocDefault), "hipHostMalloc(staging)")) return nullptr;
            stage_cap = n + n / 4 + 4096;
        }
        return stage_p;
    }
    void d2d(void* d, const void* s, size_t n) { if (n) check(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, stream), "hipMemcpy D2D"); }
    void* pinned_alloc(size_t n) { void* p = nullptr; if (!check(hipHostMalloc(&p, n ? n : 1, hipHostMallocDefault), "hipHostMalloc")) return nullptr; return p; }
    void pinned_free(void* p) { if (p) (void)hipHostFree(p); }
    void h2d_staged(void* d, const void* s, size_t n) {
        bytes_h2d += n; if (n) check(hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, stream), "hipMemcpy H2D (staged)");
        if (getenv("PARSNP_COPY_LOG")) fprintf(stderr, "[copy] h2d_staged %8zu B\n", n);
    }

    // copy threads of the genome upload: the staging copy into page-locked memory (~5 GB/s per thread) is what the upload waits for
    static int upload_threads() { const char* e = getenv("PARSNP_UPLOAD_THREADS"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= 32 ? v : 4; }
    // see Engine::load_genomes.  kThreads host threads, two staging slots each (page-locked host block + device block + stream)
    bool stage_genomes(int n, const uint8_t* const* seqs, const int64_t* lens, const std::vector<char>& take, const std::vector<int64_t>& goff,
                       pm::SeqBlock* blk, int64_t maxlen) {
        int64_t total = 0;
        for (int g = 0; g < n; g++) if (take[(size_t)g]) total += lens[g];
        const int kThreads = total < (8 << 20) ? 1 : upload_threads();      // a handful of short sequences: one thread, two slots
        constexpr int kPer = 2;
        int device = 0;
        if (!check(hipGetDevice(&device), "hipGetDevice")) return false;
        check(hipStreamSynchronize(stream), "sync");      // the memset of the packed buffer precedes the pack kernels
        struct Slot { uint8_t* host = nullptr; uint8_t* dev = nullptr; hipStream_t st = nullptr; hipEvent_t done = nullptr; bool used = false; };
        std::vector<Slot> slots((size_t)kThreads * kPer);
        bool ok = true;
        const size_t cap = (size_t)std::max<int64_t>(maxlen, 1);
        for (auto& sl : slots)
            ok = ok && hipHostMalloc((void**)&sl.host, cap, hipHostMallocDefault) == hipSuccess && hipMalloc((void**)&sl.dev, cap) == hipSuccess &&
                 hipStreamCreateWithFlags(&sl.st, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&sl.done, hipEventDisableTiming) == hipSuccess;
        std::vector<int> failed((size_t)kThreads, 0);
        if (ok) {
            auto work = [&](int t) {
                if (hipSetDevice(device) != hipSuccess) { failed[(size_t)t] = 1; return; }
                int turn = 0;
                for (int g = t; g < n; g += kThreads) {
                    if (!take[(size_t)g]) continue;
                    Slot& sl = slots[(size_t)t * kPer + (size_t)(turn++ % kPer)];
                    if (sl.used && hipEventSynchronize(sl.done) != hipSuccess) { failed[(size_t)t] = 1; return; }
                    memcpy(sl.host, seqs[g], (size_t)lens[g]);
                    bytes_h2d += (uint64_t)lens[g];
                    if (hipMemcpyAsync(sl.dev, sl.host, (size_t)lens[g], hipMemcpyHostToDevice, sl.st) != hipSuccess) { failed[(size_t)t] = 1; return; }
                    const int64_t nblk = (lens[g] + 31) / 32;
                    for (int s = 0; s < 2; s++)
                        hipLaunchKernelGGL(pm_kernel<pm::PackStrand>, dim3((unsigned)((nblk + 255) / 256)), dim3(256), 0, sl.st,
                                           pm::PackStrand{sl.dev, lens[g], s, blk, goff[2 * (size_t)g + (size_t)s] / 32}, nblk);
                    if (hipGetLastError() != hipSuccess || hipEventRecord(sl.done, sl.st) != hipSuccess) { failed[(size_t)t] = 1; return; }
                    sl.used = true;
                }
            };
            std::vector<std::thread> th;
            for (int t = 1; t < kThreads; t++) th.emplace_back(work, t);
            work(0);
            for (auto& x : th) x.join();
        }
        for (auto& sl : slots) {
            if (sl.st) { if (hipStreamSynchronize(sl.st) != hipSuccess) ok = false; (void)hipStreamDestroy(sl.st); }
            if (sl.done) (void)hipEventDestroy(sl.done);
            if (sl.host) (void)hipHostFree(sl.host);
            if (sl.dev) (void)hipFree(sl.dev);
        }
        for (int f : failed) if (f) ok = false;
        if (!ok && err.empty()) err = "staging the genomes failed (allocation, copy or PackStrand launch)";
        return ok;
    }

    template <class F> void launch(const char* name, int64_t n, F f) {
        if (n <= 0) return;
        int64_t blocks = (n + 255) / 256;
        if (blocks > 0x7fffffffll) { if (err.empty()) err = std::string("grid too large: ") + name; return; }
        hipLaunchKernelGGL(pm_kernel<F>, dim3((unsigned)blocks), dim3(256), 0, stream, f, n);
        check(hipGetLastError(), name);
    }
    template <class F> void launch_wave(const char* name, int64_t n, F f) {
        if (n <= 0) return;
        if (n > 0x7fffffffll) { if (err.empty()) err = std::string("grid too large: ") + name; return; }
        hipLaunchKernelGGL(pm_wave_kernel<F>, dim3((unsigned)n), dim3(64), 0, stream, f, n);
        check(hipGetLastError(), name);
    }
    void need_tmp(size_t n) {
        if (n <= tmp_cap) return;
        if (tmp) (void)hipFree(tmp);
        tmp = nullptr; tmp_cap = 0;
        if (check(hipMalloc(&tmp, n + n / 4 + 256), "hipMalloc(tmp)")) tmp_cap = n + n / 4 + 256;
    }
    void exclusive_scan(const int64_t* in, int64_t* out, size_t n) {
        size_t bytes = 0;
        check(rocprim::exclusive_scan(nullptr, bytes, in, out, (int64_t)0, n, rocprim::plus<int64_t>(), stream), "scan size");
        need_tmp(bytes);
        check(rocprim::exclusive_scan(tmp, bytes, in, out, (int64_t)0, n, rocprim::plus<int64_t>(), stream), "exclusive_scan");
    }
    void sort_pairs(uint64_t* ki, uint64_t* ko, uint64_t* vi, uint64_t* vo, size_t n, int bits) {
        size_t bytes = 0;
        check(rocprim::radix_sort_pairs(nullptr, bytes, ki, ko, vi, vo, n, 0, (unsigned)bits, stream), "sort size");
        need_tmp(bytes);
        check(rocprim::radix_sort_pairs(tmp, bytes, ki, ko, vi, vo, n, 0, (unsigned)bits, stream), "radix_sort_pairs");
    }

    // phase timing: mark(name) opens a phase, mark(nullptr) closes the last one
    bool timing_on = true;      // pm_session_tune(s, "timing", 0): no events, collect() only waits for the stream
    void mark(const char* name) {
        if (!timing_on) return;
        hipEvent_t e;
        if (!pool.empty()) { e = pool.back(); pool.pop_back(); }
        else if (!check(hipEventCreate(&e), "hipEventCreate")) return;
        check(hipEventRecord(e, stream), "hipEventRecord");
        marks.emplace_back(name, e);
    }
    std::vector<pm::PhaseTime> collect() {
        std::vector<pm::PhaseTime> out;
        sync();
        for (size_t i = 0; i + 1 < marks.size(); i++) {
            if (!marks[i].first) continue;
            float ms = 0;
            check(hipEventElapsedTime(&ms, marks[i].second, marks[i + 1].second), "hipEventElapsedTime");
            bool merged = false;
            for (auto& t : out) if (!strcmp(t.name, marks[i].first)) { t.ms += ms; merged = true; }
            if (!merged) out.push_back(pm::PhaseTime{marks[i].first, ms});
        }
        for (auto& m : marks) pool.push_back(m.second);
        marks.clear();
        return out;
    }
};

}  // namespace

typedef HipBackend PmBackend;
static const char* pm_backend_name = "hip";
static PmBackend* pm_backend_open(int device, std::string* err) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) {
        *err = std::string("no HIP device available (") + (e == hipSuccess ? "0 devices" : hipGetErrorString(e)) + "); this engine has no CPU path";
        return nullptr;
    }
    if (device < 0) {   // PARSNP_DEVICE: the GPU of this process when the caller cannot pass one (one process per GPU)
        const char* e = getenv("PARSNP_DEVICE");
        if (e && *e) device = atoi(e);
    }
    if (device >= 0) {
        if (device >= count || hipSetDevice(device) != hipSuccess) { *err = "cannot select the requested HIP device"; return nullptr; }
    }
    HipBackend* b = new HipBackend;
    (void)hipGetDevice(&b->device);
    if (hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess) { *err = "hipStreamCreate failed"; delete b; return nullptr; }
    return b;
}
#define PM_HAVE_RCCL 1
#include "abi_glue.h"
