// sigprof.c -- a sampling profiler for the host side of a bench run, for boxes without perf:
//   gcc -O2 -shared -fPIC scripts/sigprof.c -o /tmp/sigprof.so
//   LD_PRELOAD=/tmp/sigprof.so SIGPROF_OUT=gpurun_out/prof python bench.py ...      (writes gpurun_out/prof.<pid>)
//   python scripts/sigprof_report.py gpurun_out/prof.<pid>
// ITIMER_PROF ticks on the CPU time of the whole process (all threads); the handler runs on the thread that used the
// time and notes the interrupted program counter and the time.  At exit: the samples and /proc/self/maps, for the report
// to turn into (library, symbol) counts.  A measurement tool: nothing of the product links it.
#define _GNU_SOURCE
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/syscall.h>
#include <sys/time.h>
#include <time.h>
#include <ucontext.h>
#include <unistd.h>

#define CAP (1u << 21)
static struct { uint64_t pc; uint32_t tid; uint32_t ms; } *g_samples;
static volatile uint32_t g_count;
static struct timespec g_t0;

static void on_prof(int sig, siginfo_t* si, void* uc_) {
    (void)sig; (void)si;
    ucontext_t* uc = (ucontext_t*)uc_;
    uint32_t i = __atomic_fetch_add(&g_count, 1, __ATOMIC_RELAXED);
    if (i >= CAP) return;
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    g_samples[i].pc = (uint64_t)uc->uc_mcontext.gregs[REG_RIP];
    g_samples[i].tid = (uint32_t)syscall(SYS_gettid);
    g_samples[i].ms = (uint32_t)((ts.tv_sec - g_t0.tv_sec) * 1000 + (ts.tv_nsec - g_t0.tv_nsec) / 1000000);
}

__attribute__((constructor)) static void start(void) {
    const char* out = getenv("SIGPROF_OUT");
    if (!out) return;
    g_samples = calloc(CAP, sizeof *g_samples);
    clock_gettime(CLOCK_MONOTONIC, &g_t0);
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_prof;
    sa.sa_flags = SA_SIGINFO | SA_RESTART;
    sigaction(SIGPROF, &sa, NULL);
    const char* us = getenv("SIGPROF_US");
    struct itimerval it;
    it.it_interval.tv_sec = 0; it.it_interval.tv_usec = us ? atoi(us) : 1000;
    it.it_value = it.it_interval;
    setitimer(ITIMER_PROF, &it, NULL);
}

__attribute__((destructor)) static void stop(void) {
    const char* out = getenv("SIGPROF_OUT");
    if (!out || !g_samples) return;
    struct itimerval it;
    memset(&it, 0, sizeof it);
    setitimer(ITIMER_PROF, &it, NULL);
    char path[1024];
    snprintf(path, sizeof path, "%s.%d", out, (int)getpid());      // one file per process (timeout, python, children)
    if (g_count == 0) return;
    FILE* f = fopen(path, "w");
    if (!f) return;
    FILE* m = fopen("/proc/self/maps", "r");
    char line[1024];
    while (m && fgets(line, sizeof line, m)) if (strstr(line, " r-xp ") || strstr(line, " r--p ")) fprintf(f, "M %s", line);
    if (m) fclose(m);
    uint32_t n = g_count < CAP ? g_count : CAP;
    for (uint32_t i = 0; i < n; i++) fprintf(f, "S %llx %u %u\n", (unsigned long long)g_samples[i].pc, g_samples[i].tid, g_samples[i].ms);
    fclose(f);
}
