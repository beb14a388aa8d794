// hooks.h -- test hooks of the host code.
// Environment switches that force the rarely taken or the plain variant of a host shortcut -- same bytes by construction,
// compared in tests/ -- exist only in builds with -DPARSNP_TEST_HOOKS: the test binaries (tests/emu, oracle/_ref and
// parsnp_amd/bin/parsnp_core_hooks, the product's sources with the hooks compiled in).  The shipped parsnp_core reads only
// the switches INTEGRATION.md lists (device / sharding / RCCL / timing).
#pragma once
#include <cstdlib>

namespace parsnp {
#if defined(PARSNP_TEST_HOOKS)
inline const char* test_hook(const char* name) { return getenv(name); }
#else
inline const char* test_hook(const char*) { return nullptr; }
#endif
}  // namespace parsnp
