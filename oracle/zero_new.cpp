// TEST INFRASTRUCTURE. Linked only into oracle/_ref/parsnp_core_ref.
// The reference reads MasterRC[].UP before writing it (src/parsnp.cpp:1580 allocates, :1591-1597
// never initialise it, src/csgmum/mum.c:130,162 read it).  Fresh heap pages are zero, so the
// effective semantics are UP = 0; this shim makes that deterministic for every array new.
#include <cstdlib>
#include <new>
void* operator new[](std::size_t n) {
    void* p = std::calloc(n ? n : 1, 1);
    if (!p) throw std::bad_alloc();
    return p;
}
void operator delete[](void* p) noexcept { std::free(p); }
void operator delete[](void* p, std::size_t) noexcept { std::free(p); }
