// debug_build.h -- what separates libwhatshap_amd.so (the product) from libwhatshap_amd_debug.so (test infrastructure: the same sources with
// -DWHAMD_DEBUG_BUILD).  The debug build adds the CPU plan emulators and the host instantiation of the heuristic (whatshap_amd_debug.h), the
// kernel instantiations with in-kernel cycle stamps, and the environment switches of timing experiments -- some of which make results INVALID
// (WHAMD_SLOT_SKIP).  The product library does not read them: debug_env() is a constant there and the strings are not even linked in.
// What the product does read from the environment is listed in INTEGRATION.md (thread counts, NUMA binding, window sizes, alternative exact paths).
#pragma once
#include <cstdlib>

namespace whamd {
#ifdef WHAMD_DEBUG_BUILD
constexpr bool DEBUG_BUILD = true;
inline const char* debug_env(const char* name) { return getenv(name); }
#else
constexpr bool DEBUG_BUILD = false;
inline const char* debug_env(const char*) { return nullptr; }
#endif
}  // namespace whamd
