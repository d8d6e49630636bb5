// Fault-injection switches and kernel ablations exist only in the instrumented build (librsba_amd_hooks.so, -DRSBA_TEST_HOOKS:
// rsba_amd/csrc/Makefile); the release library does not read them and carries no branch for them.
#pragma once
#include <cstdlib>

namespace rsba {
#ifdef RSBA_TEST_HOOKS
inline const char* test_hook(const char* name) { return std::getenv(name); }
constexpr bool kTestHooks = true;
#else
inline const char* test_hook(const char*) { return nullptr; }
constexpr bool kTestHooks = false;
#endif
}  // namespace rsba
