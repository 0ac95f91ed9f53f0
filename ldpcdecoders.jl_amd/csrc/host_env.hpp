// host_env.hpp -- environment knobs exist in the EXPERIMENTS build only.
//
// The product library (libldpc_mi355x.so) reads no environment variable: every tuning decision is made from the
// graph, the batch and the device.  The knobs DESIGN.md lists ("Environment knobs") -- thresholds, forced
// geometries, fault injection, logging -- are compiled in with -DLDPC_EXPERIMENTS only, which is how
// libldpc_mi355x_exp.so is built (`make exp`): the tests and tools that force a code path load that build.
#pragma once
#include <cstdlib>

namespace ldpc {

inline const char *exp_env(const char *name)
{
#ifdef LDPC_EXPERIMENTS
    return std::getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

constexpr bool kExperimentsBuild =
#ifdef LDPC_EXPERIMENTS
    true;
#else
    false;
#endif

}  // namespace ldpc
