// latency_mode.hpp -- completion flag of the host-pointer entries' latency path (DESIGN.md "Latency
// path"): kernels launched with one workgroup per group on a host-mapped I/O image report through
// a flag word in that image instead of through the stream.
#pragma once
#include <hip/hip_runtime.h>

namespace ldpc {

// every thread's stores have left for system memory before the workgroup reports; the last workgroup
// re-arms the counter and raises the flag
__device__ __forceinline__ void publish_done(unsigned int *done_count, unsigned int *done_flag, unsigned int ticket)
{
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned prev = __hip_atomic_fetch_add(done_count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == gridDim.x - 1) {
            __hip_atomic_store(done_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(done_flag, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

}  // namespace ldpc
