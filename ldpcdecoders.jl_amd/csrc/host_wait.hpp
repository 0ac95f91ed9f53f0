// host_wait.hpp -- every place where the HOST waits for the device, with a bound.
//
// Device-side waits have been bounded by the 100 MHz clock since round 1 (team barrier 10 s, roll call 20 ms,
// bp_team_kernels.hpp); the host's own waits were plain hipDeviceSynchronize / hipEventSynchronize /
// hipStreamSynchronize calls, which block for as long as the device does not drain -- a stalled queue (a reset of the
// host's GPUs by another tenant's fault, a driver hiccup) blocks the caller for ever and says nothing (round 3's
// fuzz run of seed 5151 was killed at its time limit after 147 s of silence; the cause could not be established from
// the records, see DESIGN.md "The stalled fuzz run").  Now every such wait polls (hipEventQuery / hipStreamQuery) against a deadline, or
// -- hipDeviceSynchronize and the frees that synchronise implicitly have no query form -- runs in a helper thread the
// caller waits for with a deadline.  On expiry the call returns LDPC_ERR_HIP naming the wait, the device is marked
// STALLED for this process (every later entry on it fails at once with the same message instead of queueing behind
// the stall; memory the device may still be using is leaked rather than freed), and the helper thread, if any, is left
// behind detached.  The limit is per process: ldpc_set_wait_limit_ms() (include/ldpc_mi355x.h), default 600 s, 0 = wait
// for ever as before.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/ldpc_mi355x.h"

namespace ldpc_detail {

ldpc_status set_error(ldpc_status st, const std::string &msg);   // ldpc_mi355x.hip: what ldpc_last_error() returns

int64_t wait_limit_ms();                                  // 0 = no bound
bool device_stalled(int device);                          // a wait on this device has expired in this process
ldpc_status stalled_error(int device);                    // ... the status every later entry returns (message kept)

// LDPC_OK, or LDPC_ERR_HIP with "<what>: ..." (a HIP error, or the bound)
ldpc_status wait_event(hipEvent_t e, int device, const char *what);
ldpc_status wait_stream(hipStream_t s, int device, const char *what);
ldpc_status wait_device(int device, const char *what);    // hipSetDevice(device) + hipDeviceSynchronize, bounded
// cleanup paths (destructors): true = the device has drained and what it used may be freed
inline bool device_idle_for_release(int device, const char *what) { return wait_device(device, what) == LDPC_OK; }

}  // namespace ldpc_detail
