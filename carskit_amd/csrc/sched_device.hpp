// sched_device.hpp -- see sched_device.hip
#pragma once
#include <stdint.h>

#include "level_schedule.hpp"

namespace cmi {

// build_chain_schedule (level_schedule.hpp) computed on `device`: the same ChainSchedule, element for element.  u / j are HOST arrays
// (uploaded here).  false = not built (a HIP failure, more hub rows than the resident grid can own): the caller uses the host builder.
// keep != nullptr: the uploaded tuple ids and the permutation STAY on the device for the caller (who frees them with hipFree) and
// out.perm is left empty -- the tuple stream is then built on the device too (stream_build_device), nothing of size n goes back to the host.
struct ChainDeviceKeep { // owns the three device arrays: whatever path leaves cmi_set_ratings (error, exception), they are freed
    int32_t *d_u = nullptr, *d_j = nullptr, *d_perm = nullptr;
    ChainDeviceKeep() = default;
    ChainDeviceKeep(const ChainDeviceKeep &) = delete;
    ChainDeviceKeep &operator=(const ChainDeviceKeep &) = delete;
    void release();            // hipFree the arrays now (sched_device.hip)
    ~ChainDeviceKeep() { release(); }
};
bool build_chain_schedule_device(int device, void *stream, int64_t n, const int32_t *u, const int32_t *j, int32_t n_users, int32_t n_items, int hub,
                                 int max_chain, ChainSchedule &out, ChainDeviceKeep *keep = nullptr);

// The tuple stream in schedule order on the device (what cmi_set_ratings' host loop builds): su / sj / sr = u / j / rating of tuple
// perm[s], sconds[s * dmax ..] = the conditions of its context, -1 padded.  ctx / r are HOST arrays (uploaded here); d_sr is float or
// double (f64).  Returns a hipError_t.
int stream_build_device(void *stream, int64_t n, const int32_t *d_u, const int32_t *d_j, const int32_t *d_perm, const int32_t *ctx, const double *r,
                        const int32_t *d_ctx_ptr, const int32_t *d_ctx_conds, int dmax, bool f64, int32_t *d_su, int32_t *d_sj, int32_t *d_sconds,
                        void *d_sr);

// arena_positions (cmi_api.cpp) on the device: d_spoke_stream = the spoke row of every stream position (device), d_next / d_first device
// outputs (n / n_spokes entries).  Returns a hipError_t.
int arena_positions_device(void *stream, int64_t n, const int32_t *d_spoke_stream, int64_t n_spokes, int32_t *d_next, int32_t *d_first);

} // namespace cmi
