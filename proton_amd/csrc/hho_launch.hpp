// hho_launch.hpp -- host-side registry of the instantiated local-operator kernels.
#pragma once
#include <hip/hip_runtime.h>
#include "hho_device.hpp"
#include "hho_pre.hpp"
#include "hho_small.hpp"

namespace pa {

typedef hipError_t (*local_ops_launcher)(const LocalOpsArgs &, int grid, hipStream_t);
typedef hipError_t (*pre_launcher)(const PreArgs &, hipStream_t);
typedef hipError_t (*small_launcher)(const SmallOpsArgs &, hipStream_t);

struct KernelEntry {
    int cd, fd, quad, stab, lanes_per_cell;
    local_ops_launcher launch;        // lc only
    local_ops_launcher launch_split;  // any of lc / data / stab
    const void *func;          // for the occupancy query
    int lds_bytes;
    const char *name;
    int waves_per_simd;        // the occupancy the instance is tuned for (Cfg::WAVES): the grid does not exceed it
    pre_launcher launch_pre;   // the one-thread-per-cell pre-pass the kernel consumes (nullptr: all-in-one kernel)
    int pre_doubles;           // doubles per cell of its record (Cfg::Pre::NPRE)
    small_launcher launch_small;   // msize <= 9: the thread-per-cell kernel of hho_small.hpp (lc + info only; nullptr otherwise)
    const char *name_small;
    int self_pre;              // Cfg::SELF_PRE: the cooperative kernel forms the records itself, into one ring of 64 per block (no pre-pass launch)
    // condensed mode (static condensation fused behind the product; nullptr without a stabilization: A_TT singular)
    local_ops_launcher launch_cond;
    const void *func_cond;
    int lds_bytes_cond;
    const char *name_cond;
    int waves_per_simd_cond;
};

template <class C, int MODE>
hipError_t launch_local_ops(const LocalOpsArgs &a, int grid, hipStream_t s)
{
    hipLaunchKernelGGL((hho_local_ops_kernel<C, MODE>), dim3(grid), dim3(64), C::LDS_DOUBLES * sizeof(double), s, a);
    return hipGetLastError();
}

template <class C>
hipError_t launch_pre(const PreArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL((hho_cell_pre_kernel<C>), dim3((unsigned)((a.n + 63) / 64)), dim3(64), 0, s, a);
    return hipGetLastError();
}

}  // namespace pa
