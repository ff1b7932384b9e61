// hho_launch.hpp -- host-side registry of the instantiated local-operator kernels.
#pragma once
#include <hip/hip_runtime.h>
#include "hho_device.hpp"

namespace pa {

typedef hipError_t (*local_ops_launcher)(const LocalOpsArgs &, int grid, hipStream_t);

struct KernelEntry {
    int cd, fd, quad, stab, lanes_per_cell;
    local_ops_launcher launch;        // lc only
    local_ops_launcher launch_split;  // any of lc / data / stab
    const void *func;          // for the occupancy query
    int lds_bytes;
    const char *name;
    int waves_per_simd;        // the occupancy the instance is tuned for (Cfg::WAVES): the grid does not exceed it
};

template <class C, bool SPLIT>
hipError_t launch_local_ops(const LocalOpsArgs &a, int grid, hipStream_t s)
{
    hipLaunchKernelGGL((hho_local_ops_kernel<C, SPLIT>), dim3(grid), dim3(64), C::LDS_DOUBLES * sizeof(double), s, a);
    return hipGetLastError();
}

}  // namespace pa
