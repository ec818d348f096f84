// State that HIP keeps per DEVICE, cached per device index.  hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the function
// on the CURRENT device only, and the CU count sizes persistent grids: a process that drives several GPUs (one rank per device is
// the normal case, but the multi-rank tests and heterogeneous nodes exist) must not reuse the first device's answers.
#pragma once
#include <hip/hip_runtime.h>

namespace liso_dev {

constexpr int kMaxDev = 64;

struct PerDeviceFlag {
    bool set[kMaxDev] = {};
};

// runs fn() (-> true on success) the first time it is reached with each device current; false on any error
template <class F>
inline bool once_per_device(PerDeviceFlag& f, F&& fn) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return false;
    if (!f.set[dev]) {
        if (!fn()) return false;
        f.set[dev] = true;
    }
    return true;
}

// opt a kernel into `bytes` of dynamic LDS on the current device (once per device)
inline bool lds_opt_in(PerDeviceFlag& f, const void* kernel, int bytes) {
    return once_per_device(f, [&] { return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess; });
}

// compute units of the current device (0 on error)
inline int cu_count() {
    static int n_cu[kMaxDev] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return 0;
    if (!n_cu[dev]) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        n_cu[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return n_cu[dev];
}

}  // namespace liso_dev
