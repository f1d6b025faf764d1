// gdyn_once.hpp -- one-time set-up per device ordinal.
//
// Kernels that use more than 64 KB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize set before their first launch,
// on EVERY device they run on (code objects are loaded per device).  include/gdyn.h promises that independent handles may live on
// different threads and devices: the set-up therefore runs once per device ordinal under std::call_once -- a second thread that
// creates a handle on the same device while the first one is still inside the set-up waits for it to RETURN (no launch can precede
// it), a handle on another device runs its own -- and its status is kept: every later gd_create on that device sees the same result.
// Plain C++ (no HIP): tests/native/test_once.cpp exercises it on the CPU.
#pragma once
#include <mutex>

namespace gd {

template <int MAX_DEVICES = 64>
class DeviceOnce {
public:
    // runs setup(device) the first time `device` is seen (concurrent callers for the same device block until it has returned) and
    // returns its status, then and ever after; -1 for an ordinal outside [0, MAX_DEVICES)
    template <class F>
    int run(int device, F &&setup)
    {
        if (device < 0 || device >= MAX_DEVICES) return -1;
        std::call_once(_flag[device], [&] { _status[device] = setup(device); _runs[device]++; });
        return _status[device];
    }
    int runs(int device) const { return device >= 0 && device < MAX_DEVICES ? _runs[device] : 0; }

private:
    std::once_flag _flag[MAX_DEVICES];
    int _status[MAX_DEVICES] = {};
    int _runs[MAX_DEVICES] = {};
};

}  // namespace gd
