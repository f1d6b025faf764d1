// CPU unit test of the per-device set-up guard of libgdyn (csrc/gdyn_once.hpp): one set-up per device ordinal, no caller gets past
// the guard before the set-up has RETURNED, the status is kept.  Built and run by tests/test_abi_and_farm.py (plain g++, -pthread).
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

#include "gdyn_once.hpp"

int main()
{
    gd::DeviceOnce<8> once;
    std::atomic<int> calls[8];
    std::atomic<int> finished[8];
    for (auto &c : calls) c = 0;
    for (auto &f : finished) f = 0;
    std::atomic<int> early{0};
    std::vector<std::thread> th;
    for (int t = 0; t < 32; t++)
        th.emplace_back([&, t] {
            const int dev = t % 4;
            const int rc = once.run(dev, [&](int d) {
                calls[d]++;
                std::this_thread::sleep_for(std::chrono::milliseconds(30));      // a slow set-up: the others must wait for it
                finished[d] = 1;
                return d == 3 ? 77 : 0;      // device 3's set-up "fails": every caller must see that status
            });
            if (!finished[dev]) early++;      // got past the guard before the set-up had returned
            if (rc != (dev == 3 ? 77 : 0)) early += 1000;
        });
    for (auto &t : th) t.join();
    int bad = early.load();
    for (int d = 0; d < 4; d++) if (calls[d] != 1 || once.runs(d) != 1) bad += 100000;
    for (int d = 4; d < 8; d++) if (calls[d] != 0 || once.runs(d) != 0) bad += 100000;
    if (once.run(8, [](int) { return 0; }) != -1 || once.run(-1, [](int) { return 0; }) != -1) bad += 1000000;
    if (once.run(3, [](int) { return 0; }) != 77 || calls[3] != 1) bad += 10000000;      // no second attempt, same status
    std::printf("once: %s (%d)\n", bad ? "FAILED" : "ok", bad);
    return bad ? 1 : 0;
}
