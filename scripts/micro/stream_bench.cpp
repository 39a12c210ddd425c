// Host cost of forming and handing over 1M-event slices: DVS_flow (AoS ring + repack per slice) vs StreamFlow
// (pinned SoA ring, DMA hand-off).  Feeds the same binary event file several times over with shifted timestamps.
//   g++ -O2 -std=c++14 -Ibetter_flow_amd/host -Iinclude scripts/micro/stream_bench.cpp -Lbetter_flow_amd -lbf_accel ...
#include <better_flow/common.h>
#include <better_flow/dvs_flow.h>
#include <better_flow/event_reader.h>
#include <better_flow/stream_flow.h>
#include <chrono>
#include <cstdio>
#include <vector>

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    bf::sensor().res_x = 260; bf::sensor().res_y = 346;
    std::vector<unsigned> rows, cols; std::vector<unsigned long long> ts;
    bf::EventReader reader(argv[1]);
    reader.for_each_event([&](unsigned r, unsigned c, unsigned long long t) { rows.push_back(r); cols.push_back(c); ts.push_back(t); });
    const unsigned long long span = ts.back() + 1000;
    const int reps = 8;
    constexpr size_t MAX_SZ = 1100000;
    constexpr sll SPAN = 30000000;
    std::printf("%zu events per pass, %d passes\n", ts.size(), reps);
    {
        bf::StreamFlow<MAX_SZ, SPAN> sf(1u << 30, FROM_SEC(0.030));
        sf.set_want_flow(true);
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; ++r)
            for (size_t i = 0; i < ts.size(); ++i) sf.add_event(rows[i], cols[i], ts[i] + r * span);
        sf.recompute();
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("StreamFlow: %llu slices, %.1f ms per slice, %.1f Mev/s end to end (iterations %llu)\n", sf.get_slices_done(),
                    1e3 * dt / sf.get_slices_done(), reps * ts.size() / dt / 1e6, sf.get_iterations_total());
    }
    {
        static DVS_flow<MAX_SZ, SPAN> dvs(1u << 30, FROM_SEC(0.030));
        dvs.set_quiet(true);
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; ++r)
            for (size_t i = 0; i < ts.size(); ++i) { Event e(rows[i], cols[i], ts[i] + r * span); dvs.add_event(e); }
        dvs.recompute();
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("DVS_flow:   %llu slices, %.1f ms per slice, %.1f Mev/s end to end (iterations %llu)\n", dvs.get_slices_done(),
                    1e3 * dt / dvs.get_slices_done(), reps * ts.size() / dt / 1e6, dvs.get_iterations_total());
    }
    bf::DeviceContext::release();
    return 0;
}
