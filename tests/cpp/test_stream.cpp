// StreamFlow (structure-of-arrays ring, better_flow/stream_flow.h) against DVS_flow (the reference's AoS ring,
// better_flow/dvs_flow.h) on the same event stream: same trigger points, same slices (size, span trimming, the
// full-ring quirk, wrap-around into two pieces), same models, same per-event flow.  The ring is small so that
// it fills and wraps many times.  Prints one line per slice and a verdict; exit code 1 on any difference.
#include <better_flow/common.h>
#include <better_flow/dvs_flow.h>
#include <better_flow/event_reader.h>
#include <better_flow/stream_flow.h>
#include <cstdio>
#include <cstring>

template <size_t MAX_SZ, sll SPAN> static int run(const char *path, ull on_ev, ull on_time, int max_iter, bool stm_off) {
    DVS_flow<MAX_SZ, SPAN> dvs(on_ev, on_time);
    bf::StreamFlow<MAX_SZ, SPAN> sf(on_ev, on_time);
    dvs.set_quiet(true);
    dvs.set_max_iter(max_iter); sf.set_max_iter(max_iter);
    if (stm_off) { dvs.set_stm_disable(true); sf.set_stm_disable(true); }
    bf::EventReader reader(path);
    int bad = 0, slices = 0;
    unsigned long long n_ev = 0;
    reader.for_each_event([&](unsigned row, unsigned col, unsigned long long t_ns) {
        Event e(row, col, (ull)t_ns);
        const bool a = dvs.add_event(e);
        const bool b = sf.add_event(row, col, (ull)t_ns);
        ++n_ev;
        if (a != b) { if (bad++ < 5) std::printf("event %llu: trigger %d vs %d\n", n_ev, (int)a, (int)b); return; }
        if (!a) return;
        ++slices;
        ObjectModel m1 = dvs.get_last_model(), m2 = sf.get_last_model();
        bf_model a1 = m1.to_abi(), a2 = m2.to_abi();
        const bool same_model = std::memcmp(&a1, &a2, sizeof(a1)) == 0;
        const size_t sz1 = (size_t)dvs.get_buf_size(), sz2 = sf.size();
        size_t diff = 0, visited = 0;
        for (auto &ev : dvs.ev_buffer) {   // newest -> oldest, stops one short on a full ring
            const size_t i = visited++;
            const double su = sf.u(i), sv = sf.v(i);
            if (ev.fr_x != sf.row(i) || ev.fr_y != sf.col(i) || ev.timestamp != sf.timestamp(i) ||
                std::memcmp(&ev.best_u, &su, 8) != 0 || std::memcmp(&ev.best_v, &sv, 8) != 0)
                ++diff;
        }
        std::printf("slice %d at event %llu: ring %zu / %zu, iterated %zu, iterations %d, dx %.9g dy %.9g, %s, flow diffs %zu\n", slices, n_ev,
                    sz1, sz2, visited, sf.get_run_info().iterations, m2.total_dx, m2.total_dy, same_model ? "model ==" : "MODEL !=", diff);
        if (!same_model || sz1 != sz2 || diff) ++bad;
    });
    std::printf("%s: %d slices, %d problems (ring %zu, span %lld ns)\n", bad ? "FAIL" : "OK", slices, bad, (size_t)MAX_SZ, (long long)SPAN);
    return bad;
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    int bad = 0;
    bad += run<3000, 40000000>(argv[1], 1500, FROM_SEC(0.02), -1, false);    // fills, wraps, span-trims; STM chain
    bad += run<3000, 15000000>(argv[1], 1000, FROM_SEC(0.5), -1, false);      // the 15 ms span trims the ring before it fills
    bad += run<50000, 200000000>(argv[1], 4000, FROM_SEC(0.033), 10, true);   // never full; capped, STM off
    return bad ? 1 : 0;
}
