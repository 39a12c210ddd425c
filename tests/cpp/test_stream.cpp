// StreamFlow / StreamEngine (structure-of-arrays ring + slice farm, better_flow/stream_flow.h) against DVS_flow (the
// reference's AoS ring, better_flow/dvs_flow.h) on the same event stream: same trigger points, same slices (size, span
// trimming, the full-ring quirk, wrap-around into two pieces), same models, same per-event flow, same accumulated
// (-o) table -- for events added one by one, in bulk blocks, pipelined, and with several workers on independent slices.
// The streams carry what the marking rule of get_accumulated and the noise flag are sensitive to: twin events (same
// pixel, same timestamp), some of them straddling a trigger, and a stretch confined to a small window.
// The ring is small so that it fills and wraps many times.  Prints one line per slice and a verdict per run; exit code 1
// on any difference.  Linked against the oracle shim (CPU) or libbf_accel.so (GPU) by tests/test_host_cli.py.
#include <better_flow/common.h>
#include <better_flow/dvs_flow.h>
#include <better_flow/event_reader.h>
#include <better_flow/stream_flow.h>
#include <cstdio>
#include <cstring>

struct Stream {
    std::vector<uint32_t> row, col;
    std::vector<ull> t;
};

// the file's events; every `twin_every`-th event is followed by a twin (same pixel, same instant); while t < confine_ns
// only events of the sensor's top-left 60 x 90 corner are kept
static Stream load(const char *path, unsigned twin_every, ull confine_ns) {
    Stream s;
    bf::EventReader reader(path);
    unsigned long long k = 0;
    reader.for_each_event([&](unsigned row, unsigned col, unsigned long long t_ns) {
        if (t_ns < confine_ns && !(row < 60 && col < 90)) return;
        s.row.push_back(row); s.col.push_back(col); s.t.push_back((ull)t_ns);
        if (twin_every && ++k % twin_every == 0) { s.row.push_back(row); s.col.push_back(col); s.t.push_back((ull)t_ns); }
    });
    return s;
}

static bool same_model(const ObjectModel &a, const ObjectModel &b) {
    bf_model x = a.to_abi(), y = b.to_abi();
    return std::memcmp(&x, &y, sizeof(x)) == 0;
}

struct Rec {
    uint64_t first, n;
    int rc, iterations;
    ObjectModel model;
};

template <size_t MAX_SZ, sll SPAN>
static int run(const char *tag, const Stream &in, ull on_ev, ull on_time, int max_iter, bool stm_off, int workers) {
    int bad = 0, slices = 0;
    // ---- A: event by event into both front ends, compared after every slice
    DVS_flow<MAX_SZ, SPAN> dvs(on_ev, on_time);
    bf::StreamFlow<MAX_SZ, SPAN> sf(on_ev, on_time);
    dvs.set_quiet(true);
    dvs.set_max_iter(max_iter); sf.set_max_iter(max_iter);
    dvs.set_accumulate(); sf.set_accumulate();
    if (stm_off) { dvs.set_stm_disable(true); sf.set_stm_disable(true); }
    std::vector<Rec> recs;
    int skipped_guard = 0;
    sf.on_slice([&](const bf::SliceRecord &r) {
        recs.push_back(Rec{r.first_event, r.events, r.rc, (int)r.info.iterations, r.model});
        skipped_guard += r.window_guard ? 1 : 0;
    });
    for (size_t i = 0; i < in.t.size(); ++i) {
        Event e(in.row[i], in.col[i], in.t[i]);
        const bool a = dvs.add_event(e);
        const bool b = sf.add_event(in.row[i], in.col[i], in.t[i]);
        if (a != b) { if (bad++ < 5) std::printf("event %zu: trigger %d vs %d\n", i, (int)a, (int)b); continue; }
        if (!a) continue;
        ++slices;
        ObjectModel m1 = dvs.get_last_model(), m2 = sf.get_last_model();
        const size_t sz1 = (size_t)dvs.get_buf_size(), sz2 = sf.size();
        size_t diff = 0, visited = 0;
        for (auto &ev : dvs.ev_buffer) {   // newest -> oldest, stops one short on a full ring
            const size_t k = visited++;
            const double su = sf.u(k), sv = sf.v(k);
            if (ev.fr_x != sf.row(k) || ev.fr_y != sf.col(k) || ev.timestamp != sf.timestamp(k) ||
                std::memcmp(&ev.best_u, &su, 8) != 0 || std::memcmp(&ev.best_v, &sv, 8) != 0)
                ++diff;
        }
        std::printf("slice %d at event %zu: ring %zu / %zu, iterated %zu, iterations %d, dx %.9g dy %.9g, %s, flow diffs %zu\n", slices, i + 1,
                    sz1, sz2, visited, sf.get_run_info().iterations, m2.total_dx, m2.total_dy, same_model(m1, m2) ? "model ==" : "MODEL !=", diff);
        if (!same_model(m1, m2) || sz1 != sz2 || diff) ++bad;
    }
    dvs.recompute(); sf.recompute();   // the tail, as the command line does
    if (!same_model(dvs.get_last_model(), sf.get_last_model())) { ++bad; std::printf("tail slice: MODEL !=\n"); }
    // the accumulated table: every event once, first slice's flow, the reference's marking rule
    LinearEventCloudTemplate<Event> acc = dvs.get_accumulated();
    bf::FlowTable tab = sf.get_accumulated();
    size_t acc_diff = acc.size() == tab.size() ? 0 : 1;
    for (size_t i = 0; i < acc.size() && i < tab.size(); ++i) {
        Event &e = acc[i];
        if (e.timestamp != tab.timestamp[i] || e.fr_x != tab.row[i] || e.fr_y != tab.col[i] ||
            std::memcmp(&e.best_u, &tab.u[i], 8) != 0 || std::memcmp(&e.best_v, &tab.v[i], 8) != 0)
            ++acc_diff;
    }
    std::printf("accumulated: %zu vs %zu events of %zu seen, %zu differences\n", (size_t)acc.size(), tab.size(), in.t.size(), acc_diff);
    if (acc_diff) ++bad;

    // ---- B: the same stream in bulk blocks, pipelined, on `workers` slice contexts: same slices, models, table
    bf::StreamFlow<MAX_SZ, SPAN> bulk(on_ev, on_time);
    bulk.set_max_iter(max_iter);
    bulk.set_accumulate();
    bulk.set_pipelined();
    if (stm_off) bulk.set_stm_disable(true);
    if (workers > 1) bulk.set_devices(std::vector<int>{bf::DeviceContext::device()}, workers);
    std::vector<Rec> recs_b;
    bulk.on_slice([&](const bf::SliceRecord &r) { recs_b.push_back(Rec{r.first_event, r.events, r.rc, (int)r.info.iterations, r.model}); });
    for (size_t at = 0; at < in.t.size(); at += 7777) {
        const size_t m = in.t.size() - at < 7777 ? in.t.size() - at : 7777;
        bulk.add_events(in.row.data() + at, in.col.data() + at, in.t.data() + at, m);
    }
    bulk.recompute();
    bulk.drain();
    size_t rec_diff = recs.size() == recs_b.size() ? 0 : 1;
    for (size_t i = 0; i < recs.size() && i < recs_b.size(); ++i)
        if (recs[i].first != recs_b[i].first || recs[i].n != recs_b[i].n || recs[i].rc != recs_b[i].rc ||
            recs[i].iterations != recs_b[i].iterations || !same_model(recs[i].model, recs_b[i].model))
            ++rec_diff;
    bf::FlowTable tab_b = bulk.get_accumulated();
    size_t tab_diff = tab.size() == tab_b.size() ? 0 : 1;
    for (size_t i = 0; i < tab.size() && i < tab_b.size(); ++i)
        if (tab.timestamp[i] != tab_b.timestamp[i] || tab.row[i] != tab_b.row[i] || tab.col[i] != tab_b.col[i] ||
            std::memcmp(&tab.u[i], &tab_b.u[i], 8) != 0 || std::memcmp(&tab.v[i], &tab_b.v[i], 8) != 0)
            ++tab_diff;
    std::printf("bulk, pipelined, %d worker(s): %zu vs %zu slices, %zu slice differences, %zu table differences\n", workers, recs.size(),
                recs_b.size(), rec_diff, tab_diff);
    if (rec_diff || tab_diff) ++bad;
    if (workers > 1) {
        // ---- C: several workers WITHOUT accumulate: the (u, v) ring must hold, for every event, the flow of the LATEST slice
        // that solved it (DVS_flow: a later recompute overwrites best_u / best_v) -- not that of whichever overlapping slice
        // finished last on some worker.  Compared with front end A's ring (one worker, slices in order) after the tail.
        bf::StreamFlow<MAX_SZ, SPAN> ringc(on_ev, on_time);
        ringc.set_max_iter(max_iter);
        ringc.set_pipelined();
        ringc.set_stm_disable(true);
        ringc.set_devices(std::vector<int>{bf::DeviceContext::device()}, workers);
        for (size_t at = 0; at < in.t.size(); at += 7777) {
            const size_t m = in.t.size() - at < 7777 ? in.t.size() - at : 7777;
            ringc.add_events(in.row.data() + at, in.col.data() + at, in.t.data() + at, m);
        }
        ringc.recompute();
        ringc.drain();
        size_t ring_diff = sf.size() == ringc.size() ? 0 : 1;
        const size_t visit = sf.size() == MAX_SZ ? sf.size() - 1 : sf.size();   // (the oldest element of a full ring is in no slice)
        for (size_t k = 0; k < visit && k < ringc.size(); ++k) {
            const double a_u = sf.u(k), a_v = sf.v(k), c_u = ringc.u(k), c_v = ringc.v(k);
            if (sf.timestamp(k) != ringc.timestamp(k) || std::memcmp(&a_u, &c_u, 8) != 0 || std::memcmp(&a_v, &c_v, 8) != 0) ++ring_diff;
        }
        std::printf("flow ring, %d workers, no accumulate: %zu differences over %zu ring elements\n", workers, ring_diff, visit);
        if (ring_diff) ++bad;
    }
    std::printf("%s %s: %d slices + tail, %d stopped by the window guard, %d problems (ring %zu, span %lld ns)\n", bad ? "FAIL" : "OK", tag,
                slices, skipped_guard, bad, (size_t)MAX_SZ, (long long)SPAN);
    return bad;
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    int bad = 0;
    const Stream plain = load(argv[1], 0, 0), twins = load(argv[1], 97, 0);
    bad += run<3000, 40000000>("fill-wrap-trim", twins, 1500, FROM_SEC(0.02), -1, false, 1);    // fills, wraps, span-trims; STM chain
    bad += run<3000, 15000000>("short-span", plain, 1000, FROM_SEC(0.5), -1, false, 1);         // the 15 ms span trims the ring before it fills
    bad += run<50000, 200000000>("never-full", twins, 4000, FROM_SEC(0.033), 10, true, 1);      // never full; capped, STM off
    bad += run<3000, 40000000>("independent x3", twins, 1500, FROM_SEC(0.02), 10, true, 3);     // independent slices on three workers
    // a ring SMALLER than the trigger interval: every slice finds the ring full, and its oldest element -- which the slice
    // leaves out (datastructures.h:71-76) -- has never been in a slice: the reference's accumulated copy (dvs_flow.h:340-345)
    // still holds it, with zero flow
    bad += run<1000, 40000000>("ring < trigger", twins, 1500, FROM_SEC(0.02), 10, false, 1);
    bad += run<1000, 40000000>("ring < trigger x2", twins, 1500, FROM_SEC(0.02), 10, true, 2);
    {   // a sensor so large that the confined start of the stream falls under the window guard (optimizer_rolling.h:49-55):
        // those slices flag their events as noise, and the flagged events stay out of the later, overlapping slices
        bf::sensor().res_x = 1000; bf::sensor().res_y = 1400;
        const Stream confined = load(argv[1], 0, FROM_SEC(0.045));
        bad += run<6000, 60000000>("noise-chain", confined, 100000, FROM_SEC(0.01), 12, false, 1);
        bad += run<6000, 60000000>("noise-independent x2", confined, 100000, FROM_SEC(0.01), 12, true, 2);
        bf::sensor().res_x = 180; bf::sensor().res_y = 240;
    }
    return bad ? 1 : 0;
}
