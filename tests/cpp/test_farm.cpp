// bf::SliceFarm (better_flow/slice_farm.h) on its own: independent slices handed over as linear int32 arrays with
// slice-local times (the layout of bf_upload_events), cold and warm-started from a given model, on 1 and 3 workers --
// against the same slices solved one by one through the C-ABI on a single context.  Results must arrive in submission
// order and be bit-identical; an empty slice is reported as skipped.  Linked against the oracle shim (CPU) or
// libbf_accel.so (GPU) by tests/test_host_cli.py.
#include <better_flow/common.h>
#include <better_flow/event_reader.h>
#include <better_flow/slice_farm.h>
#include <cstdio>
#include <cstring>

struct Slice { std::vector<int32_t> x, y, t; };

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    // cut the file's events into 7 consecutive slices of uneven size (the last one empty)
    std::vector<Slice> slices(7);
    {
        std::vector<unsigned> row, col; std::vector<unsigned long long> ts;
        bf::EventReader reader(argv[1]);
        reader.for_each_event([&](unsigned r, unsigned c, unsigned long long t) { row.push_back(r); col.push_back(c); ts.push_back(t); });
        const size_t n = row.size();
        const size_t cuts[8] = {0, n / 9, n / 4, n / 2, n / 2 + 1500, 3 * n / 4, n, n};
        for (int k = 0; k < 7; ++k)
            for (size_t i = cuts[k]; i < cuts[k + 1]; ++i) {
                slices[k].x.push_back((int32_t)row[i]); slices[k].y.push_back((int32_t)col[i]);
                slices[k].t.push_back((int32_t)(ts[i] - ts[cuts[k]]));
            }
    }
    const int H = RES_X, W = RES_Y, s = 3;
    // reference: one context, one slice after the other; slice k > 2 warm-started from slice 1's model
    std::vector<bf_model> want(slices.size());
    std::vector<bf_run_info> winfo(slices.size());
    std::vector<int> wrc(slices.size());
    bf_model seed_model;
    std::memset(&seed_model, 0, sizeof(seed_model));
    {
        bf_ctx *c = nullptr;
        if (bf_create(0, 20000, s * H + s, s * W + s, nullptr, &c) != BF_OK) { std::printf("FAIL: no context\n"); return 1; }
        for (size_t k = 0; k < slices.size(); ++k) {
            const Slice &sl = slices[k];
            std::memset(&want[k], 0, sizeof(bf_model)); std::memset(&winfo[k], 0, sizeof(bf_run_info));
            if (sl.x.empty()) { wrc[k] = BF_SKIPPED; if (k > 2) want[k] = seed_model; continue; }
            bf_upload_events(c, sl.x.data(), sl.y.data(), sl.t.data(), nullptr, (int64_t)sl.x.size());
            bf_window w; bf_set_cloud(c, s, H, W, &w);
            if (k > 2) bf_set_model(c, &seed_model);
            bf_run_opts o; bf_run_opts_default(&o); o.res_x = H; o.res_y = W; o.max_iter = 30;
            wrc[k] = bf_run(c, &o, &want[k], &winfo[k]);
            if (k == 1) seed_model = want[k];
        }
        bf_destroy(c);
    }
    int bad = 0;
    for (int workers : {1, 3}) {
        std::vector<bf::SliceFarm::Result> got;
        {
            bf::SliceFarm farm(std::vector<int>{0}, workers, 20000, s * H + s, s * W + s,
                               [&](const bf::SliceFarm::Result &r) { got.push_back(r); });   // (in order, one at a time)
            for (size_t k = 0; k < slices.size(); ++k) {
                bf::SliceFarm::Task t;
                t.fr_x = slices[k].x.data(); t.fr_y = slices[k].y.data(); t.t_ns = slices[k].t.data();
                t.n = (int64_t)slices[k].x.size();
                t.scale = s; t.res_x = H; t.res_y = W; t.max_iter = 30;
                t.warm = k > 2 ? bf::SliceFarm::Warm::FromModel : bf::SliceFarm::Warm::Cold;
                t.start = seed_model;
                t.user = 100 + k;
                farm.submit(t);
            }
            farm.drain();
        }
        int diff = got.size() == slices.size() ? 0 : 1;
        for (size_t k = 0; k < got.size() && k < slices.size(); ++k) {
            const bf::SliceFarm::Result &r = got[k];
            const bool same = r.id == k && r.user == 100 + k && r.rc == wrc[k] && r.info.iterations == winfo[k].iterations &&
                              std::memcmp(&r.model, &want[k], sizeof(bf_model)) == 0;
            if (!same) { ++diff; std::printf("workers %d slice %zu: rc %d vs %d, iterations %d vs %d, %s\n", workers, k, r.rc, wrc[k],
                                             r.info.iterations, winfo[k].iterations, r.error.c_str()); }
        }
        std::printf("%s farm with %d worker(s): %zu results, %d differences\n", diff ? "FAIL" : "OK", workers, got.size(), diff);
        bad += diff;
    }
    {   // a farm whose contexts are too small for two of the slices: those two fail with BF_ERR_CAPACITY -- in their place in the
        // order, with the context's error text --, the others are solved as before
        std::vector<bf::SliceFarm::Result> got;
        {
            bf::SliceFarm farm(std::vector<int>{0}, 2, 2000, s * H + s, s * W + s, [&](const bf::SliceFarm::Result &r) { got.push_back(r); });
            for (size_t k = 0; k < slices.size(); ++k) {
                bf::SliceFarm::Task t;
                t.fr_x = slices[k].x.data(); t.fr_y = slices[k].y.data(); t.t_ns = slices[k].t.data();
                t.n = (int64_t)slices[k].x.size();
                t.scale = s; t.res_x = H; t.res_y = W; t.max_iter = 30;
                t.warm = k > 2 ? bf::SliceFarm::Warm::FromModel : bf::SliceFarm::Warm::Cold;
                t.start = seed_model;
                farm.submit(t);
            }
            farm.drain();
        }
        int diff = got.size() == slices.size() ? 0 : 1, failed = 0;
        for (size_t k = 0; k < got.size() && k < slices.size(); ++k) {
            const bool too_big = slices[k].x.size() > 2000;
            if (too_big) {
                ++failed;
                if (got[k].rc != BF_ERR_CAPACITY || got[k].error.find("capacity") == std::string::npos || got[k].id != k) ++diff;
            } else if (got[k].id != k || got[k].rc != wrc[k] || std::memcmp(&got[k].model, &want[k], sizeof(bf_model)) != 0) ++diff;
        }
        std::printf("%s farm with undersized contexts: %zu results, %d refused for capacity, %d differences\n", (diff || failed != 2) ? "FAIL" : "OK",
                    got.size(), failed, diff);
        bad += diff + (failed != 2);
    }
    return bad ? 1 : 0;
}
