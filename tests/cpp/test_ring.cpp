// CircularArray semantics (better_flow_amd/host/better_flow/datastructures.h vs the reference's
// datastructures.h:6-115): newest->oldest iteration, span trimming, and the full-ring quirk
// (iteration stops one element short once SZ elements are held, :71-76).
#include <better_flow/common.h>
#include <better_flow/event.h>
#include <cstdio>

template <class CA> static void dump(const char *tag, CA &ca) {
    printf("%s size=%zu iter=", tag, (size_t)ca.size());
    size_t k = 0;
    for (auto &e : ca) { printf("%s%llu", k ? "," : "", (unsigned long long)e.timestamp); ++k; }
    printf(" n_iter=%zu", k);
    if (ca.size() > 0) printf(" newest=%llu oldest=%llu", (unsigned long long)ca[0].timestamp,
                              (unsigned long long)ca[ca.size() - 1].timestamp);
    printf("\n");
}

int main() {
    {   // plain fill below capacity
        CircularArray<Event, 8, 1000> ca;
        for (ull t = 10; t <= 50; t += 10) { Event e(1, 2, t); ca.push_back(e); }
        dump("fill5", ca);
    }
    {   // span trimming: only events within 100 ns of the newest survive
        CircularArray<Event, 8, 100> ca;
        for (ull t : {0ull, 50ull, 90ull, 140ull, 200ull}) { Event e(1, 2, t); ca.push_back(e); }
        dump("span100", ca);
    }
    {   // full ring: size() == SZ but iteration visits SZ - 1 elements
        CircularArray<Event, 4, 1000000> ca;
        for (ull t = 1; t <= 6; ++t) { Event e(1, 2, t); ca.push_back(e); }
        dump("full4", ca);
    }
    {   // Event helpers
        Event a(3, 4, 1000), b(3, 4, 1000 + 99999), c(3, 4, 1000 + 100000), d(3, 5, 1000);
        printf("eq %d %d %d\n", (int)(a == b), (int)(a == c), (int)(a == d));
        a.set_local_time(400);
        b.set_local_time(200000);
        printf("local %lld %lld\n", (long long)a.t, (long long)b.t);
        printf("from_sec %llu %llu\n", FROM_SEC(0.033), FROM_SEC(0.2));
    }
    return 0;
}
