// bf::EventReader (better_flow/event_reader.h) against the reference's own extraction loop
// (`ifstream >> double >> uint >> uint >> bool`, bf_motion_compensator.cpp:186-197) on the same file:
// same number of records, bit-identical times, same integers -- including where the loop stops.
#include <better_flow/event_reader.h>
#include <cstdio>
#include <cstring>
#include <fstream>

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    std::ifstream f(argv[1], std::ifstream::in);
    bf::EventReader r(argv[1]);
    double t = 0, t2 = 0;
    unsigned x = 0, y = 0, x2 = 0, y2 = 0;
    bool p = false, p2 = false;
    unsigned long long n = 0, bad = 0;
    for (;;) {
        const bool a = (bool)(f >> t >> x >> y >> p);
        const bool b = r.next_text(t2, x2, y2, p2);
        if (a != b) { std::printf("record %llu: stream %d reader %d\n", n, (int)a, (int)b); return 1; }
        if (!a) break;
        if (std::memcmp(&t, &t2, 8) != 0 || x != x2 || y != y2 || p != p2) {
            if (bad++ < 5) std::printf("record %llu differs: %.17g %u %u %d vs %.17g %u %u %d\n", n, t, x, y, (int)p, t2, x2, y2, (int)p2);
        }
        ++n;
    }
    std::printf("records %llu mismatches %llu\n", n, bad);
    return bad ? 1 : 0;
}
