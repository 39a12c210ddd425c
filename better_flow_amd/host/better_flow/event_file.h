// event_file.h -- text event I/O (mirror of the I/O half of the reference's
// better_flow/event_file.h:141-176,265-289).  The visualisation half of that file (colour maps,
// flow arrows, OpenCV rendering) is outside the motion-compensation path.
#ifndef BF_HOST_EVENT_FILE_H
#define BF_HOST_EVENT_FILE_H

#include <better_flow/common.h>
#include <better_flow/event.h>

#include <better_flow/event_reader.h>

class EventFile {
public:
    // "t x y p" per line: seconds, column, row, polarity.  x / y are swapped on the way in
    // (event_file.h:160,167; bf_motion_compensator.cpp:192,200): fr_x = row, fr_y = column.
    template <class T> static void from_file(T *events, std::string fname) {
        std::cout << "Reading from file... (" << fname << ")" << std::endl << std::flush;
        clock_t begin = std::clock();
        bf::EventReader reader(fname);   // text (the reference's format) or binary SoA, auto-detected
        ull cnt = reader.for_each_event([&](unsigned row, unsigned col, unsigned long long t_ns) {
            events->push_back(Event(row, col, (ull)t_ns));
        });
        clock_t end = std::clock();
        if (cnt == 0) {
            std::cout << "Read " << cnt << " events, finished" << std::endl << std::endl << std::flush;
            return;
        }
        std::cout << "Read " << cnt << " events, finished" << std::endl << std::flush;
        std::cout << "Elapsed: " << double(end - begin) / CLOCKS_PER_SEC << " sec." << std::endl << std::flush;
    }

    // "t x y 1 best_v best_u", 9 decimals; x / y and u / v swapped back (event_file.h:272-276).
    template <class T> static void to_file_uv(T *events, std::string fname) {
        std::cout << "Writing events and flow to file... (" << fname << ")" << std::endl << std::flush;
        ull cnt = 0;
        clock_t begin = std::clock();
        // same characters as `ofstream << std::fixed << std::setprecision(9)` (event_file.h:272-276), formatted
        // into one buffer instead of one flushed line at a time
        std::string out;
        out.reserve(events->size() * 64 + 64);
        char line[256];
        for (auto &e : *events) {
            const int len = std::snprintf(line, sizeof(line), "%.9f %u %u 1 %.9f %.9f\n", double(e.timestamp) / 1000000000,
                                          (unsigned)e.fr_y, (unsigned)e.fr_x, e.best_v, e.best_u);
            out.append(line, (size_t)len);
            cnt++;
        }
        if (FILE *f = std::fopen(fname.c_str(), "wb")) {
            std::fwrite(out.data(), 1, out.size(), f);
            std::fclose(f);
        }
        clock_t end = std::clock();
        if (cnt == 0) {
            std::cout << "Written " << cnt << " events, finished" << std::endl << std::endl << std::flush;
            return;
        }
        std::cout << "Written " << cnt << " events, finished" << std::endl << std::flush;
        std::cout << "Elapsed: " << double(end - begin) / CLOCKS_PER_SEC << " sec." << std::endl << std::flush;
    }
};

#endif  // BF_HOST_EVENT_FILE_H
