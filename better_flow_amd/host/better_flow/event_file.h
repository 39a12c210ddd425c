// event_file.h -- text event I/O (mirror of the I/O half of the reference's
// better_flow/event_file.h:141-176,265-289).  The visualisation half of that file (colour maps,
// flow arrows, OpenCV rendering) is outside the motion-compensation path.
#ifndef BF_HOST_EVENT_FILE_H
#define BF_HOST_EVENT_FILE_H

#include <better_flow/common.h>
#include <better_flow/event.h>

class EventFile {
public:
    // "t x y p" per line: seconds, column, row, polarity.  x / y are swapped on the way in
    // (event_file.h:160,167; bf_motion_compensator.cpp:192,200): fr_x = row, fr_y = column.
    template <class T> static void from_file(T *events, std::string fname) {
        std::cout << "Reading from file... (" << fname << ")" << std::endl << std::flush;
        std::ifstream event_file(fname, std::ifstream::in);
        ull cnt = 0;
        double t = 0;
        uint x = 0, y = 0;
        bool p = false;
        double t_0 = 0;   // the earliest timestamp in the file
        clock_t begin = std::clock();
        if (event_file >> t_0 >> x >> y >> p) {
            events->push_back(Event(y, x, FROM_SEC(0)));
            cnt++;
        }
        while (event_file >> t >> x >> y >> p) {
            t -= t_0;
            events->push_back(Event(y, x, FROM_SEC(t)));
            cnt++;
        }
        clock_t end = std::clock();
        event_file.close();
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
        std::ofstream event_file(fname, std::ofstream::out);
        ull cnt = 0;
        clock_t begin = std::clock();
        for (auto &e : *events) {
            event_file << std::fixed << std::setprecision(9) << double(e.timestamp) / 1000000000 << " " << e.fr_y
                       << " " << e.fr_x << " " << 1 << " " << e.best_v << " " << e.best_u << std::endl;
            cnt++;
        }
        clock_t end = std::clock();
        event_file.close();
        if (cnt == 0) {
            std::cout << "Written " << cnt << " events, finished" << std::endl << std::endl << std::flush;
            return;
        }
        std::cout << "Written " << cnt << " events, finished" << std::endl << std::flush;
        std::cout << "Elapsed: " << double(end - begin) / CLOCKS_PER_SEC << " sec." << std::endl << std::flush;
    }
};

#endif  // BF_HOST_EVENT_FILE_H
