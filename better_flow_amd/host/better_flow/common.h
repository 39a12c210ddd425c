// common.h -- typedefs and constants of the host front end (mirror of the reference's
// better_flow/common.h:22-64, without its OpenCV dependency).
//
// RES_X / RES_Y are compile-time 180 x 240 in the reference (common.h:39-40); here they are
// run-time values (bf::sensor()) with the same defaults, because the BASELINE configs use
// 346x260, 640x480 and 1280x720 sensors.
#ifndef BF_HOST_COMMON_H
#define BF_HOST_COMMON_H

#include <cassert>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <string>
#include <vector>

#include <better_flow/datastructures.h>

typedef long int lint;
typedef long long int sll;
typedef unsigned int uint;
typedef unsigned long int ulong;
typedef unsigned long long int ull;

#define BF_VERSION "1.0-mi355x"

// Time conversion (common.h:35-36)
#define FROM_SEC(in) ull(1000000000 * (in))
#define FROM_MS(in) ull(1000000 * (in))

namespace bf {
struct Sensor {
    int res_x = 180;   // rows    (RES_X, common.h:39)
    int res_y = 240;   // columns (RES_Y, common.h:40)
};
inline Sensor &sensor() {
    static Sensor s;
    return s;
}
}  // namespace bf
#define RES_X (bf::sensor().res_x)
#define RES_Y (bf::sensor().res_y)

#ifndef VERBOSE
#define VERBOSE false
#endif

// Z (time) component of the direction vector (common.h:60) and the time divider (:64)
#define NZ 127
#define T_DIVIDER 1

class Event;
typedef LinearEventCloudTemplate<Event> LinearEventCloud;
typedef LinearEventPtrsTemplate<Event> LinearEventPtrs;

#endif  // BF_HOST_COMMON_H
