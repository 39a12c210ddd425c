// datastructures.h -- the containers that hand events to the motion-compensation path.
//
// Same class names, template parameters and member functions as the reference's
// better_flow/datastructures.h (CircularArray :6-115, LinearEventCloudTemplate :119-168,
// LinearEventPtrsTemplate :172-259), because DVS_flow, the optimizers and user code are written
// against them; the implementation is this build's.  The per-pixel EventCloudTemplate (:263-393)
// has no caller in the reference and is not provided.
#ifndef BF_HOST_DATASTRUCTURES_H
#define BF_HOST_DATASTRUCTURES_H

#include <cassert>
#include <climits>
#include <cstddef>
#include <vector>

// A time-windowed ring: holds the most recent events, at most SZ of them and none older than SPAN
// (in the units of `DType - DType`, ns) behind the newest.  Walks newest -> oldest.
//
// Behaviour callers depend on, all taken from the reference and pinned by tests/cpp/test_ring.cpp:
//   * the first element lands in slot 1, not 0 (the cursor is advanced before the store);
//   * trimming by SPAN is lazy: push_back only marks the ring stale, the next size() / begin() /
//     end() drops what is too old;
//   * a FULL ring reports size() == SZ but begin()..end() visits SZ - 1 elements: end() then names
//     the oldest slot itself instead of the slot behind it (datastructures.h:71-76).
template <class DType, size_t SZ, long long SPAN> class CircularArray final {
    typedef long long slot_t;
    static slot_t wrap(slot_t s) { s %= slot_t(SZ); return s < 0 ? s + slot_t(SZ) : s; }

protected:
    std::vector<DType> store;
    size_t held;      // elements currently in the window
    size_t cursor;    // slot of the newest element
    bool stale;       // a push happened since the window was last trimmed

    // slot `back` places behind the newest
    size_t slot_behind(slot_t back) const { return size_t(wrap(slot_t(cursor) - back)); }

    void trim() {
        if (!stale) return;
        stale = false;
        size_t oldest = slot_behind(slot_t(held) - 1);
        size_t dropped = 0;
        // the newest element is 0 away from itself, so this stops at the latest there
        while ((long long)(store[cursor] - store[oldest]) > SPAN) {
            oldest = (oldest + 1 == SZ) ? 0 : oldest + 1;
            ++dropped;
        }
        held -= dropped;
    }

public:
    typedef DType value_type;

    CircularArray() : store(SZ), held(0), cursor(0), stale(false) {}

    void push_back(DType &d) {
        cursor = (cursor + 1 == SZ) ? 0 : cursor + 1;
        store[cursor] = d;
        if (held < SZ) ++held;
        stale = true;
    }

    void fix_span() { trim(); }

    size_t size() {
        trim();
        return held;
    }

    // idx 0 is the newest element, size() - 1 the oldest
    DType &operator[](size_t idx) {
        assert(idx < held);
        return store[slot_behind(slot_t(idx))];
    }

    class iterator {
        friend class CircularArray;
        CircularArray *ring;
        size_t at;
        iterator(CircularArray *r, size_t slot) : ring(r), at(slot) {}

    public:
        DType &operator*() { return ring->store[at]; }
        DType *operator->() { return &ring->store[at]; }
        iterator &operator++() {   // one step into the past
            at = at ? at - 1 : SZ - 1;
            return *this;
        }
        bool operator==(const iterator &o) const { return at == o.at; }
        bool operator!=(const iterator &o) const { return at != o.at; }
    };

    iterator begin() {
        trim();
        return iterator(this, cursor);
    }

    iterator end() {
        trim();
        // one slot past the oldest -- except for a full ring, where that slot is the newest again
        // and the reference stops ON the oldest instead (the oldest element is never visited)
        const slot_t back = (held >= SZ) ? slot_t(held) - 1 : slot_t(held);
        return iterator(this, slot_behind(back));
    }
};

namespace bf_detail {
// Bounding box of the sensor addresses pushed so far; x = row, y = column.
struct AddressBox {
    int &x_min, &y_min, &x_max, &y_max;
    void take(int x, int y) {
        x_min = x < x_min ? x : x_min;
        x_max = x > x_max ? x : x_max;
        y_min = y < y_min ? y : y_min;
        y_max = y > y_max ? y : y_max;
    }
};
}  // namespace bf_detail

// Events by value, in insertion order, with the bounding box of their addresses kept up to date.
template <class DType> class LinearEventCloudTemplate {
protected:
    std::vector<DType> data;

public:
    int x_min, y_min, x_max, y_max;   // empty cloud: min = INT_MAX, max = INT_MIN

    LinearEventCloudTemplate() : x_min(INT_MAX), y_min(INT_MAX), x_max(INT_MIN), y_max(INT_MIN) {}

    explicit LinearEventCloudTemplate(std::vector<DType> &src) : LinearEventCloudTemplate() {
        data.reserve(src.size());
        for (auto &d : src) push_back(d);
    }

    explicit LinearEventCloudTemplate(std::vector<LinearEventCloudTemplate<DType>> &parts)
        : LinearEventCloudTemplate() {
        for (auto &part : parts)
            for (auto &d : part) push_back(d);
    }

    void push_back(DType d) {
        bf_detail::AddressBox{x_min, y_min, x_max, y_max}.take(int(d.get_x()), int(d.get_y()));
        data.push_back(d);
    }
    DType &operator[](size_t idx) {
        assert(idx < data.size());
        return data[idx];
    }
    size_t size() { return data.size(); }
    void reserve(size_t n) { data.reserve(n); }
    typename std::vector<DType>::iterator begin() { return data.begin(); }
    typename std::vector<DType>::iterator end() { return data.end(); }
};

// The same interface over events owned elsewhere (the slice views DVS_flow builds over its ring).
template <class DType> class LinearEventPtrsTemplate {
protected:
    std::vector<DType *> data;

public:
    int x_min, y_min, x_max, y_max;

    LinearEventPtrsTemplate() : x_min(INT_MAX), y_min(INT_MAX), x_max(INT_MIN), y_max(INT_MIN) {}

    explicit LinearEventPtrsTemplate(std::vector<DType> &src) : LinearEventPtrsTemplate() {
        data.reserve(src.size());
        for (auto &d : src) push_back(&d);
    }

    explicit LinearEventPtrsTemplate(std::vector<LinearEventCloudTemplate<DType>> &parts)
        : LinearEventPtrsTemplate() {
        for (auto &part : parts)
            for (auto &d : part) push_back(&d);
    }

    explicit LinearEventPtrsTemplate(std::vector<LinearEventPtrsTemplate<DType>> &parts)
        : LinearEventPtrsTemplate() {
        for (auto &part : parts)
            for (auto &d : part) push_back(&d);
    }

    void push_back(DType *d) {
        bf_detail::AddressBox{x_min, y_min, x_max, y_max}.take(int(d->get_x()), int(d->get_y()));
        data.push_back(d);
    }
    void push_back(DType &d) { push_back(&d); }

    DType &operator[](size_t idx) {
        assert(idx < data.size());
        return *data[idx];
    }
    size_t size() { return data.size(); }
    void reserve(size_t n) { data.reserve(n); }

    // dereferences twice, so range-for yields DType& as with the by-value cloud
    class iterator {
        typename std::vector<DType *>::iterator pos;

    public:
        explicit iterator(typename std::vector<DType *>::iterator p) : pos(p) {}
        DType &operator*() { return **pos; }
        DType *operator->() { return *pos; }
        iterator &operator++() {
            ++pos;
            return *this;
        }
        bool operator==(const iterator &o) const { return pos == o.pos; }
        bool operator!=(const iterator &o) const { return pos != o.pos; }
    };
    iterator begin() { return iterator(data.begin()); }
    iterator end() { return iterator(data.end()); }
};

#endif  // BF_HOST_DATASTRUCTURES_H
