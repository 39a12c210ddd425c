// datastructures.h -- containers feeding the motion-compensation path (mirror of the
// reference's better_flow/datastructures.h:6-259; the per-pixel EventCloudTemplate, :263-393,
// is unreferenced there and not provided).
#ifndef BF_HOST_DATASTRUCTURES_H
#define BF_HOST_DATASTRUCTURES_H

#include <cassert>
#include <climits>
#include <cstddef>
#include <vector>

// Ring of at most SZ elements spanning at most SPAN (ns) of time, iterated newest -> oldest
// (datastructures.h:6-115).  Faithful quirk: once the ring is full, end() stops one element
// short, so iteration visits SZ - 1 elements (:71-76).
template <class DType, size_t SZ, long long SPAN> class CircularArray final {
protected:
    std::vector<DType> data;
    size_t current_size, head_id;
    bool span_checked;
    size_t latest_id;

public:
    typedef DType value_type;

    CircularArray() : data(SZ), current_size(0), head_id(0), span_checked(true), latest_id(0) {}

    inline size_t size() {
        this->fix_span();
        return this->current_size;
    }

    inline void push_back(DType &d) {   // :31-44
        this->span_checked = false;
        this->current_size += (this->current_size >= SZ) ? 0 : 1;
        this->head_id++;
        if (this->head_id >= SZ) this->head_id = 0;
        this->data[this->head_id] = d;
        this->latest_id = this->head_id;
    }

    inline void fix_span() {   // :46-59
        if (this->span_checked) return;
        this->span_checked = true;
        size_t tail_id = ((1 - int(this->current_size - this->head_id)) + SZ) % SZ;
        size_t removed_cnt = 0;
        while ((long long)(this->data[this->latest_id] - this->data[tail_id]) > SPAN) {
            removed_cnt++;
            tail_id++;
            if (tail_id >= SZ) tail_id = 0;
        }
        this->current_size -= removed_cnt;
    }

    inline DType &operator[](size_t idx) {   // :61-64, idx 0 = newest
        assert(idx < this->current_size);
        return this->data[((int(this->head_id) - int(idx)) + SZ) % SZ];
    }

    class iterator {
        friend class CircularArray;
        CircularArray *ca;
        size_t id;
        iterator(CircularArray *c, size_t i) : ca(c), id(i) {}

    public:
        DType &operator*() { return ca->data[id]; }
        DType *operator->() { return &ca->data[id]; }
        iterator &operator++() {   // newest -> oldest, :86-96
            id = (id == 0) ? SZ - 1 : id - 1;
            return *this;
        }
        bool operator!=(const iterator &o) const { return id != o.id; }
        bool operator==(const iterator &o) const { return id == o.id; }
    };

    inline iterator begin() {   // :66-69
        this->fix_span();
        return iterator(this, this->head_id);
    }

    inline iterator end() {   // :71-76
        this->fix_span();
        int shift = (this->current_size >= SZ) ? 1 : 0;
        size_t tail_id = ((shift - int(this->current_size - this->head_id)) + SZ) % SZ;
        return iterator(this, tail_id);
    }
};

// A simple linear event cloud with a bounding box (datastructures.h:119-168).
template <class DType> class LinearEventCloudTemplate {
protected:
    std::vector<DType> data;

public:
    int x_min, y_min, x_max, y_max;

    LinearEventCloudTemplate() : x_min(INT_MAX), y_min(INT_MAX), x_max(INT_MIN), y_max(INT_MIN) {}

    inline void push_back(DType d) {
        if ((int)d.get_x() > this->x_max) this->x_max = d.get_x();
        if ((int)d.get_y() > this->y_max) this->y_max = d.get_y();
        if ((int)d.get_x() < this->x_min) this->x_min = d.get_x();
        if ((int)d.get_y() < this->y_min) this->y_min = d.get_y();
        this->data.push_back(d);
    }
    inline DType &operator[](size_t idx) {
        assert(idx < this->size());
        return this->data[idx];
    }
    inline size_t size() { return this->data.size(); }
    inline auto begin() { return this->data.begin(); }
    inline auto end() { return this->data.end(); }
    inline void reserve(size_t n) { this->data.reserve(n); }
};

// Same interface, storing pointers (datastructures.h:172-259).
template <class DType> class LinearEventPtrsTemplate {
protected:
    std::vector<DType *> data;

public:
    int x_min, y_min, x_max, y_max;

    LinearEventPtrsTemplate() : x_min(INT_MAX), y_min(INT_MAX), x_max(INT_MIN), y_max(INT_MIN) {}

    inline void push_back(DType *d) {
        if ((int)d->get_x() > this->x_max) this->x_max = d->get_x();
        if ((int)d->get_y() > this->y_max) this->y_max = d->get_y();
        if ((int)d->get_x() < this->x_min) this->x_min = d->get_x();
        if ((int)d->get_y() < this->y_min) this->y_min = d->get_y();
        this->data.push_back(d);
    }
    inline void push_back(DType &d) { this->push_back(&d); }
    inline DType &operator[](size_t idx) {
        assert(idx < this->size());
        return *(this->data[idx]);
    }
    inline size_t size() { return this->data.size(); }
    inline void reserve(size_t n) { this->data.reserve(n); }

    class iterator {
        typename std::vector<DType *>::iterator it;

    public:
        explicit iterator(typename std::vector<DType *>::iterator i) : it(i) {}
        DType &operator*() { return **it; }
        DType *operator->() { return *it; }
        iterator &operator++() {
            ++it;
            return *this;
        }
        bool operator!=(const iterator &o) const { return it != o.it; }
        bool operator==(const iterator &o) const { return it == o.it; }
    };
    inline iterator begin() { return iterator(this->data.begin()); }
    inline iterator end() { return iterator(this->data.end()); }
};

#endif  // BF_HOST_DATASTRUCTURES_H
