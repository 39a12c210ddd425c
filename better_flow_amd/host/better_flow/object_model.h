// object_model.h -- per-slice motion model (mirror of the reference's
// better_flow/object_model.h:8-72).  The moment reduction itself (object_model.cpp:4-39,
// 103-126) runs on the GPU (k_stencil*/model_update); see AccelLib::fast_model.
#ifndef BF_HOST_OBJECT_MODEL_H
#define BF_HOST_OBJECT_MODEL_H

#include <better_flow/common.h>
#include <bf_accel.h>

class ObjectModel {
public:
    double cx, cy, dx, dy, rot, div;
    uint cnt;
    double total_dx, total_dy, total_rot, total_div;

    ObjectModel()
        : cx(0), cy(0), dx(0), dy(0), rot(0), div(0), cnt(0), total_dx(0), total_dy(0), total_rot(0),
          total_div(0) {}

    explicit ObjectModel(const bf_model &m)
        : cx(m.cx), cy(m.cy), dx(m.dx), dy(m.dy), rot(m.rot), div(m.div), cnt(m.cnt),
          total_dx(m.total_dx), total_dy(m.total_dy), total_rot(m.total_rot), total_div(m.total_div) {}

    bf_model to_abi() const {
        bf_model m;
        m.cx = cx; m.cy = cy; m.dx = dx; m.dy = dy; m.rot = rot; m.div = div;
        m.cnt = cnt; m._pad = 0;
        m.total_dx = total_dx; m.total_dy = total_dy; m.total_rot = total_rot; m.total_div = total_div;
        return m;
    }

    // same text as the reference (object_model.h:55-63)
    friend std::ostream &operator<<(std::ostream &output, const ObjectModel &M) {
        output << "C: (" << M.cx << ", " << M.cy << "); " << std::endl
               << "\t Shift: (" << M.dx << ", " << M.dy << "); "
               << " total: (" << M.total_dx << ", " << M.total_dy << ");" << std::endl
               << "\t Rot: " << M.rot << " total: " << M.total_rot << std::endl
               << "\t Div: " << M.div << " total: " << M.total_div << std::endl
               << "\t cnt: " << M.cnt << std::endl;
        return output;
    }
};

#endif  // BF_HOST_OBJECT_MODEL_H
