// Ordered list of primitives (reference: GeometryList.hpp:14-36).  Order is
// significant: on an exact distance tie the later primitive wins.
#pragma once
#include "Geometry.hpp"

namespace craytracer {

class GeometryList : public Geometry {
    Geometry** list = nullptr;
    int list_size = 0;

public:
    GeometryList() = default;
    GeometryList(Geometry** l, int n) : list(l), list_size(n) {}
    int size() const { return list_size; }
    Geometry* operator[](int i) const { return list[i]; }

    void flatten(std::vector<crt_geometry>& out) const override {
        for (int i = 0; i < list_size; ++i)
            if (list[i]) list[i]->flatten(out);
    }
};
} // namespace craytracer
