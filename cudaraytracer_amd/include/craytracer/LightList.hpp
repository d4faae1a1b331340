// Ordered list of lights whose contributions are summed (reference: LightList.hpp:14-37).
#pragma once
#include "Light.hpp"

namespace craytracer {

class LightList : public Light {
    Light** list = nullptr;
    int list_size = 0;

public:
    LightList() = default;
    LightList(Light** l, int n) : list(l), list_size(n) {}
    int size() const { return list_size; }

    void flatten(std::vector<crt_light>& out) const override {
        for (int i = 0; i < list_size; ++i)
            if (list[i]) list[i]->flatten(out);
    }
};
} // namespace craytracer
