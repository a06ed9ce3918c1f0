// image.h — host texture image (mirror of the reference's class image, src/image.h:8-51): float RGB pixels, row 0 = bottom row.
#pragma once
#include <string>
#include <vector>
#include "scene.h"

class image {
public:
    int width = 0, height = 0;
    std::vector<ptk::vec3> pixels;
    std::string error;                                   // non-empty when loading failed (the reference sets pixels = nullptr)

    image(int x, int y);
    explicit image(const std::string& filePath, float gamma = 1.f);      // image.cpp:20-39 (stbi_loadf, 3 channels, flipped)
    bool ok() const { return !pixels.empty(); }
    size_t byteSize() const { return sizeof(ptk::vec3) * (size_t)width * height; }
    const ptk::vec3* data() const { return pixels.data(); }
    ptk::vec3 getValue(int x, int y) const { return pixels[(size_t)y * width + x]; }
};
