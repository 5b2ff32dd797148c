// png_write.hpp — see png_write.cpp.
#ifndef RT_AMD_PNG_WRITE_HPP
#define RT_AMD_PNG_WRITE_HPP
#include <cstdint>
// Writes w*h RGBA8 pixels (row 0 first unless flip) as a PNG. 0 on success.
int rt_write_png_rgba(const char *path, int w, int h, const uint8_t *rgba, bool flip);
#endif
