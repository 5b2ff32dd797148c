// tiling.hpp — how a frame is split over N GPUs (the reference is single-GPU and has no counterpart).
// Rank r owns one tile of a columns x rows grid; the grid is the factorisation of N whose largest tile, grown by the
// filter halo, has the fewest pixels (what a rank traces beyond its own tile is redundant work); ties go to the grid
// with fewer columns.  The same rule as raytracer_amd/parallel.py (tests/test_host_scene.py checks they agree).
#ifndef RT_AMD_TILING_HPP
#define RT_AMD_TILING_HPP

#include <algorithm>

#include "../../include/rt_launcher.h"

namespace rttile {

inline void stripRange(int extent, int parts, int index, int &lo, int &hi) { // contiguous, sizes differ by at most one
    const int base = extent / parts, extra = extent % parts;
    lo = index * base + std::min(index, extra);
    hi = lo + base + (index < extra ? 1 : 0);
}

inline void tileGrid(int width, int height, int world, int halo, int &gx, int &gy) {
    long long best = -1;
    gx = 1;
    gy = world;
    for (int cx = 1; cx <= world; ++cx) {
        if (world % cx) continue;
        const int cy = world / cx;
        const int tw = (width + cx - 1) / cx, th = (height + cy - 1) / cy;
        const long long area = (long long)std::min(tw + (cx > 2 ? 2 : cx - 1) * halo, width) *
                               (long long)std::min(th + (cy > 2 ? 2 : cy - 1) * halo, height);
        if (best < 0 || area < best) {
            best = area;
            gx = cx;
            gy = cy;
        }
    }
}

inline rt_rect tileRect(int width, int height, int world, int rank, int halo) { // ranks in row-major order
    int gx, gy;
    tileGrid(width, height, world, halo, gx, gy);
    rt_rect r;
    stripRange(width, gx, rank % gx, r.x0, r.x1);
    stripRange(height, gy, rank / gx, r.y0, r.y1);
    return r;
}

inline rt_rect growRect(rt_rect r, int halo, int width, int height) {
    return rt_rect{std::max(r.x0 - halo, 0), std::max(r.y0 - halo, 0), std::min(r.x1 + halo, width), std::min(r.y1 + halo, height)};
}

inline int filterHalo(int iterations) { // = rt_filter_halo: 1 + 3 + 7 + ...
    int h = 0;
    for (int i = 0; i < iterations; ++i) h += (1 << (i + 1)) - 1;
    return h;
}

} // namespace rttile

#endif
