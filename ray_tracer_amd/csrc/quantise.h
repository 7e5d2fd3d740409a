// quantise.h -- host side of the 8-bit node formats: a 4-wide node's child boxes on the node's own grid
// (CQNode, device_types.h), and the 48-byte packed form of such a node (PNode).
#ifndef RT_QUANTISE_H
#define RT_QUANTISE_H

#include <algorithm>
#include <cmath>
#include "device_types.h"

namespace rt {

// 8-bit copy of a 4-wide node: origin = lower corner of the union of the child boxes, per axis a grid step 2^(e - 127)
// such that 255 steps span the extent, child minima rounded down and maxima rounded up, so every stored box contains the
// f32 box it came from.  Empty slots (RT_REF_NONE) get an inverted box.
inline void quantise_node(const QNode& s, CQNode& q) {
    int n = 0;
    for (int j = 0; j < 4; ++j) { q.child[j] = s.child[j]; if (s.child[j] != RT_REF_NONE) n = j + 1; }
    q._pad0 = 0; q._pad[0] = q._pad[1] = 0;
    for (int a = 0; a < 3; ++a) {
        float lo = s.lo[a][0], hi = s.hi[a][0];
        for (int j = 1; j < n; ++j) { lo = std::min(lo, s.lo[a][j]); hi = std::max(hi, s.hi[a][j]); }
        q.origin[a] = lo;
        const double extent = (double)hi - (double)lo;
        int e = 1; // biased exponent of the grid step (a normal float): 255 steps must span the extent
        if (extent > 0.0) {
            int ex;
            std::frexp(extent / 255.0, &ex);
            e = std::min(254, std::max(1, ex + 127));
            while (e < 254 && std::ldexp(255.0, e - 127) < extent) ++e;
        }
        q.exp[a] = (uint8_t)e;
        const double step = std::ldexp(1.0, e - 127);
        for (int j = 0; j < 4; ++j) {
            if (j >= n) { q.qlo[a][j] = 255; q.qhi[a][j] = 0; continue; } // never tested: the slot's child is RT_REF_NONE
            const double l = (double)s.lo[a][j] - (double)lo, h = (double)s.hi[a][j] - (double)lo;
            double ql = std::floor(l / step), qh = std::ceil(h / step);
            while (ql > 0.0 && ql * step > l) ql -= 1.0;
            while (qh < 255.0 && qh * step < h) qh += 1.0;
            q.qlo[a][j] = (uint8_t)std::min(255.0, std::max(0.0, ql));
            q.qhi[a][j] = (uint8_t)std::min(255.0, std::max(0.0, qh));
        }
    }
}

// Surface-area ratio of child j's 8-bit box to its f32 box, capped at 16 (-1: no area to compare).
inline double quantised_area_ratio(const QNode& s, const CQNode& q, int j) {
    double e32[3], e8[3];
    for (int a = 0; a < 3; ++a) {
        const double step = std::ldexp(1.0, (int)q.exp[a] - 127);
        e32[a] = (double)s.hi[a][j] - (double)s.lo[a][j];
        e8[a] = ((double)q.qhi[a][j] - (double)q.qlo[a][j]) * step;
    }
    const double a32 = e32[0] * e32[1] + e32[1] * e32[2] + e32[2] * e32[0], a8 = e8[0] * e8[1] + e8[1] * e8[2] + e8[2] * e8[0];
    if (!(a32 > 0.0) || !std::isfinite(a8)) return -1.0;
    return std::min(a8 / a32, 16.0);
}

} // namespace rt
#endif
