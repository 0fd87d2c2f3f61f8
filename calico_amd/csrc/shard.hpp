// shard.hpp — multi-GPU partition rule (host only, no HIP).
//
// Residual blocks are independent given the parameters; ranks exchange only
// the sums JᵀJ, Jᵀr and cost. Observations are split by TIME: rank r owns the
// residual blocks whose spline segment falls in its window, windows being
// contiguous segment ranges balanced by block count. A rank then touches only
// the control points of its window (+ k-1 halo points), so its contribution
// to the banded part of the normal equations is local.
#pragma once
#include <cstdint>
#include <vector>

namespace cal {

// blocks_per_segment[s] = number of residual blocks whose segment is s.
// Returns world+1 boundaries b: rank r owns segments [b[r], b[r+1]).
inline std::vector<int> shard_windows(const std::vector<int64_t>& blocks_per_segment, int world) {
  const int nseg = int(blocks_per_segment.size());
  std::vector<int> b(size_t(world) + 1, nseg);
  b[0] = 0;
  int64_t total = 0;
  for (int64_t v : blocks_per_segment) total += v;
  int64_t acc = 0;
  int r = 1;
  for (int s = 0; s < nseg && r < world; ++s) {
    // cut before segment s once the running count reaches r/world of the total
    while (r < world && acc * world >= total * r) { b[size_t(r)] = s; ++r; }
    acc += blocks_per_segment[size_t(s)];
  }
  while (r < world) { b[size_t(r)] = nseg; ++r; }
  return b;
}

}  // namespace cal
