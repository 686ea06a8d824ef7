// tilewalk.h -- XCD-aware order in which a persistent workgroup walks the conv tiles.
//
// Workgroups are dealt to the 8 XCDs round-robin (workgroup i runs on XCD i % 8) and every XCD has its own L2.  With
// the plain walk  tile = blockIdx.x + k * gridDim.x  the x- and y-neighbours of a tile are always processed on OTHER
// XCDs, so the halo rows/columns two tiles share -- and the 128-byte lines their 16-byte halo chunks drag in -- are
// fetched once per L2.  Here XCD x owns the contiguous tile range [x T/8, (x+1) T/8) and its gridDim.x / 8 workgroups
// walk it together, so the 32 tiles in flight on an XCD are 4 adjacent tile rows of one image and halos hit in L2.
// Falls back to the plain walk when the counts do not divide.
#pragma once

namespace pnp {

struct TileWalk { int first, step, limit; };

__device__ __forceinline__ TileWalk tile_walk(int ntiles) {
    const int G = (int)gridDim.x, bid = (int)blockIdx.x;
    if ((G & 7) == 0 && ntiles % G == 0) {
        const int per_xcd = ntiles >> 3, xcd = bid & 7, slot = bid >> 3;
        return {xcd * per_xcd + slot, G >> 3, (xcd + 1) * per_xcd};
    }
    return {bid, G, ntiles};
}

}  // namespace pnp
