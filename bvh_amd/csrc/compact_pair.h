// Compact traversal records (EXPERIMENTAL, opt-in: BVH_AMD_PAIRS=compact; float / 3D only).
//
// The traversal kernel is bound by the number of 16-byte L1 requests a lane issues per visited node pair (four for the
// 64-byte PairNode, profiles/README.md). The boxes of a sibling pair are redundant with the box of their parent: the parent
// box is the union of the two (reference bbox.h:25-30 via top_down_sah_builder.h:59-72 / reinsertion_optimizer.h:215-226),
// so each of the parent's six planes is numerically equal to that plane of the left or of the right child. A pair is
// therefore six "fresh" planes (the ones that are not the parent's) plus, per plane, one bit saying which child inherits
// the parent's plane: 6 x 4 + 2 x 4 bytes = 32 bytes = TWO requests, with the six bits in the three spare high bits of
// each index word. Decoding needs the box of the node whose children are being fetched; a lane has it when it has just
// descended into that node and does not have it after a stack pop (the stack holds index words only), in which case it
// reads the full 64-byte record instead.
//
// Exactness: a decoded plane is numerically equal to the child's plane and may differ from it only in the sign of a zero.
// The slab test (node.h:68-88) only feeds comparisons, and (+-0 - org) * inv_dir / fma(+-0, inv_dir, inv_org) compare the
// same whichever zero goes in (a zero operand changes at most the sign of a zero result, and 0 * inf is NaN for both), so
// the visit order and the hits are unchanged. Planes are compared numerically, not bitwise, by the encoder; a pair with a
// plane that neither child shares with the parent (hand-edited boxes, NaNs) makes the whole BVH fall back to PairNode.
//
// This header is plain C++ (no HIP intrinsics): the device code and tests/cpp/compact_pair_test.cpp compile the same functions.
#pragma once

#include <cstdint>

#if defined(__HIPCC__)
#define BVH_AMD_HD __host__ __device__ inline
#else
#define BVH_AMD_HD inline
#endif

namespace bvh_amd {

constexpr uint32_t kCompactMaskShift = 29;                         // three inheritance bits above a 29-bit index word
constexpr uint32_t kCompactIndexMask = (1u << kCompactMaskShift) - 1u;

struct alignas(32) CompactPair {
    float fresh[6];                            // plane k of whichever child does NOT inherit the parent's plane k
    uint32_t li, ri;                           // index words (first_id << 4 | count); bits 29..31: inheritance bits of planes 0..2 / 3..5
};
static_assert(sizeof(CompactPair) == 32, "two 16-byte requests");

BVH_AMD_HD uint32_t compact_float_bits(float x) { return __builtin_bit_cast(uint32_t, x); }
BVH_AMD_HD float compact_bits_float(uint32_t u) { return __builtin_bit_cast(float, u); }

// Bounds are {minx, maxx, miny, maxy, minz, maxz} (node.h:31-37). Returns false when the pair is not representable.
BVH_AMD_HD bool compact_encode(const float parent[6], const float lb[6], const float rb[6], uint32_t li, uint32_t ri, CompactPair& out) {
    if ((li | ri) & ~kCompactIndexMask) return false;
    uint32_t mask = 0;
    for (int k = 0; k < 6; ++k) {
        if (lb[k] == parent[k]) { mask |= 1u << k; out.fresh[k] = rb[k]; }        // bit set: the LEFT child inherits plane k
        else if (rb[k] == parent[k]) out.fresh[k] = lb[k];
        else return false;
    }
    out.li = li | ((mask & 7u) << kCompactMaskShift);
    out.ri = ri | ((mask >> 3) << kCompactMaskShift);
    return true;
}

// Unpacks what a lane fetched into the two child boxes and index words.
//   have_box: w[0..7] is the CompactPair of the node whose box is `box`;
//   otherwise: w[0..13] are the first fourteen words of the PairNode<float> (lb[6], rb[6], li, ri); `box` is not read.
BVH_AMD_HD void compact_unpack(bool have_box, const float box[6], const uint32_t w[14], float lb[6], float rb[6], uint32_t& li, uint32_t& ri) {
    const uint32_t mask = have_box ? ((w[6] >> kCompactMaskShift) | ((w[7] >> kCompactMaskShift) << 3)) : 0u;
    // (written out plane by plane: every index is a constant, so the arrays stay in registers on the device)
#define BVH_AMD_UNPACK_PLANE(k)                                                                        \
    {                                                                                                  \
        const bool left_inherits = (mask & (1u << (k))) != 0;                                          \
        const float x = compact_bits_float(w[k]);                                                      \
        lb[k] = left_inherits ? box[k] : x;                                                            \
        rb[k] = have_box ? (left_inherits ? x : box[k]) : compact_bits_float(w[6 + (k)]);              \
    }
    BVH_AMD_UNPACK_PLANE(0) BVH_AMD_UNPACK_PLANE(1) BVH_AMD_UNPACK_PLANE(2)
    BVH_AMD_UNPACK_PLANE(3) BVH_AMD_UNPACK_PLANE(4) BVH_AMD_UNPACK_PLANE(5)
#undef BVH_AMD_UNPACK_PLANE
    li = have_box ? (w[6] & kCompactIndexMask) : w[12];
    ri = have_box ? (w[7] & kCompactIndexMask) : w[13];
}

} // namespace bvh_amd
