// Compact traversal records (EXPERIMENTAL, opt-in: BVH_AMD_PAIRS=compact; 3D, float and double).
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

template <typename T> struct CompactPairT;
template <> struct alignas(32) CompactPairT<float> {
    float fresh[6];                            // plane k of whichever child does NOT inherit the parent's plane k
    uint32_t li, ri;                           // index words (first_id << 4 | count); bits 29..31: inheritance bits of planes 0..2 / 3..5
};
template <> struct alignas(64) CompactPairT<double> {     // three 16-byte requests for the planes + one for the index words (PairNode<double>: seven)
    double fresh[6];
    uint32_t li, ri;
    uint32_t pad[2];
};
using CompactPair = CompactPairT<float>;
static_assert(sizeof(CompactPairT<float>) == 32 && sizeof(CompactPairT<double>) == 64, "two / four 16-byte requests");

BVH_AMD_HD uint32_t compact_float_bits(float x) { return __builtin_bit_cast(uint32_t, x); }
BVH_AMD_HD float compact_bits_float(uint32_t u) { return __builtin_bit_cast(float, u); }

// Bounds are {minx, maxx, miny, maxy, minz, maxz} (node.h:31-37). Returns false when the pair is not representable.
template <typename T>
BVH_AMD_HD bool compact_encode(const T parent[6], const T lb[6], const T rb[6], uint32_t li, uint32_t ri, CompactPairT<T>& out) {
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
//   have_box: x[0..5] = CompactPair::fresh, cli / cri = its index words, `box` = the box of the node whose children these are;
//   otherwise: x[0..5] = PairNode::lb, y[0..5] = PairNode::rb, fli / fri = its index words; `box`, cli, cri are not looked at.
// (written out plane by plane: every index is a constant, so the arrays stay in registers on the device)
template <typename T>
BVH_AMD_HD void compact_unpack_planes(bool have_box, const T box[6], const T x[6], const T y[6], uint32_t cli, uint32_t cri, uint32_t fli, uint32_t fri,
                                      T lb[6], T rb[6], uint32_t& li, uint32_t& ri) {
    const uint32_t mask = have_box ? ((cli >> kCompactMaskShift) | ((cri >> kCompactMaskShift) << 3)) : 0u;
#define BVH_AMD_UNPACK_PLANE(k)                                                                        \
    {                                                                                                  \
        const bool left_inherits = (mask & (1u << (k))) != 0;                                          \
        lb[k] = left_inherits ? box[k] : x[k];                                                         \
        rb[k] = have_box ? (left_inherits ? x[k] : box[k]) : y[k];                                     \
    }
    BVH_AMD_UNPACK_PLANE(0) BVH_AMD_UNPACK_PLANE(1) BVH_AMD_UNPACK_PLANE(2)
    BVH_AMD_UNPACK_PLANE(3) BVH_AMD_UNPACK_PLANE(4) BVH_AMD_UNPACK_PLANE(5)
#undef BVH_AMD_UNPACK_PLANE
    li = have_box ? (cli & kCompactIndexMask) : fli;
    ri = have_box ? (cri & kCompactIndexMask) : fri;
}

// float: w[0..7] is the CompactPair (have_box) or w[0..13] the first fourteen words of the PairNode<float>.
BVH_AMD_HD void compact_unpack(bool have_box, const float box[6], const uint32_t w[14], float lb[6], float rb[6], uint32_t& li, uint32_t& ri) {
    const float x[6] = { compact_bits_float(w[0]), compact_bits_float(w[1]), compact_bits_float(w[2]),
                         compact_bits_float(w[3]), compact_bits_float(w[4]), compact_bits_float(w[5]) };
    const float y[6] = { compact_bits_float(w[6]), compact_bits_float(w[7]), compact_bits_float(w[8]),
                         compact_bits_float(w[9]), compact_bits_float(w[10]), compact_bits_float(w[11]) };
    compact_unpack_planes<float>(have_box, box, x, y, w[6], w[7], w[12], w[13], lb, rb, li, ri);
}

} // namespace bvh_amd
