// TEST INFRASTRUCTURE (CPU, no GPU): exercises bvh_amd/csrc/compact_pair.h — the encoder and unpacker that the EXPERIMENTAL
// compact-record traversal kernel (trace_kernel_compact, bvh_amd/csrc/trace_body.inc) compiles for the device — by walking
// rays with one scalar "lane" that follows the kernel's own state machine (box held after a descent, lost after a pop) over
// records encoded exactly as k_compact_encode (bvh_amd/csrc/upload.hip) encodes them. tests/test_compact_pairs.py compares
// the hits and the visit counters with the oracle's traversal of the same tree. Nothing here is shipped.
//
// Built by the test with: g++ -std=c++20 -O2 -mavx2 -mfma -ffp-contract=off -shared -fPIC (the oracle's pinned flags).
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../bvh_amd/csrc/compact_pair.h"

namespace {

struct Node28 { float bounds[6]; uint32_t index; };              // reference node.h:31-37 (float, 3D)
struct Pair64 { float lb[6], rb[6]; uint32_t li, ri, pad[2]; };   // bvh_amd/csrc/common.h PairNode<float>
static_assert(sizeof(Node28) == 28 && sizeof(Pair64) == 64);

constexpr uint32_t kCountBits = 4, kCountMask = 15u, kInvalid = 0xFFFFFFFFu;
constexpr float kFltMax = 3.402823466e+38f, kEps = 1.1920928955078125e-07f;

inline float pick_min(float a, float b) { return a < b ? a : b; }   // utils.h:41-43
inline float pick_max(float a, float b) { return a > b ? a : b; }
inline float dot3(const float* a, const float* b) { return ((0.0f + a[0] * b[0]) + a[1] * b[1]) + a[2] * b[2]; }   // vec.h:98-100
inline bool finite_(float x) { return (bvh_amd::compact_float_bits(x) & 0x7F800000u) != 0x7F800000u; }

} // namespace

extern "C" {

// Encodes every pair like k_compact_encode. Returns 0, or 1 / 2 for a pair that is not representable / an index out of range.
// out_pairs: n_pairs x 64 bytes, out_compact: n_pairs x 32 bytes (zero where no parent pair points: the root's pair).
int compact_encode_tree(const void* nodes28, size_t node_count, void* out_pairs, void* out_compact) {
    const Node28* nodes = static_cast<const Node28*>(nodes28);
    const size_t n_pairs = (node_count - 1) / 2;
    Pair64* pairs = static_cast<Pair64*>(out_pairs);
    bvh_amd::CompactPair* recs = static_cast<bvh_amd::CompactPair*>(out_compact);
    std::memset(recs, 0, n_pairs * sizeof(bvh_amd::CompactPair));
    for (size_t p = 0; p < n_pairs; ++p) {                            // relayout_pairs (upload.hip)
        Pair64& r = pairs[p];
        for (int k = 0; k < 6; ++k) { r.lb[k] = nodes[2 * p + 1].bounds[k]; r.rb[k] = nodes[2 * p + 2].bounds[k]; }
        r.li = nodes[2 * p + 1].index; r.ri = nodes[2 * p + 2].index; r.pad[0] = r.pad[1] = 0;
    }
    int bad = 0;
    for (size_t q = 0; q < n_pairs; ++q) {
        for (int side = 0; side < 2; ++side) {
            const uint32_t word = side ? pairs[q].ri : pairs[q].li;
            if (word & kCountMask) continue;
            const uint32_t p = word >> (kCountBits + 1);
            if (p >= n_pairs) { bad |= 2; continue; }
            bvh_amd::CompactPair c;
            if (bvh_amd::compact_encode(side ? pairs[q].rb : pairs[q].lb, pairs[p].lb, pairs[p].rb, pairs[p].li, pairs[p].ri, c)) recs[p] = c;
            else bad |= 1;
        }
    }
    return bad;
}

// One lane of trace_body.inc with BVH_TRACE_COMPACT 1. counters: [0] pairs visited, [1] primitive tests, [2] pair fetches made
// with a box (two requests), [3] without (four requests).
// far_cache != 0: additionally model a one-entry register cache holding the box of the most recently pushed far child (the variant
// considered in DESIGN.md §8): a pop that takes exactly that entry keeps its box.
static int g_far_cache = 0;
void compact_walk_set_far_cache(int on) { g_far_cache = on; }

void compact_walk(const void* pairs64, const void* compact32, uint32_t root_index, const float* tris12, const float* rays8, size_t n_rays,
                  int any, int robust, float* hits4, uint64_t* counters) {
    const Pair64* pairs = static_cast<const Pair64*>(pairs64);
    const bvh_amd::CompactPair* recs = static_cast<const bvh_amd::CompactPair*>(compact32);
    std::vector<uint32_t> stack;
    for (size_t r = 0; r < n_rays; ++r) {
        const float* ray = rays8 + 8 * r;
        float org[3], dir[3], inv[3], aux[3];
        bool oct[3];
        for (int k = 0; k < 3; ++k) {                                 // bvh.h:162-165, ray.h:29-48
            org[k] = ray[k]; dir[k] = ray[3 + k];
            const float d = dir[k];
            const float iv = robust ? 1.0f / d : (std::fabs(d) <= kEps ? std::copysign(kFltMax, d) : 1.0f / d);
            inv[k] = iv;
            aux[k] = robust ? (finite_(iv) ? bvh_amd::compact_bits_float(bvh_amd::compact_float_bits(iv) + 2u) : iv) : (-iv) * org[k];
            oct[k] = std::signbit(d);
        }
        const float tmin = ray[6];
        float tmax = ray[7];
        uint32_t hit_prim = kInvalid;
        float hit_t = tmax, hit_u = 0, hit_v = 0;
        stack.clear();
        uint32_t top = root_index;
        float box[6] = {0, 0, 0, 0, 0, 0};
        bool have_box = false, done = false;
        float fbox[6] = {0, 0, 0, 0, 0, 0};            // far-box cache: the box of stack entry number fsp - 1
        size_t fsp = 0;                                // 0 = empty
        auto pop_box = [&]() {                         // called right after stack.pop_back(): stack.size() is the popped entry's number
            have_box = false;
            if (g_far_cache && fsp == stack.size() + 1) { for (int k = 0; k < 6; ++k) box[k] = fbox[k]; have_box = true; }
            fsp = 0;
        };
        while (!done) {
            if ((top & kCountMask) == 0) {
                const uint32_t p = top >> (kCountBits + 1);
                uint32_t w[14] = {0};
                if (have_box) { std::memcpy(w, &recs[p], 32); ++counters[2]; }           // words 0..7 only: the two requests
                else { std::memcpy(w, &pairs[p], 56); ++counters[3]; }
                float lb[6], rb[6];
                uint32_t li, ri;
                bvh_amd::compact_unpack(have_box, box, w, lb, rb, li, ri);
                ++counters[0];
                float l0 = tmin, l1 = tmax, r0 = tmin, r1 = tmax;                       // node.h:105-117
                for (int k = 0; k < 3; ++k) {
                    const float ln = oct[k] ? lb[2 * k + 1] : lb[2 * k], lf = oct[k] ? lb[2 * k] : lb[2 * k + 1];
                    const float rn = oct[k] ? rb[2 * k + 1] : rb[2 * k], rf = oct[k] ? rb[2 * k] : rb[2 * k + 1];
                    float la, lz, ra, rz;
                    if (robust) {                                                       // node.h:74-75
                        la = (ln - org[k]) * inv[k]; lz = (lf - org[k]) * aux[k];
                        ra = (rn - org[k]) * inv[k]; rz = (rf - org[k]) * aux[k];
                    } else {                                                            // node.h:85-86
                        la = std::fma(ln, inv[k], aux[k]); lz = std::fma(lf, inv[k], aux[k]);
                        ra = std::fma(rn, inv[k], aux[k]); rz = std::fma(rf, inv[k], aux[k]);
                    }
                    l0 = pick_max(la, l0); l1 = pick_min(lz, l1);
                    r0 = pick_max(ra, r0); r1 = pick_min(rz, r1);
                }
                const bool hl = l0 <= l1, hr = r0 <= r1;                                // bvh.h:177-180
                if (hl) {
                    uint32_t near_i = li;
                    for (int k = 0; k < 6; ++k) box[k] = lb[k];
                    have_box = true;
                    if (hr) {
                        uint32_t far_i = ri;
                        for (int k = 0; k < 6; ++k) fbox[k] = rb[k];
                        if (!any && l0 > r0) {
                            near_i = ri; far_i = li;
                            for (int k = 0; k < 6; ++k) { box[k] = rb[k]; fbox[k] = lb[k]; }
                        }
                        stack.push_back(far_i);
                        fsp = stack.size();
                    }
                    top = near_i;
                } else if (hr) {
                    top = ri;
                    for (int k = 0; k < 6; ++k) box[k] = rb[k];
                    have_box = true;
                } else if (stack.empty()) {
                    done = true;
                } else {
                    top = stack.back(); stack.pop_back();
                    pop_box();
                }
            } else {
                const uint32_t first = top >> kCountBits, count = top & kCountMask;
                for (uint32_t i = first; i < first + count; ++i) {                      // tri.h:56-74
                    ++counters[1];
                    const float* p = tris12 + 12ull * i;
                    const float c[3] = { p[0] - org[0], p[1] - org[1], p[2] - org[2] };
                    const float rr[3] = { dir[1] * c[2] - dir[2] * c[1], dir[2] * c[0] - dir[0] * c[2], dir[0] * c[1] - dir[1] * c[0] };
                    const float inv_det = 1.0f / dot3(p + 9, dir);
                    const float u = dot3(rr, p + 6) * inv_det;
                    const float v = dot3(rr, p + 3) * inv_det;
                    const float w = 1.0f - u - v;
                    const float tol = -kEps;
                    if (u >= tol && v >= tol && w >= tol) {
                        const float t = dot3(p + 9, c) * inv_det;
                        if (t >= tmin && t <= tmax) { tmax = t; hit_t = t; hit_u = u; hit_v = v; hit_prim = i; }
                    }
                }
                if (any && hit_prim != kInvalid) done = true;
                else if (stack.empty()) done = true;
                else { top = stack.back(); stack.pop_back(); pop_box(); }
            }
        }
        float* out = hits4 + 4 * r;
        std::memcpy(out, &hit_prim, 4);
        out[1] = hit_t; out[2] = hit_u; out[3] = hit_v;
    }
}

} // extern "C"
