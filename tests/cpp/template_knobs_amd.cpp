// Template arguments of the reference that are not its defaults, through the C++ mirror:
//   BinnedSahBuilder<Node, BinCount>            BinCount = 4, 16, 32            (binned_sah_builder.h:18)
//   Node<T, Dim, IndexBits, PrimCountBits>      32-bit index with a 2-bit count, 64-bit index with a 6-bit count (node.h:21-22, index.h:32-41)
// usage: template_knobs_amd <boxes+centers file> <n> <out dir>     the input holds n x {min xyz, max xyz} then n x center, float
// Writes one Bvh::serialize stream per case; tests/test_cpp_mirror.py compares each with the reference's stream for the same build
// (re-packed to the Node's Index for the second group). Also checks that what a Node cannot represent is refused, not truncated.
#include <bvh/v2/bvh.h>
#include <bvh/v2/node.h>
#include <bvh/v2/stream.h>
#include <bvh/v2/ray.h>
#include <bvh/v2/tri.h>
#include <bvh/v2/stack.h>
#include <bvh/v2/binned_sah_builder.h>
#include <bvh/v2/sweep_sah_builder.h>
#include <bvh/v2/default_builder.h>

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

using Scalar = float;
using Vec3 = bvh::v2::Vec<Scalar, 3>;
using BBox = bvh::v2::BBox<Scalar, 3>;

template <typename Bvh>
static bool save(const Bvh& bvh, const std::string& path) {
    std::ofstream file(path, std::ofstream::binary);
    if (!file) return false;
    bvh::v2::StdOutputStream stream(file);
    bvh.serialize(stream);
    return true;
}

template <size_t BinCount>
static bool binned(const std::vector<BBox>& bb, const std::vector<Vec3>& cc, const std::string& dir) {
    using Node = bvh::v2::Node<Scalar, 3>;
    typename bvh::v2::BinnedSahBuilder<Node, BinCount>::Config config;
    auto bvh = bvh::v2::BinnedSahBuilder<Node, BinCount>::build(bb, cc, config);
    config.min_leaf_size = 2; config.max_leaf_size = 5;
    auto small = bvh::v2::BinnedSahBuilder<Node, BinCount>::build(bb, cc, config);
    return save(bvh, dir + "/bins" + std::to_string(BinCount) + ".bin") && save(small, dir + "/bins" + std::to_string(BinCount) + "_leaf2to5.bin");
}

int main(int argc, char** argv) {
    if (argc < 4) { std::fprintf(stderr, "usage: %s <input> <n> <out dir>\n", argv[0]); return 1; }
    const size_t n = std::strtoul(argv[2], nullptr, 10);
    const std::string dir = argv[3];
    std::vector<BBox> bb(n);
    std::vector<Vec3> cc(n);
    {
        std::ifstream in(argv[1], std::ifstream::binary);
        std::vector<float> raw(9 * n);
        if (!in.read(reinterpret_cast<char*>(raw.data()), static_cast<std::streamsize>(raw.size() * sizeof(float)))) { std::fprintf(stderr, "short input\n"); return 1; }
        for (size_t i = 0; i < n; ++i) {
            bb[i] = BBox(Vec3(raw[6 * i], raw[6 * i + 1], raw[6 * i + 2]), Vec3(raw[6 * i + 3], raw[6 * i + 4], raw[6 * i + 5]));
            cc[i] = Vec3(raw[6 * n + 3 * i], raw[6 * n + 3 * i + 1], raw[6 * n + 3 * i + 2]);
        }
    }
    if (!binned<4>(bb, cc, dir) || !binned<16>(bb, cc, dir) || !binned<32>(bb, cc, dir)) { std::fprintf(stderr, "cannot write to %s\n", dir.c_str()); return 1; }
    // the default through the same template is the plain builder
    if (!(bvh::v2::BinnedSahBuilder<bvh::v2::Node<Scalar, 3>, 8>::build(bb, cc) == bvh::v2::BinnedSahBuilder<bvh::v2::Node<Scalar, 3>>::build(bb, cc))) {
        std::fprintf(stderr, "BinCount = 8 differs from the default\n"); return 2;
    }

    // ---- Node<float, 3, 32, 2>: leaves of at most 3 primitives ----------------------------------------------------------------
    {
        using Node = bvh::v2::Node<Scalar, 3, 32, 2>;
        static_assert(sizeof(Node) == 28 && Node::Index::max_prim_count == 3 && Node::Index::max_first_id == (1u << 30) - 1);
        typename bvh::v2::SweepSahBuilder<Node>::Config config;
        config.max_leaf_size = 3;
        auto bvh = bvh::v2::SweepSahBuilder<Node>::build(bb, cc, config);
        for (auto& node : bvh.nodes) if (node.index.prim_count() > 3) { std::fprintf(stderr, "a leaf of more than 3 primitives\n"); return 2; }
        if (!save(bvh, dir + "/count2bits_sweep.bin")) return 1;
        // host edit -> device -> host keeps the packing (refit pushes the mirror's nodes and pulls them back)
        auto copy = bvh.nodes;
        bvh.refit();
        if (!(copy == bvh.nodes)) { std::fprintf(stderr, "refit changed a finished tree\n"); return 2; }
        // the device walks the same tree: one ray through the middle of the first primitive's box
        const auto mid = (bb[bvh.prim_ids[0]].min + bb[bvh.prim_ids[0]].max) * Scalar(0.5);
        bvh::v2::Ray<Scalar, 3> ray(mid - Vec3(0, 0, 10), Vec3(0, 0, 1), 0, 100);
        bvh::v2::SmallStack<typename Node::Index, 64> stack;
        size_t visited = 0;
        bvh.template intersect<false, true>(ray, bvh.get_root().index, stack, [&](size_t b, size_t e) { visited += e - b; return false; });
        if (!visited) { std::fprintf(stderr, "the ray met no leaf\n"); return 2; }
        // a tree this Node cannot hold is refused, not truncated: six primitives that must stay one leaf (top_down_sah_builder.h:89)
        config.min_leaf_size = 6; config.max_leaf_size = 8;
        const std::vector<BBox> six_bb(bb.begin(), bb.begin() + 6);
        const std::vector<Vec3> six_cc(cc.begin(), cc.begin() + 6);
        bool refused = false;
        try { (void)bvh::v2::SweepSahBuilder<Node>::build(six_bb, six_cc, config); } catch (const std::exception&) { refused = true; }
        if (!refused) { std::fprintf(stderr, "a 6-primitive leaf accepted by a 2-bit count\n"); return 2; }
        using Wide = bvh::v2::Node<Scalar, 3, 64, 6>;         // ... and held by a Node that has the bits
        typename bvh::v2::SweepSahBuilder<Wide>::Config wide_config;
        wide_config.min_leaf_size = 6; wide_config.max_leaf_size = 8;
        const auto one_leaf = bvh::v2::SweepSahBuilder<Wide>::build(six_bb, six_cc, wide_config);
        if (one_leaf.nodes.size() != 1 || one_leaf.nodes[0].index.prim_count() != 6) { std::fprintf(stderr, "expected one leaf of six primitives\n"); return 2; }
    }
    // ---- Node<float, 3, 64, 6>: 64-bit index word behind float bounds ---------------------------------------------------------
    {
        using Node = bvh::v2::Node<Scalar, 3, 64, 6>;
        static_assert(sizeof(typename Node::Index::Type) == 8 && Node::Index::max_prim_count == 63);
        typename bvh::v2::DefaultBuilder<Node>::Config config;
        config.quality = bvh::v2::DefaultBuilder<Node>::Quality::High;
        auto bvh = bvh::v2::DefaultBuilder<Node>::build(bb, cc, config);
        if (!save(bvh, dir + "/index64_high.bin")) return 1;
        auto sub = bvh.extract_bvh(bvh.get_root().index.first_id());
        if (!save(sub, dir + "/index64_high_sub.bin")) return 1;
    }
    std::printf("ok\n");
    return 0;
}
