// The flow of the reference's test/serialize.cpp against the MI355X mirror: build, Bvh::serialize to a file through
// StdOutputStream, Bvh::deserialize through StdInputStream, compare. Added checks: the stream equals the C-ABI's
// bvh3f_serialize byte for byte, a 64-bit IndexType stream round-trips, and refit(leaf_fn) runs the leaf callback.
#include <bvh/v2/bvh.h>
#include <bvh/v2/vec.h>
#include <bvh/v2/ray.h>
#include <bvh/v2/tri.h>
#include <bvh/v2/node.h>
#include <bvh/v2/stream.h>
#include <bvh/v2/stack.h>
#include <bvh/v2/default_builder.h>

#include <fstream>
#include <iostream>
#include <optional>
#include <cstdlib>
#include <sstream>

using Scalar = float;
using Vec3   = bvh::v2::Vec<Scalar, 3>;
using BBox   = bvh::v2::BBox<Scalar, 3>;
using Tri    = bvh::v2::Tri<Scalar, 3>;
using Node   = bvh::v2::Node<Scalar, 3>;
using Bvh    = bvh::v2::Bvh<Node>;

static bool save_bvh(const Bvh& bvh, const std::string& path) {
    std::ofstream file(path, std::ofstream::binary);
    if (!file) return false;
    bvh::v2::StdOutputStream stream(file);
    bvh.serialize(stream);
    return true;
}

static std::optional<Bvh> load_bvh(const std::string& path) {
    std::ifstream file(path, std::ifstream::binary);
    if (!file) return std::nullopt;
    bvh::v2::StdInputStream stream(file);
    return std::make_optional(Bvh::deserialize(stream));
}

int main(int argc, char** argv) {
    const std::string path = argc > 1 ? argv[1] : "bvh.bin";
    std::vector<Tri> tris;
    for (int i = 0; i < 40; ++i) {                            // a strip of quads, two triangles each
        const Scalar x = static_cast<Scalar>(i);
        tris.emplace_back(Vec3(x + 1, -1, 1), Vec3(x + 1, 1, 1), Vec3(x, 1, 1));
        tris.emplace_back(Vec3(x + 1, -1, 1), Vec3(x, -1, 1), Vec3(x, 1, 1));
    }
    std::vector<BBox> bboxes(tris.size());
    std::vector<Vec3> centers(tris.size());
    for (size_t i = 0; i < tris.size(); ++i) { bboxes[i] = tris[i].get_bbox(); centers[i] = tris[i].get_center(); }

    auto bvh = bvh::v2::DefaultBuilder<Node>::build(bboxes, centers);
    if (!save_bvh(bvh, path)) { std::cerr << "Cannot write " << path << std::endl; return 1; }
    auto other = load_bvh(path);
    if (!other) { std::cerr << "Cannot load bvh file" << std::endl; return 1; }
    if (!(bvh == *other) || bvh != *other) { std::cerr << "The deserialized BVH does not match the original one" << std::endl; return 1; }

    // the same bytes as the C-ABI's serializer (which is checked against the reference's stream in tests/)
    std::ostringstream mem;
    bvh::v2::StdOutputStream mem_stream(mem);
    bvh.serialize(mem_stream);
    const std::string mine = mem.str();
    std::string theirs(bvh3f_serialize(bvh.device(), nullptr, 0), '\0');
    bvh3f_serialize(bvh.device(), theirs.data(), theirs.size());
    if (mine != theirs) { std::cerr << "Bvh::serialize differs from bvh3f_serialize" << std::endl; return 1; }

    // 64-bit index stream
    std::stringstream wide;
    bvh::v2::StdOutputStream wide_out(wide);
    bvh.serialize<uint64_t>(wide_out);
    bvh::v2::StdInputStream wide_in(wide);
    if (!(Bvh::deserialize<uint64_t>(wide_in) == bvh)) { std::cerr << "64-bit index stream does not round-trip" << std::endl; return 1; }

    // a short stream yields default values instead of garbage
    std::istringstream empty;
    bvh::v2::StdInputStream empty_in(empty);
    if (empty_in.read<uint32_t>(7u) != 7u) { std::cerr << "short read did not return the default" << std::endl; return 1; }

    // the deserialized tree is usable on the device: refit with a leaf callback that leaves the boxes alone is the identity,
    // one that inflates the leaves grows the root
    size_t leaves = 0;
    other->refit([&](Node&) { ++leaves; });
    if (!(*other == bvh) || leaves == 0) { std::cerr << "refit(leaf_fn) changed a consistent tree" << std::endl; return 1; }
    other->refit([](Node& leaf) { auto b = leaf.get_bbox(); b.min[2] -= 1; leaf.set_bbox(b); });
    if (other->get_root().get_bbox().min[2] != 0) { std::cerr << "refit(leaf_fn) did not propagate" << std::endl; return 1; }

    // traverse_bottom_up: every node once, children before parents; a refit written with it equals Bvh::refit
    {
        std::vector<int> order(bvh.nodes.size(), -1);
        int clock = 0;
        size_t n_leaves = 0, n_inner = 0;
        std::istringstream again(mine);
        bvh::v2::StdInputStream again_in(again);
        Bvh copy = Bvh::deserialize(again_in);
        copy.traverse_bottom_up(
            [&](Node& leaf) { order[static_cast<size_t>(&leaf - copy.nodes.data())] = clock++; ++n_leaves; },
            [&](Node& inner) {
                const size_t id = static_cast<size_t>(&inner - copy.nodes.data()), first = inner.index.first_id();
                if (order[first] < 0 || order[first + 1] < 0) { std::cerr << "parent before child" << std::endl; std::exit(1); }
                order[id] = clock++; ++n_inner;
                inner.set_bbox(copy.nodes[first].get_bbox().extend(copy.nodes[first + 1].get_bbox()));
            });
        if (n_leaves + n_inner != copy.nodes.size() || order[0] != clock - 1 || !(copy == bvh)) {
            std::cerr << "traverse_bottom_up did not visit the tree bottom-up" << std::endl; return 1;
        }
    }

    bvh::v2::GrowingStack<Node::Index> stack;
    stack.push(bvh.get_root().index);
    if (stack.is_empty() || !(stack.pop() == bvh.get_root().index) || !stack.is_empty()) return 1;
    if (Bvh::get_left_sibling_id(2) != 1 || Bvh::get_right_sibling_id(1) != 2 || Bvh::get_sibling_id(1) != 2) return 1;

    std::cout << "The deserialized BVH is the same as the original one (" << bvh.nodes.size() << " nodes, " << mine.size() << " bytes)" << std::endl;
    return 0;
}
