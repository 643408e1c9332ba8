// TEST PROGRAM (plain g++, no GPU needed): Bvh::traverse_top_down of the C++20 mirror with a user InnerFn that steers the descent
// (reference bvh.h:68-70, :125-157) on a hand-made tree. Prints the leaf ranges in visit order, one walk per line:
//   line 1: both children everywhere, left first          line 2: both children, RIGHT first (should_swap = true)
//   line 3: only children whose box lies left of x = 2.5   line 4: any-hit walk that stops at the leaf holding primitive 3
#include <bvh/v2/bvh.h>
#include <bvh/v2/node.h>
#include <bvh/v2/stack.h>

#include <cstdio>
#include <tuple>

using Node = bvh::v2::Node<float, 3>;
using Bvh = bvh::v2::Bvh<Node>;
using Index = Node::Index;

static Node make(float x0, float x1, Index index) {
    Node n;
    n.bounds = { x0, x1, 0.f, 1.f, 0.f, 1.f };
    n.index = index;
    return n;
}

int main() {
    // 0: [0, 5] inner -> (1, 2); 1: [0, 3] inner -> (3, 4); 2: [3, 5] inner -> (5, 6); leaves 3..6 hold primitives 0..4
    Bvh bvh;
    bvh.nodes = {
        make(0, 5, Index::make_inner(1)),
        make(0, 3, Index::make_inner(3)), make(3, 5, Index::make_inner(5)),
        make(0, 1, Index::make_leaf(0, 1)), make(1, 3, Index::make_leaf(1, 2)),
        make(3, 4, Index::make_leaf(3, 1)), make(4, 5, Index::make_leaf(4, 1)),
    };
    bvh.prim_ids = { 0, 1, 2, 3, 4 };
    bvh::v2::SmallStack<Index, 8> stack;
    auto leaf = [](size_t begin, size_t end) { std::printf("[%zu,%zu) ", begin, end); return false; };
    bvh.traverse_top_down<false>(bvh.get_root().index, stack, leaf, [](const Node&, const Node&) { return std::make_tuple(true, true, false); });
    std::printf("\n");
    bvh.traverse_top_down<false>(bvh.get_root().index, stack, leaf, [](const Node&, const Node&) { return std::make_tuple(true, true, true); });
    std::printf("\n");
    bvh.traverse_top_down<false>(bvh.get_root().index, stack, leaf, [](const Node& l, const Node& r) {
        return std::make_tuple(l.bounds[0] < 2.5f, r.bounds[0] < 2.5f, false); });
    std::printf("\n");
    bvh.traverse_top_down<true>(bvh.get_root().index, stack,
        [](size_t begin, size_t end) { std::printf("[%zu,%zu) ", begin, end); return begin <= 3 && 3 < end; },
        [](const Node&, const Node&) { return std::make_tuple(true, true, false); });
    std::printf("\n");
    return 0;
}
