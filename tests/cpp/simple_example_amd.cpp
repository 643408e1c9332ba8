// The reference's test/simple_example.cpp flow (same includes, same types, same scene, same builder call) on top of the
// C++20 mirror: two triangles, parallel High build, permuted PrecomputedTri, one closest-hit ray with the fast slab test.
// The per-ray lambda traversal of the reference becomes one batched device call. Expected output (known answer of the
// reference): primitive 1, distance 1, barycentric coordinates -0, 0.5.
#include <bvh/v2/bvh.h>
#include <bvh/v2/vec.h>
#include <bvh/v2/ray.h>
#include <bvh/v2/node.h>
#include <bvh/v2/default_builder.h>
#include <bvh/v2/mini_tree_builder.h>
#include <bvh/v2/reinsertion_optimizer.h>
#include <bvh/v2/thread_pool.h>
#include <bvh/v2/executor.h>
#include <bvh/v2/stack.h>
#include <bvh/v2/tri.h>

#include <iostream>

using Scalar = float;
using Vec3 = bvh::v2::Vec<Scalar, 3>;
using BBox = bvh::v2::BBox<Scalar, 3>;
using Tri = bvh::v2::Tri<Scalar, 3>;
using Node = bvh::v2::Node<Scalar, 3>;
using Bvh = bvh::v2::Bvh<Node>;
using Ray = bvh::v2::Ray<Scalar, 3>;
using Hit = bvh::v2::amd::Hit<Scalar>;

int main() {
    std::vector<Tri> tris;
    tris.emplace_back(Vec3(1.0, -1.0, 1.0), Vec3(1.0, 1.0, 1.0), Vec3(-1.0, 1.0, 1.0));
    tris.emplace_back(Vec3(1.0, -1.0, 1.0), Vec3(-1.0, -1.0, 1.0), Vec3(-1.0, 1.0, 1.0));

    bvh::v2::ThreadPool thread_pool;
    bvh::v2::ParallelExecutor executor(thread_pool);

    std::vector<BBox> bboxes(tris.size());
    std::vector<Vec3> centers(tris.size());
    executor.for_each(0, tris.size(), [&](size_t begin, size_t end) {
        for (size_t i = begin; i < end; ++i) {
            bboxes[i] = tris[i].get_bbox();
            centers[i] = tris[i].get_center();
        }
    });

    typename bvh::v2::DefaultBuilder<Node>::Config config;
    config.quality = bvh::v2::DefaultBuilder<Node>::Quality::High;
    auto bvh = bvh::v2::DefaultBuilder<Node>::build(thread_pool, bboxes, centers, config);

    auto precomputed_tris = bvh::v2::amd::permuted_triangles(bvh, std::span<const Tri>(tris));

    std::vector<Ray> rays{ Ray(Vec3(0., 0., 0.), Vec3(0., 0., 1.), 0., 100.) };
    std::vector<Hit> hits(rays.size());
    bvh::v2::amd::intersect_batch<false, false>(bvh, precomputed_tris, std::span<const Ray>(rays), std::span<Hit>(hits));

    // DefaultBuilder(pool, High) == MiniTreeBuilder with pruning at 0.01 + ReinsertionOptimizer (default_builder.h:41-44, :65-73);
    // parallel_threshold = 1 keeps the two triangles on the mini-tree path instead of DefaultBuilder's serial fallback
    typename bvh::v2::DefaultBuilder<Node>::Config forced = config;
    forced.sah = bvh::v2::SplitHeuristic<Scalar>(0, 1.f);                   // top_down_sah_builder.h:29, the default spelled out
    forced.parallel_threshold = 1;
    typename bvh::v2::MiniTreeBuilder<Node>::Config mini;
    mini.parallel_threshold = 1;
    auto by_hand = bvh::v2::MiniTreeBuilder<Node>::build(thread_pool, bboxes, centers, mini);
    // MiniTreeBuilder<Node, MortonCode>: the reference only reads MortonCode in a debug assert (mini_tree_builder.h:171; the codes
    // themselves are size_t, :183-186), so a 64-bit MortonCode builds the same tree — here too
    {
        typename bvh::v2::MiniTreeBuilder<Node, uint64_t>::Config mini64;
        mini64.parallel_threshold = 1;
        if (!(by_hand == bvh::v2::MiniTreeBuilder<Node, uint64_t>::build(thread_pool, bboxes, centers, mini64))) { std::cout << "MortonCode = uint64_t mismatch" << std::endl; return 2; }
    }
    typename bvh::v2::ReinsertionOptimizer<Node>::Config reinsertion;       // reinsertion_optimizer.h:18-24, the defaults spelled out
    reinsertion.batch_size_ratio = 0.05f;
    reinsertion.max_iter_count = 3;
    bvh::v2::ReinsertionOptimizer<Node>::optimize(thread_pool, by_hand, reinsertion);
    if (!(by_hand == bvh::v2::DefaultBuilder<Node>::build(thread_pool, bboxes, centers, forced))) { std::cout << "MiniTreeBuilder mismatch" << std::endl; return 2; }

    std::cout << "nodes: " << bvh.nodes.size() << ", prim_ids: " << bvh.prim_ids[0] << " " << bvh.prim_ids[1] << "\n";
    if (hits[0].prim != Hit::invalid) {
        std::cout << "Intersection found\n"
                  << "  primitive: " << hits[0].prim << "\n"
                  << "  distance: " << hits[0].t << "\n"
                  << "  barycentric coords.: " << hits[0].u << ", " << hits[0].v << std::endl;
        return 0;
    }
    std::cout << "No intersection found" << std::endl;
    return 1;
}
