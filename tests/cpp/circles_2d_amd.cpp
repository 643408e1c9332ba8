// The 2D family through the C++ mirror: Bvh<Node<float, 2>> over circles, serial DefaultBuilder, batch traversal.
// Prints "nodes <N> hits <H> first <prim> <t0>"; tests/test_cpp_mirror.py compares with the reference (oracle/_ref).
#include <bvh/v2/bvh.h>
#include <bvh/v2/default_builder.h>
#include <bvh/v2/binned_sah_builder.h>
#include <bvh/v2/sweep_sah_builder.h>
#include <bvh/v2/sphere.h>
#include <bvh/v2/ray.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

using Scalar = float;
using Vec2 = bvh::v2::Vec<Scalar, 2>;
using BBox = bvh::v2::BBox<Scalar, 2>;
using Circle = bvh::v2::Sphere<Scalar, 2>;
using Node = bvh::v2::Node<Scalar, 2>;
using Bvh = bvh::v2::Bvh<Node>;
using Ray = bvh::v2::Ray<Scalar, 2>;

int main(int argc, char** argv) {
    const size_t n = argc > 1 ? std::strtoul(argv[1], nullptr, 10) : 4096;
    std::vector<Circle> circles(n);
    std::vector<BBox> bboxes(n);
    std::vector<Vec2> centers(n);
    unsigned long long s = 12345;                             // same LCG as the test's numpy generator
    auto next = [&] { s = s * 6364136223846793005ull + 1442695040888963407ull; return static_cast<Scalar>((s >> 40) * (1.0 / 16777216.0)); };
    for (size_t i = 0; i < n; ++i) {
        const Scalar x = next(), y = next(), r = next();      // (sequenced: argument evaluation order is unspecified)
        circles[i].center = Vec2(x, y);
        circles[i].radius = Scalar(0.001) + Scalar(0.004) * r;
        bboxes[i] = circles[i].get_bbox();
        centers[i] = circles[i].get_center();
    }
    typename bvh::v2::DefaultBuilder<Node>::Config config;
    config.quality = bvh::v2::DefaultBuilder<Node>::Quality::High;
    Bvh bvh = bvh::v2::DefaultBuilder<Node>::build(bboxes, centers, config);

    std::vector<Circle> ordered(n);
    for (size_t i = 0; i < n; ++i) ordered[i] = circles[bvh.prim_ids[i]];
    bvh::v2::amd::DeviceArray<Circle> d_circles{std::span<const Circle>(ordered)};
    std::vector<Ray> rays;
    for (int i = 0; i < 1000; ++i) rays.push_back(Ray(Vec2(Scalar(-0.1), Scalar(i) / 1000), Vec2(1, 0), 0, 100));
    std::vector<bvh::v2::amd::Hit<Scalar>> hits(rays.size());
    bvh::v2::amd::intersect_batch<false, true>(bvh, d_circles, std::span<const Ray>(rays), std::span(hits));
    size_t count = 0;
    for (auto& h : hits) count += h.prim != bvh::v2::amd::Hit<Scalar>::invalid;
    // the explicit builders: Medium == SweepSahBuilder; BinnedSahBuilder == Quality::Low (default_builder.h:49-62)
    typename bvh::v2::DefaultBuilder<Node>::Config low;
    low.quality = bvh::v2::DefaultBuilder<Node>::Quality::Low;
    const bool binned_ok = bvh::v2::BinnedSahBuilder<Node>::build(bboxes, centers) == bvh::v2::DefaultBuilder<Node>::build(bboxes, centers, low);
    typename bvh::v2::DefaultBuilder<Node>::Config med;
    med.quality = bvh::v2::DefaultBuilder<Node>::Quality::Medium;
    const bool sweep_ok = bvh::v2::SweepSahBuilder<Node>::build(bboxes, centers) == bvh::v2::DefaultBuilder<Node>::build(bboxes, centers, med);
    if (!binned_ok || !sweep_ok) { std::fprintf(stderr, "explicit builders differ from DefaultBuilder\n"); return 2; }
    std::printf("nodes %zu hits %zu first %u %.9g\n", bvh.nodes.size(), count, hits[500].prim, double(hits[500].t));
    for (auto& h : hits) std::printf("%u %.9g %.9g\n", h.prim, double(h.t), double(h.u));
    return 0;
}
