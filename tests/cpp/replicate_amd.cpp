// Multi-GPU through the C++20 mirror (bvh::v2::amd::replicate / broadcast over bvhXX_replicate / bvhXX_broadcast): the scene of
// test/simple_example.cpp grown to a strip of triangles, one copy per GPU of the box, every device traces its contiguous shard
// of the rays and the concatenation equals the whole batch traced on device 0. Plain g++, no HIP headers.
#include <bvh/v2/bvh.h>
#include <bvh/v2/vec.h>
#include <bvh/v2/ray.h>
#include <bvh/v2/node.h>
#include <bvh/v2/default_builder.h>
#include <bvh/v2/thread_pool.h>
#include <bvh/v2/tri.h>

#include <cstring>
#include <iostream>
#include <numeric>

using Scalar = float;
using Vec3 = bvh::v2::Vec<Scalar, 3>;
using BBox = bvh::v2::BBox<Scalar, 3>;
using Tri = bvh::v2::Tri<Scalar, 3>;
using Node = bvh::v2::Node<Scalar, 3>;
using Bvh = bvh::v2::Bvh<Node>;
using Ray = bvh::v2::Ray<Scalar, 3>;
using Hit = bvh::v2::amd::Hit<Scalar>;
using PTri = bvh::v2::PrecomputedTri<Scalar>;
namespace amd = bvh::v2::amd;

int main() {
    const int g = bvh_amd_device_count();
    if (g < 1) { std::cerr << "no ROCm-capable device: " << bvh_amd_last_error() << std::endl; return 3; }
    std::vector<Tri> tris;
    for (int i = 0; i < 3000; ++i) {
        const Scalar x = Scalar(i) * Scalar(0.01), z = Scalar(1) + Scalar(i % 7);
        tris.emplace_back(Vec3(x + 1, -1, z), Vec3(x + 1, 1, z), Vec3(x - 1, 1, z));
        tris.emplace_back(Vec3(x + 1, -1, z), Vec3(x - 1, -1, z), Vec3(x - 1, 1, z));
    }
    std::vector<BBox> bboxes(tris.size());
    std::vector<Vec3> centers(tris.size());
    for (size_t i = 0; i < tris.size(); ++i) { bboxes[i] = tris[i].get_bbox(); centers[i] = tris[i].get_center(); }
    bvh::v2::ThreadPool pool;
    bvh_amd_device_select(0);
    auto bvh = bvh::v2::DefaultBuilder<Node>::build(pool, bboxes, centers, {});
    auto prims = amd::permuted_triangles(bvh, std::span<const Tri>(tris));

    std::vector<Ray> rays;
    for (int j = 0; j < 10007; ++j) rays.emplace_back(Vec3(Scalar(j) * Scalar(0.003), Scalar(j % 5) * Scalar(0.3) - Scalar(0.6), 0), Vec3(0, Scalar(0.01), 1), Scalar(0), Scalar(100));
    std::vector<Hit> want(rays.size()), got(rays.size());
    amd::intersect_batch<false, true>(bvh, prims, std::span<const Ray>(rays), std::span<Hit>(want));

    std::vector<int> devices(static_cast<size_t>(g));
    std::iota(devices.begin(), devices.end(), 0);
    auto scenes = amd::replicate(bvh, prims, std::span<const int>(devices));
    const size_t per = (rays.size() + size_t(g) - 1) / size_t(g);
    for (int k = 0; k < g; ++k) {
        bvh_amd_device_select(k);
        const size_t b = std::min(rays.size(), size_t(k) * per), e = std::min(rays.size(), b + per);
        if (!(scenes[size_t(k)].bvh == bvh)) { std::cout << "copy " << k << " differs from the original BVH" << std::endl; return 2; }
        if (e == b) continue;
        amd::DeviceArray<Ray> d_rays(std::span<const Ray>(rays.data() + b, e - b));
        amd::DeviceArray<Hit> d_hits(e - b);
        amd::check(amd::Api<Scalar, 3>::trace_tri(scenes[size_t(k)].bvh.device(), scenes[size_t(k)].prims, d_rays.data(), e - b, BVH_AMD_RAY_ROBUST, d_hits.data()), "trace");
        d_hits.download(std::span<Hit>(got.data() + b, e - b));
    }
    bvh_amd_device_select(0);
    if (std::memcmp(got.data(), want.data(), want.size() * sizeof(Hit)) != 0) { std::cout << "sharded hits differ" << std::endl; return 2; }

    // the one-process-per-GPU form with a communicator of one rank
    unsigned char id[BVH_AMD_COMM_ID_BYTES];
    amd::check(bvh_amd_comm_unique_id(id), "comm_unique_id");
    bvh_amd_comm* comm = bvh_amd_comm_create(id, 1, 0);
    if (!comm) { std::cout << "comm_create: " << bvh_amd_last_error() << std::endl; return 2; }
    {
        auto mine = amd::broadcast<Node, PTri>(comm, 0, &bvh, &prims);
        if (!(mine.bvh == bvh) || mine.prim_count != tris.size()) { std::cout << "broadcast returned another scene" << std::endl; return 2; }
    }
    bvh_amd_comm_destroy(comm);
    size_t hits = 0;
    for (const Hit& h : want) hits += h.prim != Hit::invalid;
    std::cout << "replicate_amd ok: " << g << " device(s), " << hits << " hits of " << rays.size() << std::endl;
    return hits > 100 ? 0 : 2;
}
