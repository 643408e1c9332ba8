// Developer measurement: how fast does ONE host core run the candidate-heap loop of reinsertion_optimizer.h:88-105 (libstdc++ pop_heap / push_heap)?
// g++ -O2 -std=c++17 tools/heap_host_time.cpp && ./a.out [n_nodes]   (k = n / 20 like the default batch_size_ratio 0.05)
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <functional>
#include <random>
#include <vector>
struct C { size_t id; float cost; bool operator>(const C& o) const { return cost > o.cost; } };
int main(int argc, char** argv) {
    size_t n = argc > 1 ? atol(argv[1]) : 1866438, k = n / 20;
    std::mt19937 rng(1); std::vector<float> cost(n);
    // node areas of a BVH in node order: roughly decreasing with depth, noisy
    for (size_t i = 0; i < n; ++i) cost[i] = std::generate_canonical<float, 24>(rng) / (1.0f + float(i % 4096) * 0.01f);
    for (int rep = 0; rep < 3; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        std::vector<C> h; h.reserve(k);
        for (size_t i = 1; i < k + 1; ++i) h.push_back(C{ i, cost[i] });
        std::make_heap(h.begin(), h.end(), std::greater<>{});
        size_t repl = 0;
        for (size_t i = k + 1; i < n; ++i) if (h.front().cost < cost[i]) { std::pop_heap(h.begin(), h.end(), std::greater<>{}); h.back() = C{ i, cost[i] }; std::push_heap(h.begin(), h.end(), std::greater<>{}); ++repl; }
        auto t1 = std::chrono::steady_clock::now();
        std::printf("n=%zu k=%zu replacements=%zu %.2f ms (%.1f ns per replacement)\n", n, k, repl, std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::nano>(t1 - t0).count() / repl);
    }
}
