// Developer tool (CPU): checks the pipelined formulation of the reinsertion optimizer's candidate heap sketched in DESIGN.md §8
// (idea (a)) against libstdc++ itself. NOT product code, NOT a port of anything: it exists to find out whether the scheme is exact
// before anyone writes a kernel for it.
//
// The reference (reinsertion_optimizer.h:89-106) keeps the k best candidates in a libstdc++ min-heap and, for every later node
// whose cost beats the minimum, does pop_heap + overwrite the last slot + push_heap. The ARRAY the heap ends up in is part of the
// result. Here one such replacement is decomposed as:
//   SIFT(v):  v = the old last element; hole at the root; descend while the smaller child (ties: right child; range [0, k-2])
//             has cost <= v, moving it up; drop v where the descent stops            (== __adjust_heap + __push_heap of pop_heap)
//   PUSH(x):  carry x down the ancestor chain of the last slot, swapping from the first chain element with cost > carry onwards;
//             the carry-out becomes the last element                                   (== push_heap's sift-up, top-down)
// and executed as a PIPELINE: a controller owns the chain c[0..L] (root .. last slot) and the siblings of its nodes; it performs
// the on-chain steps of each SIFT and the whole PUSH in its own copies, and hands the rest of the SIFT (a top-down pass through
// an off-chain subtree) to a level-synchronous pipeline in which every in-flight pass advances one level per tick. The simulator
// asserts that no tick has two agents touching the same heap node, and that the final array equals libstdc++'s.
//
//   g++ -std=c++20 -O2 tools/heap_chain_sim.cpp -o /tmp/heap_chain_sim && /tmp/heap_chain_sim [seeds] [max_k]
#include <algorithm>
#include <cassert>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <random>
#include <string>
#include <vector>

struct Cand {
    size_t id = 0;
    float cost = 0;
    bool operator>(const Cand& o) const { return cost > o.cost; }
};

// ---- the reference: libstdc++ (reinsertion_optimizer.h:89-106 with costs given directly) -------------------------------------------
static std::vector<Cand> reference(const std::vector<float>& cost, size_t k, size_t* replacements) {
    std::vector<Cand> h;
    const size_t n = cost.size(), first = std::min(n, k);
    for (size_t i = 0; i < first; ++i) h.push_back(Cand{i, cost[i]});
    std::make_heap(h.begin(), h.end(), std::greater<>{});
    size_t r = 0;
    for (size_t i = first; i < n; ++i) {
        if (h.front().cost < cost[i]) {
            std::pop_heap(h.begin(), h.end(), std::greater<>{});
            h.back() = Cand{i, cost[i]};
            std::push_heap(h.begin(), h.end(), std::greater<>{});
            ++r;
        }
    }
    *replacements = r;
    return h;
}

// ---- the pipelined formulation -------------------------------------------------------------------------------------------------------
struct Pass {                       // an off-chain SIFT pass in flight: hole at `pos`, value to place `v`
    size_t pos;
    Cand v;
    bool live;
};

struct Sim {
    std::vector<Cand> h;            // the heap array; chain nodes and chain siblings are AUTHORITATIVE in the controller's copies below
    size_t k;
    std::vector<size_t> c;          // chain: c[0] = 0 ... c[L] = k - 1
    std::vector<Cand> cv;           // controller copy of h[c[i]]
    std::vector<size_t> sib;        // sib[i] = sibling of c[i] (i >= 1) or SIZE_MAX
    std::vector<Cand> sv;           // controller copy of h[sib[i]]
    std::vector<int> sv_ready_tick; // tick from which sv[i] is valid again (a pass entered sib[i] and fills it one tick later)
    std::vector<Pass> stage;        // stage[l] = the pass whose hole is at level l (at most one per level per tick)
    long tick = 0, stalls = 0, issued = 0, max_in_flight = 0, chain_steps = 0;

    static int level_of(size_t p) { int l = 0; for (size_t q = p + 1; q > 1; q >>= 1) ++l; return l; }

    explicit Sim(std::vector<Cand> heap) : h(std::move(heap)), k(h.size()) {
        std::vector<size_t> up;
        for (size_t p = k - 1;; p = (p - 1) / 2) { up.push_back(p); if (p == 0) break; }
        c.assign(up.rbegin(), up.rend());
        const size_t L = c.size() - 1;
        cv.resize(L + 1); sib.assign(L + 1, SIZE_MAX); sv.resize(L + 1); sv_ready_tick.assign(L + 1, 0);
        for (size_t i = 0; i <= L; ++i) cv[i] = h[c[i]];
        for (size_t i = 1; i <= L; ++i) {
            const size_t s = (c[i] & 1) ? c[i] + 1 : c[i] - 1;     // odd index = left child: the sibling is the right child
            if (s < k) { sib[i] = s; sv[i] = h[s]; }
        }
        stage.assign(level_of(k - 1) + 2, Pass{0, Cand{}, false});
    }

    // one tick of the off-chain pipeline: every live pass does one step; all read the state of the tick's start
    void advance() {
        ++tick;
        std::vector<Pass> next(stage.size(), Pass{0, Cand{}, false});
        std::vector<std::pair<size_t, Cand>> writes;
        std::vector<size_t> touched;                               // nodes read or written this tick, for the hazard assertion
        for (size_t l = 0; l < stage.size(); ++l) {
            if (!stage[l].live) continue;
            const Pass& p = stage[l];
            const size_t left = 2 * p.pos + 1, right = left + 1, range = k - 1;        // SIFT works on [0, k - 2]
            touched.push_back(p.pos);
            size_t m = SIZE_MAX;
            if (right < range) { m = (h[right].cost > h[left].cost) ? left : right; touched.push_back(left); touched.push_back(right); }
            else if (left < range) { m = left; touched.push_back(left); }
            if (m != SIZE_MAX && !(h[m].cost > p.v.cost)) {       // the smaller child is <= v: it moves up, the hole moves down
                writes.push_back({p.pos, h[m]});
                assert(l + 1 < next.size() && !next[l + 1].live);
                next[l + 1] = Pass{m, p.v, true};
            } else {
                writes.push_back({p.pos, p.v});
            }
        }
        std::sort(touched.begin(), touched.end());
        // hazard check: a node may be touched by one pass only (its own hole, or as a child it reads)
        for (size_t i = 1; i < touched.size(); ++i) assert(touched[i] != touched[i - 1] && "two passes touched one node in the same tick");
        for (auto& w : writes) {
            h[w.first] = w.second;
            for (size_t i = 1; i < sib.size(); ++i)               // the pass that entered a chain sibling reports its new value
                if (sib[i] == w.first) sv[i] = w.second;
        }
        stage.swap(next);
        long live = 0;
        for (auto& p : stage) live += p.live;
        max_in_flight = std::max(max_in_flight, live);
    }

    bool pipeline_empty() const { for (auto& p : stage) if (p.live) return false; return true; }

    // one replacement: SIFT(v = last element) then PUSH(x), all chain work in the controller's copies
    void replace(const Cand& x) {
        const size_t L = c.size() - 1;
        const Cand v = cv[L];
        size_t d = 0;
        for (;;) {
            // children of c[d] inside the SIFT range [0, k - 2]: the chain child c[d + 1] (never the last slot itself) and sib[d + 1]
            const bool has_chain = d + 1 < L;
            const bool has_sib = d + 1 <= L && sib[d + 1] != SIZE_MAX && sib[d + 1] < k - 1;
            if (has_sib) while (tick < sv_ready_tick[d + 1]) { advance(); ++stalls; }   // the sibling is being filled by an earlier pass
            int pick = 0;                                          // 0 none, 1 chain child, 2 sibling
            if (has_chain && has_sib) {
                const bool chain_is_right = (c[d + 1] & 1) == 0;
                const Cand& right = chain_is_right ? cv[d + 1] : sv[d + 1];
                const Cand& left = chain_is_right ? sv[d + 1] : cv[d + 1];
                const bool take_left = right.cost > left.cost;
                pick = (take_left == chain_is_right) ? 2 : 1;
            } else if (has_chain) pick = 1;
            else if (has_sib) pick = 2;
            if (pick == 1 && !(cv[d + 1].cost > v.cost)) { cv[d] = cv[d + 1]; ++d; ++chain_steps; continue; }
            if (pick == 2 && !(sv[d + 1].cost > v.cost)) {
                cv[d] = sv[d + 1];
                // the hole enters the sibling's subtree: an off-chain pass starts there; nobody may be at that level already
                const int lvl = level_of(sib[d + 1]);
                while (stage[lvl].live || stage[lvl + 1].live) { advance(); ++stalls; }   // one pass per level, and lag two behind the pass ahead
                stage[lvl] = Pass{sib[d + 1], v, true};
                sv_ready_tick[d + 1] = static_cast<int>(tick) + 1;
                ++issued;
                break;
            }
            cv[d] = v;
            break;
        }
        // PUSH: x down the chain (controller registers only); the carry-out is the new last element = the next v
        Cand carry = x;
        bool swapped = false;
        for (size_t i = 0; i < L; ++i)
            if (swapped || cv[i].cost > carry.cost) { std::swap(cv[i], carry); swapped = true; }
        cv[L] = carry;
        advance();                                                 // the controller spends a tick per replacement
    }

    std::vector<Cand> finish() {
        while (!pipeline_empty()) advance();
        for (size_t i = 0; i < c.size(); ++i) h[c[i]] = cv[i];
        for (size_t i = 1; i < sib.size(); ++i) if (sib[i] != SIZE_MAX) assert(h[sib[i]].cost == sv[i].cost && h[sib[i]].id == sv[i].id);
        return h;
    }
};

static bool run_case(const std::vector<float>& cost, size_t k, bool verbose) {
    size_t r = 0;
    std::vector<Cand> want = reference(cost, k, &r);
    // same start as the reference: make_heap of the first k (level-parallel on the device already)
    std::vector<Cand> h;
    const size_t n = cost.size(), first = std::min(n, k);
    for (size_t i = 0; i < first; ++i) h.push_back(Cand{i, cost[i]});
    std::make_heap(h.begin(), h.end(), std::greater<>{});
    if (h.size() < 2) return true;
    Sim sim(h);
    for (size_t i = first; i < n; ++i)
        if (sim.cv[0].cost < cost[i]) sim.replace(Cand{i, cost[i]});
    std::vector<Cand> got = sim.finish();
    bool ok = got.size() == want.size();
    for (size_t i = 0; ok && i < got.size(); ++i) ok = got[i].id == want[i].id && got[i].cost == want[i].cost;
    if (verbose || !ok)
        std::printf("%s n=%zu k=%zu replacements=%zu ticks=%ld (%.2f per replacement) stalls=%ld off-chain passes=%ld on-chain steps=%ld max in flight=%ld\n",
                    ok ? "ok  " : "FAIL", n, k, r, sim.tick, r ? double(sim.tick) / r : 0.0, sim.stalls, sim.issued, sim.chain_steps, sim.max_in_flight);
    return ok;
}

// `--file costs.f32 ratio`: the node costs of a real tree in node order (float32, node 0 = the root is skipped like the reference does,
// reinsertion_optimizer.h:93), k = max(1, ratio * node count) as in ReinsertionOptimizer::optimize (:233-234)
static int run_file(const char* path, double ratio) {
    FILE* f = std::fopen(path, "rb");
    if (!f) { std::perror(path); return 2; }
    std::vector<float> all;
    float buf[4096];
    size_t got;
    while ((got = std::fread(buf, 4, 4096, f)) > 0) all.insert(all.end(), buf, buf + got);
    std::fclose(f);
    const size_t nodes = all.size();
    const size_t k = std::max<size_t>(1, static_cast<size_t>(static_cast<float>(nodes) * static_cast<float>(ratio)));
    std::vector<float> cost(all.begin() + 1, all.end());
    return run_case(cost, k, true) ? 0 : 1;
}

int main(int argc, char** argv) {
    if (argc > 2 && std::string(argv[1]) == "--file") return run_file(argv[2], argc > 3 ? std::atof(argv[3]) : 0.05);
    const int seeds = argc > 1 ? std::atoi(argv[1]) : 2000;
    const size_t max_k = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 3000;
    int bad = 0;
    for (int seed = 0; seed < seeds; ++seed) {
        std::mt19937_64 rng(seed);
        const size_t k = 2 + rng() % max_k;
        const size_t n = k + rng() % (20 * k + 1);
        std::vector<float> cost(n);
        const int kind = seed % 5;
        for (size_t i = 0; i < n; ++i) {
            switch (kind) {
            case 0: cost[i] = float(rng() % 1000003) / 1000003.0f; break;                       // no ties to speak of
            case 1: cost[i] = float(rng() % 7); break;                                          // ties everywhere
            case 2: cost[i] = float(rng() % 64) + (rng() % 3 == 0 ? 0.5f : 0.0f); break;        // many ties
            case 3: cost[i] = float(i % 97) * 0.25f + float(rng() % 2); break;                  // structured + ties
            default: cost[i] = 1.0f / float(1 + (i % 1000)) + float(rng() % 3) * 1e-3f; break;  // decreasing runs (tree order)
            }
        }
        if (!run_case(cost, k, seed < 5)) ++bad;
    }
    std::printf("%d seeds, %d failures\n", seeds, bad);
    return bad != 0;
}
