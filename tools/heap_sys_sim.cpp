// Developer tool (CPU): lane-by-lane model of the systolic candidate-heap kernel (bvh_amd/csrc/reinsert.hip: k_heap_select_sys),
// checked against libstdc++ itself. NOT product code. The kernel is a transcription of this model: every array of 64 below is a
// VGPR (one value per lane), every loop over lanes is one wave-wide instruction sequence, `lds` / `glob` are the two memories.
//
// One replacement of reinsertion_optimizer.h:96-103 = pop_heap + back() = x + push_heap is executed as
//   controller (lanes = levels of the ancestor chain of the last slot; chain values cv[] and the values sv[] of the chain nodes'
//               siblings live in registers for the whole loop, their memory copies are stale):
//      v = cv[L]; the part of the pop that runs along the chain is decided for all chain levels at once (ballot), the chain
//      shifts up, and where the hole leaves the chain into a sibling's subtree an off-chain PASS (hole position, v) starts;
//      then x is inserted into the chain (push_heap top-down: the carry-out becomes the new last element);
//   pipeline   (lanes = heap levels; at most one pass per level, a new pass only enters behind two free levels): every tick each
//               pass looks at the two children of its hole, moves the smaller one up if it is <= v (ties: the right child) and
//               follows it, else drops v; the first write of a pass is the new value of the sibling it started at (-> sv[]);
//   deferred   a pass that reaches the last level held in LDS is parked as a task (hole, v) and its LDS entry marked as an open
//               hole; tasks in different subtrees are independent and are finished 64 at a time by one lane each; a pass (or the
//               controller) that is about to read an open hole / a stale sibling resolves the tasks first.
//
//   g++ -std=c++20 -O2 tools/heap_sys_sim.cpp -o /tmp/heap_sys_sim && /tmp/heap_sys_sim [seeds] [max_k]
#include <algorithm>
#include <cassert>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <random>
#include <vector>

struct Ent { float cost = 0; uint32_t id = 0; };
struct Cand { size_t id = 0; float cost = 0; bool operator>(const Cand& o) const { return cost > o.cost; } };
constexpr uint32_t kOpenHole = 0xffffffffu;

static std::vector<Cand> reference(const std::vector<float>& cost, size_t k, size_t* replacements) {
    std::vector<Cand> h;
    const size_t n = cost.size(), first = std::min(n, k);
    for (size_t i = 0; i < first; ++i) h.push_back(Cand{i, cost[i]});
    std::make_heap(h.begin(), h.end(), std::greater<>{});
    size_t r = 0;
    for (size_t i = first; i < n; ++i)
        if (h.front().cost < cost[i]) {
            std::pop_heap(h.begin(), h.end(), std::greater<>{});
            h.back() = Cand{i, cost[i]};
            std::push_heap(h.begin(), h.end(), std::greater<>{});
            ++r;
        }
    *replacements = r;
    return h;
}

struct Model {
    std::vector<Ent> mem;                   // positions [0, cap) = LDS, the rest = HBM
    uint32_t k = 0, cap = 0; int cap_level = 0, L = 0;
    // chain lanes
    Ent cv[64], sv[64];
    uint32_t cpos[64] = {}, spos[64] = {};
    bool has_sib[64] = {}, cir[64] = {};
    // pipeline lanes
    bool live[64] = {}, first[64] = {};
    uint32_t pos[64] = {};
    Ent pv[64];
    // deferred tasks (lane i keeps task i)
    uint32_t n_tasks = 0, task_pos[64] = {};
    Ent task_val[64];
    uint64_t stale = 0;
    long ticks = 0, stalls = 0, resolves = 0, replacements = 0, tasks_made = 0;

    static int level_of(uint32_t p) { int l = 0; for (uint32_t q = p + 1; q > 1; q >>= 1) ++l; return l; }

    Model(const std::vector<Ent>& heap, int lds_levels) : mem(heap), k(static_cast<uint32_t>(heap.size())) {
        cap = (1u << lds_levels) - 1;                           // levels 0 .. lds_levels - 1 in LDS
        cap_level = lds_levels - 1;
        L = level_of(k - 1);
        assert(L <= 62);
        for (int i = 0; i <= L; ++i) {
            cpos[i] = (k >> (L - i)) - 1;
            cv[i] = mem[cpos[i]];
            if (i >= 1) {
                cir[i] = (cpos[i] & 1u) == 0;
                spos[i] = cir[i] ? cpos[i] - 1 : cpos[i] + 1;
                has_sib[i] = spos[i] < k - 1;
                if (has_sib[i]) sv[i] = mem[spos[i]];
            }
        }
    }

    // literal __adjust_heap by one lane (stl_heap.h:223-248) on [0, len): the deferred part of a pass
    void lane_adjust(uint32_t hole, uint32_t len, Ent value) {
        const uint32_t top = hole;
        uint32_t child = hole;
        while (child < (len - 1) / 2) {
            child = 2 * (child + 1);
            if (mem[child].cost > mem[child - 1].cost) --child;
            mem[hole] = mem[child];
            hole = child;
        }
        if ((len & 1u) == 0 && child == (len - 2) / 2) {
            child = 2 * (child + 1);
            mem[hole] = mem[child - 1];
            hole = child - 1;
        }
        while (hole > top) {
            const uint32_t parent = (hole - 1) / 2;
            if (!(mem[parent].cost > value.cost)) break;
            mem[hole] = mem[parent];
            hole = parent;
        }
        mem[hole] = value;
    }

    void resolve_tasks() {
        if (n_tasks) {
            ++resolves;
            for (uint32_t t = 0; t < n_tasks; ++t) lane_adjust(task_pos[t], k - 1, task_val[t]);
            n_tasks = 0;
        }
        for (int i = 1; i <= L; ++i) if ((stale >> i) & 1u) sv[i] = mem[spos[i]];
        stale = 0;
    }

    void add_task(uint32_t p, Ent v) {
        task_pos[n_tasks] = p; task_val[n_tasks] = v;
        if (p < cap) mem[p].id = kOpenHole;
        ++tasks_made;
        if (++n_tasks == 64) resolve_tasks();
    }

    void tick() {
        ++ticks;
        const uint32_t range = k - 1;
        // (open holes can only sit on the last LDS level: the lane above it checks before it reads)
        bool owed = false;
        for (int l = 0; l < 64; ++l) {
            if (!live[l] || l != cap_level - 1) continue;
            const uint32_t left = 2 * pos[l] + 1, right = left + 1;
            if ((left < range && mem[left].id == kOpenHole) || (right < range && mem[right].id == kOpenHole)) owed = true;
        }
        if (owed) resolve_tasks();
        bool nlive[64] = {}; uint32_t npos[64] = {}; Ent nv[64];
        for (int l = 0; l + 1 < 64; ++l)                           // hazard check: nobody reads a node another pass is filling
            if (live[l] && live[l + 1]) assert((pos[l + 1] - 1) / 2 != pos[l]);
        for (int l = 0; l < 64; ++l) {
            if (!live[l]) continue;
            const uint32_t left = 2 * pos[l] + 1, right = left + 1;
            assert(!(left < range) || right < cap + 0u || (right >= range && left < cap));   // a ticking pass reads children held in LDS
            bool have = false; uint32_t m = 0;
            if (right < range) { have = true; m = (mem[right].cost > mem[left].cost) ? left : right; }
            else if (left < range) { have = true; m = left; }
            assert(!have || mem[m].id != kOpenHole);
            const bool go = have && !(mem[m].cost > pv[l].cost);
            const Ent wr = go ? mem[m] : pv[l];
            mem[pos[l]] = wr;
            if (first[l]) { sv[l] = wr; first[l] = false; }
            if (go) { assert(l + 1 < 64 && !nlive[l + 1]); nlive[l + 1] = true; npos[l + 1] = m; nv[l + 1] = pv[l]; }
        }
        for (int l = 0; l < 64; ++l) { live[l] = nlive[l]; pos[l] = npos[l]; pv[l] = nv[l]; first[l] = false; }
        // a pass that arrives at a level whose children lie outside LDS is parked
        for (int l = 0; l < 64; ++l)
            if (live[l] && 2 * pos[l] + 1 >= cap && 2 * pos[l] + 1 < range) { live[l] = false; add_task(pos[l], pv[l]); }
    }

    void replace(Ent x) {
        ++replacements;
        const Ent v = cv[L];
        int dstar = 0; bool enters = false;
        for (;;) {
            bool cont[64] = {}, ent[64] = {};
            for (int d = 0; d < L; ++d) {                          // lane d looks at level d + 1 (values of the next lane)
                const bool hc = d + 1 < L, hs = has_sib[d + 1];
                int pick = 0;                                      // 1 chain child, 2 sibling
                if (hc && hs) {
                    const Ent& right = cir[d + 1] ? cv[d + 1] : sv[d + 1];
                    const Ent& left = cir[d + 1] ? sv[d + 1] : cv[d + 1];
                    const bool take_left = right.cost > left.cost;
                    pick = (take_left == cir[d + 1]) ? 2 : 1;
                } else if (hc) pick = 1;
                else if (hs) pick = 2;
                cont[d] = pick == 1 && !(cv[d + 1].cost > v.cost);
                ent[d] = pick == 2 && !(sv[d + 1].cost > v.cost);
            }
            dstar = 0;
            while (cont[dstar]) ++dstar;
            enters = ent[dstar];
            uint64_t used = 0;                                     // sibling values the decision looked at: levels 1 .. dstar + 1
            for (int d = 0; d <= dstar; ++d) used |= uint64_t{1} << (d + 1);
            if (stale & used) { resolve_tasks(); continue; }
            break;
        }
        Ent ncv[64];
        for (int d = 0; d <= L; ++d) ncv[d] = cv[d];
        for (int d = 0; d < dstar; ++d) ncv[d] = cv[d + 1];
        ncv[dstar] = enters ? sv[dstar + 1] : v;
        for (int d = 0; d <= L; ++d) cv[d] = ncv[d];
        if (enters) {
            const int lvl = dstar + 1;
            const uint32_t p = spos[lvl];
            const bool kids_in_lds = 2 * p + 1 < cap || 2 * p + 1 >= k - 1;     // (no children in range: the pass ends at once)
            if (kids_in_lds && p < cap) {
                // one pass per level; the pass one level further down must not be filling one of the new hole's children
                // (a conflict that is absent now cannot appear later: both passes move one level per tick)
                while (live[lvl] || (live[lvl + 1] && (pos[lvl + 1] - 1) / 2 == p)) { tick(); ++stalls; }
                live[lvl] = true; first[lvl] = true; pos[lvl] = p; pv[lvl] = v;
            } else {
                add_task(p, v);
                stale |= uint64_t{1} << lvl;
                if (n_tasks == 0) { /* add_task resolved: sv reloaded */ }
            }
        }
        // push_heap(x) top-down along the chain
        int i0 = L;
        for (int i = 0; i < L; ++i) if (cv[i].cost > x.cost) { i0 = i; break; }
        for (int d = 0; d <= L; ++d) ncv[d] = cv[d];
        for (int d = i0 + 1; d <= L; ++d) ncv[d] = cv[d - 1];
        ncv[i0] = x;
        for (int d = 0; d <= L; ++d) cv[d] = ncv[d];
        tick();
    }

    std::vector<Ent> finish() {
        for (;;) { bool any = false; for (int l = 0; l < 64; ++l) any |= live[l]; if (!any) break; tick(); }
        resolve_tasks();
        for (int i = 0; i <= L; ++i) mem[cpos[i]] = cv[i];
        for (int i = 1; i <= L; ++i) if (has_sib[i]) assert(mem[spos[i]].cost == sv[i].cost && mem[spos[i]].id == sv[i].id);
        return mem;
    }
};

static bool run_case(const std::vector<float>& cost, size_t k, int lds_levels, bool verbose) {
    size_t r = 0;
    std::vector<Cand> want = reference(cost, k, &r);
    std::vector<Cand> h;
    const size_t n = cost.size(), first = std::min(n, k);
    for (size_t i = 0; i < first; ++i) h.push_back(Cand{i, cost[i]});
    std::make_heap(h.begin(), h.end(), std::greater<>{});
    if (h.size() < 2) return true;
    std::vector<Ent> start(h.size());
    for (size_t i = 0; i < h.size(); ++i) start[i] = Ent{h[i].cost, static_cast<uint32_t>(h[i].id)};
    Model m(start, lds_levels);
    for (size_t i = first; i < n; ++i)
        if (m.cv[0].cost < cost[i]) m.replace(Ent{cost[i], static_cast<uint32_t>(i)});
    std::vector<Ent> got = m.finish();
    bool ok = got.size() == want.size();
    for (size_t i = 0; ok && i < got.size(); ++i) ok = got[i].id == want[i].id && got[i].cost == want[i].cost;
    if (verbose || !ok)
        std::printf("%s n=%zu k=%zu lds_levels=%d replacements=%zu ticks=%ld (%.2f per replacement) stalls=%ld tasks=%ld resolves=%ld\n", ok ? "ok  " : "FAIL",
                    n, k, lds_levels, r, m.ticks, r ? double(m.ticks) / r : 0.0, m.stalls, m.tasks_made, m.resolves);
    return ok;
}

int main(int argc, char** argv) {
    const int seeds = argc > 1 ? std::atoi(argv[1]) : 3000;
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
            case 0: cost[i] = float(rng() % 1000003) / 1000003.0f; break;
            case 1: cost[i] = float(rng() % 7); break;
            case 2: cost[i] = float(rng() % 64) + (rng() % 3 == 0 ? 0.5f : 0.0f); break;
            case 3: cost[i] = float(i % 97) * 0.25f + float(rng() % 2); break;
            default: cost[i] = 1.0f / float(1 + (i % 1000)) + float(rng() % 3) * 1e-3f; break;
            }
        }
        const int lds_levels = 1 + static_cast<int>(rng() % 13);   // small LDS parts drive the deferred-task and stale-sibling paths hard
        if (!run_case(cost, k, lds_levels, seed < 6)) ++bad;
    }
    std::printf("%d seeds, %d failures\n", seeds, bad);
    return bad != 0;
}
