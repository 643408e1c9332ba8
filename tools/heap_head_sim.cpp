// Developer tool (CPU): lane-by-lane model of the round-6 candidate-heap replay (bvh_amd/csrc/heap_head.inc: k_heap_select_head),
// checked against libstdc++ itself. NOT product code. The kernel is a transcription of this model: every array of 64 below is a
// VGPR (one value per lane), every loop over lanes is one wave-wide instruction sequence, `mem` is the heap array (positions
// [0, cap) in LDS, the rest in HBM), the three actors are three wavefronts of one workgroup and the scheduler below interleaves
// them at random, so that every ordering of "token sent / first step published / hole closed" the hardware can produce is met.
//
// One replacement of reinsertion_optimizer.h:96-103 = pop_heap + back() = x + push_heap:
//   HEAD (one wave, everything the NEXT replacement depends on, in registers): one lane per PARENT position of the "head tree" =
//        heap levels 0 .. HL-1 complete + the ancestors ("spine") of the last position k-1 below them. A lane keeps BOTH children of its
//        parent {cost, id}, so the child choice of __adjust_heap (stl_heap.h:223-248) is lane-local; the min-child path is then ONE ballot
//        of the choices tested against per-lane ancestor masks, the landing level of the popped value one more ballot, and every path lane
//        takes its new child value from its chosen child's lane (a value that was fetched before the replacement began). The push
//        (stl_heap.h:134-148) is a sorted insert into the spine: one ballot and one lane shift. Children that are not head parents
//        are roots of subtrees the head never looks into:
//   TAIL (one wave, lanes = pops in flight below the head): a pop that leaves the head tree becomes a token (sub-root e, value v). The tail
//        sifts it down one heap level per iteration inside LDS; the FIRST step decides the new value of position e, which it publishes to
//        the head (until then the head's copy of e is "pending" and a pop whose path needs it waits). A token that reaches the last LDS
//        level is parked as a
//   DEEP task (one wave, lane per task, the libstdc++ loop literally on HBM): the LDS entry is marked as an open hole until done.
//
//   g++ -std=c++20 -O2 tools/heap_head_sim.cpp -o /tmp/heap_head_sim && /tmp/heap_head_sim [seeds] [max_k]
#include <algorithm>
#include <cassert>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <deque>
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

static int level_of(uint32_t p) { int l = 0; for (uint32_t q = p + 1; q > 1; q >>= 1) ++l; return l; }
static int ffs64(uint64_t m) { return m ? __builtin_ctzll(m) : 64; }

enum Kind : int { kNone = 0, kHead = 1, kTail = 2, kBottom = 3 };

struct Token { uint32_t pos, slot; Ent v; };
struct DeepTask { uint32_t pos; int slot; Ent v; };           // slot < 0: nothing to publish

struct Model {
    std::vector<Ent> mem;
    uint32_t k = 0, len = 0, cap = 0;
    int HL = 5, LL = 14, D = 0;
    // ---- head: lane state -----------------------------------------------------------------------------------------------------
    int n_lanes = 0;
    uint32_t pos[64] = {};
    Ent Lv[64], Rv[64];
    int kindL[64] = {}, kindR[64] = {}, clL[64] = {}, clR[64] = {};
    uint64_t ancmask[64] = {}, ancval[64] = {};
    bool ischain[64] = {}; int side[64] = {};                 // chain lanes 0 .. D-1: which child continues the spine (0 = left)
    int pend[64] = {};                                        // bit 0: left child owed by the tail, bit 1: right child
    // derived (recomputed after every replacement)
    bool b[64] = {}, ex[64] = {}; Ent C[64], NC[64]; int nl[64] = {}; uint64_t B = 0;
    Ent root;
    // ---- shared "LDS" ------------------------------------------------------------------------------------------------------------
    std::deque<Token> ring; size_t ring_cap = 8;
    std::deque<DeepTask> deep_ring; size_t deep_cap = 16;
    Ent slot_val[128]; bool slot_flag[128] = {};
    // ---- tail lanes ---------------------------------------------------------------------------------------------------------------
    bool t_live[64] = {}; uint32_t t_pos[64] = {}, t_root[64] = {}; int t_steps[64] = {}, t_slot[64] = {}; Ent t_v[64];
    long head_stalls = 0, tail_stalls = 0, tokens = 0, deep_tasks = 0, replacements = 0, forwarded = 0;

    Model(const std::vector<Ent>& heap, int head_levels, int lds_levels) : mem(heap), k(uint32_t(heap.size())), HL(head_levels), LL(lds_levels) {
        len = k - 1;
        cap = (1u << LL) - 1;
        assert(len >= cap && HL >= 1 && HL < LL);
        D = level_of(k - 1);
        assert(D >= LL);
        auto spine = [&](int j) { return (k >> (D - j)) - 1; };
        // lane assignment: spine parents q_0 .. q_{D-1} -> lanes 0 .. D-1; the other parents of levels 0 .. HL-1 behind them in BFS order
        std::vector<uint32_t> lane_pos;
        for (int j = 0; j < D; ++j) lane_pos.push_back(spine(j));
        for (uint32_t p = 0; p < (1u << HL) - 1; ++p) if (spine(level_of(p)) != p) lane_pos.push_back(p);
        n_lanes = int(lane_pos.size());
        assert(n_lanes <= 64);
        auto lane_of = [&](uint32_t p) { for (int i = 0; i < n_lanes; ++i) if (lane_pos[i] == p) return i; return -1; };
        for (int i = 0; i < n_lanes; ++i) {
            pos[i] = lane_pos[i];
            ischain[i] = i < D;
            const uint32_t l = 2 * pos[i] + 1, r = l + 1;
            auto kind = [&](uint32_t c) { return c == k - 1 ? kBottom : c >= len ? kNone : lane_of(c) >= 0 ? kHead : kTail; };
            kindL[i] = kind(l); kindR[i] = kind(r);
            clL[i] = kindL[i] == kHead ? lane_of(l) : 64; clR[i] = kindR[i] == kHead ? lane_of(r) : 64;
            if (kindL[i] != kNone) Lv[i] = mem[l];
            if (kindR[i] != kNone) Rv[i] = mem[r];
            if (ischain[i]) side[i] = spine(i + 1) == r ? 1 : 0;
            // ancestors: walk up from pos[i]
            for (uint32_t c = pos[i]; c != 0;) {
                const uint32_t par = (c - 1) / 2; const int pl = lane_of(par);
                assert(pl >= 0 && pl < i);                   // lane order = depth order along every path
                ancmask[i] |= uint64_t{1} << pl;
                if (c == 2 * par + 2) ancval[i] |= uint64_t{1} << pl;
                c = par;
            }
        }
        root = mem[0];
        recompute();
    }

    void recompute() {
        B = 0;
        for (int i = 0; i < n_lanes; ++i) {
            const bool exL = kindL[i] == kHead || kindL[i] == kTail, exR = kindR[i] == kHead || kindR[i] == kTail;
            b[i] = exR && !(Rv[i].cost > Lv[i].cost);         // comp(second, second - 1): take the left child iff right > left
            C[i] = b[i] ? Rv[i] : Lv[i];
            ex[i] = b[i] ? exR : exL;
            const int ck = b[i] ? kindR[i] : kindL[i];
            nl[i] = ck == kHead ? (b[i] ? clR[i] : clL[i]) : 64;
            if (b[i]) B |= uint64_t{1} << i;
        }
        for (int i = 0; i < n_lanes; ++i) NC[i] = nl[i] < 64 ? C[nl[i]] : Ent{};      // ds_bpermute
    }
    void fold() {                                             // take what the tail has published
        for (int i = 0; i < n_lanes; ++i)
            for (int s = 0; s < 2; ++s)
                if ((pend[i] >> s & 1) && slot_flag[2 * i + s]) {
                    (s ? Rv[i] : Lv[i]) = slot_val[2 * i + s];
                    slot_flag[2 * i + s] = false;
                    pend[i] &= ~(1 << s);
                }
    }

    // one replacement; returns false when it has to wait (for a publish or for ring space) - nothing was changed then
    bool head_replace(Ent x) {
        fold(); recompute();                                  // (the kernel folds after the push of the previous replacement)
        const Ent v = ischain[D - 1] ? (side[D - 1] ? Rv[D - 1] : Lv[D - 1]) : Ent{};
        uint64_t P = 0, PG = 0, Pbad = 0;
        for (int i = 0; i < n_lanes; ++i) {
            const bool on = ((B ^ ancval[i]) & ancmask[i]) == 0;
            const bool moves = ex[i] && !(C[i].cost > v.cost);
            if (on) P |= uint64_t{1} << i;
            if (on && moves) PG |= uint64_t{1} << i;
            if (on && pend[i]) Pbad |= uint64_t{1} << i;
        }
        const int pstar = ffs64(P & ~PG), fb = ffs64(Pbad);
        if (fb != 64 && fb <= pstar) { ++head_stalls; return false; }
        const bool exits = pstar == 64;
        if (exits && ring.size() >= ring_cap) { ++head_stalls; return false; }
        const uint64_t M = exits ? PG : PG & ((uint64_t{1} << pstar) - 1);
        const Ent C0 = C[0];
        int z = -1;
        for (int i = 0; i < n_lanes; ++i) {
            if (!(M >> i & 1)) continue;
            const int ck = b[i] ? kindR[i] : kindL[i];
            if (ck == kHead) (b[i] ? Rv[i] : Lv[i]) = nl[i] < pstar ? NC[i] : v;
            else { assert(ck == kTail && exits); assert(z < 0); z = i; }
        }
        root = (M & 1) ? C0 : v;
        if (exits) {
            assert(z >= 0 && z == 63 - __builtin_clzll(P));
            Token t; t.pos = 2 * pos[z] + 1 + (b[z] ? 1 : 0); t.slot = uint32_t(2 * z + (b[z] ? 1 : 0)); t.v = v;
            ring.push_back(t); ++tokens;
            pend[z] |= 1 << (b[z] ? 1 : 0);
        }
        // push x: sorted insert into the spine (levels 0 .. D; lane i keeps level i + 1, the root is level 0)
        Ent cv[64], up[64];
        for (int i = 0; i < D; ++i) cv[i] = side[i] ? Rv[i] : Lv[i];
        for (int i = 0; i < D; ++i) up[i] = i == 0 ? root : cv[i - 1];                 // DPP wave_shr:1, lane 0 keeps `old` = root
        uint64_t W = root.cost > x.cost ? 1 : 0;
        for (int i = 0; i + 1 < D; ++i) if (cv[i].cost > x.cost) W |= uint64_t{2} << i;      // bit j = level j moves down
        const uint64_t Z = ~W & ((uint64_t{1} << D) - 1);
        const int J = Z ? 64 - __builtin_clzll(Z) : 0;        // landing level: below the deepest ancestor that is not greater
        for (int i = 0; i < D; ++i)
            if (i + 1 >= J) (side[i] ? Rv[i] : Lv[i]) = i + 1 == J ? x : up[i];
        if (J == 0) root = x;
        ++replacements;
        return true;
    }

    // ---- tail: one iteration of the wave ---------------------------------------------------------------------------------------
    void tail_iteration(std::mt19937& rng) {
        // pick up new tokens into free lanes (in ring order; a token that follows another one into the same sub-heap keeps two levels behind)
        int picked = 0;
        while (!ring.empty() && picked < 2) {
            const Token t = ring.front();
            bool clash = false; int free_lane = -1;
            for (int i = 0; i < 64; ++i) {
                if (t_live[i] && t_root[i] == t.pos && t_steps[i] < 2) clash = true;
                if (!t_live[i] && free_lane < 0) free_lane = i;
            }
            if (clash || free_lane < 0) break;
            if (level_of(t.pos) >= LL - 1) {                  // the sub-root has no children inside LDS: straight to the deep wave
                if (deep_ring.size() >= deep_cap) break;
                deep_ring.push_back(DeepTask{t.pos, int(t.slot), t.v}); ++forwarded;
                ring.pop_front(); ++picked;
                continue;
            }
            ring.pop_front(); ++picked;
            t_live[free_lane] = true; t_pos[free_lane] = t.pos; t_root[free_lane] = t.pos; t_steps[free_lane] = 0; t_slot[free_lane] = int(t.slot); t_v[free_lane] = t.v;
        }
        (void)rng;
        // an open hole among the children any lane is about to read stalls the whole wave (a later token could otherwise overtake)
        for (int i = 0; i < 64; ++i)
            if (t_live[i] && (mem[2 * t_pos[i] + 1].id == kOpenHole || mem[2 * t_pos[i] + 2].id == kOpenHole)) { ++tail_stalls; return; }
        // lanes about to park need room in the deep ring: count them first (reads happen before writes in lockstep)
        Ent rl[64], rr[64];
        for (int i = 0; i < 64; ++i) if (t_live[i]) { rl[i] = mem[2 * t_pos[i] + 1]; rr[i] = mem[2 * t_pos[i] + 2]; }
        size_t parks = 0;
        for (int i = 0; i < 64; ++i)
            if (t_live[i]) {
                const bool right = !(rr[i].cost > rl[i].cost);
                const Ent c = right ? rr[i] : rl[i];
                if (!(c.cost > t_v[i].cost) && level_of(2 * t_pos[i] + 1) == LL - 1) ++parks;
            }
        if (deep_ring.size() + parks > deep_cap) { ++tail_stalls; return; }
        for (int i = 0; i < 64; ++i) {
            if (!t_live[i]) continue;
            const bool right = !(rr[i].cost > rl[i].cost);
            const Ent c = right ? rr[i] : rl[i];
            const uint32_t cpos = 2 * t_pos[i] + 1 + (right ? 1 : 0);
            Ent written;
            if (c.cost > t_v[i].cost) { written = t_v[i]; mem[t_pos[i]] = written; t_live[i] = false; }
            else {
                written = c; mem[t_pos[i]] = written;
                t_pos[i] = cpos;
                if (level_of(cpos) == LL - 1) {               // last LDS level: its children are in HBM
                    mem[cpos].id = kOpenHole;
                    deep_ring.push_back(DeepTask{cpos, -1, t_v[i]}); ++deep_tasks;
                    t_live[i] = false;
                }
            }
            if (t_steps[i] == 0) { slot_val[t_slot[i]] = written; slot_flag[t_slot[i]] = true; }
            ++t_steps[i];
        }
    }

    // ---- deep: a batch of tasks, one lane each, the libstdc++ loop literally ----------------------------------------------------
    void deep_batch() {
        const size_t n = std::min<size_t>(deep_ring.size(), 64);
        for (size_t t = 0; t < n; ++t) {
            const DeepTask task = deep_ring[t];
            uint32_t hole = task.pos; const uint32_t top = hole; uint32_t child = hole;
            while (child < (len - 1) / 2) {
                child = 2 * (child + 1);
                if (mem[child].cost > mem[child - 1].cost) --child;
                mem[hole] = mem[child]; hole = child;
            }
            if ((len & 1u) == 0 && child == (len - 2) / 2) { child = 2 * (child + 1); mem[hole] = mem[child - 1]; hole = child - 1; }
            while (hole > top) {
                const uint32_t parent = (hole - 1) / 2;
                if (!(mem[parent].cost > task.v.cost)) break;
                mem[hole] = mem[parent]; hole = parent;
            }
            mem[hole] = task.v;
            if (task.slot >= 0) { slot_val[task.slot] = mem[task.pos]; slot_flag[task.slot] = true; }
        }
        deep_ring.erase(deep_ring.begin(), deep_ring.begin() + long(n));
    }

    bool idle() const {
        if (!ring.empty() || !deep_ring.empty()) return false;
        for (int i = 0; i < 64; ++i) if (t_live[i]) return false;
        return true;
    }
    void flush() {
        fold();
        for (int i = 0; i < n_lanes; ++i) assert(!pend[i]);
        mem[0] = root;
        for (int i = 0; i < n_lanes; ++i) {
            if (kindL[i] == kHead || kindL[i] == kBottom) mem[2 * pos[i] + 1] = Lv[i];
            if (kindR[i] == kHead || kindR[i] == kBottom) mem[2 * pos[i] + 2] = Rv[i];
        }
    }
};

static bool run_case(uint32_t seed, size_t n, size_t k, int HL, int LL, int distinct, int head_bias, bool verbose) {
    std::mt19937 rng(seed);
    std::vector<float> cost(n);
    for (auto& c : cost) c = distinct ? float(rng() % uint32_t(distinct)) : float(rng() >> 8) * 0x1p-24f;
    if (seed % 3 == 1) {                                      // a rising stream: many replacements
        for (size_t i = 0; i < n; ++i) cost[i] += float(i) * (distinct ? float(distinct) / float(n) : 1.0f / float(n));
    }
    size_t ref_repl = 0;
    const auto ref = reference(cost, k, &ref_repl);
    std::vector<Ent> heap(k);
    {
        std::vector<Cand> h;
        for (size_t i = 0; i < k; ++i) h.push_back(Cand{i, cost[i]});
        std::make_heap(h.begin(), h.end(), std::greater<>{});
        for (size_t i = 0; i < k; ++i) heap[i] = Ent{h[i].cost, uint32_t(h[i].id)};
    }
    Model m(heap, HL, LL);
    size_t i = k;
    long guard = 0;
    while (i < n || !m.idle()) {
        if (++guard > 400000000L) { std::printf("  livelock\n"); return false; }
        const unsigned pick = rng() % 16;
        if (pick < unsigned(head_bias)) {
            if (i < n) {
                if (!(m.root.cost < cost[i])) { ++i; continue; }
                if (m.head_replace(Ent{cost[i], uint32_t(i)})) ++i;
            }
        } else if (pick < unsigned(head_bias) + (16 - unsigned(head_bias)) * 2 / 3) m.tail_iteration(rng);
        else m.deep_batch();
    }
    m.flush();
    bool ok = size_t(m.replacements) == ref_repl;
    for (size_t j = 0; j < k && ok; ++j) ok = m.mem[j].id == uint32_t(ref[j].id) && m.mem[j].cost == ref[j].cost;
    if (!ok || verbose)
        std::printf("%s seed %u n %zu k %zu HL %d LL %d distinct %d: %ld replacements (ref %zu), %ld tokens (%ld forwarded), %ld deep, head stalls %ld, tail stalls %ld\n",
                    ok ? "ok  " : "FAIL", seed, n, k, HL, LL, distinct, m.replacements, ref_repl, m.tokens, m.forwarded, m.deep_tasks, m.head_stalls, m.tail_stalls);
    return ok;
}

int main(int argc, char** argv) {
    const int seeds = argc > 1 ? std::atoi(argv[1]) : 3000;
    const size_t max_k = argc > 2 ? size_t(std::atoll(argv[2])) : 6000;
    std::mt19937 rng(12345);
    int fails = 0;
    for (int s = 0; s < seeds; ++s) {
        const int HL = 1 + int(rng() % 5);                    // 1 .. 5 complete head levels
        const int LL = HL + 1 + int(rng() % 4);               // LDS levels
        const size_t kmin = (size_t{1} << LL) + 1;            // len = k - 1 >= cap
        if (kmin >= max_k) { --s; continue; }
        const size_t k = kmin + rng() % (max_k - kmin);
        const size_t n = k + 1 + rng() % (8 * k);
        const int distinct = (s % 4 == 0) ? 0 : (s % 4 == 1) ? 3 + int(rng() % 5) : (s % 4 == 2) ? 50 : 1000;
        const int head_bias = 2 + int(rng() % 12);
        if (!run_case(uint32_t(s), n, k, HL, LL, distinct, head_bias, s < 8)) ++fails;
    }
    std::printf("%d cases, %d failures\n", seeds, fails);
    return fails != 0;
}
