// Developer tool (CPU): lane-by-lane model of the round-6 candidate-heap replay (bvh_amd/csrc/heap_head.inc: k_heap_select_head),
// checked against libstdc++ itself. NOT product code. The kernel is a transcription of this model: every array of 64 below is a
// VGPR (one value per lane), every 64-bit mask a scalar register pair, every loop over lanes one wave-wide instruction sequence, `mem`
// the heap array (positions [0, cap) in LDS, the rest in HBM), the four actors are four wavefronts of one workgroup and the scheduler
// below interleaves them at random, so that every ordering of "token written / first step done / value written back / hole closed"
// the hardware can produce is met.
//
// One replacement of reinsertion_optimizer.h:96-103 = pop_heap + back() = x + push_heap:
//   HEAD (wave A, everything the NEXT replacement depends on, in registers): one lane per PARENT position of the "head tree" = a virtual
//        parent of the root (lane 0) + the ancestors ("spine") of the last position k-1 (lanes 1 .. D) + the other parents of heap levels
//        0 .. HL-1. A lane keeps BOTH children of its parent, so the child choice of __adjust_heap (stl_heap.h:223-248) is lane-local
//        (mask B); the min-child path is B tested against per-lane ancestor masks, the landing level of the popped value one more
//        compare plus bit arithmetic, and every path lane takes its new child from its chosen child's lane (a value fetched before the
//        replacement began). The push (stl_heap.h:134-148) is a sorted insert into the spine: one compare, one count of leading zeros, one
//        lane shift. Children that are not head parents are roots of sub-heaps the head never looks into; their values live in WORDS
//        the head re-reads every replacement. A pop that leaves the head tree writes a TOKEN {v | tag} into the word: the child is pending.
//   T1   polls the words; for a token it does the first step below the head (that fixes the sub-root's new value), writes the value back
//        into the word and hands the rest of the pop (hole, v) to
//   T2   lanes = pops in flight inside the LDS levels, one level per iteration. The position a hole stands at is marked (kOpenHole)
//        until its final entry is written; a pop that finds a marked child waits. A hole that reaches the last LDS level goes, marked, to
//   DEEP one lane per task, the same top-down sift on the HBM levels.
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
constexpr uint32_t kOpenHole = 0xffffffffu, kTokenTag = 0x80000000u, kHandedTag = 0x40000000u, kNoSlot = 0xffffffffu;

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

struct Tok { uint32_t pos, slot; Ent v; };

struct Model {
    std::vector<Ent> mem;
    uint32_t k = 0, len = 0, cap = 0, first_parked = 0;
    int HL = 5, LL = 14, D = 0;
    // ---- head: static lane state --------------------------------------------------------------------------------------------
    int n_lanes = 0;
    bool valid[64] = {}; uint32_t lpos[64] = {}, rpos[64] = {};
    int kind_l[64] = {}, kind_r[64] = {}, nl_l[64] = {}, nl_r[64] = {};
    uint64_t anc_mask[64] = {}, anc_val[64] = {};
    uint64_t VALID = 0, EXL = 0, EXR = 0, HLm = 0, HRm = 0, TL = 0, TR = 0, CHAIN = 0, CHAINCMP = 0, SIDE = 0;
    // ---- head: dynamic -----------------------------------------------------------------------------------------------------------
    float lc[64] = {}, rc[64] = {}; uint32_t li[64] = {}, ri[64] = {};
    uint64_t B = 0, EX = 0, CHH = 0, PEND = 0;
    float chc[64] = {}, nchc[64] = {}; uint32_t chi[64] = {}, nchi[64] = {}; int nl[64] = {};
    float vc = 0; uint32_t vi = 0; float root_c = 0;
    // ---- shared "LDS" ------------------------------------------------------------------------------------------------------------
    Ent words[128];
    std::deque<Tok> deep_ring; size_t deep_cap = 64;             // (each at least one entry per lane: a whole look / iteration must fit, like kTokCap, kDeepCap)
    long head_blocked = 0, tokens = 0, first_steps = 0, deep_tasks = 0, replacements = 0, forwarded = 0;

    uint32_t spine(int level) const { return (k >> (D - level)) - 1; }
    bool on_spine(uint32_t p) const { const int lv = level_of(p); return lv <= D && spine(lv) == p; }
    std::vector<uint32_t> lane_pos;                           // parent position per lane (lane 0: the virtual parent)
    int lane_of(uint32_t p) const { for (int i = 1; i < int(lane_pos.size()); ++i) if (lane_pos[i] == p && valid[i]) return i; return -1; }
    int kind_of(uint32_t c) const { return c == k - 1 ? 3 : c >= k - 1 ? 0 : lane_of(c) >= 0 ? 1 : 2; }

    Model(const std::vector<Ent>& heap, int head_levels, int lds_levels) : mem(heap), k(uint32_t(heap.size())), HL(head_levels), LL(lds_levels) {
        len = k - 1;
        cap = (1u << LL) - 1;
        first_parked = cap >> 1;
        assert(len >= cap && HL >= 1 && HL < LL);
        D = level_of(k - 1);
        assert(D >= LL);
        // lanes: 0 virtual, 1 .. D spine parents q_0 .. q_{D-1}, then the other parents of levels 0 .. HL-2, then those of level HL-1
        lane_pos.assign(1, 0xffffffffu);
        for (int j = 0; j < D; ++j) lane_pos.push_back(spine(j));
        for (uint32_t p = 0; p < (1u << (HL - 1)) - 1; ++p) if (!on_spine(p)) lane_pos.push_back(p);
        for (uint32_t p = (1u << (HL - 1)) - 1; p < (1u << HL) - 1; ++p) if (!on_spine(p)) lane_pos.push_back(p);
        n_lanes = int(lane_pos.size());
        assert(n_lanes <= 64);
        for (int i = 0; i < n_lanes; ++i) valid[i] = true;
        for (int i = 0; i < n_lanes; ++i) {
            const bool super = i == 0;
            lpos[i] = super ? 0 : 2 * lane_pos[i] + 1; rpos[i] = lpos[i] + 1;
            kind_l[i] = super ? 1 : kind_of(lpos[i]); kind_r[i] = super ? 0 : kind_of(rpos[i]);
            nl_l[i] = kind_l[i] == 1 ? (super ? 1 : lane_of(lpos[i])) : 64; nl_r[i] = kind_r[i] == 1 ? lane_of(rpos[i]) : 64;
            if (kind_l[i]) { lc[i] = mem[lpos[i]].cost; li[i] = mem[lpos[i]].id; }
            if (kind_r[i]) { rc[i] = mem[rpos[i]].cost; ri[i] = mem[rpos[i]].id; }
            if (kind_l[i] == 2) words[2 * i] = mem[lpos[i]];
            if (kind_r[i] == 2) words[2 * i + 1] = mem[rpos[i]];
            if (!super)
                for (uint32_t c = lane_pos[i]; c != 0;) {
                    const uint32_t par = (c - 1) / 2; const int pl = lane_of(par);
                    assert(pl >= 1 && pl < i);               // lane order = depth order along every path
                    anc_mask[i] |= uint64_t{1} << pl;
                    if (c == 2 * par + 2) anc_val[i] |= uint64_t{1} << pl;
                    c = par;
                }
            const uint64_t bit = uint64_t{1} << i;
            VALID |= bit;
            if (kind_l[i] == 1 || kind_l[i] == 2) EXL |= bit;
            if (kind_r[i] == 1 || kind_r[i] == 2) EXR |= bit;
            if (kind_l[i] == 1) HLm |= bit;
            if (kind_r[i] == 1) HRm |= bit;
            if (kind_l[i] == 2) TL |= bit;
            if (kind_r[i] == 2) TR |= bit;
            if (i >= 1 && i <= D && spine(i) == rpos[i]) SIDE |= bit;
        }
        CHAIN = (uint64_t{2} << D) - 1; CHAINCMP = (uint64_t{1} << D) - 1;
        recompute();
        vc = (SIDE >> D & 1) ? rc[D] : lc[D]; vi = (SIDE >> D & 1) ? ri[D] : li[D];
        root_c = lc[0];
    }

    void recompute() {
        B = 0;
        for (int i = 0; i < 64; ++i) if (!(rc[i] > lc[i])) B |= uint64_t{1} << i;       // comp(second, second - 1): take the left child iff right > left
        B &= EXR;
        for (int i = 0; i < 64; ++i) {
            const bool b = B >> i & 1;
            chc[i] = b ? rc[i] : lc[i]; chi[i] = b ? ri[i] : li[i]; nl[i] = b ? nl_r[i] : nl_l[i];
        }
        EX = EXL ^ (B & (EXL ^ EXR));
        CHH = HLm ^ (B & (HLm ^ HRm));
        for (int i = 0; i < 64; ++i) { nchc[i] = chc[nl[i] & 63]; nchi[i] = chi[nl[i] & 63]; }       // ds_bpermute (lane = address / 4 mod 64)
    }
    void fold(const Ent* w) {                                 // w = the words as read (two per lane)
        PEND = 0;
        for (int i = 0; i < 64; ++i) {
            if (TL >> i & 1) { lc[i] = w[2 * i].cost; li[i] = w[2 * i].id; }
            if (TR >> i & 1) { rc[i] = w[2 * i + 1].cost; ri[i] = w[2 * i + 1].id; }
            if (int32_t(li[i] | ri[i]) < 0) PEND |= uint64_t{1} << i;
        }
    }

    // one replacement; returns false when it has to wait for a pending child (registers re-folded, nothing else changed)
    bool head_replace(float xc, uint32_t xi) {
        uint64_t P = 0, G = 1;
        for (int i = 0; i < 64; ++i) {
            if (((B ^ anc_val[i]) & anc_mask[i]) == 0) P |= uint64_t{1} << i;
            if (!(chc[i] > vc)) G |= uint64_t{1} << i;
        }
        P &= VALID;
        const uint64_t PG = P & G & EX, stop = P & ~PG, below = stop - 1, M = PG & below;
        if (P & PEND & (stop ^ below)) { ++head_blocked; fold(words); recompute(); return false; }
        const uint32_t pstar = stop ? uint32_t(ffs64(stop)) : 0xffffffffu;
        const uint64_t MH = M & CHH, TOK = M & ~CHH;
        for (int i = 0; i < 64; ++i) {
            const bool from_below = uint32_t(nl[i]) < pstar;
            const float wc = from_below ? nchc[i] : vc; const uint32_t wi = from_below ? nchi[i] : vi;
            if ((MH & B) >> i & 1) { rc[i] = wc; ri[i] = wi; }
            if ((MH & ~B) >> i & 1) { lc[i] = wc; li[i] = wi; }
        }
        if (TOK) {
            assert(stop == 0 && __builtin_popcountll(TOK) == 1);
            const int z = ffs64(TOK);
            assert(z == 63 - __builtin_clzll(P));
            const int s = 2 * z + int(B >> z & 1);
            assert(!(words[s].id & kTokenTag));
            words[s] = Ent{vc, vi | kTokenTag};
            ++tokens;
        } else assert(stop != 0);
        const Ent w_after[128] = {};                          // (the kernel reads the words here; the model folds from `words` below)
        (void)w_after;
        // push x: a sorted insert into the spine (lane i keeps level i, level D = position k - 1)
        float cvc[64], upc[64]; uint32_t cvi[64], upi[64];
        for (int i = 0; i < 64; ++i) { const bool sd = SIDE >> i & 1; cvc[i] = sd ? rc[i] : lc[i]; cvi[i] = sd ? ri[i] : li[i]; }
        for (int i = 0; i < 64; ++i) { upc[i] = i ? cvc[i - 1] : cvc[0]; upi[i] = i ? cvi[i - 1] : cvi[0]; }       // DPP wave_shr:1
        uint64_t gt = 0;
        for (int i = 0; i < 64; ++i) if (cvc[i] > xc) gt |= uint64_t{1} << i;
        const uint64_t stays = ~gt & CHAINCMP;
        const int land = stays ? 64 - __builtin_clzll(stays) : 0;
        const uint64_t SH = CHAIN & (~uint64_t{0} << land);
        float wcD = 0; uint32_t wiD = 0;
        for (int i = 0; i < 64; ++i) {
            const bool not_here = i != land;
            const float wc = not_here ? upc[i] : xc; const uint32_t wi = not_here ? upi[i] : xi;
            if ((SH & SIDE) >> i & 1) { rc[i] = wc; ri[i] = wi; }
            if ((SH & ~SIDE) >> i & 1) { lc[i] = wc; li[i] = wi; }
            if (i == D) { wcD = wc; wiD = wi; }
        }
        vc = wcD; vi = wiD;
        fold(words);
        recompute();
        root_c = lc[0];
        ++replacements;
        return true;
    }

    // ---- T1: one look at the words (one lane per sub-root; the rest of a pop goes into the mailbox of one of the walker waves, in turns) ----
    static constexpr int kWalkWaves = 2;
    struct Box { uint32_t tag = 0, pos = 0; Ent v; };         // tag: 0 empty, 1 a pop that goes on at pos
    Box boxes[kWalkWaves][128];
    int turn[128] = {};
    void t1_look() {
        for (int i = 0; i < n_lanes; ++i)
            for (int side = 0; side < 2; ++side) {
                if ((side ? kind_r[i] : kind_l[i]) != 2) continue;
                const uint32_t widx = uint32_t(2 * i + side), e = side ? rpos[i] : lpos[i];
                if ((words[widx].id & (kTokenTag | kHandedTag)) != kTokenTag) continue;
                Ent v = words[widx]; v.id &= ~kTokenTag;
                if (e >= first_parked) {                      // no children inside LDS: the deep wave does all of it
                    if (deep_ring.size() >= deep_cap) continue;
                    words[widx].id = v.id | kTokenTag | kHandedTag;
                    deep_ring.push_back(Tok{e, widx, v}); ++forwarded;
                    continue;
                }
                const uint32_t c_l = 2 * e + 1;
                const Ent cl = mem[c_l], cr = mem[c_l + 1];
                if (cl.id == kOpenHole || cr.id == kOpenHole) continue;      // the pop ahead in this sub-heap has not moved on yet
                const bool right = !(cr.cost > cl.cost);
                const Ent c = right ? cr : cl; const uint32_t cpos = c_l + (right ? 1 : 0);
                const bool lands = c.cost > v.cost;
                if (!lands && cpos >= first_parked) {         // the hole would stand on the last LDS level: a task for the deep wave
                    if (deep_ring.size() >= deep_cap) continue;
                    words[widx] = c; mem[cpos].id = kOpenHole;
                    deep_ring.push_back(Tok{cpos, kNoSlot, v}); ++deep_tasks; ++first_steps;
                    continue;
                }
                Box& box = boxes[turn[widx]][widx];
                if (!lands && box.tag) continue;              // a pop that goes on needs the mailbox empty
                words[widx] = lands ? v : c;
                ++first_steps;
                if (!lands) { mem[cpos].id = kOpenHole; box = Box{1, cpos, v}; turn[widx] = (turn[widx] + 1) % kWalkWaves; }
            }
    }
    void t1_flush() {                                         // the sub-roots' own heap entries: nobody's input, written once at the end
        for (int i = 0; i < n_lanes; ++i) {
            if (kind_l[i] == 2 && lpos[i] < first_parked) mem[lpos[i]] = words[2 * i];
            if (kind_r[i] == 2 && rpos[i] < first_parked) mem[rpos[i]] = words[2 * i + 1];
        }
    }

    // ---- a walker wave: one look (lane j walks pops of sub-root j) ----------------------------------------------------------------------
    bool w_live[kWalkWaves][128] = {}; uint32_t w_pos[kWalkWaves][128] = {}; Ent w_v[kWalkWaves][128];
    void walker_look(int ww) {
        struct Plan { bool act = false, lands = false, parks = false; uint32_t cpos = 0; Ent c; } pl[128];
        size_t n_parks = 0;
        for (int j = 0; j < 128; ++j) {
            if (!w_live[ww][j]) continue;
            const uint32_t p = w_pos[ww][j];
            const Ent cl = mem[2 * p + 1], cr = mem[2 * p + 2];
            Plan& q = pl[j];
            q.act = cl.id != kOpenHole && cr.id != kOpenHole;
            const bool right = !(cr.cost > cl.cost);
            q.c = right ? cr : cl; q.cpos = 2 * p + 1 + (right ? 1 : 0);
            q.lands = q.c.cost > w_v[ww][j].cost;
            q.parks = q.act && !q.lands && q.cpos >= first_parked;
            if (q.parks) ++n_parks;
        }
        const bool room = deep_ring.size() + n_parks <= deep_cap;       // (no room: the parking pops wait a look)
        for (int j = 0; j < 128; ++j) {
            const Plan& q = pl[j];
            if (w_live[ww][j] && q.act && (room || !q.parks)) {
                mem[w_pos[ww][j]] = q.lands ? w_v[ww][j] : q.c;
                if (!q.lands) mem[q.cpos].id = kOpenHole;
                if (q.parks) { deep_ring.push_back(Tok{q.cpos, kNoSlot, w_v[ww][j]}); ++deep_tasks; }
                w_live[ww][j] = !q.lands && !q.parks;
                w_pos[ww][j] = q.cpos;
            }
            Box& box = boxes[ww][j];
            if (!w_live[ww][j] && box.tag) { w_live[ww][j] = true; w_pos[ww][j] = box.pos; w_v[ww][j] = box.v; box.tag = 0; }
        }
    }

    // ---- deep: a batch of tasks, one lane each, top-down ------------------------------------------------------------------------------
    void deep_batch() {
        const size_t n = std::min<size_t>(deep_ring.size(), 64);
        for (size_t t = 0; t < n; ++t) {
            const Tok task = deep_ring[t];
            uint32_t p = task.pos; Ent top_val = task.v;
            for (;;) {
                const uint32_t l = 2 * p + 1;
                Ent w = task.v; uint32_t next = 0; bool lands = true;
                if (l < len) {
                    Ent c = mem[l]; next = l;
                    if (l + 1 < len) { const Ent r = mem[l + 1]; if (!(r.cost > c.cost)) { c = r; next = l + 1; } }
                    assert(c.id != kOpenHole);
                    if (!(c.cost > task.v.cost)) { w = c; lands = false; }
                }
                if (p == task.pos) top_val = w;
                mem[p] = w;
                if (lands) break;
                p = next;
            }
            if (task.slot != kNoSlot) words[task.slot] = top_val;
        }
        deep_ring.erase(deep_ring.begin(), deep_ring.begin() + long(n));
    }

    bool idle() const {
        if (!deep_ring.empty()) return false;
        for (int w = 0; w < kWalkWaves; ++w) for (int i = 0; i < 128; ++i) if (boxes[w][i].tag || w_live[w][i]) return false;
        for (int i = 0; i < 128; ++i) if (words[i].id & kTokenTag) return false;
        return true;
    }
    void flush() {
        t1_flush();
        for (int i = 0; i < n_lanes; ++i) {
            if (kind_l[i] == 1 || kind_l[i] == 3) mem[lpos[i]] = Ent{lc[i], li[i]};
            if (kind_r[i] == 1 || kind_r[i] == 3) mem[rpos[i]] = Ent{rc[i], ri[i]};
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
        if (++guard > 3000000L + 200L * long(n)) { std::printf("  livelock at i=%zu of %zu: deep %zu PEND %llx, blocked %ld tokens %ld first %ld\n", i, n, m.deep_ring.size(), (unsigned long long)m.PEND, m.head_blocked, m.tokens, m.first_steps); for (int q = 0; q < 128; ++q) if (m.words[q].id & kTokenTag) std::printf("    word %d id %x\n", q, m.words[q].id); return false; }
        const unsigned pick = rng() % 16;
        if (pick < unsigned(head_bias)) {
            if (i < n) {
                if (!(m.root_c < cost[i])) { ++i; continue; }
                if (m.head_replace(cost[i], uint32_t(i))) ++i;
            }
        } else {
            const unsigned r = rng() % 4;
            if (r < 2) m.t1_look(); else if (r == 2) m.walker_look(int(rng() % Model::kWalkWaves)); else m.deep_batch();
        }
    }
    m.flush();
    bool ok = size_t(m.replacements) == ref_repl;
    for (size_t j = 0; j < k && ok; ++j) ok = m.mem[j].id == uint32_t(ref[j].id) && m.mem[j].cost == ref[j].cost;
    if (!ok || verbose)
        std::printf("%s seed %u n %zu k %zu HL %d LL %d distinct %d: %ld replacements (ref %zu), %ld tokens (%ld passed on whole), %ld first steps, %ld deep, head blocked %ld\n",
                    ok ? "ok  " : "FAIL", seed, n, k, HL, LL, distinct, m.replacements, ref_repl, m.tokens, m.forwarded, m.first_steps, m.deep_tasks, m.head_blocked);
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
    // the kernel's own configuration
    for (size_t k : {16385u, 16386u, 32768u, 50000u})
        for (int distinct : {0, 5, 1000})
            if (!run_case(uint32_t(k + distinct), k * 3, k, 5, 14, distinct, 8, true)) ++fails;
    std::printf("%d failures\n", fails);
    return fails != 0;
}
