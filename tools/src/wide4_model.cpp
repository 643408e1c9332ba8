// Developer tool (CPU model, not product code): what would a closest-hit walk cost in RECORD FETCHES if the reference's binary tree were
// collapsed two levels at a time into 4-wide nodes whose child boxes are quantized OUTWARD to 8 bits (one 64-byte record per wide node:
// origin 12 B + 3 exponents + 4 x (6 B box + 4 B index)), visited nearest child first — against the reference's own pair-record walk
// (bvh.h:125-157) on the same tree and rays? VERDICT r5 "Next 2", step A. Input: tools/wide4_model.py dumps the reference-built tree,
// the triangles and a sample of the bench's rays.
//   g++ -O2 -std=c++20 tools/src/wide4_model.cpp -o /tmp/wide4_model && /tmp/wide4_model /tmp/wide4_<scene>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

struct Node { float b[6]; uint32_t index; };                  // bounds = {minx, maxx, miny, maxy, minz, maxz}; index = first_id << 4 | prim_count
struct Tri { float p[9]; };
struct Ray { float o[3], d[3], tmin, tmax; };

template <typename T> static std::vector<T> slurp(const std::string& path) {
    FILE* f = fopen(path.c_str(), "rb"); if (!f) { perror(path.c_str()); exit(1); }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<T> v(size_t(n) / sizeof(T)); if (fread(v.data(), sizeof(T), v.size(), f) != v.size()) exit(1); fclose(f); return v;
}

static bool slab(const float* b, const Ray& r, const float* inv, float tmax, float& t_in) {
    float t0 = r.tmin, t1 = tmax;
    for (int a = 0; a < 3; ++a) {
        float lo = (b[2 * a] - r.o[a]) * inv[a], hi = (b[2 * a + 1] - r.o[a]) * inv[a];
        if (lo > hi) std::swap(lo, hi);
        t0 = std::max(t0, lo); t1 = std::min(t1, hi * 1.0000003f);     // (a little slack in place of the robust padding: a model, not a parity tool)
    }
    t_in = t0; return t0 <= t1;
}
static bool tri_hit(const Tri& t, const Ray& r, float& tmax) {
    const float* p0 = t.p; const float e1[3] = {p0[0] - t.p[3], p0[1] - t.p[4], p0[2] - t.p[5]}, e2[3] = {t.p[6] - p0[0], t.p[7] - p0[1], t.p[8] - p0[2]};
    const float n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    const float c[3] = {p0[0] - r.o[0], p0[1] - r.o[1], p0[2] - r.o[2]};
    const float rr[3] = {r.d[1] * c[2] - r.d[2] * c[1], r.d[2] * c[0] - r.d[0] * c[2], r.d[0] * c[1] - r.d[1] * c[0]};
    const float inv_det = 1.0f / (n[0] * r.d[0] + n[1] * r.d[1] + n[2] * r.d[2]);
    const float u = (rr[0] * e2[0] + rr[1] * e2[1] + rr[2] * e2[2]) * inv_det, v = (rr[0] * e1[0] + rr[1] * e1[1] + rr[2] * e1[2]) * inv_det, w = 1.0f - u - v;
    if (u >= 0 && v >= 0 && w >= 0) { const float tt = (n[0] * c[0] + n[1] * c[1] + n[2] * c[2]) * inv_det; if (tt >= r.tmin && tt <= tmax) { tmax = tt; return true; } }
    return false;
}

struct Wide { float box[4][6]; uint32_t index[4]; int n; };   // child boxes DEQUANTIZED (outward); index as in Node (inner: first_id of its pair -> wide node of that pair)

int main(int argc, char** argv) {
    const std::string base = argc > 1 ? argv[1] : "/tmp/wide4_soup";
    const auto nodes = slurp<Node>(base + ".nodes"); const auto tris = slurp<Tri>(base + ".tris"); const auto ids = slurp<uint64_t>(base + ".prim_ids"); const auto rays = slurp<Ray>(base + ".rays");
    auto is_leaf = [&](const Node& n) { return (n.index & 15u) != 0; };
    // wide node per inner binary node X (identified by first_id of its child pair)
    std::vector<int32_t> wide_of(nodes.size(), -1);
    std::vector<Wide> wides;
    double grow = 0; size_t boxes = 0;
    std::vector<uint32_t> todo{0};
    wide_of[0] = 0; wides.emplace_back();
    for (size_t q = 0; q < todo.size(); ++q) {
        const uint32_t x = todo[q];
        const uint32_t first = nodes[x].index >> 4;
        uint32_t kids[4]; int nk = 0;
        for (uint32_t c = first; c < first + 2; ++c) {
            if (is_leaf(nodes[c])) kids[nk++] = c;
            else { const uint32_t f2 = nodes[c].index >> 4; kids[nk++] = f2; kids[nk++] = f2 + 1; }
        }
        Wide w{}; w.n = nk;
        const float* pb = nodes[x].b;
        float scale[3];
        for (int a = 0; a < 3; ++a) { const float ext = pb[2 * a + 1] - pb[2 * a]; int e; frexpf(ext / 255.0f, &e); scale[a] = ext > 0 ? ldexpf(1.0f, e) : 1.0f; }   // 2^e >= ext / 255
        for (int i = 0; i < nk; ++i) {
            const Node& k = nodes[kids[i]];
            for (int a = 0; a < 3; ++a) {
                const float qlo = std::floor((k.b[2 * a] - pb[2 * a]) / scale[a]), qhi = std::ceil((k.b[2 * a + 1] - pb[2 * a]) / scale[a]);
                w.box[i][2 * a] = pb[2 * a] + std::clamp(qlo, 0.0f, 255.0f) * scale[a];
                w.box[i][2 * a + 1] = pb[2 * a] + std::clamp(qhi, 0.0f, 255.0f) * scale[a];
            }
            auto area = [](const float* b) { const float d0 = b[1] - b[0], d1 = b[3] - b[2], d2 = b[5] - b[4]; return (d0 + d1) * d2 + d0 * d1; };
            if (area(k.b) > 0) { grow += area(w.box[i]) / area(k.b); ++boxes; }
            w.index[i] = k.index;
            if (!is_leaf(k) && wide_of[kids[i]] < 0) { wide_of[kids[i]] = int32_t(wides.size()); wides.emplace_back(); todo.push_back(kids[i]); }
        }
        wides[size_t(wide_of[x])] = w;
        // (remember which binary node each inner child is, to find its wide node)
        for (int i = 0; i < nk; ++i) if (!is_leaf(nodes[kids[i]])) wides[size_t(wide_of[x])].index[i] = (uint32_t(wide_of[kids[i]]) << 4);
    }
    size_t pairs = 0; for (const Node& n : nodes) if (!is_leaf(n)) ++pairs;
    double P = 0, T = 0, W = 0, T4 = 0, B4 = 0; size_t mism = 0, hits = 0;
    for (const Ray& r : rays) {
        float inv[3]; for (int a = 0; a < 3; ++a) inv[a] = 1.0f / r.d[a];
        // the reference's walk: near child first (by entry distance), far child pushed
        float tmax = r.tmax; int best = -1;
        { std::vector<uint32_t> st{nodes[0].index >> 4};
          while (!st.empty()) {
              const uint32_t f = st.back(); st.pop_back(); P += 1;
              float t[2]; bool h[2]; for (int i = 0; i < 2; ++i) h[i] = slab(nodes[f + i].b, r, inv, tmax, t[i]);
              int order[2] = {0, 1}; if (h[0] && h[1] && t[1] < t[0]) std::swap(order[0], order[1]);
              uint32_t push[2]; int np = 0;
              for (int oi = 0; oi < 2; ++oi) { const int i = order[oi]; if (!h[i]) continue; const Node& n = nodes[f + i];
                  if (is_leaf(n)) { const uint32_t b0 = n.index >> 4, cnt = n.index & 15u; for (uint32_t j = b0; j < b0 + cnt; ++j) { T += 1; if (tri_hit(tris[ids[j]], r, tmax)) best = int(ids[j]); } }
                  else push[np++] = n.index >> 4; }
              for (int i = np - 1; i >= 0; --i) st.push_back(push[i]);
          } }
        // the 4-wide walk: children by entry distance, leaves tested in that order, inner children pushed far to near with their entry distance
        float tmax4 = r.tmax; int best4 = -1;
        { struct E { uint32_t w; float t; }; std::vector<E> st{{0u, r.tmin}};
          while (!st.empty()) {
              const E e = st.back(); st.pop_back(); if (e.t > tmax4) continue;
              const Wide& w = wides[e.w]; W += 1; B4 += w.n;
              struct C { float t; int i; } c[4]; int nc = 0;
              for (int i = 0; i < w.n; ++i) { float t; if (slab(w.box[i], r, inv, tmax4, t)) c[nc++] = C{t, i}; }
              std::sort(c, c + nc, [](const C& a, const C& b) { return a.t < b.t; });
              E push[4]; int np = 0;
              for (int k = 0; k < nc; ++k) { const uint32_t idx = w.index[c[k].i];
                  if (idx & 15u) { if (c[k].t > tmax4) continue; const uint32_t b0 = idx >> 4, cnt = idx & 15u; for (uint32_t j = b0; j < b0 + cnt; ++j) { T4 += 1; if (tri_hit(tris[ids[j]], r, tmax4)) best4 = int(ids[j]); } }
                  else push[np++] = E{idx >> 4, c[k].t}; }
              for (int i = np - 1; i >= 0; --i) st.push_back(push[i]);
          } }
        if (best >= 0) ++hits;
        if (best != best4 && !(best >= 0 && best4 >= 0 && tmax == tmax4)) ++mism;
    }
    const double n = double(rays.size());
    std::printf("%s: %zu binary nodes (%zu pair records = %.1f MB), %zu wide records (%.1f MB); quantized child boxes are %.3fx the exact ones (half area)\n", base.c_str(), nodes.size(), pairs,
                pairs * 64e-6, wides.size(), wides.size() * 64e-6, grow / double(boxes));
    std::printf("  %zu rays, %zu hit; closest primitive differs on %zu rays (ties aside)\n", rays.size(), hits, mism);
    std::printf("  reference walk: %.2f pair records + %.2f primitive tests per ray = %.2f record-equivalents (48-byte primitives count 3/4)\n", P / n, T / n, P / n + 0.75 * T / n);
    std::printf("  4-wide walk   : %.2f wide records (%.2f child boxes) + %.2f primitive tests per ray = %.2f record-equivalents -> %.2fx fewer fetches\n", W / n, B4 / n, T4 / n,
                W / n + 0.75 * T4 / n, (P / n + 0.75 * T / n) / (W / n + 0.75 * T4 / n));
    return 0;
}
