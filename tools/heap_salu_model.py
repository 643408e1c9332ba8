"""Cycle model (no kernel) of one more formulation of the candidate-heap replay of Quality::High (reinsertion_optimizer.h:88-105 over
libstdc++'s pop_heap / push_heap): the top of the heap in SGPRs, driven by the scalar unit — s_movrels / s_movreld for the indexed
entries, s_cmp_lt_u32 on the cost bits (half areas are >= 0, so the IEEE order is the unsigned order), s_cselect — instead of the
ballot -> s_ff1 -> v_readlane -> v_cndmask chains of the shipped two-wave loop (0.80 us per replacement measured, bench `build.high`).

Acceptance bar set in VERDICT round 2: build it only if the model shows < 0.3 us per replacement. It does not (0.32 us, and the
same model under-prices the shipped loop 2.8x; five heap levels through s_movrel cost MORE dependent cycles than the one LDS round trip
they replace), so nothing was built and `< 1 s @ 10M` stays a Low / Medium claim.

    python tools/heap_salu_model.py
"""
import math

CLK_GHZ = 2.4
# measured on MI355X (profiles/r02_traversal_experiments.md section 5, MI355X_MICROARCH.md "Per-instruction cycle constants"):
SALU_DEP_CLK = 5       # one lone wave, dependent SALU instructions back to back (issue + M0 hazard): ~4-5 clk each
VALU_SALU_DEP_CLK = 11 # one lone wave on VALU <-> SALU ping-pong code (ballot, ffs, readlane, DPP): measured 11 clk per instruction
LDS_LATENCY_CLK = 64   # ds_read issue -> use, dependent chain, one wave (guide: ~50; 64 with the address VALU in front)
HBM_TASK_CLK = 100     # the levels below LDS are deferred and sifted 64 at a time: amortised cost per replacement (measured ~100)
SGPR_ENTRIES = 31      # cost + id per entry = 2 SGPRs; 31 entries (levels 0-4) = 62 of the ~100 SGPRs a wave can address
LDS_LEVELS = 14        # the shipped kernel keeps the top 14 levels in LDS (16,383 entries x 8 B = 128 KB)


def model(k, name):
    depth = int(math.floor(math.log2(k)))                      # __adjust_heap walks the hole from the root to the bottom: `depth` levels
    sgpr_levels = int(math.log2(SGPR_ENTRIES + 1))            # 5
    lds_levels = max(0, min(depth, LDS_LEVELS) - sgpr_levels)
    hbm_levels = max(0, depth - LDS_LEVELS)
    # SGPR part: per level two s_movrels (children's costs, M0 set before each), s_cmp, s_cselect of the index, two s_movreld (cost + id up)
    # + index arithmetic: ~10 dependent scalar instructions
    sgpr_clk = sgpr_levels * 10 * SALU_DEP_CLK
    # LDS part, as shipped: five levels per round trip (each lane of a 31-lane BFS loads its two children, the path is five dependent
    # readlane + compare steps on the VALU <-> SALU chain)
    trips = math.ceil(lds_levels / 5)
    lds_clk = trips * (LDS_LATENCY_CLK + 5 * 2 * VALU_SALU_DEP_CLK)
    # a floor for ANY sequential formulation: one LDS round trip per five levels and nothing else
    floor_clk = sgpr_levels * 4 * SALU_DEP_CLK + trips * LDS_LATENCY_CLK
    push_clk = 2 * 6 * SALU_DEP_CLK                            # push_heap: the new value rises ~2 levels of the register-resident ancestor chain
    total = sgpr_clk + lds_clk + (HBM_TASK_CLK if hbm_levels else 0) + push_clk
    print(f"{name}: k = {k} candidates, path of {depth} levels = {sgpr_levels} in SGPRs + {lds_levels} in LDS + {hbm_levels} below LDS")
    print(f"   SGPR top {sgpr_clk} clk + LDS {trips} round trips {lds_clk} clk + deferred HBM tasks {HBM_TASK_CLK if hbm_levels else 0} clk + push {push_clk} clk "
          f"= {total} clk = {total / CLK_GHZ / 1e3:.2f} us per replacement (shipped two-wave loop: 0.80 us measured)")
    print(f"   floor of a one-at-a-time replay (scalar selects at 4 instructions per level, bare LDS latency, nothing else): {floor_clk} clk = "
          f"{floor_clk / CLK_GHZ / 1e3:.2f} us")
    # the same arithmetic applied to the SHIPPED formulation (levels 0 .. 13 in LDS, five per round trip): what the model is worth
    shipped_trips = math.ceil(min(depth, LDS_LEVELS) / 5)
    shipped_model = shipped_trips * (LDS_LATENCY_CLK + 5 * 2 * VALU_SALU_DEP_CLK) + (HBM_TASK_CLK if hbm_levels else 0) + push_clk
    print(f"   calibration: the same model prices the shipped loop at {shipped_model} clk = {shipped_model / CLK_GHZ / 1e3:.2f} us; it MEASURES 0.80 us "
          f"({0.80 * CLK_GHZ * 1e3 / shipped_model:.1f}x the model: waits between the two waves, queue tokens, the stream of costs). The SGPR top replaces ONE "
          f"{LDS_LATENCY_CLK + 5 * 2 * VALU_SALU_DEP_CLK}-clk LDS round trip by {sgpr_clk} clk of dependent scalar instructions: slower even on paper")
    return total / CLK_GHZ / 1e3


if __name__ == "__main__":
    a = model(93_321, "1M-triangle soup (1.87M nodes x 5 %)")
    b = model(758_716, "10M-triangle soup (15.2M nodes x 5 %)")
    print(f"bar: < 0.30 us per replacement -> {'met' if max(a, b) < 0.30 else 'NOT met'}; at 2 M replacements per replayed iteration of the 10M build "
          f"even the model's {b:.2f} us is {2.0 * b:.2f} s per iteration")
