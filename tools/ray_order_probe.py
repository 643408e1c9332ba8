"""Developer experiment: how much faster does the closest-hit kernel run when the ray batch is reordered for coherence?
    python tools/ray_order_probe.py [soup|terrain|sponza] [n_tris] [n_rays]
Orders tried: as generated; origin Morton (10 bits/axis); origin Morton (b bits/axis) + direction octant; direction octant first."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bvh_amd
from bvh_amd import synth

def spread(v, bits):
    out = torch.zeros_like(v)
    for b in range(bits):
        out |= ((v >> b) & 1) << (3 * b)
    return out

def morton(org, lo, hi, bits):
    q = ((org - lo) / (hi - lo)).clamp(0, 1 - 1e-7)
    q = (q * (1 << bits)).to(torch.int64)
    return spread(q[:, 0], bits) | (spread(q[:, 1], bits) << 1) | (spread(q[:, 2], bits) << 2)

def main():
    scene = sys.argv[1] if len(sys.argv) > 1 else "soup"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    nr = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 24
    tris = {"soup": synth.soup, "terrain": synth.terrain, "sponza": synth.sponza_proxy, "proc": synth.procedural_10m}[scene](n)
    d_tris = torch.from_numpy(tris).cuda()
    bb, cc = bvh_amd.tri_bounds(d_tris)
    bvh = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality(int(os.environ.get("Q", "2")))), thread_pool=bvh_amd.ThreadPool())
    prims = bvh_amd.precompute_tris(d_tris, bvh.device_prim_ids())
    lo, hi = synth.scene_bounds(tris)
    any_hit = len(sys.argv) > 4 and sys.argv[4] == "any"
    rays = torch.from_numpy((synth.rays_shadow if any_hit else synth.rays_closest)(nr, lo, hi)).cuda()
    olo, ohi = rays[:, :3].min(0).values, rays[:, :3].max(0).values
    octant = ((rays[:, 3] < 0).to(torch.int64) | ((rays[:, 4] < 0).to(torch.int64) << 1) | ((rays[:, 5] < 0).to(torch.int64) << 2))
    orders = {"as generated": None}
    m = lambda bits: morton(rays[:, :3], olo, ohi, bits)
    def fine(bits_coarse, bits_fine):                         # the low `bits_fine` bits per axis below a coarse cell of `bits_coarse` bits per axis
        full = m(bits_coarse + bits_fine)
        return full >> (3 * bits_fine), full & ((1 << (3 * bits_fine)) - 1)
    dq = ((rays[:, 3:6] / rays[:, 3:6].abs().max(1, keepdim=True).values * 0.5 + 0.5).clamp(0, 1 - 1e-6) * 4).to(torch.int64)
    dkey = dq[:, 0] | (dq[:, 1] << 2) | (dq[:, 2] << 4)
    keys = {
        "origin 4b + octant (library)": (m(4) << 3) | octant,
        "origin 5b + octant": (m(5) << 3) | octant,
        "origin 6b + octant": (m(6) << 3) | octant,
        "octant + origin 4b": (octant << 12) | m(4),
        "octant + origin 5b": (octant << 15) | m(5),
        "origin 3b + octant + fine 1b": (fine(3, 1)[0] << 6) | (octant << 3) | fine(3, 1)[1],
        "origin 3b + octant + fine 2b": (fine(3, 2)[0] << 9) | (octant << 6) | fine(3, 2)[1],
        "origin 2b + octant + fine 2b": (fine(2, 2)[0] << 9) | (octant << 6) | fine(2, 2)[1],
        "origin 2b + octant + fine 3b": (fine(2, 3)[0] << 12) | (octant << 9) | fine(2, 3)[1],
        "origin 1b + octant + fine 4b": (fine(1, 4)[0] << 15) | (octant << 12) | fine(1, 4)[1],
        "origin 4b + dir 6b": (m(4) << 6) | dkey,
        "origin 3b + dir 6b + fine 1b": (fine(3, 1)[0] << 9) | (dkey << 3) | fine(3, 1)[1],
    }
    for name, k in keys.items():
        orders[name] = torch.argsort(k, stable=True)
    out = torch.empty((nr, 4), dtype=torch.float32, device="cuda")
    base = None
    for name, perm in orders.items():
        r = rays if perm is None else rays[perm].contiguous()
        for _ in range(2):
            bvh_amd.intersect(bvh, prims, r, any_hit, not any_hit, out=out, sort_rays=False)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(5):
            bvh_amd.intersect(bvh, prims, r, any_hit, not any_hit, out=out, sort_rays=False)
        ev1.record(); torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / 5
        if perm is None:
            base = out.clone()
        else:
            back = torch.empty_like(out); back[perm] = out
            assert torch.equal(back.view(torch.int32), base.view(torch.int32)), name
        print(f"ORDER {scene} n={n} rays={nr} any={int(any_hit)} {name:34s} {ms:8.3f} ms {nr / ms / 1e3:8.1f} Mrays/s", flush=True)

if __name__ == "__main__":
    main()
