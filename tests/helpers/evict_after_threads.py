"""Developer probe (round 4): host threads that build Quality::Low trees end (their worker streams are destroyed), then the main thread's
builds push the scratch cache over a small bound so that the blocks those workers left are evicted.   BVH_AMD_CACHE_MB=64 python tests/helpers/evict_after_threads.py"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bvh_amd
from bvh_amd import synth

t0 = time.time()
def say(*a): print(f"[{time.time() - t0:6.2f}s]", *a, flush=True)
small = bvh_amd.tri_bounds(torch.from_numpy(synth.soup(150_000)).cuda())
big = bvh_amd.tri_bounds(torch.from_numpy(synth.soup(1_000_000)).cuda())
low = bvh_amd.Config(quality=bvh_amd.Quality.Low)
def work():
    torch.cuda.set_device(0)
    for _ in range(3):
        bvh_amd.DefaultBuilder.build(*small, low, thread_pool=bvh_amd.ThreadPool())
    torch.cuda.synchronize()
threads = [threading.Thread(target=work) for _ in range(3)]
for th in threads: th.start()
for th in threads: th.join()
say("threads done")
for i in range(4):
    b = bvh_amd.DefaultBuilder.build(*big, bvh_amd.Config(quality=bvh_amd.Quality(i % 2)), thread_pool=bvh_amd.ThreadPool())
    torch.cuda.synchronize()
    say("main build", i, b.node_count)
say("done")
