# Developer script (GPU box): DefaultBuilder timings with k_sweep_medium on (default) and off.
for m in 1 0; do
  echo "== BVH_AMD_SWEEP_MEDIUM=$m"
  export BVH_AMD_SWEEP_MEDIUM=$m
  python tools/build_profile.py soup 1000000 0 1 7 | grep BUILD
  python tools/build_profile.py soup 1000000 1 1 7 | grep BUILD
  python tools/build_profile.py soup 10000000 0 1 5 | grep BUILD
  python tools/build_profile.py soup 10000000 1 1 5 | grep BUILD
  python tools/build_profile.py soup 1000000 1 0 5 | grep BUILD
  python tools/build_profile.py terrain 1000000 0 1 7 | grep BUILD
  python tools/build_profile.py terrain 1000000 1 1 7 | grep BUILD
  python tools/build_profile.py sponza 262144 1 0 5 | grep BUILD
  python tools/build_profile.py sponza 262144 1 1 5 | grep BUILD
done
