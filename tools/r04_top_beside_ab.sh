# Developer script (GPU box): Quality::Low / no-pruning builds with the top level built beside the forest (default) and after it.
for m in 1 0; do
  echo "== BVH_AMD_TOP_BESIDE=$m"
  export BVH_AMD_TOP_BESIDE=$m
  python tools/build_profile.py soup 1000000 0 1 9 | grep BUILD
  python tools/build_profile.py soup 10000000 0 1 5 | grep BUILD
  python tools/build_profile.py terrain 1000000 0 1 7 | grep BUILD
  python tools/build_profile.py sponza 262144 0 1 7 | grep BUILD
  python tools/build_profile.py soup 100000 0 1 7 | grep BUILD
  python tools/build_profile.py soup 1000000 1 1 5 | grep BUILD
done
