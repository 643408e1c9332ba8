# Developer script (GPU box): pruned builds with the per-node extraction (default) and with the walk per cut (BVH_AMD_EXTRACT=walk).
for m in node walk; do
  echo "== BVH_AMD_EXTRACT=$m"
  export BVH_AMD_EXTRACT=$m
  python tools/build_profile.py soup 1000000 1 1 7 | grep BUILD
  python tools/build_profile.py terrain 1000000 1 1 7 | grep BUILD
  python tools/build_profile.py sponza 262144 1 1 7 | grep BUILD
  python tools/build_profile.py soup 10000000 1 1 5 | grep BUILD
  python tools/build_profile.py terrain 10000000 1 1 5 | grep BUILD
done
