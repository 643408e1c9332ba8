# Developer script (GPU box): std::sort replay with the many-block partition step for segments > 16384 ids (default) and without.
for m in 1 0; do
  echo "== BVH_AMD_SORT_HUGE=$m"
  export BVH_AMD_SORT_HUGE=$m
  python tools/build_profile.py soup 1000000 1 1 7 | grep BUILD
  python tools/build_profile.py terrain 1000000 1 1 7 | grep BUILD
  python tools/build_profile.py soup 10000000 1 1 5 | grep BUILD
  python tools/build_profile.py terrain 10000000 1 1 5 | grep BUILD
  python tools/build_profile.py soup 1000000 1 0 5 | grep BUILD
  python tools/build_profile.py sponza 262144 1 0 5 | grep BUILD
  python tools/build_profile.py soup 10000000 1 0 3 | grep BUILD
done
