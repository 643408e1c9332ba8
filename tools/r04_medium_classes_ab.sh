# Developer script (GPU box): which k_medium kernel serves the 257..512 / 513..1024 lists: by population (default 0), always own (1), never (2).
for m in 0 1 2; do
  echo "== BVH_AMD_MEDIUM_CLASSES=$m"
  export BVH_AMD_MEDIUM_CLASSES=$m
  python tools/build_profile.py soup 1000000 0 1 9 | grep BUILD
  python tools/build_profile.py terrain 1000000 0 1 7 | grep BUILD
  python tools/build_profile.py sponza 262144 0 1 7 | grep BUILD
  python tools/build_profile.py soup 1000000 1 1 7 | grep BUILD
  python tools/build_profile.py terrain 1000000 1 1 7 | grep BUILD
  python tools/build_profile.py soup 10000000 0 1 5 | grep BUILD
  python tools/build_profile.py soup 1000000 0 0 7 | grep BUILD
done
