# Developer script (GPU box): builds with the one-launch LDS std::sort for <= 4096 ids (default) and with the partition replay + radix sort.
for m in 1 0; do
  echo "== BVH_AMD_SORT_SMALL=$m"
  export BVH_AMD_SORT_SMALL=$m
  python tools/build_profile.py soup 1000000 0 1 9 | grep BUILD
  python tools/build_profile.py soup 10000000 0 1 5 | grep BUILD
  python tools/build_profile.py terrain 1000000 0 1 7 | grep BUILD
  python tools/build_profile.py sponza 262144 0 1 7 | grep BUILD
done
