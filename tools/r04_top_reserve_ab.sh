# Developer script (GPU box): Low builds with Phase B on a CU-masked stream that leaves N CUs to the top-level worker.
for m in 8 0 4 16 32; do
  echo "== BVH_AMD_TOP_RESERVE=$m"
  export BVH_AMD_TOP_RESERVE=$m
  python tools/build_profile.py soup 1000000 0 1 9 | grep BUILD
  python tools/build_profile.py soup 10000000 0 1 5 | grep BUILD
  python tools/build_profile.py terrain 1000000 0 1 7 | grep BUILD
  python tools/build_profile.py sponza 262144 0 1 7 | grep BUILD
done
