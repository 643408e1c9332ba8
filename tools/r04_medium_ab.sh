# Developer script (GPU box): DefaultBuilder timings with k_medium on (default) and off.  bash tools/r04_medium_ab.sh
for m in 1 0; do
  echo "== BVH_AMD_MEDIUM=$m"
  export BVH_AMD_MEDIUM=$m
  for q in 0 1; do python tools/build_profile.py soup 1000000 $q 1 7; python tools/build_profile.py soup 10000000 $q 1 5; done 2>&1 | grep BUILD
  python tools/build_profile.py soup 1000000 0 0 7 | grep BUILD
  python tools/build_profile.py soup 1000000 1 0 7 | grep BUILD
  python tools/build_profile.py terrain 1000000 0 1 7 | grep BUILD
  python tools/build_profile.py terrain 1000000 1 1 7 | grep BUILD
  python tools/build_profile.py terrain 10000000 0 1 5 | grep BUILD
  python tools/build_profile.py sponza 262144 0 0 7 | grep BUILD
done
