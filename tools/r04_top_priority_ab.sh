for p in 1 0; do
  echo "== BVH_AMD_TOP_PRIORITY=$p"
  export BVH_AMD_TOP_PRIORITY=$p
  python tools/build_profile.py soup 1000000 0 1 9 | grep BUILD
  python tools/build_profile.py soup 10000000 0 1 5 | grep BUILD
  python tools/build_profile.py terrain 1000000 0 1 7 | grep BUILD
done
