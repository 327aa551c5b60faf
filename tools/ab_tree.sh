# A/B of the hierarchical cell list on one box: tools/ab_tree.sh "<build flags>;<ENV=val ...>" ...   (each variant is compiled on the box)
for v in "$@"; do
  flags="${v%%;*}"; envs="${v#*;}"
  echo "[$flags | $envs]"
  env PCT_EXTRA_FLAGS="$flags" $envs timeout -k 10 300 python tools/density_probe.py 1000000 "${AB_CLOUD:-1/r^2}" 2>/dev/null | grep TREE
done
