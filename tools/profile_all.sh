#!/bin/bash
# All profiling passes of tools/profile.sh for every bench workload (one gpurun call): tools/profile_all.sh <tag>
TAG=${1:-r02}
for WL in pool_n16_4k pool_4k cubes_1080p v7_1080p synthetic_4k spheres40_1080p; do
  echo "== $WL"
  bash tools/profile.sh $TAG $WL || exit $?
done
