#!/bin/bash
# build A/B variants that differ in ONE source: tools/ab_fast.sh <source-stem> tag[:-Dx=y,...] ...
# (objects of the default build are reused for every other source; then `python tools/ab.py run ...` on the GPU box)
set -e
cd "$(dirname "$0")/.."
stem=$1; shift
python -m makani_amd.build > /dev/null
for spec in "$@"; do
  tag=${spec%%:*}
  mkdir -p makani_amd/build_$tag
  cp -p makani_amd/build/*.o makani_amd/build_$tag/
  rm -f makani_amd/build_$tag/$stem.o makani_amd/libmakani_amd_$tag.so
done
python tools/ab.py build "$@"
