#!/bin/bash
# builds variants of commit.hip's object with extra -D flags and links them beside the library: sibelia_amd/lib/var_<name>.so
# usage: tools/build_variants.sh name1:"-DX=1 -DY" name2:"-DZ" ...   (run after the normal build)
cd $(dirname $0)/..
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c sibelia_amd/csrc/commit.hip -o sibelia_amd/lib/obj/commit_$name.o || exit 1
  objs=$(ls sibelia_amd/lib/obj/*.o | grep -v "/commit" | tr '\n' ' ')
  hipcc --offload-arch=gfx950 -shared -fPIC -o sibelia_amd/lib/var_$name.so $objs sibelia_amd/lib/obj/commit_$name.o || exit 1
  echo built var_$name.so
done
