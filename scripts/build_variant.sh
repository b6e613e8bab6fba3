#!/bin/bash
# build a variant of the library into _b_var/<name>.so (git-ignored; travels with gpurun) for same-box A/B runs (scripts/ab_libs.sh):
#   scripts/build_variant.sh auto4 -DPICLES_AUTO_WAVES=4
set -e
name=$1; shift
src=picles_amd/csrc
out=_b_var; mkdir -p $out/obj_$name
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -munsafe-fp-atomics -fPIC -fvisibility=hidden -w $*"
for u in picles_hip k_step_explicit k_step_auto k_advance; do
  /opt/rocm/bin/hipcc $FLAGS -c $src/$u.hip -o $out/obj_$name/$u.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $out/obj_$name/*.o -o $out/$name.so
rm -rf $out/obj_$name
ls -la $out/$name.so
