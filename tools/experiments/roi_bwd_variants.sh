#!/bin/bash
# Experiment builds of libovis_hip.so that differ in csrc/roi_align_bwd_plane.hip only (the other objects are the regular
# build's): bash tools/experiments/roi_bwd_variants.sh name:"flags" ...  ->  tools/experiments/variants/libovis_hip_<name>.so
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"
make -C "$ROOT/cvpr22_cross_modal_pseudo_labeling_amd/csrc" -j8 -s
OUT="$ROOT/tools/experiments/variants"; mkdir -p "$OUT"
FLAGS="-mllvm -amdgpu-mfma-vgpr-form=1 --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -I$ROOT/include -I$ROOT/cvpr22_cross_modal_pseudo_labeling_amd/csrc"
OTHERS=$(ls "$ROOT"/build/ovis_hip/*.o | grep -v roi_align_bwd_plane.o)
for spec in "$@"; do
  name="${spec%%:*}"; extra="${spec#*:}"; [ "$extra" = "$spec" ] && extra=""
  /opt/rocm/bin/hipcc $FLAGS $extra -c "$ROOT/cvpr22_cross_modal_pseudo_labeling_amd/csrc/roi_align_bwd_plane.hip" -o "$OUT/bwd_$name.o"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libovis_hip_$name.so" $OTHERS "$OUT/bwd_$name.o"
  rm -f "$OUT/bwd_$name.o"
  echo "built $name ($extra)"
done
