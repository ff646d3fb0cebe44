#!/bin/bash
# Experiment builds of libovis_hip.so with different compile-time settings of csrc/split_gemm.hip (the other objects are
# the regular build's): bash tools/experiments/build_variants.sh  ->  tools/experiments/variants/libovis_hip_<name>.so
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"
make -C "$ROOT/cvpr22_cross_modal_pseudo_labeling_amd/csrc" -j8 -s
OUT="$ROOT/tools/experiments/variants"; mkdir -p "$OUT"
FLAGS="-mllvm -amdgpu-mfma-vgpr-form=1 --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -I$ROOT/include -I$ROOT/cvpr22_cross_modal_pseudo_labeling_amd/csrc"
OTHERS=$(ls "$ROOT"/build/ovis_hip/*.o | grep -v split_gemm.o)
build() { # name, extra flags
  /opt/rocm/bin/hipcc $FLAGS $2 -c "$ROOT/cvpr22_cross_modal_pseudo_labeling_amd/csrc/split_gemm.hip" -o "$OUT/split_gemm_$1.o"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libovis_hip_$1.so" $OTHERS "$OUT/split_gemm_$1.o"
  rm -f "$OUT/split_gemm_$1.o"
  echo "built $1 ($2)"
}
for v in "$@"; do
  case "$v" in
    nbuf4) build nbuf4 "-DOVIS_EPI_NBUF=4" ;;
    nbuf5) build nbuf5 "-DOVIS_EPI_NBUF=5" ;;
    dual3) build dual3 "-DOVIS_EPI_NBUF_DUAL=3" ;;
    nbuf4dual3) build nbuf4dual3 "-DOVIS_EPI_NBUF=4 -DOVIS_EPI_NBUF_DUAL=3" ;;
    base) build base "" ;;
    *:*) build "${v%%:*}" "${v#*:}" ;;   # name:"compiler flags"
    *) echo "unknown variant $v"; exit 1 ;;
  esac
done
