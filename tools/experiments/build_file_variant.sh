#!/bin/bash
# Link a variant of libovis_hip.so in which ONE object is compiled from a given scratch source file:
#   bash tools/experiments/build_file_variant.sh <object stem, e.g. roi_align_bwd_plane> <scratch source> <variant name> ["extra flags"]
# -> tools/experiments/variants/libovis_hip_<name>.so   (same-box A/B: tools/bench_ops.py --lib ..., ab_bench.py lib:<name>)
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"
CSRC="$ROOT/cvpr22_cross_modal_pseudo_labeling_amd/csrc"
stem="$1"; src="$2"; name="$3"; extra="${4:-}"
OUT="$ROOT/tools/experiments/variants"; mkdir -p "$OUT"
FLAGS="-mllvm -amdgpu-mfma-vgpr-form=1 --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -I$ROOT/include -I$CSRC -I$ROOT/build/ovis_hip"
[ "$stem" = roi_align_bwd_plane ] && FLAGS="$FLAGS -fno-slp-vectorize"
TMP="$(mktemp -d)"
/opt/rocm/bin/hipcc $FLAGS $extra -c "$src" -o "$TMP/$stem.o"
OTHERS=$(ls "$ROOT"/build/ovis_hip/*.o | grep -v "/$stem.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libovis_hip_$name.so" $OTHERS "$TMP/$stem.o"
rm -rf "$TMP"
echo "built $name"
