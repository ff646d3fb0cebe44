#!/bin/bash
# Experiment builds of libovis_hip.so.  The product sources carry NO compile-time switch (tests/test_abi.py): the probes
# (wrong-output timing stamps, ablations) and knobs of earlier rounds live as patches under tools/experiments/patches and
# are applied here to a scratch COPY of one source file, compiled with the variant's -D flags and linked with the regular
# build's other objects into a separate library -- never into cvpr22_cross_modal_pseudo_labeling_amd/libovis_hip.so.
#
#   bash tools/experiments/build_variants.sh <file>:<name>[:"flags"[:<patch file>]] ...
#     <file> = split_gemm | roi_align_bwd_plane | nms  ->  tools/experiments/variants/libovis_hip_<name>.so
#     <patch file> (under tools/experiments/patches) replaces the file's default patch, e.g. split_gemm:tn_gmajor:"-DOVIS_TN_FORCE_GMAJOR=1":tn_tile_order.patch
#   e.g.  split_gemm:gw16:"-DOVIS_SG_GW=16"  split_gemm:nbuf4:"-DOVIS_EPI_NBUF=4"  split_gemm:tn_noshift:"-DOVIS_TN_ABL_NOSHIFT"
#         roi_align_bwd_plane:regplane:"":roi_bwd_register_plane.patch   (round 5: plane strips in the matrix core's accumulators)
#         roi_align_bwd_plane:probe_time:"-DOVIS_ROI_PROBE_TIME"  roi_align_bwd_plane:kri6:"-DOVIS_ROI_KRI=6 -DOVIS_ROI_KRING=2"
# tools/experiments/ab_bench.py lib:<name> loads such a library for a same-box A/B.
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"
CSRC="$ROOT/cvpr22_cross_modal_pseudo_labeling_amd/csrc"
make -C "$CSRC" -j8 -s
OUT="$ROOT/tools/experiments/variants"; mkdir -p "$OUT"
FLAGS="-mllvm -amdgpu-mfma-vgpr-form=1 --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -I$ROOT/include -I$CSRC -I$ROOT/build/ovis_hip"
declare -A PATCH=( [split_gemm]=split_gemm_knobs.patch [roi_align_bwd_plane]=roi_bwd_probes.patch [nms]=nms_wide_stores.patch )
declare -A EXTRA=( [split_gemm]="" [roi_align_bwd_plane]="-fno-slp-vectorize" [nms]="" )
for spec in "$@"; do
  IFS=: read -r file name flags patchfile <<< "$spec"
  [ -n "${PATCH[$file]:-}" ] || { echo "unknown source $file"; exit 1; }
  [ -n "${patchfile:-}" ] && PATCH[$file]="$patchfile"
  TMP="$(mktemp -d)"; cp "$CSRC/$file.hip" "$TMP/$file.hip"
  patch -s "$TMP/$file.hip" "$ROOT/tools/experiments/patches/${PATCH[$file]}"
  /opt/rocm/bin/hipcc $FLAGS ${EXTRA[$file]} ${flags:-} -c "$TMP/$file.hip" -o "$TMP/$file.o"
  OTHERS=$(ls "$ROOT"/build/ovis_hip/*.o | grep -v "/$file.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libovis_hip_$name.so" $OTHERS "$TMP/$file.o"
  rm -rf "$TMP"
  echo "built $name from $file.hip + ${PATCH[$file]} (${flags:-no flags})"
done
