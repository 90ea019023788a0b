#!/bin/bash
# Dev box: build a named variant of libhfpf.so for same-box A/B runs (tools/ab_kernels.sh name=build/variants/name.so ...).
#   tools/build_variant.sh <name> [extra hipcc flags, e.g. -DHFPF_UPD2_CHUNK=12]      from the working tree
#   tools/build_variant.sh --rev <git rev> <name> [flags]                            from a committed revision
# build/ is git-ignored but travels to the GPU box with the snapshot.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
SRC=$R/high-fidelity-pointcloud-fusion_amd/csrc
if [ "$1" = "--rev" ]; then
  REV=$2; shift 2
  TMP=$(mktemp -d)
  mkdir -p $TMP/high-fidelity-pointcloud-fusion_amd/csrc $TMP/include
  for f in det_math.hpp geometry.hpp tables.hpp stats.hpp kernels.hpp hfpf.hip; do
    git -C $R show $REV:high-fidelity-pointcloud-fusion_amd/csrc/$f > $TMP/high-fidelity-pointcloud-fusion_amd/csrc/$f
  done
  for f in hfpf.h hfpf_probe.h; do git -C $R show $REV:include/$f > $TMP/include/$f; done
  SRC=$TMP/high-fidelity-pointcloud-fusion_amd/csrc
fi
NAME=$1; shift
mkdir -p $R/build/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fPIC -Wall -Wno-unused-function -Wno-unused-result "$@" \
  -shared -o $R/build/variants/$NAME.so $SRC/hfpf.hip
echo "built build/variants/$NAME.so ($*)"
