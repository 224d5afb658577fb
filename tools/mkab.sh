#!/bin/bash
# Build a variant of the library HERE (the build container cross-compiles gfx950) into abl/<name>.so, which travels to the GPU box
# with the snapshot (git-ignored as *.so, not gpurun-ignored), for tools/abn.sh:
#   tools/mkab.sh name [-DZZ_X=1 ...]          the working tree
#   tools/mkab.sh name@<git-ref> [-D...]       that commit's sources (git archive into a temporary directory)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
spec=$1; shift
name=${spec%@*}
mkdir -p $R/abl
src=$R
if [[ "$spec" == *@* ]]; then
  ref=${spec#*@}
  src=$(mktemp -d)
  git -C $R archive $ref zzflate_amd/csrc include | tar -x -C $src
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared "$@" -o $R/abl/$name.so $src/zzflate_amd/csrc/zz_api.hip $src/zzflate_amd/csrc/zz_cxx_shim.cpp
[[ "$spec" == *@* ]] && rm -rf $src
echo abl/$name.so
