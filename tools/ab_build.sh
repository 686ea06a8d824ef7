#!/bin/bash
# A/B builds for same-box timing: tools/ab_build.sh NAME "sed-expression" [FILE (default dncnn_wino44.hip)]
# -> pnp_svrg_amd/lib/ab/NAME.so (git-ignored; selected with PNP_HIP_LIB).  The other objects are reused from csrc/build.
# AB_FLAGS adds compiler flags.  (If the expression touches a header, name the .hip file that includes it and pass
# AB_HEADER=header.h: the patched header is compiled in place of the original through -include ordering.)
set -e
cd "$(dirname "$0")/../pnp_svrg_amd/csrc"
name=$1; expr=$2; file=${3:-dncnn_wino44.hip}
mkdir -p ../lib/ab /tmp/ab_$name
extra=""
[ "$file" = dncnn_wino44.hip ] && extra="-mllvm -pragma-unroll-threshold=200000 -fno-slp-vectorize"
[ "$file" = prox.hip ] || [ "$file" = nlm.hip ] && extra="-ffp-contract=off"
if [ -n "$AB_HEADER" ]; then
  mkdir -p /tmp/ab_$name/inc && cp *.h /tmp/ab_$name/inc/ && sed -e "$expr" $AB_HEADER > /tmp/ab_$name/inc/$AB_HEADER
  cp $file /tmp/ab_$name/inc/_ab_src.hip
  inc=$(cd ../../include && pwd)
  ( cd /tmp/ab_$name/inc && sed -i 's#"../../include/pnp_hip.h"#"'"$inc"'/pnp_hip.h"#' common.h && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function $extra $AB_FLAGS -x hip -c _ab_src.hip -o /tmp/ab_$name/obj.o )
else
  sed -e "$expr" $file > ./_ab_$name.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function $extra $AB_FLAGS -x hip -c ./_ab_$name.hip -o /tmp/ab_$name/obj.o || { rm -f ./_ab_$name.hip; exit 1; }
  rm -f ./_ab_$name.hip
fi
objs=$(ls build/*.o | grep -v "build/$file.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/ab/$name.so $objs /tmp/ab_$name/obj.o
echo built ../lib/ab/$name.so
