#!/bin/bash
# A/B builds of the F(4x4,3x3) kernel for same-box timing: tools/ab_build.sh NAME "sed-expression on dncnn_wino44.hip"
# -> pnp_svrg_amd/lib/ab/NAME.so (git-ignored; selected with PNP_HIP_LIB).  The other objects are reused from csrc/build.
set -e
cd "$(dirname "$0")/../pnp_svrg_amd/csrc"
name=$1; expr=$2
mkdir -p ../lib/ab /tmp/ab_$name
sed -e "$expr" dncnn_wino44.hip > ./_ab_$name.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function -mllvm -pragma-unroll-threshold=200000 $AB_FLAGS -x hip -c ./_ab_$name.hip -o /tmp/ab_$name/w44.o
rm -f ./_ab_$name.hip
objs=$(ls build/*.o | grep -v dncnn_wino44.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/ab/$name.so $objs /tmp/ab_$name/w44.o
echo built ../lib/ab/$name.so
