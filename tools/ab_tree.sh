#!/bin/bash
# A/B build of the whole library from a patched copy of the source tree:
#     tools/ab_tree.sh NAME 'shell commands run inside the copy of pnp_svrg_amd/csrc'
# -> pnp_svrg_amd/lib/ab/NAME.so (git-ignored; selected with PNP_HIP_LIB).  Objects of unchanged files are reused (make).
# Examples:  tools/ab_tree.sh oldmed 'git -C /root/repo show HEAD:pnp_svrg_amd/csrc/prox_tv.h > prox_tv.h'
#            tools/ab_tree.sh rs130 "sed -i 's/F_RS = 129/F_RS = 130/' csmri_fused.hip"
set -e
root="$(cd "$(dirname "$0")/.." && pwd)"
name=$1; cmds=$2
work=/tmp/ab_tree_$name
rm -rf $work && mkdir -p $work/pnp_svrg_amd $work/include $root/pnp_svrg_amd/lib/ab
cp -r $root/pnp_svrg_amd/csrc $work/pnp_svrg_amd/csrc
cp $root/include/pnp_hip.h $work/include/
( cd $work/pnp_svrg_amd/csrc && eval "$cmds" && make -j4 OUT=$root/pnp_svrg_amd/lib/ab/$name.so >/dev/null )
echo built $root/pnp_svrg_amd/lib/ab/$name.so
