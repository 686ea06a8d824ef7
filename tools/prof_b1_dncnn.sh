#!/bin/bash
# rocprofv3 kernel stats of DnCNN-17 forward passes on ONE 256 x 256 image (the reference's own usage)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/b1_dncnn
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 tools/time_fwd_b1.py > $O/out.txt 2> $O/err.txt
echo done
