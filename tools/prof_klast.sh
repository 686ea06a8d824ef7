#!/bin/bash
# per-kernel time of k_first / k_last for A/B library builds: tools/prof_klast.sh NAME...
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for n in "$@"; do
  rm -rf gpurun_out/pk_$n
  PNP_HIP_LIB=$GRAFT_REPO_ROOT/pnp_svrg_amd/lib/ab/$n.so rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pk_$n -- python3 tools/time_klast.py > /dev/null 2>&1
  f=$(find gpurun_out/pk_$n -name "*kernel_stats.csv" | head -1)
  echo "$n: $(grep -E 'k_last|k_first' $f | awk -F',' '{print $1, $4}' | cut -c1-30,120-)"
  grep -E 'k_last|k_first' $f | awk -F'","' '{print "   ", substr($1,1,40), $4}' 
done
