#!/bin/bash
# Diagnostic (GPU box): rocprofv3 counter passes of one probe, one small counter group per run (a group that exceeds the
# hardware's capabilities aborts the profiler), each bounded by its own timeout.
# usage: pmc_passes.sh <out tag> <python script + args> -- <group1 counters> -- <group2 counters> ...
tag=$1; shift
cmd=(); while [ "$1" != "--" ] && [ $# -gt 0 ]; do cmd+=("$1"); shift; done
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
g=0
while [ $# -gt 0 ]; do
  shift; grp=(); while [ "$1" != "--" ] && [ $# -gt 0 ]; do grp+=("$1"); shift; done
  g=$((g+1))
  timeout -k 5 150 rocprofv3 --pmc "${grp[@]}" --kernel-trace --output-format csv -d $root/gpurun_out/pmc_${tag}_$g -- python3 "${cmd[@]/#tools/$root/tools}" > $root/gpurun_out/pmc_${tag}_$g.log 2>&1
  echo "group $g (${grp[*]}): rc=$?" | tee -a $root/gpurun_out/pmc_${tag}.rc
done
