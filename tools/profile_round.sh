#!/bin/bash
# GPU box, from the repository root: everything behind profiles/<round>_* in one call.
#   tools/profile_round.sh r03
# 1. the default bench line; 2. the same under rocprofv3 --kernel-trace --stats; 3. counter passes of
# `bench.py --pairs 32768 --steps 1 --warmup 0 --no-cpu` (FETCH_SIZE | WRITE_SIZE | SQ group: separate passes, as
# MI355X_MICROARCH.md prescribes); 4. TCP / TCC passes of the pose scorer alone (tools/score_probe.py).
# Every profiler run is bounded by its own timeout and writes under gpurun_out/<round>/; tools/collect_profiles.py turns
# that into profiles/<round>_* in the build container.
rnd=${1:-r03}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/$rnd
mkdir -p $out
cd $root
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench default rc=$?" | tee $out/rc.log
cd /tmp && export TMPDIR=/tmp
B="python3 $root/bench.py --pairs 32768 --steps 1 --warmup 0 --no-cpu"
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py --no-cpu > $out/bench_profiled.json 2> $out/stats.log; echo "stats rc=$?" | tee -a $out/rc.log
timeout -k 5 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- $B > $out/pmc_fetch.log 2>&1; echo "fetch rc=$?" | tee -a $out/rc.log
timeout -k 5 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- $B > $out/pmc_write.log 2>&1; echo "write rc=$?" | tee -a $out/rc.log
timeout -k 5 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_sq -- $B > $out/pmc_sq.log 2>&1; echo "sq rc=$?" | tee -a $out/rc.log
S="python3 $root/tools/score_probe.py"
timeout -k 5 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/score_sq -- $S > $out/score_sq.log 2>&1; echo "score sq rc=$?" | tee -a $out/rc.log
timeout -k 5 150 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum --kernel-trace --output-format csv -d $out/score_tcp -- $S > $out/score_tcp.log 2>&1; echo "score tcp rc=$?" | tee -a $out/rc.log
timeout -k 5 150 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum --kernel-trace --output-format csv -d $out/score_tcp2 -- $S > $out/score_tcp2.log 2>&1; echo "score tcp2 rc=$?" | tee -a $out/rc.log
timeout -k 5 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $out/score_tcc -- $S > $out/score_tcc.log 2>&1; echo "score tcc rc=$?" | tee -a $out/rc.log
cat $out/rc.log
