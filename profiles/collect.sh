#!/bin/bash
# Profile collection for one round (run on the GPU box from the repo root):
#     bash profiles/collect.sh r3 [passes] [config] [batch]
# passes: any of trace fetch write sq train (default: all).  config / batch: a bench.py workload other than the headline
# one (vit_s16 256, vit_b16_384 64 ...); its output goes to gpurun_out/prof_<round>_<config>.  A pass that is killed at
# its time limit ends the script; run the missing passes again in a call of their own.
# Pass 1: kernel trace + stats of the bench command.  Passes 2-4: PMC counters, each in a run of its own with
# kernel-trace only (TCC FETCH_SIZE and WRITE_SIZE do not fit one pass; SQ counters a third).
set -e
R=${1:-r3}
PASSES=${2:-"trace fetch write sq train"}
CONFIG=${3:-vit_b16}
BATCH=${4:-256}
OUT=gpurun_out/prof_$R
if [ "$CONFIG" != vit_b16 ]; then OUT=gpurun_out/prof_${R}_$CONFIG; fi
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
WL="--config $CONFIG --batch $BATCH"
BENCH="python3 bench.py $WL --steps 10 --warmup 3 --no-cpu-baseline"
# counter passes: no long warm-up.  What was seen once (no log kept): the --pmc passes run with the FULL bench command
# (~23 k dispatches, ~13 k of them queued by the 2-second warm-up before the first synchronize) did not finish inside
# their 150-200 s limits, while the kernel-trace pass of the same command always finished in ~5 s.  With counters on the
# profiler serialises every dispatch (the short command's ~1 k dispatches take a counter pass ~3 s, start-up included:
# on the order of a millisecond each, growing with the depth of the queue).  The library's kernels cannot be what stood
# still: none waits on another workgroup (s_barrier inside a workgroup is their only synchronisation; there is no atomic
# and no spin loop in csrc/), and the same launches complete under the trace pass and in every test.  Hence the short
# command below; VITM_BENCH_WATCHDOG=<s> (stack dump and exit) and VITM_SYNC_DEBUG=1 (name and wait for every call) are
# there to name the call in flight should a pass ever stand still again.
SHORT="python3 bench.py $WL --steps 2 --warmup 1 --no-cpu-baseline --warm-seconds 0 --min-timed-seconds 0"
want() { case " $PASSES " in *" $1 "*) echo "pass $1: $(date +%T)" >> $OUT/progress.log; return 0;; esac; return 1; }
if want trace; then
  rm -rf $OUT/trace
  timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/bench_trace.log 2>&1
fi
if want fetch; then
  rm -rf $OUT/pmc_fetch
  timeout -s KILL 90 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $SHORT > $OUT/pmc_fetch.log 2>&1
fi
if want write; then
  rm -rf $OUT/pmc_write
  timeout -s KILL 90 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $SHORT > $OUT/pmc_write.log 2>&1
fi
if want sq; then
  rm -rf $OUT/pmc_sq
  timeout -s KILL 90 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- $SHORT > $OUT/pmc_sq.log 2>&1
fi
if want train; then   # training step (informational): kernel trace + stats only
  rm -rf $OUT/train_bf16 $OUT/train_f32
  timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_bf16 -- python3 bench.py --mode train --dtype bf16 --batch 256 --steps 3 --warmup 1 > $OUT/train_bf16.log 2>&1
  timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_f32 -- python3 bench.py --mode train --dtype f32 --batch 64 --steps 3 --warmup 1 > $OUT/train_f32.log 2>&1
fi
echo "done: $(date +%T)" >> $OUT/progress.log
find $OUT -name "*.csv" | head -20
