#!/bin/bash
# Collects the round's evidence on the GPU box (everything lands under gpurun_out/prof_<round>/;
# tools/summarize_profile.py then copies the judged summaries into profiles/):
#   bench.json            the default bench line (H2D-inclusive value, resident leg, CPU baseline)
#   trace/                rocprofv3 --kernel-trace --stats of the same command
#   pmc_fetch, pmc_write  HBM byte counters, separate passes as the microarch guide prescribes
#   sq_*                  SQ counters per kernel with serial streams (tools/pmc_sq.sh)
#   c2/c3/c4.json + _trace  the other BASELINE configurations: bench line (roofline + cpu_baseline) and kernel stats
#   uniform / fresh       SURVEY 8(d)'s secondary inputs (uniform ids, fresh state)
#   emu8_rank*.json       one rank's share of the 8-GPU job, every rank in turn (tuning aid)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=${1:-r03}
OUT=gpurun_out/prof_$R
mkdir -p $OUT
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_shape.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-resident > $OUT/trace_bench.json 2> $OUT/trace_bench.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-profile --no-resident > $OUT/pmc_fetch.json 2>/dev/null
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-profile --no-resident > $OUT/pmc_write.json 2>/dev/null
python3 tools/timeline.py $OUT/trace > $OUT/timeline.txt 2>&1
python3 tools/step_gaps.py $OUT/trace > $OUT/step_gaps.txt 2>&1
# the other workloads: kernel stats and HBM counters of their own (bench.py attaches a traffic figure
# only to runs of the workload it was captured on)
pmc_of() {  # <tag> <bench args...>
  tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${tag}_trace -o bench -- python3 bench.py "$@" --steps 60 --warmup 6 --no-cpu-baseline --no-resident > /dev/null 2>&1
  cp $OUT/${tag}_trace/bench_kernel_stats.csv $OUT/${tag}_kernel_stats.csv 2>/dev/null || cp $(ls $OUT/${tag}_trace/*/*kernel_stats.csv | head -1) $OUT/${tag}_kernel_stats.csv
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$tag -o bench -- python3 bench.py "$@" --steps 8 --warmup 3 --no-cpu-baseline --no-profile --no-resident > $OUT/pmc_fetch_$tag.json 2>/dev/null
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$tag -o bench -- python3 bench.py "$@" --steps 8 --warmup 3 --no-cpu-baseline --no-profile --no-resident > /dev/null 2>&1
}
for c in c2 c3 c4; do
  python3 bench.py --config $c > $OUT/$c.json 2> $OUT/$c.err
  pmc_of $c --config $c
done
pmc_of uniform --dist uniform
python3 bench.py --dist uniform --no-cpu-baseline > $OUT/uniform.json 2>/dev/null
python3 bench.py --state fresh --no-cpu-baseline > $OUT/fresh.json 2>/dev/null
for r in 0 1 2 3 4 5 6 7; do python3 bench.py --emulate-shards 8 --emulate-rank $r --steps 40 --warmup 6 --no-profile > $OUT/emu8_rank$r.json 2>/dev/null; done
python3 bench.py --emulate-shards 8 --emulate-rank 3 --steps 40 --warmup 6 --scaling strong --no-profile > $OUT/emu8_strong_rank3.json 2>/dev/null
./tools/pmc_sq.sh $R > /dev/null 2>&1
python3 tools/pmc_sq_summary.py $R > $OUT/sq_summary.txt 2>&1
cat $OUT/bench.json | cut -c1-600
