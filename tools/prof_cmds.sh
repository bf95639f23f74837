# rocprofv3 evidence of a round (run on the GPU box through gpurun: bash tools/prof_cmds.sh <tag>)
set -x
TAG=${1:-r5}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cd $R
HEAD="--no-cpu-baseline --concurrent 0 --milp-nodes 0 --c4 0 --frontier-vars 0 --general 0"
# 1. default bench, unprofiled (the numbers DESIGN.md quotes)
python bench.py > $OUT/bench_unprofiled.json 2> $OUT/bench_unprofiled.err
# 2. kernel stats of the headline region alone (metric LP) and of the frontier leg alone
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_M -- python3 bench.py $HEAD > $OUT/bench_M_profiled.json 2> $OUT/stats_M.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_C5 -- python3 tools/wave_prof.py 8 4 > $OUT/wave_profiled.out 2> $OUT/stats_C5.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_wide -- python3 tools/wave_prof.py 6 4 0 11 > $OUT/wide_profiled.out 2> $OUT/stats_wide.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_xwide -- python3 tools/wave_prof.py 4 4 0 13 > $OUT/xwide_profiled.out 2> $OUT/stats_xwide.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_C4 -- python3 bench.py --workload C4 --steps 2 --warmup 1 $HEAD > $OUT/bench_C4_profiled.json 2> $OUT/stats_C4.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_general -- python3 tools/general_prof.py > $OUT/general_profiled.out 2> $OUT/stats_general.err
# 3. PMC passes (separate runs; counters only)
SMALL="--steps 1 --warmup 0 --no-cpu-baseline --concurrent 0 --milp-nodes 0 --c4 0 --general 0 --frontier-vars 0"   # (the metric LP alone: the frontier legs have PMC passes of their own below)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $SMALL > $OUT/pmc_f.json 2> $OUT/pmc_f.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py $SMALL > $OUT/pmc_w.json 2> $OUT/pmc_w.err
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/pmc_mfma -- python3 bench.py $SMALL > $OUT/pmc_m.json 2> $OUT/pmc_m.err
# 4. PMC passes of the C5 wave ALONE (256 children; no 2048-wide leg in the same run: one traffic figure per kernel and shape)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_C5 -- python3 tools/wave_prof.py 4 4 > $OUT/pmc_f_C5.out 2> $OUT/pmc_f_C5.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_C5 -- python3 tools/wave_prof.py 4 4 > $OUT/pmc_w_C5.out 2> $OUT/pmc_w_C5.err
du -sh $OUT
