set -x
mkdir -p gpurun_out/r2p
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cd $R
python bench.py > gpurun_out/r2p/bench_unprofiled.json 2> gpurun_out/r2p/bench_unprofiled.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2p/stats -- python3 bench.py --no-cpu-baseline > gpurun_out/r2p/bench_profiled.json 2> gpurun_out/r2p/prof.err
SMALL="--steps 1 --warmup 0 --no-cpu-baseline --concurrent 0 --milp-nodes 0 --c4 0"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r2p/pmc_fetch -- python3 bench.py $SMALL > gpurun_out/r2p/pmc_f.json 2> gpurun_out/r2p/pmc_f.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r2p/pmc_write -- python3 bench.py $SMALL > gpurun_out/r2p/pmc_w.json 2> gpurun_out/r2p/pmc_w.err
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d gpurun_out/r2p/pmc_mfma -- python3 bench.py $SMALL > gpurun_out/r2p/pmc_m.json 2> gpurun_out/r2p/pmc_m.err
ls gpurun_out/r2p/*/*/ | head -30
du -sh gpurun_out/r2p
