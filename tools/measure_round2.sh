set -u
# Round-2 measurement run (one gpurun call): bench lines of every config, rocprofv3 kernel stats of the bench command,
# SQ counters and HBM traffic (separate --pmc passes, --kernel-trace only), host-path rates.  Output: gpurun_out/r2_final/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2_final; mkdir -p $O
cd $R
nproc > $O/host.txt; grep -m1 "model name" /proc/cpuinfo >> $O/host.txt
timeout 900 python bench.py > $O/bench_english_1024.json 2> $O/bench_english_1024.err
timeout 600 python bench.py --size-mb 256 --no-cpu-baseline > $O/bench_english_256.json 2> $O/bench_english_256.err
for c in mixed code code_files; do timeout 600 python bench.py --corpus $c --size-mb 256 --steps 5 --warmup 2 > $O/bench_${c}_256.json 2> $O/bench_${c}_256.err; done
timeout 600 python bench.py --corpus mixed --pattern tekken --size-mb 256 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_mixed_tekken_256.json 2> $O/bench_mixed_tekken_256.err
timeout 900 python bench.py --corpus mixed --size-mb 1024 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_mixed_1024.json 2> $O/bench_mixed_1024.err
timeout 600 python bench.py --gpus 2 --same-gpu --dist-backend gloo --no-cpu-baseline --steps 5 --warmup 2 > $O/bench_2rank_same_gpu_1024.json 2> $O/bench_2rank.err
TD_BENCH_FORCE_DIST=1 timeout 600 python bench.py --no-cpu-baseline --steps 5 --warmup 2 > $O/bench_rccl_world1_1024.json 2> $O/bench_dist1.err
timeout 900 python tools/gpu_hostpath.py > $O/hostpath.txt 2>&1
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-verify --steps 10 --warmup 3"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats -- $B > $O/stats_bench.json 2> $O/stats.err
python $R/tools/prof_summary.py $O/stats $O/stats_english_1024.txt "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-verify --steps 10 --warmup 3 (1024 MiB English)" > /dev/null
for c in mixed code code_files; do timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats_$c -- python $R/bench.py --corpus $c --size-mb 256 --no-cpu-baseline --no-verify --steps 5 --warmup 2 > /dev/null 2> $O/stats_$c.err; python $R/tools/prof_summary.py $O/stats_$c $O/stats_${c}_256.txt "rocprofv3 --kernel-trace --stats -- python bench.py --corpus $c --size-mb 256 --no-cpu-baseline --no-verify --steps 5 --warmup 2" > /dev/null; rm -rf $O/stats_$c; done
rm -rf $O/stats
for c in FETCH_SIZE WRITE_SIZE; do rm -rf $O/traffic_english/$c; timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/traffic_english/$c -- $B > /dev/null 2> $O/traffic_$c.err; done
for w in mixed code; do for c in FETCH_SIZE WRITE_SIZE; do rm -rf $O/traffic_$w/$c; timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/traffic_$w/$c -- python $R/bench.py --corpus $w --size-mb 256 --no-cpu-baseline --no-verify --steps 5 --warmup 2 > /dev/null 2>> $O/traffic_$w.err; done; done
P="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT"
Q="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
S="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU"
rm -rf $O/pmc; mkdir -p $O/pmc
timeout 600 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/pmc/insts -- $B > /dev/null 2> $O/pmc_insts.err
timeout 600 rocprofv3 --kernel-trace --pmc $Q --output-format csv -d $O/pmc/cycles -- $B > /dev/null 2> $O/pmc_cycles.err
timeout 600 rocprofv3 --kernel-trace --pmc $S --output-format csv -d $O/pmc/lds -- $B > /dev/null 2> $O/pmc_lds.err
python $R/tools/pmc_summary.py $O/pmc > $O/pmc_english_1024.txt
# keep the merged download small: per-pass CSVs only
find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete
ls $O
