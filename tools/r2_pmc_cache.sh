set -u
# usage: tools/r2_pmc_cache.sh <corpus>: L2 (TCC) / L1 (TCP) counters per kernel of one bench run (256 MiB) -> gpurun_out/cache_<corpus>.txt
R=$GRAFT_REPO_ROOT; c=$1; O=$R/gpurun_out/cache_$c; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --corpus $c --size-mb 256 --no-cpu-baseline --no-verify --steps 5 --warmup 2"
timeout 600 rocprofv3 --kernel-trace --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d $O/tcc -- $B > /dev/null 2> $O/tcc.err
timeout 600 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum --output-format csv -d $O/tcp -- $B > /dev/null 2> $O/tcp.err
python $R/tools/pmc_summary.py $O > $R/gpurun_out/cache_$c.txt 2>&1
tail -3 $O/tcc.err $O/tcp.err
rm -rf $O
grep -E "merge_pieces|long_pieces|pack_tokens|probe_tiles|split_tiles" $R/gpurun_out/cache_$c.txt
