set -u
# usage: tools/r2_pmc_corpus.sh <corpus>: SQ counters of every kernel of one bench run on that corpus (256 MiB) -> gpurun_out/pmc_<corpus>.txt
R=$GRAFT_REPO_ROOT; c=$1; O=$R/gpurun_out/pmc_$c; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --corpus $c --size-mb 256 --no-cpu-baseline --no-verify --steps 5 --warmup 2"
P="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT"
Q="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
S="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU"
timeout 600 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/insts -- $B > /dev/null 2> $O/insts.err
timeout 600 rocprofv3 --kernel-trace --pmc $Q --output-format csv -d $O/cycles -- $B > /dev/null 2> $O/cycles.err
timeout 600 rocprofv3 --kernel-trace --pmc $S --output-format csv -d $O/lds -- $B > /dev/null 2> $O/lds.err
python $R/tools/pmc_summary.py $O > $R/gpurun_out/pmc_$c.txt
rm -rf $O
cat $R/gpurun_out/pmc_$c.txt | grep -E "merge_pieces|long_pieces|pack_tokens|probe_tiles|split_tiles"
