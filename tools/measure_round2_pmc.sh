set -u
# The SQ counter passes of tools/measure_round2.sh only (three separate --pmc runs of the headline command).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2_final; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-verify --steps 10 --warmup 3"
P="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT"
Q="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
S="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU"
rm -rf $O/pmc; mkdir -p $O/pmc
timeout 600 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/pmc/insts -- $B > /dev/null 2> $O/pmc_insts.err
timeout 600 rocprofv3 --kernel-trace --pmc $Q --output-format csv -d $O/pmc/cycles -- $B > /dev/null 2> $O/pmc_cycles.err
timeout 600 rocprofv3 --kernel-trace --pmc $S --output-format csv -d $O/pmc/lds -- $B > /dev/null 2> $O/pmc_lds.err
python $R/tools/pmc_summary.py $O/pmc > $O/pmc_english_1024.txt
find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete
grep -c td:: $O/pmc_english_1024.txt
