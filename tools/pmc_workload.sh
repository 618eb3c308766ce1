set -u
# SQ / GRBM / TCC counter passes of the bench command (one gpurun call; each pass a separate rocprofv3 --kernel-trace --pmc run, no
# other trace domains).  usage: tools/pmc_workload.sh <tag> <corpus> <MiB> [bench args] -> gpurun_out/<tag>/pmc_<corpus>_<MiB>.txt
R=$GRAFT_REPO_ROOT; tag=$1; c=$2; mb=$3; shift 3; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --corpus $c --size-mb $mb --no-cpu-baseline --no-verify --steps 3 --warmup 1 $*"
D=$O/pmc_${c}_$mb; rm -rf $D; mkdir -p $D
pass() { n=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $D/$n -- $B > /dev/null 2> $D/$n.err || echo "pass $n failed: $(tail -2 $D/$n.err)"; }
pass cycles SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
pass insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_BUSY_CU_CYCLES
pass level SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVE_CYCLES
pass l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
python $R/tools/pmc_table.py $D "$B" > $O/pmc_${c}_$mb.txt 2>&1
find $D -name "*agent_info.csv" -delete; find $D -name "*.db" -delete
cat $O/pmc_${c}_$mb.txt
