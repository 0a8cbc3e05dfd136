cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ldspmc; mkdir -p $OUT
BENCH="python bench.py --no-alt --no-cpu-baseline --min-seconds 0.05"
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT -d $OUT/a -- $BENCH > /dev/null 2> $OUT/a.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_GDS -d $OUT/b -- $BENCH > /dev/null 2> $OUT/b.err
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU -d $OUT/c -- $BENCH > /dev/null 2> $OUT/c.err
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_IFETCH -d $OUT/d -- $BENCH > /dev/null 2> $OUT/d.err
python tools/rocpd_summary.py counters k_step_philox1 $OUT/lds_pmc.json $OUT/a $OUT/b $OUT/c $OUT/d
cat $OUT/lds_pmc.json
tail -3 $OUT/a.err
rm -rf $OUT/a $OUT/b $OUT/c $OUT/d
