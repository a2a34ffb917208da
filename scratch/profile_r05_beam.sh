# The pruned-search part of scratch/profile_r05.sh alone (after a change that touches csrc/beam.hip only): kernel stats + SQ
# counters of the last expansion per configuration, then the driver's bench command.  Outputs merge into gpurun_out/prof_r05.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_r05
mkdir -p $out
for leg in "cfg4" "cfg4 --quads" "cfg3" "bruxelles3"; do
  name=$(echo $leg | tr -d ' -')
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -o beam_$name -- python scratch/cfg_beam.py $leg > $out/beam_$name.log 2>&1
  for set in "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU"; do
    tag=$(echo $set | tr ' ' '_' | cut -c1-30)
    rocprofv3 --pmc $set --output-format csv -d $out -o pmcbeam_${name}_$tag -- python scratch/cfg_beam.py $leg > /dev/null 2>&1
  done
done
python bench.py --gpus 1 --steps 20 --warmup 5 --full-json $out/bench_driver_full.json > $out/bench_driver.json 2> $out/bench_driver.err
find $out -name "*.db" -delete
find $out -name "beam_*_kernel_trace.csv" -delete
du -sh $out
