# Round-6 profile set (run on the MI355X box from the repo root).  Counters only in the --pmc passes, one pass per set,
# no other trace domain.  Outputs (CSV only: the copy-back limit is 64 MiB) under gpurun_out/prof_r06;
# scratch/collect_r06.py turns them into profiles/r06/*.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_r06
rm -rf $out && mkdir -p $out
# 1. the driver's command under the kernel trace (same flags as the driver uses)
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o r06 -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --full-json $out/bench_traced_full.json > $out/bench_traced.json 2> $out/bench_traced.err
# 2. headline kernel: HBM traffic (separate passes)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out -o pmcmt_$(echo $c | tr A-Z a-z) -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-paths --no-scaling --full-json /tmp/x.json > /dev/null 2>&1
done
# 3. dense tracer (reference API): kernel trace (its HBM-traffic record profiles/r05/pmc_trace_dense.json still describes the
#    kernel: trace_dense.hip changed only in the overflow branch of the capped queue -- re-collected all the same, order 2)
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o dense -- python bench_dense.py --max-chunks 24 > $out/dense_traced.json 2> $out/dense_traced.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out -o pmc_$(echo $c | tr A-Z a-z) -- python bench_dense.py --max-chunks 6 > /dev/null 2>&1
  rocprofv3 --pmc $c --output-format csv -d $out -o pmc3_$(echo $c | tr A-Z a-z) -- python bench_dense.py --order 3 --max-chunks 6 > /dev/null 2>&1
done
# 4. exhaustive filter kernel: executed VALU per candidate
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_SALU GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d $out -o tr_$tag -- python bench_paths.py --ranks 20000000 --steps 1 --no-cpu > $out/tr_$tag.log 2>&1
done
# 5. pruned search: kernel trace + SQ counters of the last expansion: configs[3] (triangles, coplanar pairs), configs[3] as
#    quads, configs[2], and the reference's bruxelles mesh at order 3
for leg in "cfg4" "cfg4 --quads" "cfg3" "bruxelles3"; do
  name=$(echo $leg | tr -d ' -')
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -o beam_$name -- python scratch/cfg_beam.py $leg > $out/beam_$name.log 2>&1
  for set in "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU"; do
    tag=$(echo $set | tr ' ' '_' | cut -c1-30)
    rocprofv3 --pmc $set --output-format csv -d $out -o pmcbeam_${name}_$tag -- python scratch/cfg_beam.py $leg > /dev/null 2>&1
  done
done
python bench.py --gpus 1 --steps 20 --warmup 5 --full-json $out/bench_driver_full.json > $out/bench_driver.json 2> $out/bench_driver.err
python bench.py --full-json $out/bench_default_full.json --write-strong-n1 $out/strong_n1.json > $out/bench_default.json 2> $out/bench_default.err
python bench_scaling.py --emulate-shards 8 --window 400000000 > $out/emulate_shards8.json 2> /dev/null
find $out -name "*.db" -delete
du -sh $out
