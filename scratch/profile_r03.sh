# Round-3 profile set (run on the MI355X box from the repo root): kernel trace + stats of the DRIVER-style bench
# (--steps 20 --warmup 5), FETCH_SIZE / WRITE_SIZE passes and SQ counter passes of the dense kernel, SQ passes of the
# trace filter kernel (counters only, one pass per set, no other trace domain).  Outputs under gpurun_out/prof_r03;
# scratch/collect_r03.py turns them into profiles/r03/*.{csv,json,md} with the kernel-source hashes.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_r03
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o r03 -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_traced.json 2> $out/bench_traced.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out -o r03_pmc_$(echo $c | tr A-Z a-z) -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-paths --no-scaling > /dev/null 2>&1
done
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_WR GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d $out -o sq_$tag -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-paths --no-scaling > /dev/null 2>&1
done
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_SALU GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d $out -o tr_$tag -- python bench_paths.py --ranks 20000000 --steps 1 --no-cpu > $out/tr_$tag.log 2>&1
done
python bench.py > $out/bench_default.json 2> $out/bench_default.err
python bench.py --steps 20 --warmup 5 > $out/bench_driver.json 2> $out/bench_driver.err
ls $out | head -50
