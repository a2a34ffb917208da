# Round-4 profile set of the dense-layout tracer (run on the MI355X box from the repo root): kernel trace + stats of
# bench_dense.py (24 chunks of the order-2 leg, and the same with the round-3 kernel: DRT_DENSE_LEGACY), FETCH_SIZE /
# WRITE_SIZE passes (counters only, one pass per counter, no other trace domain), image-method legs.
# Outputs under gpurun_out/prof_r04_dense; scratch/collect_r04.py turns them into profiles/r04/*.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_r04_dense
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o dense -- python bench_dense.py --max-chunks 24 > $out/dense_traced.json 2> $out/dense_traced.err
DRT_DENSE_LEGACY=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o dense_legacy -- python bench_dense.py --max-chunks 24 > $out/dense_legacy_traced.json 2> $out/dense_legacy_traced.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o dense3 -- python bench_dense.py --order 3 --max-chunks 12 > $out/dense3_traced.json 2> $out/dense3_traced.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out -o pmc_$(echo $c | tr A-Z a-z) -- python bench_dense.py --max-chunks 6 > /dev/null 2>&1
  rocprofv3 --pmc $c --output-format csv -d $out -o pmc3_$(echo $c | tr A-Z a-z) -- python bench_dense.py --order 3 --max-chunks 6 > /dev/null 2>&1
done
python bench_dense.py --all > $out/dense_all.json 2> $out/dense_all.err
ls -R $out | head -60
