# Round-6 evidence on the FINAL tree (run on the MI355X box from the repo root): the -m gpu suite, whole order-3 pair spaces
# through the exhaustive tracer (configs[3], configs[4], bruxelles, manhattan), the pruned-vs-exhaustive stress driver (half
# of the scenes triangle soups) at kappa 64 and 1, soups only, the fused tracer against the C oracle, 90-second slices of the
# other drivers.  Part "a" / "b" so that one gpurun call stays under an hour.  scratch/collect_evidence_r06.py files the results.
cd $GRAFT_REPO_ROOT
out=gpurun_out/evidence_r06
mkdir -p $out
part=${1:-a}
if [ "$part" = "a" ]; then
  timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -30 > $out/r06_pytest_gpu.txt; tail -3 $out/r06_pytest_gpu.txt
  timeout 1200 python scratch/exhaustive_pairs.py --pairs3 16 --pairs5 16 --pairs-real 8 --out $out/exhaustive_pairs.json > $out/exhaustive_pairs.log 2>&1; tail -1 $out/exhaustive_pairs.log
  timeout 1400 python scratch/beam_stress.py 1200 --seed=601 > $out/beam_stress.json 2> $out/beam_stress.err; tail -c 500 $out/beam_stress.json; echo
  timeout 700 python scratch/beam_stress.py 600 --kappa=1 --seed=602 > $out/beam_stress_kappa1.json 2> $out/beam_stress_kappa1.err; tail -c 500 $out/beam_stress_kappa1.json; echo
else
  python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; tail -1 $out/smoke.txt
  timeout 700 python scratch/beam_stress.py 600 --only-soup --seed=603 > $out/beam_stress_soups.json 2> $out/beam_stress_soups.err; tail -c 500 $out/beam_stress_soups.json; echo
  timeout 700 python scratch/trace_oracle_stress.py 600 > $out/trace_oracle_stress.json 2> $out/trace_oracle_stress.err; tail -c 500 $out/trace_oracle_stress.json; echo
  for k in 0.25 0.015625; do
    timeout 120 python scratch/beam_stress.py 60 --kappa=$k --seed=604 > $out/beam_stress_kappa_$k.json 2> /dev/null; tail -c 300 $out/beam_stress_kappa_$k.json; echo
  done
  for d in oracle_stress query_oracle_stress bvh_stress trace_stress hybrid_stress; do
    timeout 200 python scratch/$d.py 90 > $out/$d.json 2> $out/$d.err; tail -c 300 $out/$d.json; echo
  done
fi
