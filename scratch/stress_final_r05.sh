# 90-second slices of the other stress drivers on the final tree of round 5 (run on the MI355X box from the repo root)
cd $GRAFT_REPO_ROOT
out=gpurun_out/stress_final_r05
mkdir -p $out
for d in oracle_stress query_oracle_stress bvh_stress trace_stress hybrid_stress em_stress smooth_stress; do
  python scratch/$d.py 90 > $out/$d.json 2> $out/$d.err
  tail -c 400 $out/$d.json; echo
done
