# Round-5 stress records (run on the MI355X box from the repo root): rotated / shuffled random cities.
#   bash scratch/stress_r05.sh [seconds per driver]
cd $GRAFT_REPO_ROOT
S=${1:-1200}
out=gpurun_out/stress_r05
mkdir -p $out
python scratch/beam_stress.py $S > $out/beam_stress_rotated.json 2> $out/beam_stress_rotated.err
python scratch/beam_stress.py $((S / 2)) --kappa=1 > $out/beam_stress_rotated_kappa1.json 2> $out/beam_stress_rotated_kappa1.err
python scratch/trace_oracle_stress.py $S > $out/trace_oracle_stress_rotated.json 2> $out/trace_oracle_stress_rotated.err
tail -c 1500 $out/*.json
