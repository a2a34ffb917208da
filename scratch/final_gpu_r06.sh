cd $GRAFT_REPO_ROOT; out=gpurun_out/evidence_r06; mkdir -p $out
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -30 > $out/r06_pytest_gpu.txt; grep -E "passed|failed" $out/r06_pytest_gpu.txt
timeout 1000 python scratch/beam_stress.py 900 --seed=700 > $out/beam_stress_seed700.json 2> $out/beam_stress_seed700.err; tail -c 400 $out/beam_stress_seed700.json
